"""Loss and optimiser step of the training iteration on the HIP path.

FusedLoss   -- get_loss_dict (reference model.py:346-430) on ANY outputs dict: rsn_loss_forward_backward computes the 8
               terms and their gradients w.r.t. the model outputs in one pass over the samples.
FusedRayLoss-- the same for outputs of this package's training graph: the per-sample normal terms were already reduced
               per ray in the compositing epilogue, the loss is two launches over R rays.
FusedRAdam  -- torch.optim.RAdam semantics (reference config.py:50-53) as one multi-tensor launch (rsn_radam_step),
               with the reference's exponential learning-rate decay (lr 1e-3 -> 1e-4 over 50 000 steps).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch
from torch import Tensor

from . import _abi, ops
from ._abi import check, ptr

LOSS_TERMS = ("loss_mid_coarse", "loss_mid_fine", "loss_reflect_mid_coarse", "loss_reflect_mid_fine",
              "predicted_normal_loss_coarse", "predicted_normal_loss_fine", "orientation_loss_coarse",
              "orientation_loss_fine")


def _ptr_array(tensors: List[Optional[Tensor]]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class FusedLoss(torch.autograd.Function):
    """(image, coef[8], rgb x4, weights x2, normals x2, pred_normals x2, n_dot_d x2) -> scaled losses [8]."""

    @staticmethod
    def forward(ctx, image, coef, rgb_c, rgb_f, refl_c, refl_f, w_c, w_f, n_c, n_f, pn_c, pn_f, ndd_c, ndd_f):
        lib = _abi.load_library()
        f = ops._f32c
        image = f(image)
        rgb = [f(rgb_c), f(rgb_f), f(refl_c), f(refl_f)]
        R = rgb[0].shape[0]
        w = [f(w_c.reshape(R, -1)), f(w_f.reshape(R, -1))]
        Sc, Sf = w[0].shape[1], w[1].shape[1]
        nrm = [f(n_c.reshape(R, Sc, 3)), f(n_f.reshape(R, Sf, 3))]
        pn = [f(pn_c.reshape(R, Sc, 3)), f(pn_f.reshape(R, Sf, 3))]
        ndd = [f(ndd_c.reshape(R, Sc)), f(ndd_f.reshape(R, Sf))]
        dev = image.device
        losses = torch.empty(8, device=dev)
        g_rgb = [torch.empty_like(t) for t in rgb]
        g_pn = [torch.empty_like(t) for t in pn]
        g_ndd = [torch.empty_like(t) for t in ndd]
        coef_arr = (C.c_float * 8)(*[float(c) for c in coef])
        check(lib.rsn_loss_forward_backward(R, Sc, Sf, ptr(image), _ptr_array(rgb), _ptr_array(w), _ptr_array(nrm),
                                            _ptr_array(pn), _ptr_array(ndd), coef_arr, ptr(losses), _ptr_array(g_rgb),
                                            _ptr_array(g_pn), _ptr_array(g_ndd), ops._stream()))
        ctx.grads = (g_rgb, g_pn, g_ndd)
        ctx.dims = (R, Sc, Sf)
        ctx.shapes = (pn_c.shape, pn_f.shape, ndd_c.shape, ndd_f.shape)
        key = (tuple(float(c) for c in coef), dev)
        if FusedLoss._coef_cache[0] != key:  # the coefficients change twice per run (warm-up): no per-step upload
            FusedLoss._coef_cache = (key, torch.tensor([float(c) for c in coef], device=dev))
        return losses * FusedLoss._coef_cache[1]

    _coef_cache = (None, None)

    @staticmethod
    def backward(ctx, g8):
        lib = _abi.load_library()
        if ctx.grads is None:
            raise RuntimeError("FusedLoss: the gradient buffers are scaled in place; backward can run once per forward")
        g_rgb, g_pn, g_ndd = ctx.grads
        ctx.grads = None
        R, Sc, Sf = ctx.dims
        s = ctx.shapes
        # chain rule with the upstream gradients of the eight terms, in place, one launch (they stay on the device)
        check(lib.rsn_loss_scale_grads(R, Sc, Sf, ptr(ops._f32c(g8)), _ptr_array(g_rgb), _ptr_array(g_pn),
                                       _ptr_array(g_ndd), ops._stream()))
        out = [None, None]
        out += list(g_rgb)
        out += [None, None, None, None]  # weights, normals: constants of the loss (detached in the reference)
        out += [g_pn[0].reshape(s[0]), g_pn[1].reshape(s[1]), g_ndd[0].reshape(s[2]), g_ndd[1].reshape(s[3])]
        return tuple(out)


def fused_loss_dict(outputs: Dict[str, Tensor], image: Tensor, coefficients: Dict[str, float]) -> Dict[str, Tensor]:
    coef = [float(coefficients.get(k, 1.0)) for k in LOSS_TERMS]
    scaled = FusedLoss.apply(image, coef, outputs["mid_rgb_coarse"], outputs["mid_rgb_fine"],
                             outputs["mid_reflect_coarse"], outputs["mid_reflect_fine"], outputs["weights_coarse"],
                             outputs["weights_fine"], outputs["normals_coarse"], outputs["normals_fine"],
                             outputs["pred_normals_coarse"], outputs["pred_normals_fine"], outputs["n_dot_d_coarse"],
                             outputs["n_dot_d_fine"])
    return dict(zip(LOSS_TERMS, scaled.unbind(0)))  # one backward node (a stack) instead of eight select_backwards


class FusedRayLoss(torch.autograd.Function):
    """(image, coef[8], rgb x4 [R,3], pn_loss_ray x2 [R], ori_loss_ray x2 [R]) -> scaled losses [8].

    The training graph reduces the two per-sample normal terms of get_loss_dict (model.py:403-407) per ray inside the
    compositing kernel, where the weights are in registers (train_graph.FUSED_KEYS); what is left of the loss is work on
    R rays: one launch forward (rsn_loss_rays_forward), one backward.  The per-sample gradients d/d pred_normals,
    d/d n_dot_d are formed inside the field's backward kernel from the per-ray upstream gradients returned here."""

    @staticmethod
    def forward(ctx, image, coef, rgb_c, rgb_f, refl_c, refl_f, pnr_c, pnr_f, orr_c, orr_f):
        lib = _abi.load_library()
        f = ops._f32c
        image = f(image)
        rgb = [f(rgb_c), f(rgb_f), f(refl_c), f(refl_f)]
        R = rgb[0].shape[0]
        pnr, orr = [f(pnr_c), f(pnr_f)], [f(orr_c), f(orr_f)]
        dev = image.device
        losses = torch.empty(8, device=dev)
        g_rgb = [torch.empty_like(t) for t in rgb]
        coef_arr = (C.c_float * 8)(*[float(c) for c in coef])
        check(lib.rsn_loss_rays_forward(R, ptr(image), _ptr_array(rgb), _ptr_array(pnr), _ptr_array(orr), coef_arr,
                                        ptr(losses), _ptr_array(g_rgb), ops._stream()))
        ctx.g_rgb, ctx.R, ctx.coef_arr = g_rgb, R, coef_arr
        key = (tuple(float(c) for c in coef), dev)
        if FusedLoss._coef_cache[0] != key:  # the coefficients change twice per run (warm-up): no per-step upload
            FusedLoss._coef_cache = (key, torch.tensor([float(c) for c in coef], device=dev))
        return losses * FusedLoss._coef_cache[1]

    @staticmethod
    def backward(ctx, g8):
        lib = _abi.load_library()
        if ctx.g_rgb is None:
            raise RuntimeError("FusedRayLoss: the gradient buffers are scaled in place; backward can run once per forward")
        g_rgb, R = ctx.g_rgb, ctx.R
        ctx.g_rgb = None
        dev = g_rgb[0].device
        g_pn = [torch.empty(R, device=dev), torch.empty(R, device=dev)]
        g_or = [torch.empty(R, device=dev), torch.empty(R, device=dev)]
        check(lib.rsn_loss_rays_backward(R, ptr(ops._f32c(g8)), ctx.coef_arr, _ptr_array(g_rgb), _ptr_array(g_pn),
                                         _ptr_array(g_or), ops._stream()))
        return (None, None, *g_rgb, *g_pn, *g_or)


def fused_ray_loss_dict(outputs: Dict[str, Tensor], fused: Dict[str, Tensor], image: Tensor,
                        coefficients: Dict[str, float]) -> Dict[str, Tensor]:
    coef = [float(coefficients.get(k, 1.0)) for k in LOSS_TERMS]
    scaled = FusedRayLoss.apply(image, coef, outputs["mid_rgb_coarse"], outputs["mid_rgb_fine"],
                                outputs["mid_reflect_coarse"], outputs["mid_reflect_fine"], fused["pn_loss_ray_coarse"],
                                fused["pn_loss_ray_fine"], fused["ori_loss_ray_coarse"], fused["ori_loss_ray_fine"])
    return dict(zip(LOSS_TERMS, scaled.unbind(0)))


def exponential_decay_lr(step: int, lr_init: float = 1e-3, lr_final: float = 1e-4, max_steps: int = 50000) -> float:
    """nerfstudio ExponentialDecayScheduler without warm-up (reference config.py:52): log-linear interpolation."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return float(lr_init * (lr_final / lr_init) ** t)


class FusedRAdam:
    """RAdam (torch.optim.RAdam semantics, weight_decay 0) over a fixed parameter list, one kernel launch per step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-15,
                 lr_final: Optional[float] = None, max_steps: int = 50000):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.lr_final, self.max_steps = lr_final, max_steps
        self.step_count = 0
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self._sizes = (C.c_int32 * len(self.params))(*[p.numel() for p in self.params])

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # -- checkpoint / resume in torch.optim.RAdam's own format: a run of the reference (nerfstudio's RAdamOptimizerConfig builds
    #    torch.optim.RAdam; its trainer saves optimizer.state_dict()) resumes here and the other way round.
    def state_dict(self) -> dict:
        state = {}
        for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq)):
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m, "exp_avg_sq": v}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "decoupled_weight_decay": False,
                 "foreach": None, "maximize": False, "capturable": False, "differentiable": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError(f"optimizer state for {len(ids)} parameters, this optimizer has {len(self.params)}")
        g0 = groups[0]
        if any(g.get("weight_decay", 0) != 0 for g in groups):
            raise ValueError("FusedRAdam has no weight decay (the reference trains without: config.py:50-53)")
        self.lr, self.betas, self.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
        steps = set()
        for k, pid in enumerate(ids):  # k-th parameter of this optimizer <- state entry `pid` (torch numbers them in group order)
            st = sd["state"].get(pid)
            p = self.params[k]
            if st is None:  # torch creates a parameter's state at its first gradient: none yet = zeros
                self.exp_avg[k].zero_()
                self.exp_avg_sq[k].zero_()
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"parameter {k}: state of shape {tuple(st['exp_avg'].shape)}, parameter {tuple(p.shape)}")
            self.exp_avg[k] = st["exp_avg"].detach().to(device=p.device, dtype=p.dtype).clone()
            self.exp_avg_sq[k] = st["exp_avg_sq"].detach().to(device=p.device, dtype=p.dtype).clone()
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one step count")
        self.step_count = steps.pop() if steps else 0

    def current_lr(self) -> float:
        if self.lr_final is None:
            return self.lr
        return exponential_decay_lr(self.step_count, self.lr, self.lr_final, self.max_steps)

    @torch.no_grad()
    def step(self) -> None:
        lib = _abi.load_library()
        lr = self.current_lr()
        self.step_count += 1
        if self.exp_avg[0].device != self.params[0].device:  # parameters were moved after construction
            self.exp_avg = [m.to(p.device) for m, p in zip(self.exp_avg, self.params)]
            self.exp_avg_sq = [v.to(p.device) for v, p in zip(self.exp_avg_sq, self.params)]
        grads = [None if p.grad is None else ops._f32c(p.grad) for p in self.params]
        check(lib.rsn_radam_step(len(self.params), _ptr_array([p.data for p in self.params]), _ptr_array(grads),
                                 _ptr_array(self.exp_avg), _ptr_array(self.exp_avg_sq), self._sizes, self.step_count,
                                 lr, self.betas[0], self.betas[1], self.eps, ops._stream()))
        # The kernel updated the parameters behind torch's back: bump their version counters so that consumers keyed on
        # Tensor._version (the Field's packed-weights cache) see the change.
        touched = [p for p in self.params if p.grad is not None]
        setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
        if setter is not None:
            setter(touched, [p._version + 1 for p in touched])
        else:  # pragma: no cover - older torch: a no-op in-place op bumps the counter
            for p in touched:
                p.data.mul_(1.0)
