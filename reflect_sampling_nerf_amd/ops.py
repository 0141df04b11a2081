"""Thin torch-facing wrappers over the C ABI (include/rsn.h).  torch is plumbing here: it owns device
memory and the stream; every arithmetic step runs in librsn_hip.so.  No fallbacks."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _abi
from ._abi import CompositeIO, FieldOutputs, ReflectIO, check, ptr

RSN_COMP_EVAL = 1
RSN_COMP_CLIP_RGB = 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Measurement hook (bench.py, tools/): while one is installed as `ops.TIMER`, the dominant kernels' launches are
    bracketed by HIP events on the stream they are launched on (torch's current stream), tagged with a name and the
    work of the launch; `totals()` reads the events after a synchronize.  Not installed -> zero overhead."""

    def __init__(self):
        self.spans = []

    def add(self, name, start, end, work):
        self.spans.append((name, start, end, work))

    def totals(self):
        out = {}
        for name, s, e, work in self.spans:
            t = out.setdefault(name, {"calls": 0, "ms": 0.0, "work": {}})
            t["calls"] += 1
            t["ms"] += s.elapsed_time(e)
            for k, v in (work or {}).items():
                if k.endswith("_dev"):  # [(device int tensor, multiplier)]: work that depends on a device-side count
                    k, v = k[:-4], sum(int(c.item()) * m for c, m in v)
                t["work"][k] = t["work"].get(k, 0) + v
        return out


TIMER: Optional[KernelTimer] = None


def timed(name: str, work: Optional[Dict], fn):
    """Run `fn()` (one C-ABI launch); under an installed KernelTimer, between two events on the launch stream."""
    t = TIMER
    if t is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    t.add(name, s, e, work)
    return r


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.contiguous().float()
    if not t.is_cuda:
        raise _abi.RsnError("the HIP path needs tensors on a cuda (ROCm) device; there is no CPU fallback")
    return t


def sample_spaced(n_rays: int, n_dev: Optional[Tensor], n_samples: int, spacing: int, tan: float, nears: Tensor,
                  fars: Tensor, t_rand: Optional[Tensor]):
    lib = _abi.load_library()
    sb = torch.empty(n_rays, n_samples + 1, device=nears.device, dtype=torch.float32)
    eb = torch.empty_like(sb)
    if t_rand is not None:
        t_rand = _f32c(t_rand)
    check(lib.rsn_sample_spaced(n_rays, ptr(n_dev), n_samples, spacing, tan, ptr(nears), ptr(fars), ptr(t_rand),
                                ptr(sb), ptr(eb), _stream()))
    return sb, eb


def sample_pdf(n_rays: int, n_dev: Optional[Tensor], s_in: int, s_out: int, spacing: int, tan: float,
               histogram_padding: float, nears: Tensor, fars: Tensor, weights: Tensor, spacing_bins_in: Tensor,
               u_rand: Optional[Tensor]):
    lib = _abi.load_library()
    sb = torch.empty(n_rays, s_out + 1, device=nears.device, dtype=torch.float32)
    eb = torch.empty_like(sb)
    if u_rand is not None:
        u_rand = _f32c(u_rand)
    check(lib.rsn_sample_pdf(n_rays, ptr(n_dev), s_in, s_out, spacing, tan, histogram_padding, ptr(nears), ptr(fars),
                             ptr(weights), ptr(spacing_bins_in), ptr(u_rand), ptr(sb), ptr(eb), _stream()))
    return sb, eb


def composite(n_rays: int, n_dev: Optional[Tensor], n_samples: int, background: int, flags: int, sigma: Tensor,
              euclid_bins: Tensor, color: Tensor, bg_rgb: Optional[Tensor] = None, level: Optional[Dict] = None,
              surface: bool = False, want_depth: bool = True, ray_losses: bool = False) -> Dict[str, Tensor]:
    """-> weights [R,S], rgb [R,3], accumulation [R], depth [R] (+ diff/tint/normals/roughness if surface;
    + pn_loss_ray / ori_loss_ray [R] if ray_losses: the per-sample normal losses of get_loss_dict (model.py:403-407)
    reduced per ray where the weights are in registers; needs level["normals"], ["pred_normals"], ["n_dot_d"])."""
    lib = _abi.load_library()
    dev = sigma.device
    out = {
        "weights": torch.empty(n_rays, n_samples, device=dev, dtype=torch.float32),
        "rgb": torch.empty(n_rays, 3, device=dev, dtype=torch.float32),
        "accumulation": torch.empty(n_rays, device=dev, dtype=torch.float32),
    }
    if want_depth:
        out["depth"] = torch.empty(n_rays, device=dev, dtype=torch.float32)
    io = CompositeIO()
    io.sigma, io.euclid_bins, io.color, io.bg_rgb = ptr(sigma), ptr(euclid_bins), ptr(color), ptr(bg_rgb)
    io.weights, io.rgb, io.accumulation = ptr(out["weights"]), ptr(out["rgb"]), ptr(out["accumulation"])
    io.depth = ptr(out.get("depth"))
    if surface:
        assert level is not None
        out["diff"] = torch.empty(n_rays, 3, device=dev, dtype=torch.float32)
        out["tint"] = torch.empty(n_rays, 3, device=dev, dtype=torch.float32)
        out["normals"] = torch.empty(n_rays, 3, device=dev, dtype=torch.float32)
        out["roughness"] = torch.empty(n_rays, device=dev, dtype=torch.float32)
        io.diff, io.tint = ptr(level["diff"]), ptr(level["tint"])
        io.pred_normals, io.roughness = ptr(level["pred_normals"]), ptr(level["roughness"])
        io.diff_out, io.tint_out = ptr(out["diff"]), ptr(out["tint"])
        io.normals_out, io.roughness_out = ptr(out["normals"]), ptr(out["roughness"])
    if ray_losses:
        assert level is not None
        out["pn_loss_ray"] = torch.empty(n_rays, device=dev, dtype=torch.float32)
        out["ori_loss_ray"] = torch.empty(n_rays, device=dev, dtype=torch.float32)
        io.pred_normals, io.normals, io.n_dot_d = ptr(level["pred_normals"]), ptr(level["normals"]), ptr(level["n_dot_d"])
        io.pn_loss_ray, io.ori_loss_ray = ptr(out["pn_loss_ray"]), ptr(out["ori_loss_ray"])
    check(lib.rsn_composite(n_rays, ptr(n_dev), n_samples, background, flags, io, _stream()))
    return out


def reflect_setup(origins: Tensor, directions: Tensor, accumulation: Tensor, depth: Tensor, normals: Tensor,
                  roughness: Tensor, reflect_far: float) -> Dict[str, Tensor]:
    lib = _abi.load_library()
    R, dev = origins.shape[0], origins.device
    f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
    out = {
        "mask": torch.empty(R, device=dev, dtype=torch.uint8),
        "n_masked": torch.empty(1, device=dev, dtype=torch.int32),  # written by the kernel (also when R == 0)
        "ray_index": torch.empty(R, device=dev, dtype=torch.int32),
        "n_dot_d": f(R), "origins2": f(R, 3), "directions2": f(R, 3), "sqradius": f(R), "pixel_area2": f(R),
        "nears2": f(R), "fars2": f(R), "reflect_coarse": f(R, 3), "reflect_fine": f(R, 3),
    }
    io = ReflectIO()
    io.origins, io.directions, io.accumulation, io.depth = ptr(origins), ptr(directions), ptr(accumulation), ptr(depth)
    io.pred_normals, io.roughness = ptr(normals), ptr(roughness)
    for k, v in out.items():
        setattr(io, k, ptr(v))
    ws = torch.empty(max(1, lib.rsn_reflect_workspace_bytes(R) // 4), device=dev, dtype=torch.int32)
    io.workspace = ptr(ws)
    check(lib.rsn_reflect_setup(R, reflect_far, io, _stream()))
    return out


def reflect_combine(n_max: int, n_masked: Tensor, ray_index: Tensor, diff: Tensor, tint: Tensor, comp: Tensor,
                    out: Tensor) -> None:
    lib = _abi.load_library()
    check(lib.rsn_reflect_combine(n_max, ptr(n_masked), ptr(ray_index), ptr(diff), ptr(tint), ptr(comp), ptr(out),
                                  _stream()))


def field_outputs_struct(level: Dict[str, Tensor]) -> FieldOutputs:
    fo = FieldOutputs()
    for name in ("sigma", "color", "pred_normals", "n_dot_d", "diff", "tint", "roughness", "raw_density",
                 "raw_roughness"):
        setattr(fo, name, ptr(level.get(name)))
    return fo


def sh34_encode(directions: Tensor, roughness: Optional[Tensor] = None) -> Tensor:
    """IntegratedSHEncoding.forward: [..., 3], [..., 1] or None -> [..., 34] (rsn_sh34_encode)."""
    lib = _abi.load_library()
    lead = directions.shape[:-1]
    d = _f32c(directions.reshape(-1, 3))
    n = d.shape[0]
    r = _f32c(roughness.expand(*lead, 1).reshape(-1)) if roughness is not None else None
    out = torch.empty(n, 34, device=d.device, dtype=torch.float32)
    check(lib.rsn_sh34_encode(n, ptr(d), ptr(r), ptr(out), _stream()))
    return out.reshape(*lead, 34)


def ipe_encode(means: Tensor, covs: Optional[Tensor], freqs: Tensor) -> Tensor:
    """NeRFEncoding.forward(means, covs): [..., 3] (+ [..., 3, 3] or diagonal [..., 3]) -> [..., 99] (rsn_ipe_encode)."""
    lib = _abi.load_library()
    lead = means.shape[:-1]
    m = _f32c(means.reshape(-1, 3))
    n = m.shape[0]
    cd = None
    if covs is not None:
        cd = covs if covs.shape[-1] == 3 and covs.dim() == means.dim() else torch.diagonal(covs, dim1=-2, dim2=-1)
        cd = _f32c(cd.reshape(-1, 3))
    fr = (C.c_float * 16)(*[float(f) for f in freqs.tolist()])
    out = torch.empty(n, 99, device=m.device, dtype=torch.float32)
    check(lib.rsn_ipe_encode(n, ptr(m), ptr(cd), fr, ptr(out), _stream()))
    return out.reshape(*lead, 99)


class LazyOutputs(dict):
    """The output dict of get_outputs (training and eval).  One key of the reference needs the reflected-ray count M on
    the host: `depth_reflect_fine`, [M, 1], present only when M > 0 (reference model.py:341).  Here M lives on the device,
    so that entry is materialised on FIRST USE: `d["depth_reflect_fine"]`, `in`, `get`, and every way of looking at the
    dict as a whole -- `keys()`, `items()`, `values()`, iteration, `len()`, `dict(d)`, `copy()` -- so a generic consumer
    (metrics, loggers, dict copies) sees exactly the reference's key set.  That first use is one device-to-host read; a
    training step or a chunked image render that only indexes the keys it needs performs none.  `present()` lists what is
    there WITHOUT materialising (the renderer's own chunk loop uses it)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.lazy: Dict[str, tuple] = {}
        self.fused: Optional[Dict[str, Tensor]] = None  # per-ray loss reductions for get_loss_dict (training graph)

    def _materialise(self, key):
        count, per_ray = self.lazy.pop(key)
        m = int(count.item())
        if m > 0:
            dict.__setitem__(self, key, per_ray[:m].unsqueeze(-1).detach())

    def materialise(self):
        for key in list(self.lazy):
            self._materialise(key)
        return self

    def present(self):
        """(key, value) pairs already in the dict; pending lazy entries are left alone (no device-to-host read)."""
        return dict.items(self)

    def __missing__(self, key):
        if key in self.lazy:
            self._materialise(key)
            if dict.__contains__(self, key):
                return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        if key in self.lazy:
            self._materialise(key)
        return dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):
        return dict.__iter__(self.materialise())

    def __len__(self):
        return dict.__len__(self.materialise())

    def keys(self):
        return dict.keys(self.materialise())

    def items(self):
        return dict.items(self.materialise())

    def values(self):
        return dict.values(self.materialise())

    def copy(self):
        return dict(dict.items(self.materialise()))
