"""Training-mode get_outputs as ONE autograd node (reference: reflect_sampling_nerf_model.py:142-344 under
`self.training`, with torch autograd building the graph op by op).

forward  = samplers (stratified jitter) -> fused field kernels in training mode (activations saved, analytic
           normals by a dX sweep) -> compositing -> reflected rays (sized for R rays, the count M read from device memory
           by every launch; the host never reads it: the step has no device-to-host synchronisation).
backward = the same pipeline reversed through librsn_hip.so: reflect combine / composite backward (suffix scan),
           rsn_field_backward_* (transposed-weight MFMA sweep producing every layer's pre-activation gradient),
           then rsn_weight_grad: dW = dY^T X and db for every linear layer (output-stationary MFMA reduction over
           all samples), accumulated straight into nn.Linear-layout gradient tensors.

Which outputs carry gradient, and the detach points, follow the reference exactly: mid_rgb_*, mid_reflect_*
(through colour and -- for the primary levels -- through the weights), pred_normals_*, n_dot_d_*, roughness;
reflect weights, diff/tint (rendered), normals, accumulation, depth and the resampled bins are constants.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import _abi, ops
from ._abi import CompositeBwdIO, FieldGradsIn, FieldGradsOut, FieldSaved, check, ptr

ENC_SLOTS = 104
SH_SLOTS = 40


def enc_slot_columns() -> List[int]:
    """slot k (= it*8 + 4h + s) of the kernel's encoded-input order -> column of the reference's 99-wide
    NeRFEncoding output, or -1 (padding).  Mirrors csrc/rsn_pack.hip: enc_slot_to_column."""
    cols = []
    for k in range(ENC_SLOTS):
        it, h, s = k >> 3, (k >> 2) & 1, k & 3
        u = it * 4 + s
        if u < 24:
            c = (u // 8) * 16 + 8 * h + (u % 8)
        elif u < 48:
            c = 48 + ((u - 24) // 8) * 16 + 8 * h + ((u - 24) % 8)
        elif u < 51 and h == 0:
            c = 96 + (u - 48)
        else:
            c = -1
        cols.append(c)
    return cols


def sh_slot_columns() -> List[int]:
    cols = []
    for k in range(SH_SLOTS):
        it, h, s = k >> 3, (k >> 2) & 1, k & 3
        u = it * 4 + s
        cols.append(17 * h + u if u < 17 else -1)
    return cols


def _unpermute(g_slots: Tensor, cols: List[int], width: int) -> Tensor:
    """[out, n_slots] gradient in slot order -> [out, width] in reference column order."""
    live = [i for i, c in enumerate(cols) if c >= 0]
    dst = [cols[i] for i in live]
    out = g_slots.new_zeros(g_slots.shape[0], width)
    out[:, dst] = g_slots[:, live]
    return out


# ------------------------------------------------------------------------------------------------ thin wrappers
def _composite_backward(n, S, background, flags, detach_w, level, eb, weights, g_rgb, bg=None, g_rough=None,
                        want_sigma=True, want_bg=False, rough_samples=None, g_acc=None, n_dev=None):
    lib = _abi.load_library()
    dev = g_rgb.device
    out = {"g_color": torch.empty(n, S, 3, device=dev)}
    if want_sigma:
        out["g_sigma"] = torch.empty(n, S, device=dev)
    if rough_samples is not None:
        out["g_rough"] = torch.empty(n, S, device=dev)
    if want_bg:
        out["g_bg"] = torch.zeros(n, 3, device=dev)  # rows behind a device-side count stay zero
    io = CompositeBwdIO()
    io.sigma, io.euclid_bins, io.color, io.bg_rgb = ptr(level["sigma"]), ptr(eb), ptr(level["color"]), ptr(bg)
    io.roughness, io.weights, io.g_rgb, io.g_roughness = ptr(rough_samples), ptr(weights), ptr(g_rgb), ptr(g_rough)
    io.g_accumulation = ptr(g_acc)
    io.g_sigma, io.g_color = ptr(out.get("g_sigma")), ptr(out["g_color"])
    io.g_roughness_sample, io.g_bg = ptr(out.get("g_rough")), ptr(out.get("g_bg"))
    check(lib.rsn_composite_backward(n, ptr(n_dev), S, background, flags, detach_w, io, ops._stream()))
    return out


def _saved_struct(saved: Dict[str, Tensor]) -> FieldSaved:
    fs = FieldSaved()
    for k in ("enc", "act", "bott", "sh", "hid", "heads", "normals", "relu_bits"):
        setattr(fs, k, ptr(saved.get(k)))
    return fs


def _alloc_gout(field, N: int, dev, need_input: bool):
    W, L = field.width, field.mlp_base.num_layers
    wd = field.wide_dtype()  # bf16 in the reduced-precision training mode (rsn_field_grads_out)
    g = {"dz_rgb": torch.empty(N, 4, device=dev), "da_mid": torch.empty(N, 128, device=dev, dtype=wd),
         "d_bott": torch.empty(N, W, device=dev, dtype=wd), "dz_heads": torch.empty(N, 16, device=dev),
         "dy": torch.empty(L, N, W, device=dev, dtype=wd)}
    if need_input:
        g["d_input"] = torch.empty(N, device=dev)
    st = FieldGradsOut()
    for k in ("dz_rgb", "da_mid", "d_bott", "dz_heads", "dy", "d_input"):
        setattr(st, k, ptr(g.get(k)))
    return g, st


class _GradAcc:
    """Zero-initialised parameter gradients (nn.Linear layout, reference state_dict names) that rsn_weight_grad
    accumulates into across the five field evaluations of one step."""

    def __init__(self, field):
        self.field = field
        dev = next(field.parameters()).device
        W = field.param_width  # the heads block has the parameters' width (the kernels' zero-padded units have no gradient)
        # one flat zero-filled buffer (one fill kernel per step), viewed per parameter + the two heads blocks
        named = [(n, p) for n, p in field.named_parameters() if "field_output_low" not in n]
        sizes = [p.numel() for _, p in named] + [16 * W, 16]
        flat = torch.zeros(sum(sizes), device=dev)
        views = flat.split(sizes)
        self.g: Dict[str, Tensor] = {n: v.view_as(p) for (n, p), v in zip(named, views)}
        self.heads_w = views[-2].view(16, W)  # rows: 0 density, 1-3 normals, 4-6 diff, 8 roughness, 12-14 tint
        self.heads_b = views[-1]
        # slot -> reference column of the saved encoded / SH inputs: the layout of the kernel that serves this shape and mode
        field._enc_col_map, field._sh_col_map = field.train_col_maps(dev)

    def finish(self) -> Dict[str, Tensor]:
        dst, src = [], []
        for name, lo, hi in (("density", 0, 1), ("normals", 1, 4), ("diff", 4, 7), ("roughness", 8, 9),
                             ("tint", 12, 15)):
            dst += [self.g[f"field_output_{name}.net.weight"], self.g[f"field_output_{name}.net.bias"]]
            src += [self.heads_w[lo:hi], self.heads_b[lo:hi]]
        torch._foreach_copy_(dst, src)  # the head tensors received nothing else: a copy, in one multi-tensor launch
        return self.g


def _wgrad(dy: Tensor, n_out: int, x: Tensor, k_in: int, dw: Tensor, dw_col0: int, db: Optional[Tensor],
           col_map: Optional[Tensor] = None):
    """dw[:, dw_col0 + (col_map[k] or k)] += dy[:, :n_out]^T x[:, :k_in];  db += column sums of dy."""
    _wgrad_multi([(dy, x)], n_out, k_in, dw, dw_col0, db, col_map)


_WGRAD_MODE = 0  # RSN_MMA_*: set per step by _weight_grads from the Field's MMA mode
_WGRAD_JOBS_MAX = 8  # WG_MAX_JOBS of rsn_wgrad.hip


def _wgrad_multi(segs, n_out: int, k_in: int, dw: Tensor, dw_col0: int, db: Optional[Tensor],
                 col_map: Optional[Tensor] = None):
    """One weight-gradient launch: the reduction runs over the points of every (dy, x[, (count, rows per count)]) segment
    (the field evaluations of one step share their weights), so the per-launch flush is paid once per layer.  A segment
    with a device-side count holds min(rows of dy, count * rows per count) rows (rsn_weight_grad_multi_dev): the
    reflected-ray count never comes to the host."""
    lib = _abi.load_library()
    segs = [sg for sg in segs if sg[0].shape[0] > 0]
    if not segs:
        return
    ns = len(segs)
    ld_dy, ld_x = segs[0][0].stride(0), segs[0][1].stride(0)
    assert all(sg[0].stride(0) == ld_dy and sg[1].stride(0) == ld_x and sg[0].shape[0] == sg[1].shape[0] for sg in segs)
    npts = (C.c_int64 * ns)(*[sg[0].shape[0] for sg in segs])
    dys = (C.c_void_p * ns)(*[sg[0].data_ptr() for sg in segs])
    xs = (C.c_void_p * ns)(*[sg[1].data_ptr() for sg in segs])
    cnt = [sg[2] if len(sg) > 2 else None for sg in segs]
    ndev = (C.c_void_p * ns)(*[None if c is None else c[0].data_ptr() for c in cnt])
    per = (C.c_int32 * ns)(*[1 if c is None else int(c[1]) for c in cnt])
    dwp = C.c_void_p(dw.data_ptr() + 4 * dw_col0)
    bf = torch.bfloat16  # rows that are bf16 in memory (reduced-precision training): all segments alike
    assert all((sg[0].dtype == bf) == (segs[0][0].dtype == bf) and (sg[1].dtype == bf) == (segs[0][1].dtype == bf) for sg in segs)
    operand_bf16 = (1 if segs[0][1].dtype == bf else 0) | (2 if segs[0][0].dtype == bf else 0)
    work = {"point_out_in": sum(sg[0].shape[0] for sg in segs if len(sg) < 3 or sg[2] is None) * n_out * k_in}
    dev_work = [(c[0], c[1] * n_out * k_in) for c in cnt if c is not None]
    if dev_work:
        work["point_out_in_dev"] = dev_work
    ops.timed("weight_grad", work,
              lambda: check(lib.rsn_weight_grad_multi_dev(ns, npts, ndev, per, dys, ld_dy, n_out, xs, ld_x, k_in, ptr(col_map),
                                                          dwp, dw.stride(0), ptr(db), _WGRAD_MODE, operand_bf16, ops._stream())))


def _wgrad_jobs(jobs, n_out: int, k_in: int):
    """Several weight-gradient reductions of ONE shape in one launch (rsn_weight_grad_jobs): jobs = [(segs, dw, dw_col0, db[, col_map])]
    with segs as _wgrad_multi takes them, the same segment lengths / counts / leading dimensions / dtypes in every job.
    The workgroups are dealt to the jobs; the launch pays one atomic-flush phase and one ramp for all of them."""
    lib = _abi.load_library()
    jobs = [(jb[0], jb[1], jb[2], jb[3], jb[4] if len(jb) > 4 else None) for jb in jobs]
    keep = [i for i, sg in enumerate(jobs[0][0]) if sg[0].shape[0] > 0]
    if not keep:
        return
    ns = len(keep)
    ref = [jobs[0][0][i] for i in keep]
    ld_dy, ld_x = ref[0][0].stride(0), ref[0][1].stride(0)
    bf = torch.bfloat16
    operand_bf16 = (1 if ref[0][1].dtype == bf else 0) | (2 if ref[0][0].dtype == bf else 0)
    npts = (C.c_int64 * ns)(*[sg[0].shape[0] for sg in ref])
    cnt = [sg[2] if len(sg) > 2 else None for sg in ref]
    ndev = (C.c_void_p * ns)(*[None if c is None else c[0].data_ptr() for c in cnt])
    per = (C.c_int32 * ns)(*[1 if c is None else int(c[1]) for c in cnt])
    arr = (_abi.WGradJob * len(jobs))()
    hold = []  # the pointer arrays must outlive the call
    for q, (segs_, dw, c0, db, cmap) in zip(arr, jobs):
        sel = [segs_[i] for i in keep]
        assert all(a[0].shape[0] == b[0].shape[0] and a[0].stride(0) == ld_dy and a[1].stride(0) == ld_x and
                   a[0].dtype == b[0].dtype and a[1].dtype == b[1].dtype for a, b in zip(sel, ref))
        dys = (C.c_void_p * ns)(*[sg[0].data_ptr() for sg in sel])
        xs = (C.c_void_p * ns)(*[sg[1].data_ptr() for sg in sel])
        hold += [dys, xs]
        q.dy, q.x = dys, xs
        q.col_map = None if cmap is None else cmap.data_ptr()
        q.dw = dw.data_ptr() + 4 * c0
        q.ld_dw = dw.stride(0)
        q.db = None if db is None else db.data_ptr()
    work = {"point_out_in": len(jobs) * sum(sg[0].shape[0] for sg in ref if len(sg) < 3 or sg[2] is None) * n_out * k_in}
    dev_work = [(c[0], len(jobs) * c[1] * n_out * k_in) for c in cnt if c is not None]
    if dev_work:
        work["point_out_in_dev"] = dev_work
    ops.timed("weight_grad", work,
              lambda: check(lib.rsn_weight_grad_jobs(ns, npts, ndev, per, len(jobs), arr, ld_dy, n_out, ld_x, k_in,
                                                     _WGRAD_MODE, operand_bf16, ops._stream())))


def _weight_grads(field, levels, acc: _GradAcc):
    """dW = dY^T X (+ db) for every linear layer, reduced over all field evaluations of the step at once.
    levels: list of (saved activations, backward-sweep outputs, with_heads[, (device count, rows per count)]).  Buffers of an
    evaluation that was launched with a device-side ray count are sized for the upper bound; the weight-gradient kernel
    reads the count itself and takes the first count * rows-per-count rows."""
    global _WGRAD_MODE
    _WGRAD_MODE = int(field.mma_mode)  # bf16x6: split operands (fp32-equivalent); bf16: rounded operands (reduced precision)
    # W: the PARAMETER width (n_out / k_in of the reductions); the operand rows are [N, field.width] (leading dimension = the
    # kernels' padded width, taken from the tensors' strides)
    L, W = field.mlp_base.num_layers, field.param_width
    skip = field.field_desc().skip_layer
    enc_map, sh_map = field._enc_col_map, field._sh_col_map
    g = acc.g
    cnt = [lv[3] if len(lv) > 3 else None for lv in levels]

    def segs(pick, only_heads=False):
        return [(*pick(lv[0], lv[1]), c) for lv, c in zip(levels, cnt) if lv[2] or not only_heads]

    # the W x W reductions (trunk layers >= 1 and the bottleneck) share one shape: one job-parallel launch per <= 8 of them
    ww, we = [], []  # (W x W) and (W x encoded inputs) jobs
    for l in range(L):
        gw, gb = g[f"mlp_base.layers.{l}.weight"], g[f"mlp_base.layers.{l}.bias"]
        if l == 0:
            we.append((segs(lambda sv, go: (go["dy"][l], sv["enc"])), gw, 0, gb, enc_map))
        elif l == skip:
            we.append((segs(lambda sv, go: (go["dy"][l], sv["enc"])), gw, 0, gb, enc_map))
            ww.append((segs(lambda sv, go: (go["dy"][l], sv["act"][l - 1])), gw, 99, None))
        else:
            ww.append((segs(lambda sv, go: (go["dy"][l], sv["act"][l - 1])), gw, 0, gb))
    ww.append((segs(lambda sv, go: (go["d_bott"], sv["act"][L - 1])), g["field_output_bottleneck.net.weight"], 0,
               g["field_output_bottleneck.net.bias"]))
    _wgrad_jobs(we, W, int(enc_map.numel()))
    for i in range(0, len(ww), _WGRAD_JOBS_MAX):
        _wgrad_jobs(ww[i:i + _WGRAD_JOBS_MAX], W, W)
    _wgrad_multi(segs(lambda sv, go: (go["da_mid"], sv["sh"])), 128, int(sh_map.numel()), g["mlp_mid.layers.0.weight"], 0,
                 g["mlp_mid.layers.0.bias"], sh_map)
    _wgrad_multi(segs(lambda sv, go: (go["da_mid"], sv["bott"])), 128, W, g["mlp_mid.layers.0.weight"], 34, None)
    _wgrad_multi(segs(lambda sv, go: (go["dz_rgb"], sv["hid"])), 3, 128, g["field_output_mid.net.weight"], 0,
                 g["field_output_mid.net.bias"])
    _wgrad_multi(segs(lambda sv, go: (go["dz_heads"], sv["act"][L - 1]), only_heads=True), 16, W, acc.heads_w, 0, acc.heads_b)


def _field_backward(field, rays, eb, level, gin: Dict[str, Optional[Tensor]], need_input: bool, n_dev=None,
                    work: Optional[Dict] = None):
    lib = _abi.load_library()
    o, d, pa = rays
    n, S = eb.shape[0], eb.shape[1] - 1
    gout, gst = _alloc_gout(field, n * S, o.device, need_input)
    gi = FieldGradsIn()
    for k in ("sigma", "color", "pred_normals", "n_dot_d", "roughness", "ray_pn_loss", "ray_ori_loss"):
        setattr(gi, k, ptr(gin.get(k)))
    if gin.get("ray_pn_loss") is not None or gin.get("ray_ori_loss") is not None:
        gi.weights = ptr(gin["weights"])
    fo = ops.field_outputs_struct(level)
    fs = _saved_struct(level["saved"])
    desc = field.field_desc()
    pk = field.packed_weights()
    ops.timed("field_backward_input" if need_input else "field_backward", work or {"points": n * S}, lambda: check(
        lib.rsn_field_backward_frustum(C.byref(desc), ptr(pk), n, ptr(n_dev), S, ptr(o), ptr(d), ptr(pa), ptr(eb),
                                       C.byref(fo), C.byref(fs), C.byref(gi), C.byref(gst), 1 if need_input else 0,
                                       ops._stream())))
    return gout


def _reflect_field_backward(field, rays2, sq, levels, inf_saved, g_bg, n_dev, R: int, Rb: Optional[int] = None):
    """The three backward sweeps of the reflect branch in ONE launch (rsn_field_backward_jobs): the two reflect levels
    (`levels`: [(euclid bins, level dict, upstream colour gradient, S)]) and get_inf_color (upstream: g_bg, the background
    gradient both levels' compositing backward accumulated).  -> ([gout per level], gout of get_inf_color)."""
    lib = _abi.load_library()
    o, d, pa = rays2
    dev = o.device
    Rb = R if Rb is None else Rb  # the levels' capacity (their buffers hold Rb rays); get_inf_color stays at R
    keep = []  # ctypes structures the job table points into
    jobs = (_abi.FieldBwdJob * (len(levels) + 1))()
    gouts, dev_work = [], []
    for k, (eb, lv, g_color, S) in enumerate(levels):
        gout, gst = _alloc_gout(field, Rb * S, dev, True)
        gi = FieldGradsIn()
        gi.color = ptr(g_color)
        fo, fs = ops.field_outputs_struct(lv), _saved_struct(lv["saved"])
        keep += [gst, gi, fo, fs]
        j = jobs[k]
        j.kind, j.n_rays, j.n_dev, j.n_samples, j.need_input_grad = 0, Rb, n_dev.data_ptr(), S, 1
        j.origins, j.directions, j.pixel_area, j.euclid_bins = o.data_ptr(), d.data_ptr(), pa.data_ptr(), eb.data_ptr()
        j.fwd, j.saved, j.gin, j.gout = C.pointer(fo), C.pointer(fs), C.pointer(gi), C.pointer(gst)
        gouts.append(gout)
        dev_work.append((n_dev, S))
    gout_inf, gst_inf = _alloc_gout(field, R, dev, True)
    fs_inf = _saved_struct(inf_saved)
    keep += [gst_inf, fs_inf]
    j = jobs[len(levels)]
    j.kind, j.n_rays, j.n_dev, j.n_samples, j.need_input_grad = 1, R, n_dev.data_ptr(), 1, 1
    j.directions, j.sqradius, j.g_rgb = d.data_ptr(), sq.data_ptr(), g_bg.data_ptr()
    j.saved, j.gout = C.pointer(fs_inf), C.pointer(gst_inf)
    dev_work.append((n_dev, 1))
    desc = field.field_desc()
    pk = field.packed_weights()
    ops.timed("field_backward_input", {"points": 0, "points_dev": dev_work}, lambda: check(
        lib.rsn_field_backward_jobs(C.byref(desc), ptr(pk), len(levels) + 1, jobs, ops._stream())))
    del keep
    return gouts, gout_inf


def _ray_sum(x: Tensor, n: int, S: int, n_dev=None) -> Tensor:
    lib = _abi.load_library()
    out = torch.zeros(n, device=x.device)  # rows behind a device-side count stay zero
    check(lib.rsn_ray_sum(n, ptr(n_dev), S, ptr(x), ptr(out), ops._stream()))
    return out


from .ops import LazyOutputs  # noqa: E402,F401  (kept importable from here)

_LazyAux = LazyOutputs


# ------------------------------------------------------------------------------------------------ reflect-branch capacity
def reflect_capacity(model, R: int, dev) -> int:
    """Rays the two reflect LEVELS of a training step are sized for (their saved activations and layer gradients are ~20 KB
    per sample in fp32, 64 + 64 samples per ray: 10.6 GB at 4096 rays -- a third of the step's memory).

    Default (`model.reflect_capacity is None`): R, whatever the number M of reflected rays turns out to be -- M lives on the
    device and the host never reads it, so only R is a safe bound.  `model.reflect_capacity = "auto"` (opt-in) sizes them for
    the PREVIOUS steps' count instead: every step copies its device-side M into pinned host memory behind the reflect setup
    (asynchronously, no wait); the next step's forward looks at the newest copy whose event has fired (one step old: it fired
    long ago, no stall) and takes cap = min(R, 1.25 M + 128, rounded up to 64).  The level launches clamp the device count to
    their capacity, so a step whose M exceeds it (M jumps by more than 25 % from one step to the next) renders and trains the
    reflect branch on its first `cap` reflected rays only -- detected one step late (`model.reflect_overflows` counts, a
    warning is logged, and the capacity falls back to R for the rest of the run).  That one truncated step is why this is
    opt-in: the default is exact on every step."""
    mode = getattr(model, "reflect_capacity", None)
    st = model.__dict__.setdefault("_reflect_cap_state", {"pending": [], "m_seen": None, "disabled": False, "overflows": 0})
    if mode is None or st["disabled"] or dev.type != "cuda":
        return R
    while st["pending"] and st["pending"][0][0].query():  # counts whose copy has landed: newest wins; check each against its cap
        _ev, host, cap_used, r_used = st["pending"].pop(0)
        m = int(host[0])
        st["m_seen"] = m
        if m > cap_used:
            st["overflows"] += 1
            model.reflect_overflows = st["overflows"]
            st["disabled"] = True
            import warnings

            warnings.warn(f"reflect_capacity='auto': a step reflected {m} rays with buffers for {cap_used} (of {r_used}); its "
                          "reflect branch ran on the first rays only.  Falling back to full-size buffers.")
            return R
    if st["m_seen"] is None:
        return R
    cap = (int(st["m_seen"] * 1.25) + 128 + 63) // 64 * 64
    return max(64, min(R, cap))


def _note_reflect_count(model, nm: Tensor, cap: int, R: int) -> None:
    """Asynchronous copy of this step's device-side reflected-ray count for reflect_capacity() of a later step (opt-in mode only)."""
    if getattr(model, "reflect_capacity", None) is None or not nm.is_cuda:
        return
    st = model.__dict__["_reflect_cap_state"]
    if st["disabled"] or len(st["pending"]) > 8:
        return
    host = torch.empty(1, dtype=nm.dtype, pin_memory=True)
    host.copy_(nm, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(nm.device))
    st["pending"].append((ev, host, cap, R))


# ------------------------------------------------------------------------------------------------ the autograd node
DIFF_KEYS = ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "pred_normals_coarse",
             "pred_normals_fine", "n_dot_d_coarse", "n_dot_d_fine", "roughness")
# differentiable per-ray reductions of the normal losses, handed to get_loss_dict beside the reference's keys
# (LazyOutputs.fused): sum_s w |n - n_pred|^2 and sum_s w max(0, n.d)^2 of the coarse and the fine level
FUSED_KEYS = ("pn_loss_ray_coarse", "pn_loss_ray_fine", "ori_loss_ray_coarse", "ori_loss_ray_fine")


class GetOutputsTrain(torch.autograd.Function):
    """inputs: (model, ray tensors, jitter dict or None, bins dict or None, *field parameters) -> tuple of DIFF_KEYS
    tensors.  The non-differentiable outputs are left on `model._train_aux` by forward.
    jitter: the samplers' uniform draws; bins: {"<level>_spacing", "<level>_euclid"} [n,S+1] replacing the sampler
    output of a level (the samplers' outputs are constants of the graph: PDFSampler detaches, model.py:182,317) --
    tests use both to put this pipeline and the reference / the oracle on identical sample positions."""

    @staticmethod
    def forward(ctx, model, o, d, pa, nears, fars, jitter, bins, *params):
        cfg, fld = model.config, model.field
        R, dev = o.shape[0], o.device
        Sc, Sf = cfg.num_coarse_samples, cfg.num_importance_samples
        Src, Srf = cfg.num_reflect_coarse_samples, cfg.num_reflect_importance_samples
        CLIP = ops.RSN_COMP_CLIP_RGB
        uni, rec = model.sampler_uniform.spec, model.sampler_reciprocal.spec
        jitter = jitter or {}
        bins = bins or {}

        def pad_rows(t, n):
            """[m, ...] -> [n, ...] (m <= n), zero rows behind: device-counted launches ignore them"""
            if t.shape[0] == n:
                return t
            out = torch.zeros((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
            out[: t.shape[0]] = t
            return out

        def jit(name, n, S, rows=None, n_valid=None):
            """the level's uniform draws [n, S+1]: injected ones (validated: the kernel reads n*(S+1) floats) or fresh.
            Reflect levels (launched for n = R rows with a device-side count, n_valid of them real) also accept draws per
            ORIGINAL ray [R, S+1] -- the rows of this pass's reflected rays are used -- or per reflected ray [n_valid, S+1]."""
            t = jitter.get(name)
            if t is None:
                # (a level sized below R -- reflect_capacity -- takes the first rows of the same R-row draw: the random stream of
                # a step does not depend on the capacity)
                return torch.rand(n, S + 1, device=dev) if (rows is None or n == R) else torch.rand(R, S + 1, device=dev)[:n]
            t = ops._f32c(t.to(dev))
            if rows is not None and t.shape[0] == R and n_valid != R:
                t = t[rows[:n_valid].long()].contiguous()
            if n_valid is not None:
                if tuple(t.shape) != (n_valid, S + 1):
                    raise ValueError(f"jitter[{name!r}] has shape {tuple(t.shape)}, expected {(n_valid, S + 1)}")
                return pad_rows(t, n)
            if tuple(t.shape) != (n, S + 1):
                raise ValueError(f"jitter[{name!r}] has shape {tuple(t.shape)}, expected {(n, S + 1)}")
            return t

        def level_bins(name, n, S, sample, n_valid=None):
            """the level's (spacing, euclidean) bins: injected ones if given, else the sampler launch `sample()`"""
            if name + "_euclid" in bins:
                m = n if n_valid is None else n_valid
                sb, eb = (pad_rows(ops._f32c(bins[name + k].to(dev).reshape(m, S + 1)), n) for k in ("_spacing", "_euclid"))
                return sb, eb
            return sample()

        # A. coarse, B. fine (+ per-ray surface attributes)
        sb_c, eb_c = level_bins("coarse", R, Sc, lambda: ops.sample_spaced(R, None, Sc, uni.spacing, uni.tan, nears, fars,
                                                                           jit("coarse", R, Sc)))
        lc = fld.evaluate_frustums_train(o, d, pa, eb_c, want_normals=True)
        cc = ops.composite(R, None, Sc, 1, CLIP, lc["sigma"], eb_c, lc["color"], level=lc, ray_losses=True)
        sb_f, eb_f = level_bins("fine", R, Sf, lambda: ops.sample_pdf(
            R, None, Sc, Sf, uni.spacing, uni.tan, model.sampler_pdf.histogram_padding, nears, fars, cc["weights"], sb_c,
            jit("fine", R, Sf)))
        lf = fld.evaluate_frustums_train(o, d, pa, eb_f, want_normals=True)
        cf = ops.composite(R, None, Sf, 1, CLIP, lf["sigma"], eb_f, lf["color"], level=lf, surface=True, ray_losses=True)
        rs = ops.reflect_setup(o, d, cf["accumulation"], cf["depth"], cf["normals"], cf["roughness"], float(model.far))
        # C.-F. the reflect branch runs on the M <= R rays behind the mask.  Like the eval path (model.get_outputs) every
        # launch is sized for R rays and takes the count from device memory (rs["n_masked"]).  The host never reads M: the
        # [M, 1] output is materialised on demand (LazyOutputs) and the weight-gradient kernel takes its segment lengths
        # from the same device word, so the host runs ahead of the GPU across the whole step (and into the next one).
        nm = rs["n_masked"]
        lib = _abi.load_library()
        W, L = fld.width, fld.mlp_base.num_layers
        # injected draws / bins come in the reference's shapes ([M, S + 1]): that (test) mode reads M first.  The
        # production path never does: M stays on the device for the whole step -- forward, backward, weight gradients.
        M = int(nm.item()) if (jitter or bins) else None
        # Rb: rays the reflect LEVELS are sized / launched for (R unless model.reflect_capacity = "auto": see reflect_capacity);
        # the per-ray quantities of the branch (secondary rays, get_inf_color, background) stay at R
        Rb = R if M is not None else reflect_capacity(model, R, dev)
        if M is None:
            _note_reflect_count(model, nm, Rb, R)
        o2, d2, pa2, sq = rs["origins2"], rs["directions2"], rs["pixel_area2"], rs["sqradius"]
        near2, far2 = rs["nears2"], rs["fars2"]
        o2b, d2b, pa2b, near2b, far2b = o2[:Rb], d2[:Rb], pa2[:Rb], near2[:Rb], far2[:Rb]
        # reflect-coarse level and get_inf_color (the composites' background, model.py:290): both depend only on the
        # secondary rays and run as two jobs of ONE launch (rsn_field_forward_train_jobs)
        # launches sized for R rays do the work of M: the timer resolves the device-side count when it reads its events
        work_rc, work_rf = ({"points": 0, "points_dev": [(nm, k)]} for k in (Src + 1, Srf))
        sb_rc, eb_rc = level_bins("reflect_coarse", Rb, Src, lambda: ops.sample_spaced(
            Rb, nm, Src, rec.spacing, rec.tan, near2b, far2b, jit("reflect_coarse", Rb, Src, rs["ray_index"], M)), M)
        lrc, bg, inf_saved = fld.evaluate_reflect_train(o2b, d2b, pa2b, eb_rc, nm, sq, work=work_rc, inf_directions=d2)
        crc = ops.composite(Rb, nm, Src, 2, 0, lrc["sigma"], eb_rc, lrc["color"], bg_rgb=bg, want_depth=False)
        ops.reflect_combine(Rb, nm, rs["ray_index"], cf["diff"], cf["tint"], crc["rgb"], rs["reflect_coarse"])
        sb_rf, eb_rf = level_bins("reflect_fine", Rb, Srf, lambda: ops.sample_pdf(
            Rb, nm, Src, Srf, rec.spacing, rec.tan, model.sampler_reflect_pdf.histogram_padding, near2b, far2b,
            crc["weights"], sb_rc, jit("reflect_fine", Rb, Srf, rs["ray_index"], M)), M)
        lrf = fld.evaluate_frustums_train(o2b, d2b, pa2b, eb_rf, n_dev=nm, want_normals=False, work=work_rf)
        crf = ops.composite(Rb, nm, Srf, 2, 0, lrf["sigma"], eb_rf, lrf["color"], bg_rgb=bg)
        ops.reflect_combine(Rb, nm, rs["ray_index"], cf["diff"], cf["tint"], crf["rgb"], rs["reflect_fine"])
        mask_bool = rs["mask"].bool()
        model._last_n_masked_dev = nm  # device-side; `model._last_num_reflected` reads it on demand (a host sync)

        aux = _LazyAux({
            "accumulation_coarse": cc["accumulation"].unsqueeze(-1), "accumulation_fine": cf["accumulation"].unsqueeze(-1),
            "depth_coarse": cc["depth"].unsqueeze(-1), "depth_fine": cf["depth"].unsqueeze(-1),
            "weights_coarse": cc["weights"].unsqueeze(-1), "weights_fine": cf["weights"].unsqueeze(-1),
            "normals_coarse": lc["normals"], "normals_fine": lf["normals"],
            "diff": cf["diff"], "tint": cf["tint"], "mask": mask_bool,
        })
        # [M, 1] (reference model.py:341, present only when M > 0): its shape needs M on the host, so it is materialised on
        # first access -- the training step itself never asks for it
        if M is None:
            aux.lazy["depth_reflect_fine"] = (nm, crf["depth"])
        elif M > 0:  # the count is on the host already (injected draws / bins)
            aux["depth_reflect_fine"] = crf["depth"][:M].unsqueeze(-1)
        st = dict(R=R, Rb=Rb, eb_c=eb_c, eb_f=eb_f, lc=lc, lf=lf, cc=cc, cf=cf, rs=rs, rays=(o, d, pa),
                  rays2=(o2, d2, pa2), sq=sq, bg=bg, inf_saved=inf_saved, eb_rc=eb_rc, eb_rf=eb_rf, lrc=lrc, lrf=lrf,
                  crc=crc, crf=crf)
        if getattr(model, "_keep_train_state", False):  # test hook: the sample positions this pass evaluated
            if M is None:
                M = int(nm.item())
            exp = dict(st, M=M, sb_c=sb_c, sb_f=sb_f)
            if M > 0:  # the reflect levels' buffers are sized for R rays: the hook shows their M real rows
                def trim(lv, n):
                    sv = {k: (v[:, :n] if k in ("act", "relu_bits") else v[:n]) for k, v in lv["saved"].items()}
                    return dict(lv, saved=sv)
                exp.update(sb_rc=sb_rc[:M], sb_rf=sb_rf[:M], eb_rc=eb_rc[:M], eb_rf=eb_rf[:M],
                           lrc=trim(lrc, M * Src), lrf=trim(lrf, M * Srf))
            model._train_state = exp
        ctx.st = st
        ctx.model = model
        ctx.n_params = len(params)
        model._train_aux = aux
        outs = (cc["rgb"], cf["rgb"], rs["reflect_coarse"], rs["reflect_fine"], lc["pred_normals"], lf["pred_normals"],
                lc["n_dot_d"].unsqueeze(-1), lf["n_dot_d"].unsqueeze(-1), cf["roughness"].unsqueeze(-1),
                cc["pn_loss_ray"], cf["pn_loss_ray"], cc["ori_loss_ray"], cf["ori_loss_ray"])
        return outs

    @staticmethod
    def backward(ctx, g_rgb_c, g_rgb_f, g_refl_c, g_refl_f, g_pn_c, g_pn_f, g_ndd_c, g_ndd_f, g_rough,
                 g_pnr_c=None, g_pnr_f=None, g_orr_c=None, g_orr_f=None):
        st, model = ctx.st, ctx.model
        fld = model.field
        lib = _abi.load_library()
        R = st["R"]
        dev = st["rays"][0].device
        CLIP = ops.RSN_COMP_CLIP_RGB
        cfg = model.config
        Sc, Sf = cfg.num_coarse_samples, cfg.num_importance_samples
        Src, Srf = cfg.num_reflect_coarse_samples, cfg.num_reflect_importance_samples
        z = lambda g, *shape: ops._f32c(g) if g is not None else torch.zeros(*shape, device=dev)  # noqa: E731
        g_rgb_c, g_rgb_f = z(g_rgb_c, R, 3), z(g_rgb_f, R, 3)
        g_refl_c, g_refl_f = z(g_refl_c, R, 3), z(g_refl_f, R, 3)
        acc = _GradAcc(fld)
        pending = []  # (saved, gout, with_heads[, device count]) of every field evaluation: weight gradients at the end
        cf, rs = st["cf"], st["rs"]
        g_rough_ray = z(g_rough, R, 1).reshape(R).clone()

        # The reflect branch: everything is sized for R rays and launched with the device-side count M = rs["n_masked"]
        # (see forward); rows M.. of the per-ray buffers are zero-filled so that whole-buffer sums are sums over M rows, and
        # the weight-gradient kernel takes the first M (x samples) rows of each reflect evaluation by the same count.  With
        # M = 0 every launch below is a no-op on the device: no host-side early-out, no host read.
        nm = rs["n_masked"]
        Rb = st["Rb"]  # rays the two reflect levels were sized for (R by default; forward)
        g_bg = torch.zeros(R, 3, device=dev)
        g_pa2 = torch.zeros(R, device=dev)
        jobs = []
        for eb, lv, cp, g_out, S in ((st["eb_rf"], st["lrf"], st["crf"], g_refl_f, Srf),
                                     (st["eb_rc"], st["lrc"], st["crc"], g_refl_c, Src)):
            g_comp = torch.empty(Rb, 3, device=dev)
            check(lib.rsn_reflect_combine_backward(Rb, ptr(nm), ptr(rs["ray_index"]), ptr(cf["diff"]), ptr(cf["tint"]),
                                                   ptr(cp["rgb"]), ptr(g_out), ptr(g_comp), ops._stream()))
            cb = _composite_backward(Rb, S, 2, 0, 1, lv, eb, cp["weights"], g_comp, bg=st["bg"], want_sigma=False,
                                     want_bg=True, n_dev=nm)
            g_bg[:Rb] += cb["g_bg"]
            jobs.append((eb, lv, cb["g_color"], S))
        # the sweeps of both levels and of get_inf_color (whose upstream gradient g_bg is complete now) in ONE launch:
        # their tiles share the persistent workgroups' rounds (rsn_field_backward_jobs)
        gouts, gout = _reflect_field_backward(fld, st["rays2"], st["sq"], jobs, st["inf_saved"], g_bg, nm, R, Rb)
        for (eb, lv, _g, S), go in zip(jobs, gouts):
            g_pa2[:Rb] += _ray_sum(go["d_input"], Rb, S, nm)
            pending.append((lv["saved"], go, True, (nm, S)))
        pending.append((st["inf_saved"], gout, False, (nm, 1)))
        g_r = torch.empty(R, device=dev)  # the kernel zero-fills the rays that are not reflected
        check(lib.rsn_reflect_backward(R, ptr(nm), ptr(rs["ray_index"]), ptr(rs["n_dot_d"]), ptr(cf["roughness"]),
                                       ptr(gout["d_input"]), ptr(g_pa2), ptr(g_r), ops._stream()))
        g_rough_ray += g_r
        if getattr(model, "weight_grad_groups", 1) == 2 and not getattr(model, "_keep_train_state", False):
            # Opt-in memory bound: the reflect branch's weight gradients are reduced NOW (six more reduction launches per step, each
            # with its own flush) and its saved rows and sweep outputs -- a third of the step's memory -- are released before the
            # primary levels' sweep outputs are allocated (the caching allocator hands them the same blocks; one stream: no race).
            _weight_grads(fld, pending, acc)
            pending = []
            for lv in (st["lrf"], st["lrc"]):
                lv.pop("saved", None)
            st["inf_saved"] = None
            del gouts, gout, jobs, go, lv

        # fine primary level (+ the live accumulation of the non-reflected rays' default reflect colour)
        lf = st["lf"]
        g_acc = torch.empty(R, device=dev)
        check(lib.rsn_reflect_default_backward(R, ptr(rs["mask"]), ptr(g_refl_c), ptr(g_refl_f), ptr(g_acc),
                                               ops._stream()))
        cb = _composite_backward(R, Sf, 1, CLIP, 0, lf, st["eb_f"], cf["weights"], g_rgb_f, g_rough=g_rough_ray,
                                 rough_samples=lf["roughness"], g_acc=g_acc)
        gin = {"sigma": cb["g_sigma"], "color": cb["g_color"], "roughness": cb["g_rough"],
               "pred_normals": ops._f32c(g_pn_f) if g_pn_f is not None else None,
               "n_dot_d": ops._f32c(g_ndd_f.reshape(R, Sf)) if g_ndd_f is not None else None,
               "ray_pn_loss": ops._f32c(g_pnr_f) if g_pnr_f is not None else None,
               "ray_ori_loss": ops._f32c(g_orr_f) if g_orr_f is not None else None, "weights": cf["weights"]}
        gout = _field_backward(fld, st["rays"], st["eb_f"], lf, gin, need_input=False)
        pending.append((lf["saved"], gout, True))
        # coarse primary level
        lc, cc = st["lc"], st["cc"]
        cb = _composite_backward(R, Sc, 1, CLIP, 0, lc, st["eb_c"], cc["weights"], g_rgb_c)
        gin = {"sigma": cb["g_sigma"], "color": cb["g_color"],
               "pred_normals": ops._f32c(g_pn_c) if g_pn_c is not None else None,
               "n_dot_d": ops._f32c(g_ndd_c.reshape(R, Sc)) if g_ndd_c is not None else None,
               "ray_pn_loss": ops._f32c(g_pnr_c) if g_pnr_c is not None else None,
               "ray_ori_loss": ops._f32c(g_orr_c) if g_orr_c is not None else None, "weights": cc["weights"]}
        gout = _field_backward(fld, st["rays"], st["eb_c"], lc, gin, need_input=False)
        pending.append((lc["saved"], gout, True))
        _weight_grads(fld, pending, acc)
        del pending, gout

        final = acc.finish()
        grads = [final.get(name) for name, _ in fld.named_parameters()]  # field_output_low: unused -> None
        ctx.st = None
        return (None,) * 8 + tuple(grads)
