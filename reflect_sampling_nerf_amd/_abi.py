"""ctypes binding of include/rsn.h (librsn_hip.so).

There is NO fallback: if the library is missing or does not export the ABI of include/rsn.h the
import of the product path raises.  Tensors are passed as raw device pointers (`tensor.data_ptr()`),
the stream as `torch.cuda.current_stream().cuda_stream`; the library never allocates or synchronises.
"""
import ctypes as C
import os

from ._build import LIB_PATH

RSN_ABI_VERSION = 16
RSN_ABI_DIAG_FLAG = 0x10000  # rsn_abi_version() of a -DRSN_DIAG_BUILD library (csrc/rsn_common.h)
RSN_MAX_TRUNK_LAYERS = 16
RSN_NUM_FREQS = 16
RSN_SPACING_UNIFORM = 0
RSN_SPACING_RECIPROCAL = 1
RSN_MMA_F32, RSN_MMA_BF16X6, RSN_MMA_BF16X3, RSN_MMA_BF16 = 0, 1, 2, 3

_fp = C.c_void_p  # device float*


class FieldDesc(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32),
        ("width", C.c_int32),
        ("skip_layer", C.c_int32),
        ("mid_width", C.c_int32),
        ("density_bias", C.c_float),
        ("freqs", C.c_float * RSN_NUM_FREQS),
        ("mma_mode", C.c_int32),
        ("param_width", C.c_int32),
    ]


class FieldParams(C.Structure):
    _fields_ = [
        ("trunk_w", _fp * RSN_MAX_TRUNK_LAYERS),
        ("trunk_b", _fp * RSN_MAX_TRUNK_LAYERS),
        ("density_w", _fp), ("density_b", _fp),
        ("normals_w", _fp), ("normals_b", _fp),
        ("roughness_w", _fp), ("roughness_b", _fp),
        ("diff_w", _fp), ("diff_b", _fp),
        ("tint_w", _fp), ("tint_b", _fp),
        ("bottleneck_w", _fp), ("bottleneck_b", _fp),
        ("mid_w", _fp), ("mid_b", _fp),
        ("rgb_w", _fp), ("rgb_b", _fp),
    ]


class FieldOutputs(C.Structure):
    _fields_ = [(n, _fp) for n in
                ("sigma", "color", "pred_normals", "n_dot_d", "diff", "tint", "roughness", "raw_density",
                 "raw_roughness")]


class FieldSaved(C.Structure):
    _fields_ = [(n, _fp) for n in ("enc", "act", "bott", "sh", "hid", "heads", "normals", "relu_bits")]


class FieldJob(C.Structure):
    """rsn_field_job: one evaluation of a multi-evaluation training launch (rsn_field_forward_train_jobs)."""
    _fields_ = [("kind", C.c_int32), ("n_rays", C.c_int32), ("n_dev", C.c_void_p), ("n_samples", C.c_int32),
                ("origins", _fp), ("directions", _fp), ("pixel_area", _fp), ("euclid_bins", _fp), ("sqradius", _fp),
                ("out_rgb", _fp), ("out", C.POINTER(FieldOutputs)), ("saved", C.POINTER(FieldSaved))]


class FieldGradsIn(C.Structure):
    _fields_ = [(n, _fp) for n in ("sigma", "color", "pred_normals", "n_dot_d", "roughness", "ray_pn_loss", "ray_ori_loss",
                                   "weights")]


class FieldGradsOut(C.Structure):
    _fields_ = [(n, _fp) for n in ("dz_rgb", "da_mid", "d_bott", "dz_heads", "dy", "d_input")]


class FieldBwdJob(C.Structure):
    """rsn_field_bwd_job: one evaluation of a multi-evaluation backward launch (rsn_field_backward_jobs)."""
    _fields_ = [("kind", C.c_int32), ("n_rays", C.c_int32), ("n_dev", C.c_void_p), ("n_samples", C.c_int32),
                ("need_input_grad", C.c_int32), ("origins", _fp), ("directions", _fp), ("pixel_area", _fp),
                ("euclid_bins", _fp), ("sqradius", _fp), ("g_rgb", _fp), ("fwd", C.POINTER(FieldOutputs)),
                ("saved", C.POINTER(FieldSaved)), ("gin", C.POINTER(FieldGradsIn)), ("gout", C.POINTER(FieldGradsOut))]


class WGradJob(C.Structure):
    """rsn_wgrad_job: one reduction of a job-parallel weight-gradient launch (rsn_weight_grad_jobs)."""
    _fields_ = [("dy", C.POINTER(C.c_void_p)), ("x", C.POINTER(C.c_void_p)), ("col_map", C.c_void_p), ("dw", C.c_void_p),
                ("ld_dw", C.c_int32), ("db", C.c_void_p)]


class CompositeBwdIO(C.Structure):
    _fields_ = [(n, _fp) for n in ("sigma", "euclid_bins", "color", "bg_rgb", "roughness", "weights", "g_rgb",
                                   "g_roughness", "g_accumulation", "g_sigma", "g_color", "g_roughness_sample",
                                   "g_bg")]


class CompositeIO(C.Structure):
    _fields_ = [(n, _fp) for n in
                ("sigma", "euclid_bins", "color", "bg_rgb", "diff", "tint", "pred_normals", "roughness",
                 "weights", "rgb", "accumulation", "depth", "diff_out", "tint_out", "normals_out", "roughness_out",
                 "normals", "n_dot_d", "pn_loss_ray", "ori_loss_ray")]


class ReflectIO(C.Structure):
    _fields_ = [(n, _fp) for n in
                ("origins", "directions", "accumulation", "depth", "pred_normals", "roughness", "mask", "n_masked",
                 "ray_index", "n_dot_d", "origins2", "directions2", "sqradius", "pixel_area2", "nears2", "fars2",
                 "reflect_coarse", "reflect_fine", "workspace")]


_SIGNATURES = {
    "rsn_abi_version": (C.c_int, []),
    "rsn_last_error": (C.c_char_p, []),
    "rsn_packed_weights_bytes": (C.c_size_t, [C.POINTER(FieldDesc)]),
    "rsn_pack_weights": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), _fp, C.c_size_t, C.c_void_p]),
    "rsn_pack_table_bytes": (C.c_size_t, []),
    "rsn_pack_weights_table": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), _fp, C.c_size_t, _fp, C.c_size_t,
                                         C.c_int32, C.c_void_p]),
    "rsn_sample_spaced": (C.c_int, [C.c_int32, _fp, C.c_int32, C.c_int32, C.c_float, _fp, _fp, _fp, _fp, _fp,
                                    C.c_void_p]),
    "rsn_sample_pdf": (C.c_int, [C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _fp, _fp,
                                 _fp, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_field_forward_frustum": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, C.c_int32, _fp, _fp, _fp,
                                            _fp, C.POINTER(FieldOutputs), C.c_void_p]),
    "rsn_field_forward_frustum_train": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, C.c_int32, _fp, _fp,
                                                  _fp, _fp, C.POINTER(FieldOutputs), C.POINTER(FieldSaved),
                                                  C.c_void_p]),
    "rsn_field_backward_frustum": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, C.c_int32, _fp, _fp, _fp, _fp,
                                             C.POINTER(FieldOutputs), C.POINTER(FieldSaved),
                                             C.POINTER(FieldGradsIn), C.POINTER(FieldGradsOut), C.c_int32,
                                             C.c_void_p]),
    "rsn_field_forward_inf_train": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp, _fp,
                                              C.POINTER(FieldSaved), C.c_void_p]),
    "rsn_field_backward_inf": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp, C.POINTER(FieldSaved),
                                         _fp, C.POINTER(FieldGradsOut), C.c_int32, C.c_void_p]),
    "rsn_composite_backward": (C.c_int, [C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(CompositeBwdIO), C.c_void_p]),
    "rsn_weight_grad": (C.c_int, [C.c_int64, _fp, C.c_int32, C.c_int32, _fp, C.c_int32, C.c_int32, _fp, _fp, C.c_int32,
                                  _fp, C.c_void_p]),
    "rsn_sh34_encode": (C.c_int, [C.c_int64, _fp, _fp, _fp, C.c_void_p]),
    "rsn_ipe_encode": (C.c_int, [C.c_int64, _fp, _fp, C.POINTER(C.c_float), _fp, C.c_void_p]),
    "rsn_weight_grad_multi": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                        C.POINTER(C.c_void_p), C.c_int32, C.c_int32, _fp, C.c_void_p, C.c_int32, _fp,
                                        C.c_void_p]),
    "rsn_weight_grad_multi_mode": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                             C.POINTER(C.c_void_p), C.c_int32, C.c_int32, _fp, C.c_void_p, C.c_int32, _fp,
                                             C.c_int32, C.c_void_p]),
    "rsn_field_forward_train_jobs": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, C.POINTER(FieldJob), C.c_void_p]),
    "rsn_field_backward_jobs": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, C.POINTER(FieldBwdJob), C.c_void_p]),
    "rsn_weight_grad_multi_dev": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                            C.c_int32, _fp, C.c_void_p, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_void_p]),
    "rsn_weight_grad_jobs": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32,
                                       C.POINTER(WGradJob), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]),
    "rsn_loss_forward_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, C.POINTER(_fp), C.POINTER(_fp),
                                            C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_float), _fp,
                                            C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.c_void_p]),
    "rsn_loss_rays_forward": (C.c_int, [C.c_int32, _fp, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_float),
                                        _fp, C.POINTER(_fp), C.c_void_p]),
    "rsn_loss_rays_backward": (C.c_int, [C.c_int32, _fp, C.POINTER(C.c_float), C.POINTER(_fp), C.POINTER(_fp),
                                         C.POINTER(_fp), C.c_void_p]),
    "rsn_loss_scale_grads": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp),
                                       C.c_void_p]),
    "rsn_radam_step": (C.c_int, [C.c_int32, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp),
                                 C.POINTER(C.c_int32), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_void_p]),
    "rsn_colsum": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _fp, _fp, C.c_int32, C.c_void_p]),
    "rsn_ray_sum": (C.c_int, [C.c_int32, _fp, C.c_int32, _fp, _fp, C.c_void_p]),
    "rsn_reflect_backward": (C.c_int, [C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_reflect_default_backward": (C.c_int, [C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_reflect_combine_backward": (C.c_int, [C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_field_forward_inf": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_field_forward_embedding": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp,
                                              C.POINTER(FieldOutputs), C.c_void_p]),
    "rsn_gaussians": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_contract": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_reflection": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_field_forward_gaussians": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp,
                                              C.POINTER(FieldOutputs), _fp, C.c_void_p]),
    "rsn_field_forward_gaussians_train": (C.c_int, [C.POINTER(FieldDesc), _fp, C.c_int32, _fp, _fp, _fp,
                                                    C.POINTER(FieldOutputs), _fp, C.POINTER(FieldSaved), C.c_void_p]),
    "rsn_composite": (C.c_int, [C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CompositeIO),
                                C.c_void_p]),
    "rsn_reflect_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "rsn_reflect_setup": (C.c_int, [C.c_int32, C.c_float, C.POINTER(ReflectIO), C.c_void_p]),
    "rsn_reflect_combine": (C.c_int, [C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]),
    "rsn_train_saved_layout": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class RsnError(RuntimeError):
    pass


_lib = None


def load_library(path: str = LIB_PATH):
    """Load librsn_hip.so and bind every symbol of include/rsn.h.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RSN_LIBRARY", path)  # tools/: an experimental build of the same ABI (tools/_variant.py)
    if not os.path.exists(path):
        raise RsnError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or reflect_sampling_nerf_amd/_build.py). There is no CPU/eager fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RsnError(f"{path} does not export {name} (include/rsn.h)") from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.rsn_abi_version()
    diag, ver = bool(ver & RSN_ABI_DIAG_FLAG), ver & ~RSN_ABI_DIAG_FLAG
    if ver != RSN_ABI_VERSION:
        raise RsnError(f"ABI version mismatch: library {ver}, binding {RSN_ABI_VERSION}")
    if diag and os.path.abspath(path) == os.path.abspath(LIB_PATH):
        # -DRSN_DIAG_BUILD libraries (timing ablations with wrong results, phase counters) are tools-only: they are
        # loaded by explicit path (tools/_variant.py, RSN_LIBRARY), never from the product location
        raise RsnError(f"{path} is a diagnostic build (RSN_DIAG_BUILD): rebuild it with reflect_sampling_nerf_amd/_build.py")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load_library().rsn_last_error()
        raise RsnError(f"librsn_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
