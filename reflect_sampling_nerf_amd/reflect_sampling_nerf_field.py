"""ReflectSamplingNeRFNerfField on MI355X: same constructor knobs, parameter names and method names as
the reference Field (reflect_sampling_nerf_field.py:28-207), evaluated by the fused HIP kernels of
librsn_hip.so instead of eager torch ops.

Parameter layout / names (state_dict interchange with the reference):
    mlp_base.layers.{i}.{weight,bias}, field_output_density.net.*, field_output_low.net.* (unused, kept),
    field_output_bottleneck.net.*, mlp_mid.layers.0.*, field_output_mid.net.*, field_output_normals.net.*,
    field_output_roughness.net.*, field_output_diff.net.*, field_output_tint.net.*
Modules are created in the reference's order so that the same torch seed gives the same random init.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _abi, ops
from ._abi import FieldDesc, FieldParams, FieldSaved, check, ptr
from .nerfstudio_compat import Field
from .reflect_sampling_nerf_components import IntegratedSHEncoding, NeRFEncoding


class _MLP(nn.Module):
    """Parameter container shaped like nerfstudio's MLP (`.layers` ModuleList of nn.Linear)."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, skip_connections: Tuple[int, ...] = ()):
        super().__init__()
        self.in_dim, self.num_layers, self.layer_width = in_dim, num_layers, layer_width
        self.out_dim = layer_width
        skips = set(skip_connections or ())
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, layer_width))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    assert i not in skips, "Skip connection at layer 0 doesn't make sense."
                    layers.append(nn.Linear(in_dim, layer_width))
                elif i in skips:
                    layers.append(nn.Linear(layer_width + in_dim, layer_width))
                else:
                    layers.append(nn.Linear(layer_width, layer_width))
            layers.append(nn.Linear(layer_width, layer_width))
        self.layers = nn.ModuleList(layers)

    def get_out_dim(self) -> int:
        return self.out_dim


class _Head(nn.Module):
    """Parameter container shaped like nerfstudio's FieldHead (`.net` nn.Linear)."""

    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.net = nn.Linear(in_dim, out_dim)


class ReflectSamplingNeRFNerfField(Field):
    def __init__(
        self,
        position_encoding=None,
        direction_encoding=None,
        base_mlp_num_layers: int = 8,
        base_mlp_layer_width: int = 256,
        skip_connections: Tuple[int, ...] = (4,),
        head_mlp_num_layers: int = 1,
        head_mlp_layer_width: int = 128,
        spatial_distortion=None,
        density_bias: float = 0.5,
        roughness_bias: float = -1.0,
    ) -> None:
        super().__init__()
        self.position_encoding = position_encoding if position_encoding is not None else NeRFEncoding()
        self.direction_encoding = direction_encoding if direction_encoding is not None else IntegratedSHEncoding()
        pe = self.position_encoding
        if (getattr(pe, "num_frequencies", None) != 16 or pe.get_out_dim() != 99
                or float(getattr(pe, "min_freq", 0.0)) != 0.0):
            raise NotImplementedError("the HIP field kernel fuses NeRFEncoding(3, 16, 0.0, max, include_input=True)")
        if self.direction_encoding.get_out_dim() != 34:
            raise NotImplementedError("the HIP field kernel fuses the 34-channel IntegratedSHEncoding")
        if head_mlp_num_layers != 1:
            raise NotImplementedError("head_mlp_num_layers != 1 is not fused (reference default: 1)")
        # field.py:49,92-94: applied to the Gaussians in get_blob.  The reference model passes None (model.py:103-106); a field
        # built with one serves the granular API (get_blob -> contract -> get_density -> heads); the fused level kernels build
        # their Gaussians in-kernel and refuse it (_no_distortion).
        self.spatial_distortion = spatial_distortion
        self.skip_connections = tuple(skip_connections or ())
        if len([s for s in self.skip_connections if 0 < s <= base_mlp_num_layers - 1]) > 1:
            raise NotImplementedError("at most one live skip connection is fused")

        W = base_mlp_layer_width
        # creation order == reference field.py:54-86
        self.mlp_base = _MLP(99, base_mlp_num_layers, W, self.skip_connections)
        self.field_output_density = _Head(W, 1)
        self.density_bias = density_bias
        self.field_output_low = _Head(W, 3)  # never evaluated by the model (field.py:67), kept for state_dict parity
        self.field_output_bottleneck = _Head(W, W)
        self.mlp_mid = _MLP(34 + W, head_mlp_num_layers, head_mlp_layer_width)
        self.field_output_mid = _Head(head_mlp_layer_width, 3)
        self.field_output_normals = _Head(W, 3)
        self.field_output_roughness = _Head(W, 1)
        self.roughness_bias = roughness_bias  # stored, never applied (field.py:82)
        self.field_output_diff = _Head(W, 3)
        self.field_output_tint = _Head(W, 3)

        self._packed: Optional[Tensor] = None
        self._packed_key = None
        self._pack_table: Optional[Tensor] = None
        self._pack_table_key = None
        self._desc: Optional[FieldDesc] = None
        self.mma_mode = _abi.RSN_MMA_F32

    MMA_MODES = {"f32": _abi.RSN_MMA_F32, "bf16x6": _abi.RSN_MMA_BF16X6, "bf16x3": _abi.RSN_MMA_BF16X3,
                 "bf16": _abi.RSN_MMA_BF16}

    def set_mma_mode(self, mode: str) -> None:
        """Arithmetic of the dense GEMMs in the eval field kernel: "f32" (exact fp32 MFMA), "bf16x6" (fp32
        emulation by 3-way bf16 splits, fp32-equivalent), "bf16x3" (2-way split, reduced precision, opt-in) or "bf16"
        (plain bf16 operands, fp32 accumulate: BASELINE configs[3])."""
        self.mma_mode = self.MMA_MODES[mode]
        self._desc = None
        self._packed_key = None  # rsn_pack_weights writes the split-bf16 segments only for the modes that read them

    # ------------------------------------------------------------------ C-ABI plumbing
    @property
    def param_width(self) -> int:
        """base_mlp_layer_width: the width of the parameter tensors (reference field.py:41)."""
        return self.mlp_base.layer_width

    @property
    def width(self) -> int:
        """The width the kernels run at: the next of 64 / 128 / 256 at or above base_mlp_layer_width.  Units beyond
        param_width are zero-padded by rsn_pack_weights (zero weights, zero biases: their activations and gradients are exact
        zeros); every wide buffer of the C ABI ([N, W] rows) has this many columns."""
        pw = self.mlp_base.layer_width
        for w in (64, 128, 256):
            if pw <= w:
                return w
        raise NotImplementedError(f"base_mlp_layer_width={pw}: the fused kernels hold at most 256 units per layer")

    def field_desc(self) -> FieldDesc:
        if self._desc is None:
            L = self.mlp_base.num_layers
            live = [s for s in self.skip_connections if 0 < s <= L - 1]
            d = FieldDesc()
            d.num_layers = L
            d.width = self.width
            d.param_width = self.param_width
            d.skip_layer = live[0] if live else -1
            d.mid_width = self.mlp_mid.layer_width
            d.density_bias = float(self.density_bias)
            d.mma_mode = int(self.mma_mode)
            pe = self.position_encoding
            freqs = 2 ** torch.linspace(float(pe.min_freq), float(pe.max_freq), int(pe.num_frequencies))
            for i in range(16):
                d.freqs[i] = float(freqs[i])
            self._desc = d
        return self._desc

    def _param_struct(self) -> FieldParams:
        p = FieldParams()
        for i, layer in enumerate(self.mlp_base.layers):
            p.trunk_w[i] = layer.weight.data_ptr()
            p.trunk_b[i] = layer.bias.data_ptr()
        for name, mod in (("density", self.field_output_density.net), ("normals", self.field_output_normals.net),
                          ("roughness", self.field_output_roughness.net), ("diff", self.field_output_diff.net),
                          ("tint", self.field_output_tint.net), ("bottleneck", self.field_output_bottleneck.net),
                          ("mid", self.mlp_mid.layers[0]), ("rgb", self.field_output_mid.net)):
            setattr(p, name + "_w", mod.weight.data_ptr())
            setattr(p, name + "_b", mod.bias.data_ptr())
        return p

    def packed_weights(self) -> Tensor:
        """Parameters re-laid into MFMA fragment order (rsn_pack_weights); re-packed only when a parameter
        changed (torch bumps Tensor._version on every in-place optimiser update)."""
        params = [p for p in self.parameters()]
        dev = params[0].device
        if dev.type != "cuda":
            raise _abi.RsnError("the Field must live on a cuda (ROCm) device: there is no CPU fallback")
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _abi.RsnError("Field parameters must be contiguous fp32")
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is None or self._packed_key != key or self._packed.device != dev:
            lib = _abi.load_library()
            desc = self.field_desc()
            nbytes = lib.rsn_packed_weights_bytes(C.byref(desc))
            if nbytes == 0:
                check(-1)
            if self._packed is None or self._packed.numel() * 4 != nbytes or self._packed.device != dev:
                self._packed = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
            ps = self._param_struct()
            # one launch for all segments; the job table is re-uploaded only when a pointer or the shape changed
            tbytes = lib.rsn_pack_table_bytes()
            if self._pack_table is None or self._pack_table.device != dev:
                self._pack_table = torch.empty(tbytes, device=dev, dtype=torch.uint8)
                self._pack_table_key = None
            tkey = (tuple(p.data_ptr() for p in params), self._packed.data_ptr(), int(self.mma_mode))
            rebuild = 1 if tkey != self._pack_table_key else 0
            check(lib.rsn_pack_weights_table(C.byref(desc), C.byref(ps), ptr(self._packed), nbytes,
                                             ptr(self._pack_table), tbytes, rebuild, ops._stream()))
            self._pack_table_key = tkey
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ fused evaluation (what the Model calls)
    def evaluate_frustums(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, euclid_bins: Tensor,
                          n_dev: Optional[Tensor] = None, full: bool = True) -> Dict[str, Tensor]:
        """One sampling level.  origins/directions [R,3], pixel_area [R], euclid_bins [R,S+1] ->
        per-sample sigma [R,S], color [R,S,3] and (full) pred_normals, n_dot_d, diff, tint, roughness."""
        self._no_distortion()
        lib = _abi.load_library()
        R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
        dev = origins.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        level = {"sigma": f(R, S), "color": f(R, S, 3)}
        if full:
            level.update({"pred_normals": f(R, S, 3), "n_dot_d": f(R, S), "diff": f(R, S, 3), "tint": f(R, S, 3),
                          "roughness": f(R, S)})
        fo = ops.field_outputs_struct(level)
        desc = self.field_desc()
        pk = self.packed_weights()
        ops.timed("field_forward_eval" if full else "field_forward_eval_color", {"points": R * S}, lambda: check(
            lib.rsn_field_forward_frustum(C.byref(desc), ptr(pk), R, ptr(n_dev), S, ptr(origins), ptr(directions),
                                          ptr(pixel_area), ptr(euclid_bins), C.byref(fo), ops._stream())))
        return level

    def evaluate_frustums_train(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, euclid_bins: Tensor,
                                n_dev: Optional[Tensor] = None, want_normals: bool = True,
                                work: Optional[Dict] = None) -> Dict[str, Tensor]:
        """Training-mode level: same per-sample outputs as evaluate_frustums plus `raw_density`, the analytic
        `normals` (Field.get_normals, when want_normals) and `saved` = the activations the backward pass needs."""
        self._no_distortion()
        lib = _abi.load_library()
        R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
        N, W, L = R * S, self.width, self.mlp_base.num_layers
        dev = origins.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        level = {"sigma": f(R, S), "color": f(R, S, 3), "pred_normals": f(R, S, 3), "n_dot_d": f(R, S),
                 "diff": f(R, S, 3), "tint": f(R, S, 3), "roughness": f(R, S), "raw_density": f(R, S)}
        saved = self.alloc_saved(N, dev)
        if want_normals:
            saved["normals"] = f(R, S, 3)
        fo = ops.field_outputs_struct(level)
        fs = FieldSaved()
        for k, v in saved.items():
            setattr(fs, k, ptr(v))
        desc = self.field_desc()
        pk = self.packed_weights()
        if work is None:  # (a caller launching with a device-side ray count fills in the evaluated points later)
            work = {"points": N}
        ops.timed("field_forward_train_normals" if want_normals else "field_forward_train", work, lambda: check(
            lib.rsn_field_forward_frustum_train(C.byref(desc), ptr(pk), R, ptr(n_dev), S, ptr(origins), ptr(directions),
                                                ptr(pixel_area), ptr(euclid_bins), C.byref(fo), C.byref(fs),
                                                ops._stream())))
        level["saved"] = saved
        if want_normals:
            level["normals"] = saved["normals"]
        return level

    def evaluate_reflect_train(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, euclid_bins: Tensor,
                               n_dev: Tensor, sqradius: Tensor, work: Optional[Dict] = None,
                               inf_directions: Optional[Tensor] = None):
        """The first two field evaluations of the reflect branch in ONE launch (rsn_field_forward_train_jobs): the
        reflect-coarse level on the reflected rays (model.py:292-296) and get_inf_color of the same rays (model.py:290,
        field.py:190-201).  Both depend only on the secondary rays; as jobs of one launch get_inf_color's few tiles fill
        the level's last, partial round of tiles instead of occupying a launch of their own.
        -> (level dict as evaluate_frustums_train(want_normals=False), inf rgb [R,3], inf saved)."""
        self._no_distortion()
        lib = _abi.load_library()
        R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1  # R here: the rays the LEVEL is sized for
        # get_inf_color runs on ALL secondary rays (`inf_directions` / sqradius may be longer than the level's inputs when the
        # level's capacity is below the batch size: train_graph.reflect_capacity)
        d_inf = directions if inf_directions is None else inf_directions
        R_inf = d_inf.shape[0]
        dev = origins.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        level = {"sigma": f(R, S), "color": f(R, S, 3), "pred_normals": f(R, S, 3), "n_dot_d": f(R, S),
                 "diff": f(R, S, 3), "tint": f(R, S, 3), "roughness": f(R, S), "raw_density": f(R, S)}
        saved, inf_saved = self.alloc_saved(R * S, dev), self.alloc_saved(R_inf, dev)
        bg = f(R_inf, 3)
        fo = ops.field_outputs_struct(level)
        fs, fs_inf = FieldSaved(), FieldSaved()
        for k, v in saved.items():
            setattr(fs, k, ptr(v))
        for k, v in inf_saved.items():
            setattr(fs_inf, k, ptr(v))
        jobs = (_abi.FieldJob * 2)()
        jobs[0].kind, jobs[0].n_rays, jobs[0].n_dev, jobs[0].n_samples = 0, R, n_dev.data_ptr(), S
        jobs[0].origins, jobs[0].directions = origins.data_ptr(), directions.data_ptr()
        jobs[0].pixel_area, jobs[0].euclid_bins = pixel_area.data_ptr(), euclid_bins.data_ptr()
        jobs[0].out, jobs[0].saved = C.pointer(fo), C.pointer(fs)
        jobs[1].kind, jobs[1].n_rays, jobs[1].n_dev, jobs[1].n_samples = 1, R_inf, n_dev.data_ptr(), 1
        jobs[1].directions, jobs[1].sqradius, jobs[1].out_rgb = d_inf.data_ptr(), sqradius.data_ptr(), bg.data_ptr()
        jobs[1].saved = C.pointer(fs_inf)
        desc = self.field_desc()
        pk = self.packed_weights()
        ops.timed("field_forward_train", work if work is not None else {"points": R * (S + 1)}, lambda: check(
            lib.rsn_field_forward_train_jobs(C.byref(desc), ptr(pk), 2, jobs, ops._stream())))
        level["saved"] = saved
        return level, bg, inf_saved

    def wide_dtype(self) -> torch.dtype:
        """dtype of the WIDE training buffers (saved act / bott / hid, layer gradients dy / d_bott / da_mid): bf16 in the
        reduced-precision training mode (RSN_MMA_BF16: its GEMMs round these values to bf16 anyway -- half the step's HBM
        stream and activation memory), fp32 otherwise (include/rsn.h: rsn_field_saved)."""
        return torch.bfloat16 if int(self.mma_mode) == _abi.RSN_MMA_BF16 else torch.float32

    def train_layout(self) -> Dict:
        """Layout of the narrow saved buffers (encoded inputs, SH inputs) of a training forward for this Field's shape and
        MMA mode, from the library (rsn_train_saved_layout, include/rsn.h): row lengths, dtype and the slot -> reference
        column maps the weight gradients of trunk layer 0 / the skip layer / mlp_mid's SH part scatter by."""
        key = (int(self.mma_mode), self.width, self.mlp_base.num_layers)
        lay = getattr(self, "_train_layout", None)
        if lay is None or lay["key"] != key:
            lib = _abi.load_library()
            desc = self.field_desc()
            enc_cols, sh_cols, narrow = C.c_int32(), C.c_int32(), C.c_int32()
            enc_map, sh_map = (C.c_int32 * 128)(), (C.c_int32 * 64)()
            check(lib.rsn_train_saved_layout(C.byref(desc), C.byref(enc_cols), C.byref(sh_cols), C.byref(narrow), enc_map, sh_map))
            lay = {"key": key, "enc_cols": enc_cols.value, "sh_cols": sh_cols.value,
                   "narrow_dtype": torch.bfloat16 if narrow.value else torch.float32,
                   "enc_map": list(enc_map)[: enc_cols.value], "sh_map": list(sh_map)[: sh_cols.value], "dev": {}}
            self._train_layout = lay
        return lay

    def train_col_maps(self, dev):
        """(enc_map, sh_map) of train_layout() as int32 device tensors (cached per device)."""
        lay = self.train_layout()
        if dev not in lay["dev"]:
            lay["dev"][dev] = (torch.tensor(lay["enc_map"], dtype=torch.int32, device=dev),
                               torch.tensor(lay["sh_map"], dtype=torch.int32, device=dev))
        return lay["dev"][dev]

    def alloc_saved(self, N: int, dev) -> Dict[str, Tensor]:
        """The buffers a training-mode forward over N points fills (rsn_field_saved)."""
        W, L = self.width, self.mlp_base.num_layers
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        wd = self.wide_dtype()
        lay = self.train_layout()
        nd = lay["narrow_dtype"]
        return {"enc": torch.empty(N, lay["enc_cols"], device=dev, dtype=nd), "act": torch.empty(L, N, W, device=dev, dtype=wd),
                "bott": torch.empty(N, W, device=dev, dtype=wd), "sh": torch.empty(N, lay["sh_cols"], device=dev, dtype=nd),
                "hid": torch.empty(N, 128, device=dev, dtype=wd), "heads": f(N, 8),
                "relu_bits": torch.empty(L + 1, N, 2, max(W // 64, 2), device=dev, dtype=torch.int32)}

    def evaluate_inf(self, directions: Tensor, sqradius: Tensor, n_dev: Optional[Tensor] = None) -> Tensor:
        """get_inf_color on M (<= len) rays: directions [R,3], sqradius [R] -> rgb [R,3]."""
        lib = _abi.load_library()
        R = directions.shape[0]
        out = torch.empty(R, 3, device=directions.device, dtype=torch.float32)
        desc = self.field_desc()
        check(lib.rsn_field_forward_inf(C.byref(desc), ptr(self.packed_weights()), R, ptr(n_dev), ptr(directions),
                                        ptr(sqradius), ptr(out), ops._stream()))
        return out

    def evaluate_gaussians(self, means: Tensor, cov_diag: Optional[Tensor], view_dirs: Optional[Tensor],
                           want_embedding: bool = False, want_normals: bool = False) -> Dict[str, Tensor]:
        """Granular evaluation on explicit (already contracted) Gaussians: means [N,3], cov_diag [N,3].  want_normals: the
        training kernel (rsn_field_forward_gaussians_train) with its analytic-normal sweep; level["normals"] [N,3]."""
        lib = _abi.load_library()
        N, dev = means.shape[0], means.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        level = {"sigma": f(N), "color": f(N, 3), "pred_normals": f(N, 3), "n_dot_d": f(N), "diff": f(N, 3),
                 "tint": f(N, 3), "roughness": f(N), "raw_density": f(N)}
        emb = f(N, self.width) if want_embedding else None
        fo = ops.field_outputs_struct(level)
        desc = self.field_desc()
        if want_normals:
            saved = self.alloc_saved(N, dev)
            saved["normals"] = f(N, 3)
            fs = _abi.FieldSaved()
            for k in ("enc", "act", "bott", "sh", "hid", "heads", "normals", "relu_bits"):
                setattr(fs, k, ptr(saved[k]))
            check(lib.rsn_field_forward_gaussians_train(C.byref(desc), ptr(self.packed_weights()), N, ptr(means), ptr(cov_diag),
                                                        ptr(view_dirs), C.byref(fo), ptr(emb), C.byref(fs), ops._stream()))
            level["normals"] = saved["normals"]
        else:
            check(lib.rsn_field_forward_gaussians(C.byref(desc), ptr(self.packed_weights()), N, ptr(means), ptr(cov_diag),
                                                  ptr(view_dirs), C.byref(fo), ptr(emb), ops._stream()))
        if emb is not None:
            level["embedding"] = emb
        return level

    # ------------------------------------------------------------------ reference method names (granular API)
    def get_density(self, mean: Tensor, cov: Optional[Tensor] = None, requires_density_grad: bool = False):
        """field.py:122-137: (density [...,1], embedding [...,W]).  With `requires_density_grad` in training mode
        (field.py:125-127,133-134) the analytic normals -normalize(d raw_density / d mean) of the means handed in are
        computed by the same launch (the training kernel's in-kernel sweep) and handed out by get_normals(), as the
        reference's autograd does it -- forward VALUES: the granular calls carry no autograd graph (training runs through
        the model's fused graph, train_graph.py); exact-fp32 fields only."""
        shp = mean.shape[:-1]
        m = ops._f32c(mean.reshape(-1, 3))
        cd = None
        if cov is not None:
            cd = ops._f32c(torch.diagonal(cov, dim1=-2, dim2=-1).reshape(-1, 3))
        self._normals = None
        if requires_density_grad and self.training:
            if int(self.mma_mode) != _abi.RSN_MMA_F32:
                raise NotImplementedError("get_density(requires_density_grad=True) on explicit Gaussians runs on the exact-fp32 "
                                          "kernels only (set_mma_mode('f32')); the other modes train through the fused path")
            lv = self.evaluate_gaussians(m, cd, None, want_embedding=True, want_normals=True)
            self._normals = lv.pop("normals").reshape(*shp, 3)
        else:
            lv = self.evaluate_gaussians(m, cd, None, want_embedding=True)
        lv["embedding"] = lv["embedding"][:, : self.param_width]  # the kernels' padded units (exact zeros) are not part of the API
        self._last_level = {k: v.reshape(*shp, -1) for k, v in lv.items()}
        return self._last_level["sigma"], self._last_level["embedding"]

    # -- geometry ------------------------------------------------------------------------------------------------
    def get_blob(self, ray_samples):
        """field.py:90-96: (mean [...,3], cov [...,3,3]) of the conical frustums of `ray_samples` (any object with a
        `.frustums` carrying origins, directions, starts, ends, pixel_area; broadcastable)."""
        lib = _abi.load_library()
        fr = ray_samples.frustums
        shp = torch.broadcast_shapes(fr.origins.shape[:-1], fr.directions.shape[:-1], fr.starts.shape[:-1],
                                     fr.ends.shape[:-1], fr.pixel_area.shape[:-1])
        ex = lambda t, c: ops._f32c(t.expand(*shp, c).reshape(-1, c))  # noqa: E731
        o, d = ex(fr.origins, 3), ex(fr.directions, 3)
        pa, t0, t1 = ex(fr.pixel_area, 1).reshape(-1), ex(fr.starts, 1).reshape(-1), ex(fr.ends, 1).reshape(-1)
        n = o.shape[0]
        mean = torch.empty(n, 3, device=o.device)
        cov = torch.empty(n, 3, 3, device=o.device)
        check(lib.rsn_gaussians(n, ptr(o), ptr(d), ptr(pa), ptr(t0), ptr(t1), ptr(mean), ptr(cov), ops._stream()))
        mean, cov = mean.reshape(*shp, 3), cov.reshape(*shp, 3, 3)
        if self.spatial_distortion is not None:  # field.py:93-94
            from .nerfstudio_compat import Gaussians

            g = self.spatial_distortion(Gaussians(mean=mean, cov=cov))
            mean, cov = g.mean, g.cov
        return mean, cov

    def _no_distortion(self) -> None:
        if self.spatial_distortion is not None:
            raise NotImplementedError("the fused level kernels form the conical-frustum Gaussians in-kernel: a field with a "
                                      "spatial_distortion serves the granular API only (get_blob, contract, get_density, heads)")

    def contract(self, mean: Tensor, cov: Tensor, mask_return: bool = False):
        """field.py:98-119: mip-NeRF-360 contraction of Gaussians (J cov J, diagonal clamped >= 0)."""
        lib = _abi.load_library()
        shp = mean.shape[:-1]
        m = ops._f32c(mean.reshape(-1, 3))
        c = ops._f32c(cov.reshape(-1, 9))
        mo, co = torch.empty_like(m), torch.empty_like(c)
        check(lib.rsn_contract(m.shape[0], ptr(m), ptr(c), ptr(mo), ptr(co), ops._stream()))
        mo, co = mo.reshape(*shp, 3), co.reshape(*shp, 3, 3)
        if mask_return:
            return mo, co, torch.linalg.vector_norm(mean, dim=-1, keepdim=True) > 1
        return mo, co

    def get_reflection(self, directions: Tensor, normals: Tensor):
        """field.py:203-207: (normalised mirror reflection [...,3], n_dot_d [...,1])."""
        lib = _abi.load_library()
        shp = torch.broadcast_shapes(directions.shape[:-1], normals.shape[:-1])
        d = ops._f32c(directions.expand(*shp, 3).reshape(-1, 3))
        nrm = ops._f32c(normals.expand(*shp, 3).reshape(-1, 3))
        refl, ndd = torch.empty_like(d), torch.empty(d.shape[0], device=d.device)
        check(lib.rsn_reflection(d.shape[0], ptr(d), ptr(nrm), ptr(refl), ptr(ndd), ops._stream()))
        return refl.reshape(*shp, 3), ndd.reshape(*shp, 1)

    # -- head getters on a caller-supplied embedding (eval-mode forward; training goes through the fused graph) --
    def _heads(self, embedding: Tensor, view_dirs: Optional[Tensor] = None, roughness: Optional[Tensor] = None,
               mid_only: bool = False) -> Dict[str, Tensor]:
        lib = _abi.load_library()
        shp = embedding.shape[:-1]
        e = embedding.reshape(-1, self.param_width)
        if self.param_width != self.width:  # zero-padded units of the kernels' width
            e = torch.nn.functional.pad(e, (0, self.width - self.param_width))
        e = ops._f32c(e)
        N, dev = e.shape[0], e.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        lv = {"color": f(N, 3)}
        if not mid_only:
            lv.update({"pred_normals": f(N, 3), "diff": f(N, 3), "tint": f(N, 3), "roughness": f(N),
                       "raw_density": f(N), "sigma": f(N), "raw_roughness": f(N)})
        vd = None if view_dirs is None else ops._f32c(view_dirs.expand(*shp, 3).reshape(-1, 3))
        rg = None if roughness is None else ops._f32c(roughness.expand(*shp, 1).reshape(-1))
        fo = ops.field_outputs_struct(lv)
        desc = self.field_desc()
        check(lib.rsn_field_forward_embedding(C.byref(desc), ptr(self.packed_weights()), N, ptr(e), ptr(vd), ptr(rg),
                                              C.byref(fo), ops._stream()))
        return {k: v.reshape(*shp, -1) for k, v in lv.items()}

    def get_pred_normals(self, embedding: Tensor) -> Tensor:
        """field.py:139-144."""
        return self._heads(embedding)["pred_normals"]

    def get_normals(self) -> Tensor:
        """field.py:146-147 -> nerfstudio Field.get_normals: -normalize(d density-before-activation / d sample locations) of
        the last get_density(mean, cov, requires_density_grad=True) call in training mode (the model's fused training
        forward returns its own as outputs["normals_*"])."""
        n = getattr(self, "_normals", None)
        if n is None:
            raise RuntimeError("get_normals(): call get_density(mean, cov, requires_density_grad=True) in training mode first "
                               "(reference field.py:125-127)")
        return n

    def get_roughness(self, embedding: Tensor, activation: Optional[nn.Module] = None) -> Tensor:
        """field.py:150-155: activation(roughness head); default Sigmoid."""
        heads = self._heads(embedding)
        if activation is None or isinstance(activation, nn.Sigmoid):
            return heads["roughness"]
        return activation(heads["raw_roughness"])  # the caller's activation on the raw head output (any magnitude)

    def get_diff(self, embedding: Tensor) -> Tensor:
        """field.py:176-180."""
        return self._heads(embedding)["diff"]

    def get_tint(self, embedding: Tensor) -> Tensor:
        """field.py:182-186."""
        return self._heads(embedding)["tint"]

    def get_mid(self, directions: Tensor, roughness: Tensor, embedding: Tensor, use_bottleneck: bool = True) -> Tensor:
        """field.py:167-174: sigmoid RGB of mlp_mid(cat[SH34(directions, roughness), bottleneck(embedding)])."""
        if not use_bottleneck:
            raise NotImplementedError("use_bottleneck=False is not used by the reference model")
        return self._heads(embedding, view_dirs=directions, roughness=roughness, mid_only=True)["color"]

    def get_low(self, embedding: Tensor, use_bottleneck: bool = True) -> Tensor:
        """field.py:158-164 (unused by the model): get_mid with the SH inputs zeroed."""
        if not use_bottleneck:
            raise NotImplementedError("use_bottleneck=False is not used by the reference model")
        return self._heads(embedding, view_dirs=None, roughness=None, mid_only=True)["color"]

    def get_inf_color(self, directions: Tensor, sqradius: Tensor) -> Tensor:
        """field.py:190-201."""
        shp = directions.shape[:-1]
        out = self.evaluate_inf(ops._f32c(directions.reshape(-1, 3)), ops._f32c(sqradius.reshape(-1)))
        return out.reshape(*shp, 3)
