"""ReflectSamplingNeRFModel on MI355X: the reference's Model hooks, config fields and output keys
(reflect_sampling_nerf_model.py:38-430), with get_outputs driven through librsn_hip.so.

get_outputs (reference model.py:142-344) becomes a fixed sequence of asynchronous kernel launches on the
current stream -- samplers, the fused field kernel, per-ray compositing, device-side stable compaction of
the reflected rays (dynamic M read from device memory by the later launches) -- with NO host sync: the one
output whose SHAPE depends on M (`depth_reflect_fine`, [M, 1]) is materialised on first use (ops.LazyOutputs).
The reference's debug prints / six syncs (model.py:230,263-265,342) are not reproduced.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Type

import torch
from torch import Tensor, nn
from torch.nn import Parameter

from . import ops
from .nerfstudio_compat import Model, ModelConfig
from .reflect_sampling_nerf_components import (
    IntegratedSHEncoding,
    NeRFEncoding,
    PDFSampler,
    ReciprocalSampler,
    UniformSampler,
)
from .reflect_sampling_nerf_field import ReflectSamplingNeRFNerfField

_LOSS_COEFFICIENTS = {
    "loss_low_coarse": 1e-1,
    "loss_low_fine": 1e-1,
    "loss_mid_coarse": 1.0,
    "loss_mid_fine": 1.0,
    "loss_reflect_low_coarse": 1e-1,
    "loss_reflect_low_fine": 1e-1,
    "loss_reflect_mid_coarse": 1.0,
    "loss_reflect_mid_fine": 1.0,
    "predicted_normal_loss_coarse": 3e-5,
    "predicted_normal_loss_fine": 3e-4,
    "orientation_loss_coarse": 1e-2,
    "orientation_loss_fine": 1e-1,
}


@dataclass
class ReflectSamplingNeRFModelConfig(ModelConfig):
    """Same field names and defaults as the reference config (model.py:38-75)."""

    num_coarse_samples: int = 128
    num_importance_samples: int = 128
    num_reflect_coarse_samples: int = 64
    num_reflect_importance_samples: int = 64
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: dict(_LOSS_COEFFICIENTS))
    enable_temporal_distortion: bool = False
    temporal_distortion_params: Dict[str, Any] = field(default_factory=lambda: {"kind": "dnerf"})
    # Field size knobs (constructor arguments of the reference Field, field.py:40-41).  The reference Model
    # always builds the default 8 x 256 Field; BASELINE.json's configs turn these two.
    base_mlp_num_layers: int = 8
    base_mlp_layer_width: int = 256
    _target: Type = field(default_factory=lambda: ReflectSamplingNeRFModel)


class ReflectSamplingNeRFModel(Model):
    config: ReflectSamplingNeRFModelConfig

    def __init__(self, config: ReflectSamplingNeRFModelConfig, **kwargs) -> None:
        self.field = None
        assert config.collider_params is not None, "MipNeRF model requires bounding box collider parameters."
        super().__init__(config=config, **kwargs)
        assert self.config.collider_params is not None, "mip-NeRF requires collider parameters to be set."

    # ------------------------------------------------------------------ construction (model.py:93-132)
    def populate_modules(self):
        super().populate_modules()
        position_encoding = NeRFEncoding(in_dim=3, num_frequencies=16, min_freq_exp=0.0, max_freq_exp=16.0,
                                         include_input=True)
        direction_encoding = IntegratedSHEncoding()
        self.field = ReflectSamplingNeRFNerfField(
            position_encoding=position_encoding,
            direction_encoding=direction_encoding,
            base_mlp_num_layers=getattr(self.config, "base_mlp_num_layers", 8),
            base_mlp_layer_width=getattr(self.config, "base_mlp_layer_width", 256),
        )
        self.sampler_uniform = UniformSampler(num_samples=self.config.num_coarse_samples)
        self.sampler_pdf = PDFSampler(num_samples=self.config.num_importance_samples, include_original=False)
        self.sampler_reciprocal = ReciprocalSampler(num_samples=self.config.num_reflect_coarse_samples, tan=0.25)
        self.sampler_reflect_pdf = PDFSampler(num_samples=self.config.num_reflect_importance_samples,
                                              include_original=False)
        self.far = 2**8
        self.near = 1.0 / 16
        self.background_color = torch.tensor([1.0, 1.0, 1.0])  # colors.WHITE
        self.rgb_loss = nn.MSELoss()
        # Checkpoints of the reference: its Model also owns torchmetrics modules (model.py:131-133: psnr, ssim,
        # lpips -- the LPIPS network's weights sit in every pipeline checkpoint as `_model.lpips.net.*`).  This Model has no such
        # modules (SSIM / LPIPS are out of scope), so their entries are dropped from an incoming state dict before nn.Module's strict
        # key check sees them -- also when the checkpoint is loaded through a parent (nerfstudio's pipeline: prefix `_model.`).
        self._register_load_state_dict_pre_hook(self._drop_reference_metric_state)

    _REFERENCE_METRIC_MODULES = ("lpips", "psnr", "ssim")

    @staticmethod
    def _drop_reference_metric_state(state_dict, prefix, *_unused) -> None:
        drop = tuple(prefix + m + "." for m in ReflectSamplingNeRFModel._REFERENCE_METRIC_MODULES)
        for k in [k for k in state_dict if k.startswith(drop)]:
            del state_dict[k]

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        if self.field is None:
            raise ValueError("populate_fields() must be called before get_param_groups")
        return {"fields": list(self.field.parameters())}

    # ------------------------------------------------------------------ the hot path (model.py:142-344)
    def get_outputs(self, ray_bundle) -> Dict[str, Tensor]:
        if self.field is None:
            raise ValueError("populate_fields() must be called before get_outputs")
        if self.training:
            return self._get_outputs_train(ray_bundle)
        return self._get_outputs_eval(ray_bundle)

    def _get_outputs_train(self, ray_bundle, jitter: Optional[Dict[str, Tensor]] = None,
                           bins: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """Training mode: one autograd node (train_graph.GetOutputsTrain) over the HIP forward/backward kernels.
        `jitter` optionally injects the samplers' uniform draws and `bins` whole sampler outputs
        ({"fine_spacing", "fine_euclid", ...}); tests share them with the oracle / the reference's logged values."""
        from .train_graph import DIFF_KEYS, FUSED_KEYS, GetOutputsTrain

        R = ray_bundle.origins.shape[0]
        o = ops._f32c(ray_bundle.origins.reshape(R, 3))
        d = ops._f32c(ray_bundle.directions.reshape(R, 3))
        pa = ops._f32c(ray_bundle.pixel_area.reshape(R))
        nears = ops._f32c(ray_bundle.nears.reshape(R))
        fars = ops._f32c(ray_bundle.fars.reshape(R))
        outs = GetOutputsTrain.apply(self, o, d, pa, nears, fars, jitter, bins, *self.field.parameters())
        aux = self._train_aux
        self._train_aux = None
        outputs = type(aux)(zip(DIFF_KEYS, outs))  # LazyOutputs: depth_reflect_fine ([M, 1]) on first access
        for k, v in aux.present():  # (not items(): that would materialise the lazy [M, 1] entry -- a host read)
            outputs[k] = v.detach() if v.dtype.is_floating_point else v
        outputs.lazy = aux.lazy
        # not keys of the reference's dict: the normal losses of get_loss_dict, reduced per ray by the compositing kernel
        outputs.fused = dict(zip(FUSED_KEYS, outs[len(DIFF_KEYS):]))
        return outputs

    @property
    def _last_num_reflected(self) -> int:
        """M of the last training-mode get_outputs.  The count lives on the device (`_last_n_masked_dev`): reading it
        here is a device-to-host synchronisation -- diagnostics only, the training step never does."""
        nm = getattr(self, "_last_n_masked_dev", None)
        return 0 if nm is None else int(nm.item())

    @torch.no_grad()
    def _get_outputs_eval(self, ray_bundle) -> Dict[str, Tensor]:
        cfg, fld = self.config, self.field
        R = ray_bundle.origins.shape[0]
        o = ops._f32c(ray_bundle.origins.reshape(R, 3))
        d = ops._f32c(ray_bundle.directions.reshape(R, 3))
        pa = ops._f32c(ray_bundle.pixel_area.reshape(R))
        nears = ops._f32c(ray_bundle.nears.reshape(R))
        fars = ops._f32c(ray_bundle.fars.reshape(R))
        Sc, Sf = cfg.num_coarse_samples, cfg.num_importance_samples
        Src, Srf = cfg.num_reflect_coarse_samples, cfg.num_reflect_importance_samples
        EVAL, CLIP = ops.RSN_COMP_EVAL, ops.RSN_COMP_CLIP_RGB
        uni, rec = self.sampler_uniform.spec, self.sampler_reciprocal.spec

        # A. coarse primary (model.py:148-177)
        sb_c, eb_c = ops.sample_spaced(R, None, Sc, uni.spacing, uni.tan, nears, fars, None)
        lc = fld.evaluate_frustums(o, d, pa, eb_c)
        cc = ops.composite(R, None, Sc, 1, EVAL | CLIP, lc["sigma"], eb_c, lc["color"])
        # B. fine primary (model.py:182-211) + C. per-ray surface attributes (model.py:215-227)
        sb_f, eb_f = ops.sample_pdf(R, None, Sc, Sf, uni.spacing, uni.tan, self.sampler_pdf.histogram_padding, nears,
                                    fars, cc["weights"], sb_c, None)
        lf = fld.evaluate_frustums(o, d, pa, eb_f)
        cf = ops.composite(R, None, Sf, 1, EVAL | CLIP, lf["sigma"], eb_f, lf["color"], level=lf, surface=True)
        # mask, stable compaction, secondary rays, default reflect colours (model.py:222-229,240-241,267-289)
        rs = ops.reflect_setup(o, d, cf["accumulation"], cf["depth"], cf["normals"], cf["roughness"], float(self.far))
        n_dev = rs["n_masked"]

        outputs = ops.LazyOutputs({
            "mid_rgb_coarse": cc["rgb"],
            "mid_rgb_fine": cf["rgb"],
            "mid_reflect_coarse": rs["reflect_coarse"],
            "mid_reflect_fine": rs["reflect_fine"],
            "accumulation_coarse": cc["accumulation"].unsqueeze(-1),
            "accumulation_fine": cf["accumulation"].unsqueeze(-1),
            "depth_coarse": cc["depth"].unsqueeze(-1),
            "depth_fine": cf["depth"].unsqueeze(-1),
            "weights_coarse": cc["weights"].unsqueeze(-1),
            "weights_fine": cf["weights"].unsqueeze(-1),
            "pred_normals_coarse": lc["pred_normals"],
            "pred_normals_fine": lf["pred_normals"],
            "normals_coarse": lc["pred_normals"],  # eval: normals == predicted normals (model.py:161-162)
            "normals_fine": lf["pred_normals"],
            "n_dot_d_coarse": lc["n_dot_d"].unsqueeze(-1),
            "n_dot_d_fine": lf["n_dot_d"].unsqueeze(-1),
            "diff": cf["diff"],
            "tint": cf["tint"],
            "roughness": cf["roughness"].unsqueeze(-1),
            "mask": rs["mask"].bool(),
        })

        # E-G. reflected rays; every launch below reads M from device memory (no host sync)
        o2, d2, pa2, near2, far2 = rs["origins2"], rs["directions2"], rs["pixel_area2"], rs["nears2"], rs["fars2"]
        bg = fld.evaluate_inf(d2, rs["sqradius"], n_dev)
        sb_rc, eb_rc = ops.sample_spaced(R, n_dev, Src, rec.spacing, rec.tan, near2, far2, None)
        lrc = fld.evaluate_frustums(o2, d2, pa2, eb_rc, n_dev, full=False)
        crc = ops.composite(R, n_dev, Src, 2, EVAL, lrc["sigma"], eb_rc, lrc["color"], bg_rgb=bg, want_depth=False)
        ops.reflect_combine(R, n_dev, rs["ray_index"], cf["diff"], cf["tint"], crc["rgb"], rs["reflect_coarse"])
        sb_rf, eb_rf = ops.sample_pdf(R, n_dev, Src, Srf, rec.spacing, rec.tan,
                                      self.sampler_reflect_pdf.histogram_padding, near2, far2, crc["weights"], sb_rc,
                                      None)
        lrf = fld.evaluate_frustums(o2, d2, pa2, eb_rf, n_dev, full=False)
        crf = ops.composite(R, n_dev, Srf, 2, EVAL, lrf["sigma"], eb_rf, lrf["color"], bg_rgb=bg)
        ops.reflect_combine(R, n_dev, rs["ray_index"], cf["diff"], cf["tint"], crf["rgb"], rs["reflect_fine"])

        # depth_reflect_fine is [M, 1] and present only when M > 0 (model.py:341): M stays on the device and the entry is
        # materialised on first use (ops.LazyOutputs) -- get_outputs itself issues no device-to-host read, so a chunked
        # image render enqueues every chunk before anything waits
        outputs.lazy["depth_reflect_fine"] = (n_dev, crf["depth"])
        return outputs

    # ------------------------------------------------------------------ eval image, chunked (config.py:41; "next" row §8(f).4)
    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle) -> Dict[str, Tensor]:
        """Chunked full-image rendering: `eval_num_rays_per_chunk` rays per forward (reference config.py:41: 1024),
        per-ray outputs concatenated and reshaped to the image -- what nerfstudio's base Model does, overridden here
        (for nerfstudio's own base class too) so that EVERY CHUNK IS ENQUEUED BEFORE ANYTHING IS READ BACK: get_outputs
        issues no device-to-host read (the one output whose shape needs the reflected-ray count, depth_reflect_fine
        [M, 1], is lazy and is no image anyway; the base class's `.items()` would materialise it once per chunk), so the
        GPU runs chunk after chunk while the host is already launching the following ones."""
        chunk = self.config.eval_num_rays_per_chunk
        image_shape = camera_ray_bundle.origins.shape[:-1]
        n = 1
        for s_ in image_shape:
            n *= int(s_)
        lists: Dict[str, list] = {}
        n_chunks = 0
        # Small chunks (the reference's 1024 rays) leave the GPU with ramps, tails and a dozen 10-us launches per 3 ms of field
        # kernels: consecutive chunks are independent, so they alternate between two side streams and one chunk's small
        # launches / last partial round of tiles run beside the other's field kernels.
        dev = camera_ray_bundle.origins.device
        side = None
        if dev.type == "cuda" and n > chunk:
            self.field.packed_weights()  # packed once, on the caller's stream, before the side streams read it
            main = torch.cuda.current_stream(dev)
            n_side = int(getattr(self, "eval_side_streams", 3))  # measured: 278 / 310 / 325 / 312 K rays/s with 1 / 2 / 3 / 4 at 1024-ray chunks
            side = getattr(self, "_eval_streams", None)
            if side is None or side[0].device != dev or len(side) != n_side:
                side = self._eval_streams = tuple(torch.cuda.Stream(device=dev) for _ in range(n_side))
            for st in side:
                st.wait_stream(main)
        for i in range(0, n, chunk):
            rb = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, min(i + chunk, n))
            ctx = torch.cuda.stream(side[n_chunks % len(side)]) if side is not None else contextlib.nullcontext()
            n_chunks += 1
            with ctx:
                out = self.forward(rb)
            present = out.present() if hasattr(out, "present") else out.items()
            for k, v in present:
                if isinstance(v, Tensor) and v.shape[:1] == (len(rb),):
                    lists.setdefault(k, []).append(v)
        if side is not None:
            for st in side:
                main.wait_stream(st)
            for vs in lists.values():  # allocated on a side stream, read by the caller's stream from here on
                for v in vs:
                    v.record_stream(main)
        # per-ray outputs only: a key that is not [rays, ...] in every chunk (depth_reflect_fine is [M,1]) is no image
        return {k: torch.cat(v).view(*image_shape, *v[0].shape[1:]) for k, v in lists.items()
                if len(v) == n_chunks and k != "depth_reflect_fine"}

    # ------------------------------------------------------------------ loss (model.py:346-430; "next" row §8(f).1)
    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, Tensor]:
        """model.py:346-430: four MSE terms against the (white-blended) image, two predicted-normal and two
        orientation terms, scaled by config.loss_coefficients.  One fused HIP pass (train_ops.FusedLoss)."""
        from .train_ops import fused_loss_dict, fused_ray_loss_dict

        image = batch["image"].to(self.device)
        if image.shape[-1] == 4:  # RGBRenderer.blend_background with the white background colour
            image = image[..., :3] * image[..., 3:] + (1.0 - image[..., 3:])
        fused = getattr(outputs, "fused", None)
        if fused is not None:  # outputs of this model's own training graph: the per-sample terms are already per-ray sums
            return fused_ray_loss_dict(outputs, fused, image, self.config.loss_coefficients)
        return fused_loss_dict(outputs, image, self.config.loss_coefficients)

    # ------------------------------------------------------------------ eval image (model.py:432-482; "next" row §8(f).4)
    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
        """Working equivalent of the reference hook (which raises KeyError: 'low_coarse' at model.py:438): PSNR of the
        coarse render and of the reflect-fine render against the white-blended ground truth, and the three panels the
        reference assembles (rgb | accumulation | depth, coarse and fine side by side).  SSIM / LPIPS need the
        torchmetrics networks and are out of scope."""
        assert self.config.collider_params is not None, "mip-NeRF requires collider parameters to be set."
        image = batch["image"].to(outputs["mid_rgb_coarse"].device)
        if image.shape[-1] == 4:
            image = image[..., :3] * image[..., 3:] + (1.0 - image[..., 3:])
        rgb_coarse = torch.clip(outputs["mid_rgb_coarse"], 0.0, 1.0)
        rgb_fine = torch.clip(outputs["mid_reflect_fine"], 0.0, 1.0)

        def psnr(a, b):
            return float(10.0 * torch.log10(1.0 / torch.mean((a - b) ** 2).clamp(min=1e-12)))

        near, far = self.config.collider_params["near_plane"], self.config.collider_params["far_plane"]

        def depth_panel(d, acc):
            return (((d - near) / (far - near)).clamp(0.0, 1.0) * acc + (1.0 - acc)).expand(*d.shape[:-1], 3)

        gray = lambda a: a.expand(*a.shape[:-1], 3)  # noqa: E731
        images = {
            "img": torch.cat([image, rgb_coarse, rgb_fine], dim=1),
            "accumulation": torch.cat([gray(outputs["accumulation_coarse"]), gray(outputs["accumulation_fine"])], dim=1),
            "depth": torch.cat([depth_panel(outputs["depth_coarse"], outputs["accumulation_coarse"]),
                                depth_panel(outputs["depth_fine"], outputs["accumulation_fine"])], dim=1),
        }
        fine_psnr = psnr(image, rgb_fine)
        metrics = {"psnr": fine_psnr, "coarse_psnr": psnr(image, rgb_coarse), "fine_psnr": fine_psnr}
        return metrics, images
