"""The slice of the Nerfstudio plugin surface this method needs.

When `nerfstudio` is importable its own classes are used, so the Model/Field below register and run
inside `ns-train` as the same method (reference pyproject.toml:12-13).  When it is not (this image
and the GPU box have no nerfstudio and no network), minimal stand-ins with the same names, fields
and hook semantics are defined here so that the hot path, its tests and the benchmark run without it.
Only what the reference's model/field touch is provided.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Type

import torch
from torch import Tensor, nn

try:  # pragma: no cover
    from nerfstudio.utils.math import Gaussians  # type: ignore
except ImportError:

    @dataclass
    class Gaussians:
        """Stand-in for nerfstudio.utils.math.Gaussians: what a SpatialDistortion receives and returns (field.py:92-95)."""

        mean: Tensor
        cov: Tensor


try:  # pragma: no cover - nerfstudio is not installed in the build image
    from nerfstudio.cameras.rays import RayBundle  # type: ignore
    from nerfstudio.fields.base_field import Field  # type: ignore
    from nerfstudio.model_components.scene_colliders import NearFarCollider  # type: ignore
    from nerfstudio.models.base_model import Model, ModelConfig  # type: ignore

    HAVE_NERFSTUDIO = True
except ImportError:
    HAVE_NERFSTUDIO = False

    @dataclass
    class RayBundle:
        """Per-ray tensors, batch shape [R]: origins/directions [R,3], pixel_area/nears/fars [R,1]."""

        origins: Tensor
        directions: Tensor
        pixel_area: Tensor
        camera_indices: Optional[Tensor] = None
        nears: Optional[Tensor] = None
        fars: Optional[Tensor] = None
        metadata: Optional[Dict[str, Tensor]] = None
        times: Optional[Tensor] = None

        def __len__(self) -> int:
            return self.origins.shape[0]

        @property
        def shape(self):
            return tuple(self.origins.shape[:-1])

        def _map(self, fn):
            kw = {}
            for name in ("origins", "directions", "pixel_area", "camera_indices", "nears", "fars", "times"):
                v = getattr(self, name)
                kw[name] = fn(v) if isinstance(v, Tensor) else v
            return RayBundle(metadata=self.metadata, **kw)

        def __getitem__(self, idx):
            return self._map(lambda t: t[idx])

        def to(self, device):
            return self._map(lambda t: t.to(device))

        def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
            flat = self._map(lambda t: t.reshape(-1, t.shape[-1]))
            return flat._map(lambda t: t[start_idx:end_idx])

    class Field(nn.Module):
        """Base Field: only the two stashes Field.get_normals consumes."""

        def __init__(self) -> None:
            super().__init__()
            self._sample_locations: Optional[Tensor] = None
            self._density_before_activation: Optional[Tensor] = None

    class NearFarCollider(nn.Module):
        """Fills nears/fars with fixed planes when the bundle has none; near plane reset to 0 in eval."""

        def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True) -> None:
            super().__init__()
            self.near_plane = near_plane
            self.far_plane = far_plane
            self.reset_near_plane = reset_near_plane

        def set_nears_and_fars(self, ray_bundle):
            ones = torch.ones_like(ray_bundle.origins[..., 0:1])
            near_plane = self.near_plane if (self.training or not self.reset_near_plane) else 0
            ray_bundle.nears = ones * near_plane
            ray_bundle.fars = ones * self.far_plane
            return ray_bundle

        def forward(self, ray_bundle):
            if ray_bundle.nears is not None and ray_bundle.fars is not None:
                return ray_bundle
            return self.set_nears_and_fars(ray_bundle)

    @dataclass
    class ModelConfig:
        _target: Type = field(default_factory=lambda: Model)
        enable_collider: bool = True
        collider_params: Optional[Dict[str, float]] = field(
            default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
        loss_coefficients: Dict[str, float] = field(
            default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
        eval_num_rays_per_chunk: int = 4096
        prompt: Optional[str] = None

        def setup(self, **kwargs) -> Any:
            return self._target(self, **kwargs)

    class Model(nn.Module):
        """Base Model: construction protocol and forward() = collider + get_outputs."""

        config: ModelConfig

        def __init__(self, config: ModelConfig, scene_box=None, num_train_data: int = 0, **kwargs) -> None:
            super().__init__()
            self.config = config
            self.scene_box = scene_box
            self.render_aabb = None
            self.num_train_data = num_train_data
            self.kwargs = kwargs
            self.collider = None
            self.populate_modules()
            self.callbacks = None
            self.device_indicator_param = nn.Parameter(torch.empty(0))

        @property
        def device(self):
            return self.device_indicator_param.device

        def populate_modules(self):
            if self.config.enable_collider:
                assert self.config.collider_params is not None
                self.collider = NearFarCollider(near_plane=self.config.collider_params["near_plane"],
                                                far_plane=self.config.collider_params["far_plane"])

        def forward(self, ray_bundle) -> Dict[str, Tensor]:
            if self.collider is not None:
                ray_bundle = self.collider(ray_bundle)
            return self.get_outputs(ray_bundle)

        def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
            return {}

        @torch.no_grad()
        def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle) -> Dict[str, Tensor]:
            """Chunked full-image rendering (eval_num_rays_per_chunk rays per forward), as nerfstudio's base Model does it."""
            chunk = self.config.eval_num_rays_per_chunk
            image_shape = camera_ray_bundle.origins.shape[:-1]
            n = int(torch.tensor(image_shape).prod())
            lists: Dict[str, list] = {}
            n_chunks = 0
            for i in range(0, n, chunk):
                rb = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, min(i + chunk, n))
                n_chunks += 1
                for k, v in self.forward(rb).items():
                    if isinstance(v, Tensor) and v.shape[:1] == (len(rb),):
                        lists.setdefault(k, []).append(v)
            return {k: torch.cat(v).view(*image_shape, *v[0].shape[1:]) for k, v in lists.items() if len(v) == n_chunks}
