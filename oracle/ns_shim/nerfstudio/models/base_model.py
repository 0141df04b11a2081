"""Oracle shim: nerfstudio.models.base_model (Model, ModelConfig) and the near/far collider
(SURVEY.md §8(a) row N12).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at this boundary.

Reference call sites: reflect_sampling_nerf_model.py:34,38-39,78-95.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Type

import torch
from torch import nn

from nerfstudio.cameras.rays import RayBundle
from nerfstudio.configs.config_utils import to_immutable_dict


class NearFarCollider(nn.Module):
    """Sets every ray's nears/fars to fixed planes; in eval mode the near plane is reset to 0
    when `reset_near_plane` (the later-0.3.x default; SURVEY §8(c) lists this as unverifiable)."""

    def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True) -> None:
        super().__init__()
        self.near_plane = near_plane
        self.far_plane = far_plane
        self.reset_near_plane = reset_near_plane

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        ones = torch.ones_like(ray_bundle.origins[..., 0:1])
        near_plane = self.near_plane if (self.training or not self.reset_near_plane) else 0
        ray_bundle.nears = ones * near_plane
        ray_bundle.fars = ones * self.far_plane
        return ray_bundle

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


@dataclass
class ModelConfig:
    _target: Type = field(default_factory=lambda: Model)
    enable_collider: bool = True
    collider_params: Optional[Dict[str, float]] = to_immutable_dict({"near_plane": 2.0, "far_plane": 6.0})
    loss_coefficients: Dict[str, float] = to_immutable_dict({"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096
    prompt: Optional[str] = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


class Model(nn.Module):
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
        # to keep track of which device the nn.Module is on
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        if self.config.enable_collider:
            assert self.config.collider_params is not None
            self.collider = NearFarCollider(
                near_plane=self.config.collider_params["near_plane"], far_plane=self.config.collider_params["far_plane"]
            )

    def forward(self, ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        return {}
