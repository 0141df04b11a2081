"""Oracle shim: nerfstudio.model_components.renderers.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at this boundary.
Reference call sites: reflect_sampling_nerf_model.py:118-124,155-156,176,189-190,210,215,217,
220,226,311,337,341,360-391,437.
"""
from typing import Literal, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from nerfstudio.cameras.rays import RaySamples
from nerfstudio.utils import colors
from nerfstudio.utils.math import safe_normalize

BackgroundColor = Union[Literal["random", "last_sample", "black", "white"], Tensor]
BACKGROUND_COLOR_OVERRIDE: Optional[Tensor] = None


class RGBRenderer(nn.Module):
    def __init__(self, background_color: BackgroundColor = "random") -> None:
        super().__init__()
        self.background_color: BackgroundColor = background_color

    @classmethod
    def combine_rgb(cls, rgb: Tensor, weights: Tensor, background_color: BackgroundColor = "random",
                    ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None) -> Tensor:
        if ray_indices is not None and num_rays is not None:
            raise NotImplementedError("packed samples are not used by the reference")
        comp_rgb = torch.sum(weights * rgb, dim=-2)
        accumulated_weight = torch.sum(weights, dim=-2)
        if BACKGROUND_COLOR_OVERRIDE is not None:
            background_color = BACKGROUND_COLOR_OVERRIDE
        if isinstance(background_color, str) and background_color == "random":
            # "random": the composite is returned unblended (as over black)
            return comp_rgb
        if isinstance(background_color, str) and background_color == "last_sample":
            background_color = rgb[..., -1, :]
        background_color = cls.get_background_color(background_color, shape=comp_rgb.shape, device=comp_rgb.device)
        assert isinstance(background_color, torch.Tensor)
        comp_rgb = comp_rgb + background_color * (1.0 - accumulated_weight)
        return comp_rgb

    @classmethod
    def get_background_color(cls, background_color: BackgroundColor, shape: Tuple[int, ...], device) -> Tensor:
        assert background_color not in {"last_sample", "random"} if isinstance(background_color, str) else True
        assert shape[-1] == 3, "Background color must be RGB."
        if BACKGROUND_COLOR_OVERRIDE is not None:
            background_color = BACKGROUND_COLOR_OVERRIDE
        if isinstance(background_color, str) and background_color in colors.COLORS_DICT:
            background_color = colors.COLORS_DICT[background_color]
        assert isinstance(background_color, Tensor)
        return background_color.expand(shape).to(device)

    def blend_background(self, image: Tensor, background_color: Optional[BackgroundColor] = None) -> Tensor:
        if image.size(-1) < 4:
            return image
        rgb, opacity = image[..., :3], image[..., 3:]
        if background_color is None:
            background_color = self.background_color
            if isinstance(background_color, str) and background_color in {"last_sample", "random"}:
                background_color = "black"
        background_color = self.get_background_color(background_color, shape=rgb.shape, device=rgb.device)
        assert isinstance(background_color, torch.Tensor)
        return rgb * opacity + background_color.to(rgb.device) * (1 - opacity)

    def blend_background_for_loss_computation(self, pred_image: Tensor, pred_accumulation: Tensor,
                                              gt_image: Tensor) -> Tuple[Tensor, Tensor]:
        background_color = self.background_color
        if isinstance(background_color, str) and background_color == "last_sample":
            background_color = "black"  # No background blending for GT
        elif isinstance(background_color, str) and background_color == "random":
            background_color = torch.rand_like(pred_image)
            pred_image = pred_image + background_color * (1.0 - pred_accumulation)
        gt_image = self.blend_background(gt_image, background_color=background_color)
        return pred_image, gt_image

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, background_color: Optional[BackgroundColor] = None) -> Tensor:
        if background_color is None:
            background_color = self.background_color
        if not self.training:
            rgb = torch.nan_to_num(rgb)
        rgb = self.combine_rgb(rgb, weights, background_color=background_color, ray_indices=ray_indices,
                               num_rays=num_rays)
        if not self.training:
            torch.clamp_(rgb, min=0.0, max=1.0)
        return rgb


class AccumulationRenderer(nn.Module):
    @classmethod
    def forward(cls, weights: Tensor, ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None) -> Tensor:
        return torch.sum(weights, dim=-2)


class DepthRenderer(nn.Module):
    def __init__(self, method: Literal["median", "expected"] = "median") -> None:
        super().__init__()
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None) -> Tensor:
        if self.method == "median":
            steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
            cumulative_weights = torch.cumsum(weights[..., 0], dim=-1)  # [..., S]
            split = torch.ones((*weights.shape[:-2], 1), device=weights.device) * 0.5  # [..., 1]
            median_index = torch.searchsorted(cumulative_weights, split, side="left")  # [..., 1]
            median_index = torch.clamp(median_index, 0, steps.shape[-2] - 1)
            median_depth = torch.gather(steps[..., 0], dim=-1, index=median_index)
            return median_depth
        if self.method == "expected":
            eps = 1e-10
            steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
            depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + eps)
            depth = torch.clip(depth, steps.min(), steps.max())
            return depth
        raise NotImplementedError(f"Method {self.method} not implemented")


class NormalsRenderer(nn.Module):
    @classmethod
    def forward(cls, normals: Tensor, weights: Tensor, normalize: bool = True) -> Tensor:
        n = torch.sum(weights * normals, dim=-2)
        if normalize:
            n = safe_normalize(n)
        return n


class SemanticRenderer(nn.Module):
    @classmethod
    def forward(cls, semantics: Tensor, weights: Tensor) -> Tensor:
        return torch.sum(weights * semantics, dim=-2)
