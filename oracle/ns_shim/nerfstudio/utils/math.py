"""Oracle shim: restatement of the nerfstudio math helpers the reference relies on.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  nerfstudio (pyproject.toml:6 of the
reference, `nerfstudio >= 0.3.0`, un-vendored, not installable here) is restated from
its published 0.3.x semantics as summarised in SURVEY.md §8(a) rows N2/N3.
PARITY UNPINNED at this boundary: no nerfstudio source or golden vectors exist offline.

Used by the reference at reflect_sampling_nerf_field.py:12 (import) and, through
Frustums.get_gaussian_blob, at reflect_sampling_nerf_field.py:93.
"""
from dataclasses import dataclass

import torch
from torch import Tensor


@dataclass
class Gaussians:
    """Stores Gaussians: mean [..., 3] and covariance [..., 3, 3]."""

    mean: Tensor
    cov: Tensor


def compute_3d_gaussian(directions: Tensor, means: Tensor, dir_variance: Tensor, radius_variance: Tensor) -> Gaussians:
    """Covariance = var_t * d d^T + var_r * (I - d d^T / |d|^2)  (SURVEY §8(a) N3)."""
    dir_outer_product = directions[..., :, None] * directions[..., None, :]
    eye = torch.eye(directions.shape[-1], device=directions.device)
    dir_mag_sq = torch.clamp(torch.sum(directions**2, dim=-1, keepdim=True), min=1e-10)
    null_outer_product = eye - directions[..., :, None] * (directions / dir_mag_sq)[..., None, :]
    dir_cov_diag = dir_variance[..., None] * dir_outer_product[..., :, :]
    radius_cov_diag = radius_variance[..., None] * null_outer_product[..., :, :]
    cov = dir_cov_diag + radius_cov_diag
    return Gaussians(mean=means, cov=cov)


def conical_frustum_to_gaussian(origins: Tensor, directions: Tensor, starts: Tensor, ends: Tensor, radius: Tensor) -> Gaussians:
    """mip-NeRF conical frustum -> Gaussian."""
    mu = (starts + ends) / 2.0
    hw = (ends - starts) / 2.0
    means = origins + directions * (mu + (2.0 * mu * hw**2.0) / (3.0 * mu**2.0 + hw**2.0))
    dir_variance = (hw**2) / 3 - (4 / 15) * ((hw**4 * (12 * mu**2 - hw**2)) / (3 * mu**2 + hw**2) ** 2)
    radius_variance = radius**2 * ((mu**2) / 4 + (5 / 12) * hw**2 - 4 / 15 * (hw**4) / (3 * mu**2 + hw**2))
    return compute_3d_gaussian(directions, means, dir_variance, radius_variance)


def expected_sin(x_means: Tensor, x_vars: Tensor) -> Tensor:
    """E[sin(x)] for x ~ N(x_means, x_vars)."""
    return torch.exp(-0.5 * x_vars) * torch.sin(x_means)


def safe_normalize(vectors: Tensor, eps: float = 1e-10) -> Tensor:
    return vectors / (torch.norm(vectors, dim=-1, keepdim=True) + eps)
