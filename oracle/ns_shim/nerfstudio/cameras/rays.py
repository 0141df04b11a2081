"""Oracle shim: RayBundle / RaySamples / Frustums (nerfstudio.cameras.rays).

TEST INFRASTRUCTURE ONLY.  Restated from nerfstudio 0.3.x semantics (SURVEY.md §8(a)
rows N3, N4, N5); PARITY UNPINNED at this boundary.  Only the behaviour the reference
exercises is provided: stride-0 broadcasting of per-ray fields over the sample axis,
boolean/slice indexing, `get_ray_samples`, `get_gaussian_blob`, `get_weights`.

Reference call sites: reflect_sampling_nerf_model.py:148,154,182,188,267-289,292,296,317,322;
reflect_sampling_nerf_field.py:93.
"""
from dataclasses import dataclass, fields
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from nerfstudio.utils.math import Gaussians, conical_frustum_to_gaussian


class _TensorFields:
    """Very small stand-in for nerfstudio's TensorDataclass: every Tensor field has shape
    [*batch, C]; fields are broadcast (expand, stride 0) to a common batch shape."""

    def _tensor_items(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, Tensor):
                yield f.name, v

    def _broadcast(self):
        shapes = [v.shape[:-1] for _, v in self._tensor_items()]
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, _TensorFields):
                shapes.append(v.shape)
        if not shapes:
            return
        batch = torch.broadcast_shapes(*shapes)
        for name, v in list(self._tensor_items()):
            setattr(self, name, v.broadcast_to((*batch, v.shape[-1])))
        self._shape = tuple(batch)

    @property
    def shape(self):
        return self._shape

    def __len__(self):
        return self._shape[0]

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, Tensor):
                kw[f.name] = v[idx + (slice(None),)]
            elif isinstance(v, _TensorFields):
                kw[f.name] = v[idx]
            else:
                kw[f.name] = v
        return type(self)(**kw)


@dataclass
class Frustums(_TensorFields):
    origins: Tensor
    directions: Tensor
    starts: Tensor
    ends: Tensor
    pixel_area: Tensor
    offsets: Optional[Tensor] = None

    def __post_init__(self):
        self._broadcast()

    def get_gaussian_blob(self) -> Gaussians:
        """Conical frustum -> Gaussian; cone radius = sqrt(pixel_area)/sqrt(pi)."""
        cone_radius = torch.sqrt(self.pixel_area) / 1.7724538509055159
        if self.offsets is not None:
            raise NotImplementedError()
        return conical_frustum_to_gaussian(
            origins=self.origins,
            directions=self.directions,
            starts=self.starts,
            ends=self.ends,
            radius=cone_radius,
        )


@dataclass
class RaySamples(_TensorFields):
    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None

    def __post_init__(self):
        self._broadcast()

    def get_weights(self, densities: Tensor) -> Tensor:
        """Volume-rendering weights from densities [..., S, 1]."""
        delta_density = self.deltas * densities
        alphas = 1 - torch.exp(-delta_density)
        transmittance = torch.cumsum(delta_density[..., :-1, :], dim=-2)
        transmittance = torch.cat(
            [torch.zeros((*transmittance.shape[:1], 1, 1), device=densities.device), transmittance], dim=-2
        )
        transmittance = torch.exp(-transmittance)
        weights = alphas * transmittance
        weights = torch.nan_to_num(weights)
        return weights


@dataclass
class RayBundle(_TensorFields):
    origins: Tensor
    directions: Tensor
    pixel_area: Tensor
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None

    def __post_init__(self):
        self._broadcast()

    def get_ray_samples(
        self,
        bin_starts: Tensor,
        bin_ends: Tensor,
        spacing_starts: Optional[Tensor] = None,
        spacing_ends: Optional[Tensor] = None,
        spacing_to_euclidean_fn: Optional[Callable] = None,
    ) -> RaySamples:
        deltas = bin_ends - bin_starts
        camera_indices = self.camera_indices[..., None, :] if self.camera_indices is not None else None
        frustums = Frustums(
            origins=self.origins[..., None, :],  # [..., 1, 3]
            directions=self.directions[..., None, :],  # [..., 1, 3]
            starts=bin_starts,  # [..., S, 1]
            ends=bin_ends,  # [..., S, 1]
            pixel_area=self.pixel_area[..., None, :],  # [..., 1, 1]
        )
        return RaySamples(
            frustums=frustums,
            camera_indices=camera_indices,
            deltas=deltas,
            spacing_starts=spacing_starts,
            spacing_ends=spacing_ends,
            spacing_to_euclidean_fn=spacing_to_euclidean_fn,
            metadata=None,
            times=None,
        )
