"""Oracle shim: nerfstudio.fields.base_field.Field.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at this boundary.
Reference call sites: reflect_sampling_nerf_field.py:25,28,146-147.
"""
from typing import Optional

import torch
from torch import Tensor, nn


class Field(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self._sample_locations: Optional[Tensor] = None
        self._density_before_activation: Optional[Tensor] = None

    def get_normals(self) -> Tensor:
        """Analytic normals = -normalize(d raw_density / d sample_location)."""
        assert self._sample_locations is not None, "Sample locations must be set before calling get_normals."
        assert self._density_before_activation is not None, "Density must be set before calling get_normals."
        assert (
            self._sample_locations.shape[:-1] == self._density_before_activation.shape[:-1]
        ), "Sample locations and density must have the same shape besides the last dimension."
        normals = torch.autograd.grad(
            self._density_before_activation,
            self._sample_locations,
            grad_outputs=torch.ones_like(self._density_before_activation),
            retain_graph=True,
        )[0]
        normals = -torch.nn.functional.normalize(normals, dim=-1)
        return normals
