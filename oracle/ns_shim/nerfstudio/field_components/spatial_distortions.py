"""Oracle shim: only the SpatialDistortion type name (reflect_sampling_nerf_field.py:24);
the model passes spatial_distortion=None (reflect_sampling_nerf_model.py:103-106)."""
from torch import nn


class SpatialDistortion(nn.Module):
    def forward(self, positions):
        raise NotImplementedError
