"""Oracle shim: nerfstudio.field_components.encodings (Encoding, Identity, NeRFEncoding).

TEST INFRASTRUCTURE ONLY.  Restated from nerfstudio 0.3.x semantics (SURVEY.md §8(a)
row N2); PARITY UNPINNED at this boundary.

Reference call sites: reflect_sampling_nerf_model.py:98-100 (NeRFEncoding(3, 16, 0, 16,
include_input=True)), reflect_sampling_nerf_field.py:129-131 (forward with covs),
reflect_sampling_nerf_components.py:38-50 (Encoding base class).
"""
from typing import Optional

import torch
from torch import Tensor, nn

from nerfstudio.utils.math import expected_sin


class FieldComponent(nn.Module):
    def __init__(self, in_dim: Optional[int] = None, out_dim: Optional[int] = None) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim

    def get_out_dim(self) -> int:
        if self.out_dim is None:
            raise ValueError("Output dimension has not been set")
        return self.out_dim


class Encoding(FieldComponent):
    def __init__(self, in_dim: int) -> None:
        if in_dim <= 0:
            raise ValueError("Input dimension should be greater than zero")
        super().__init__(in_dim=in_dim)


class Identity(Encoding):
    def get_out_dim(self) -> int:
        return self.in_dim

    def forward(self, in_tensor: Tensor) -> Tensor:
        return in_tensor


class NeRFEncoding(Encoding):
    """Multi-scale sinusoidal encoding; with `covs` it is the integrated (mip-NeRF) form.

    out = [ exp(-var/2) * sin(2*pi*x*f) , exp(-var/2) * sin(2*pi*x*f + pi/2) , x ]
    with f = 2**linspace(min_exp, max_exp, num_frequencies), coordinate-major / frequency-minor,
    var = diag(cov) * f**2, raw input appended LAST.
    """

    def __init__(self, in_dim, num_frequencies, min_freq_exp, max_freq_exp, include_input=False, implementation="torch"):
        super().__init__(in_dim)
        self.num_frequencies = num_frequencies
        self.min_freq = min_freq_exp
        self.max_freq = max_freq_exp
        self.include_input = include_input

    def get_out_dim(self) -> int:
        out_dim = self.in_dim * self.num_frequencies * 2
        if self.include_input:
            out_dim += self.in_dim
        return out_dim

    def forward(self, in_tensor: Tensor, covs: Optional[Tensor] = None) -> Tensor:
        scaled_in_tensor = 2 * torch.pi * in_tensor
        freqs = 2 ** torch.linspace(self.min_freq, self.max_freq, self.num_frequencies, device=in_tensor.device)
        scaled_inputs = scaled_in_tensor[..., None] * freqs
        scaled_inputs = scaled_inputs.view(*scaled_inputs.shape[:-2], -1)
        if covs is None:
            encoded_inputs = torch.sin(torch.cat([scaled_inputs, scaled_inputs + torch.pi / 2.0], dim=-1))
        else:
            input_var = torch.diagonal(covs, dim1=-2, dim2=-1)[..., :, None] * freqs[None, :] ** 2
            input_var = input_var.reshape((*input_var.shape[:-2], -1))
            encoded_inputs = expected_sin(
                torch.cat([scaled_inputs, scaled_inputs + torch.pi / 2.0], dim=-1), torch.cat(2 * [input_var], dim=-1)
            )
        if self.include_input:
            encoded_inputs = torch.cat([encoded_inputs, in_tensor], dim=-1)
        return encoded_inputs


class SHEncoding(Encoding):  # imported (model.py:20) but never used by the reference
    def __init__(self, levels: int = 4, implementation="torch"):
        super().__init__(in_dim=3)
        raise NotImplementedError("SHEncoding is not used by the reference hot path")
