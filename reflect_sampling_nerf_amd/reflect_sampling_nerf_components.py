"""Host-side mirrors of the reference's components (reflect_sampling_nerf_components.py:14-140) and of
the two nerfstudio component configs the model instantiates (NeRFEncoding, Uniform/PDF samplers).

They carry configuration only; the arithmetic runs in librsn_hip.so (rsn_sample_spaced,
rsn_sample_pdf, and the SH-34 / IPE stages fused into rsn_field_forward_*).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn

from . import ops
from ._abi import RSN_SPACING_RECIPROCAL, RSN_SPACING_UNIFORM


class NeRFEncoding(nn.Module):
    """Config of the integrated positional encoding the model builds (reference model.py:98-100).
    Only the (3, 16, 0.0, 16.0, include_input=True) shape is fused into the HIP field kernel."""

    def __init__(self, in_dim: int = 3, num_frequencies: int = 16, min_freq_exp: float = 0.0,
                 max_freq_exp: float = 16.0, include_input: bool = True) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.num_frequencies = num_frequencies
        self.min_freq = min_freq_exp
        self.max_freq = max_freq_exp
        self.include_input = include_input

    def get_out_dim(self) -> int:
        return self.in_dim * self.num_frequencies * 2 + (self.in_dim if self.include_input else 0)

    def frequencies(self) -> Tensor:
        # same expression as nerfstudio's NeRFEncoding so the table is bit-identical to the reference's
        return 2 ** torch.linspace(self.min_freq, self.max_freq, self.num_frequencies)

    def forward(self, in_tensor: Tensor, covs: Optional[Tensor] = None) -> Tensor:
        """[..., 3] (+ covs [..., 3, 3], only the diagonal is used) -> [..., 99] via rsn_ipe_encode: what the Field's
        `self.position_encoding(mean, covs=cov)` returns in the reference (field.py:129-131).  Eval only."""
        if (self.in_dim, self.num_frequencies, self.include_input) != (3, 16, True):
            raise NotImplementedError("only NeRFEncoding(3, 16, ..., include_input=True) is implemented")
        return ops.ipe_encode(in_tensor, covs, self.frequencies())


class IntegratedSHEncoding(nn.Module):
    """Roughness-attenuated real SH, bands l in {1,2,4,8}: 34 channels (components.py:38-140).
    Evaluated inside the HIP field kernel (sh34_attenuated in csrc/rsn_field.hip)."""

    def __init__(self) -> None:
        super().__init__()
        self.in_dim = 3

    def get_out_dim(self) -> int:
        return 34

    @torch.no_grad()
    def pytorch_fwd(self, directions: Tensor) -> Tensor:
        """components.py:52-129: the 34 unattenuated real-SH components of `directions` [..., 3] (same kernel, no roughness)."""
        assert directions.shape[-1] == 3, f"Direction input should have three dimensions. Got {directions.shape[-1]}"
        return ops.sh34_encode(directions, None)

    def forward(self, in_tensor: Tensor, roughness: Optional[Tensor] = None) -> Tensor:
        """in_tensor = directions [..., 3], roughness [..., 1] (or None) -> [..., 34] via rsn_sh34_encode (components.py:52-140;
        no gradient flows through the encoding in the reference either: computed under no_grad)."""
        return ops.sh34_encode(in_tensor, roughness)


@dataclass
class _SamplerSpec:
    num_samples: Optional[int]
    spacing: int
    tan: float = 1.0
    train_stratified: bool = True
    single_jitter: bool = False


class _SpacedSamplerBase(nn.Module):
    """generate bins with rsn_sample_spaced; returns (spacing_bins [R,S+1], euclidean_bins [R,S+1])."""

    def __init__(self, spec: _SamplerSpec) -> None:
        super().__init__()
        self.spec = spec
        self.num_samples = spec.num_samples

    def forward(self, ray_bundle, num_samples: Optional[int] = None, n_dev: Optional[Tensor] = None,
                t_rand: Optional[Tensor] = None):
        S = num_samples or self.num_samples
        R = ray_bundle.origins.shape[0]
        nears = ray_bundle.nears.reshape(R).contiguous().float()
        fars = ray_bundle.fars.reshape(R).contiguous().float()
        if self.spec.train_stratified and self.training and t_rand is None:
            shape = (R, 1) if self.spec.single_jitter else (R, S + 1)
            t_rand = torch.rand(shape, device=nears.device).expand(R, S + 1).contiguous()
        return ops.sample_spaced(R, n_dev, S, self.spec.spacing, self.spec.tan, nears, fars, t_rand)


class UniformSampler(_SpacedSamplerBase):
    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(_SamplerSpec(num_samples, RSN_SPACING_UNIFORM, 1.0, train_stratified, single_jitter))


class ReciprocalSampler(_SpacedSamplerBase):
    """spacing s(t) = t / (1/tan + t), inverse u/tan/(1-u)  (components.py:14-36)."""

    def __init__(self, tan: float = 1.0, num_samples: Optional[int] = None, train_stratified=True,
                 single_jitter=False) -> None:
        super().__init__(_SamplerSpec(num_samples, RSN_SPACING_RECIPROCAL, float(tan), train_stratified,
                                      single_jitter))


class PDFSampler(nn.Module):
    """Inverse-CDF resampling via rsn_sample_pdf (include_original=False only, as the reference uses it)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = False, histogram_padding: float = 0.01) -> None:
        super().__init__()
        if include_original:
            raise NotImplementedError("include_original=True is not used by reflect-sampling-nerf")
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        self.histogram_padding = histogram_padding

    def forward(self, nears: Tensor, fars: Tensor, base: _SpacedSamplerBase, weights: Tensor, spacing_bins: Tensor,
                num_samples: Optional[int] = None, n_dev: Optional[Tensor] = None, u_rand: Optional[Tensor] = None):
        S_out = num_samples or self.num_samples
        R, S_in = weights.shape[0], weights.shape[1]
        if self.train_stratified and self.training and u_rand is None:
            shape = (R, 1) if self.single_jitter else (R, S_out + 1)
            u_rand = torch.rand(shape, device=weights.device).expand(R, S_out + 1).contiguous()
        return ops.sample_pdf(R, n_dev, S_in, S_out, base.spec.spacing, base.spec.tan, self.histogram_padding, nears,
                              fars, weights, spacing_bins, u_rand)
