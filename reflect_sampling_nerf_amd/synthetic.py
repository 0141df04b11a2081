"""Synthetic ray batches for benchmarks and smoke runs: a camera shell of radius 4 looking at
the origin, constant pixel area (1/800)^2.  Input data only -- no arithmetic of the method lives here."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def synthetic_rays(R: int, seed: int = 0):
    """-> origins [R,3], directions [R,3] (unit), pixel_area [R,1]; CPU tensors, seeded torch.Generator."""
    g = torch.Generator().manual_seed(seed)
    o = F.normalize(torch.randn(R, 3, generator=g), dim=-1) * 4.0 + 0.05 * torch.randn(R, 3, generator=g)
    d = F.normalize(-o + 0.3 * torch.randn(R, 3, generator=g), dim=-1)
    pa = torch.full((R, 1), (1.0 / 800.0) ** 2)
    return o, d, pa
