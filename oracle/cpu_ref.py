"""CPU oracle: plain-PyTorch fp32 restatement of the reflect-sampling-nerf hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (`reflect_sampling_nerf_amd/`)
imports this file; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may use it, and only as the checker / the timed CPU baseline.

What it restates (all paths relative to /root/reference/reflect_sampling_nerf/):
  * reflect_sampling_nerf_field.py:90-207       (get_blob, contract, get_density, heads,
                                                 get_mid, get_inf_color, get_reflection)
  * reflect_sampling_nerf_components.py:14-140  (ReciprocalSampler, IntegratedSHEncoding)
  * reflect_sampling_nerf_model.py:93-132       (construction constants)
  * reflect_sampling_nerf_model.py:142-344      (get_outputs: order of ops, detach/clip points)
and, because the reference calls them, the nerfstudio 0.3.x primitives N1-N12 of SURVEY.md
§8(a) (MLP, integrated positional encoding, samplers, weights, renderers).  nerfstudio itself
(reference pyproject.toml:6, `nerfstudio >= 0.3.0`, un-vendored) is not available offline, so
those primitives are restated from their published semantics:

    PARITY UNPINNED at the nerfstudio boundary.

Pinned part: tests/test_oracle_golden.py checks this file against golden vectors produced by
running the *reference's own modules* (imported from /root/reference on top of the import
shim in oracle/ns_shim/) -- see oracle/make_golden.py.  That pins rows F1-F13 exactly and
N1-N12 up to the shim's (independent, class-shaped) restatement of nerfstudio.
tests/test_oracle_first_principles.py checks the nerfstudio-side pieces (N3, N5, N7-N10, F2) against formulations
that share no code with this file (Monte-Carlo moments, the front-to-back recurrence, empirical sampling
distributions, autograd Jacobians): evidence, not a pin -- the label above stands.

Everything is functional: explicit tensors in, tensors out, no nerfstudio types.  Parameters
are a dict keyed by the reference Field's state_dict names ("mlp_base.layers.0.weight", ...).
The debug prints / host syncs of the reference (model.py:230,263-265,342) are not reproduced.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

# --------------------------------------------------------------------------------------
# specs
# --------------------------------------------------------------------------------------


@dataclass
class FieldSpec:
    """Constructor knobs of the reference Field (field.py:36-47) + the position encoding the
    model builds for it (model.py:98-100)."""

    num_layers: int = 8
    width: int = 256
    skip: Tuple[int, ...] = (4,)
    mid_width: int = 128
    density_bias: float = 0.5
    num_freqs: int = 16
    min_freq_exp: float = 0.0
    max_freq_exp: float = 16.0

    @property
    def in_dim(self) -> int:
        return 3 * self.num_freqs * 2 + 3


@dataclass
class ModelSpec:
    """Sample counts (model.py:46-54) and reflect-ray constants (model.py:111-114)."""

    num_coarse: int = 128
    num_fine: int = 128
    num_reflect_coarse: int = 64
    num_reflect_fine: int = 64
    reflect_tan: float = 0.25
    reflect_far: float = 2.0**8
    histogram_padding: float = 0.01  # nerfstudio PDFSampler default


def init_params(fs: FieldSpec, seed: int = 0, density_bias_shift: float = 0.0) -> Dict[str, Tensor]:
    """Random-init parameters with the reference Field's names/shapes (default nn.Linear init)."""
    g = torch.Generator().manual_seed(seed)

    def linear(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound  # == kaiming_uniform(a=sqrt(5))
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return w, b

    P: Dict[str, Tensor] = {}
    W, D = fs.width, fs.in_dim
    for i in range(fs.num_layers):
        if fs.num_layers == 1:
            in_f = D
        elif i == 0:
            in_f = D
        elif i in fs.skip and i < fs.num_layers - 1:
            in_f = W + D
        else:
            in_f = W
        P[f"mlp_base.layers.{i}.weight"], P[f"mlp_base.layers.{i}.bias"] = linear(W, in_f)
    for name, out_f, in_f in [
        ("field_output_density", 1, W),
        ("field_output_low", 3, W),
        ("field_output_bottleneck", W, W),
    ]:
        P[f"{name}.net.weight"], P[f"{name}.net.bias"] = linear(out_f, in_f)
    P["mlp_mid.layers.0.weight"], P["mlp_mid.layers.0.bias"] = linear(fs.mid_width, 34 + W)
    for name, out_f, in_f in [
        ("field_output_mid", 3, fs.mid_width),
        ("field_output_normals", 3, W),
        ("field_output_roughness", 1, W),
        ("field_output_diff", 3, W),
        ("field_output_tint", 3, W),
    ]:
        P[f"{name}.net.weight"], P[f"{name}.net.bias"] = linear(out_f, in_f)
    P["field_output_density.net.bias"] = P["field_output_density.net.bias"] + density_bias_shift
    return P


# --------------------------------------------------------------------------------------
# N3: conical frustum -> Gaussian            (field.py:90-96 -> Frustums.get_gaussian_blob)
# --------------------------------------------------------------------------------------


def gaussian_blob(origins: Tensor, directions: Tensor, pixel_area: Tensor, t0: Tensor, t1: Tensor):
    """origins/directions [R,3], pixel_area [R,1], t0/t1 [R,S] -> mean [R,S,3], cov [R,S,3,3]."""
    o = origins[:, None, :]
    d = directions[:, None, :]
    radius = (torch.sqrt(pixel_area) / 1.7724538509055159)[:, None, :]  # [R,1,1]
    starts, ends = t0[..., None], t1[..., None]
    mu = (starts + ends) / 2.0
    hw = (ends - starts) / 2.0
    mean = o + d * (mu + (2.0 * mu * hw**2.0) / (3.0 * mu**2.0 + hw**2.0))
    var_t = (hw**2) / 3 - (4 / 15) * ((hw**4 * (12 * mu**2 - hw**2)) / (3 * mu**2 + hw**2) ** 2)
    var_r = radius**2 * ((mu**2) / 4 + (5 / 12) * hw**2 - 4 / 15 * (hw**4) / (3 * mu**2 + hw**2))
    ddT = d[..., :, None] * d[..., None, :]
    eye = torch.eye(3)
    dmag = torch.clamp(torch.sum(d**2, dim=-1, keepdim=True), min=1e-10)
    null = eye - d[..., :, None] * (d / dmag)[..., None, :]
    cov = var_t[..., None] * ddT + var_r[..., None] * null
    return mean.expand(*t0.shape, 3), cov


# --------------------------------------------------------------------------------------
# F2: contraction of Gaussians                                         (field.py:98-119)
# --------------------------------------------------------------------------------------


def contract(mean: Tensor, cov: Tensor):
    n2 = torch.sum(mean**2, dim=-1, keepdim=True)
    n = torch.sqrt(n2)
    outside = n > 1
    mean_c = torch.where(outside, (2 * n - 1) / n2 * mean, mean)
    n_, n2_ = n.unsqueeze(-1), n2.unsqueeze(-1)
    outer = mean[..., :, None] * mean[..., None, :] / n2_
    eye = torch.eye(3).expand(outer.shape)
    jac = torch.where(outside[..., None], ((2 * n_ - 2) * (eye - outer) + eye) / n2_, eye)
    cov_c = torch.matmul(torch.matmul(jac, cov), jac)
    diag = torch.relu(torch.diagonal(cov_c, dim1=-2, dim2=-1))
    cov_c = cov_c.clone()
    for i in range(3):
        cov_c[..., i, i] = diag[..., i]
    return mean_c, cov_c


# --------------------------------------------------------------------------------------
# N2: integrated positional encoding                        (model.py:98-100, field.py:129)
# --------------------------------------------------------------------------------------


def frequencies(fs: FieldSpec) -> Tensor:
    return 2 ** torch.linspace(fs.min_freq_exp, fs.max_freq_exp, fs.num_freqs)


def ipe(fs: FieldSpec, mean: Tensor, cov_diag: Optional[Tensor]) -> Tensor:
    """mean [...,3], cov_diag [...,3] (diag of Sigma) -> [..., 6*F+3]; raw input appended last."""
    freqs = frequencies(fs)
    scaled = (2 * torch.pi * mean)[..., None] * freqs
    scaled = scaled.reshape(*scaled.shape[:-2], -1)
    arg = torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1)
    if cov_diag is None:
        enc = torch.sin(arg)
    else:
        var = cov_diag[..., :, None] * freqs[None, :] ** 2
        var = var.reshape(*var.shape[:-2], -1)
        enc = torch.exp(-0.5 * torch.cat([var, var], dim=-1)) * torch.sin(arg)
    return torch.cat([enc, mean], dim=-1)


# --------------------------------------------------------------------------------------
# N1 + F3: trunk MLP and density                                      (field.py:54-62,122-137)
# --------------------------------------------------------------------------------------


def trunk(P: Dict[str, Tensor], fs: FieldSpec, x_in: Tensor) -> Tensor:
    x = x_in
    L = fs.num_layers
    for i in range(L):
        if i in fs.skip and 0 < i < L - 1:
            x = torch.cat([x_in, x], dim=-1)
        elif i in fs.skip and i == L - 1 and L > 1:
            x = torch.cat([x_in, x], dim=-1)  # the reference would raise a shape error here
        x = F.linear(x, P[f"mlp_base.layers.{i}.weight"], P[f"mlp_base.layers.{i}.bias"])
        if i < L - 1:
            x = torch.relu(x)
    return torch.relu(x)  # out_activation=ReLU (field.py:59)


def head(P, name: str, emb: Tensor) -> Tensor:
    return F.linear(emb, P[f"{name}.net.weight"], P[f"{name}.net.bias"])


def density_from_encoding(P, fs: FieldSpec, enc: Tensor):
    emb = trunk(P, fs, enc)
    raw = head(P, "field_output_density", emb)
    return F.softplus(raw + fs.density_bias), emb, raw


# --------------------------------------------------------------------------------------
# F11: roughness-attenuated 34-term real SH                        (components.py:38-140)
# --------------------------------------------------------------------------------------

_C1 = 0.48860251190291992
_C2 = (1.09254843059207907, 0.31539156525252001, 0.54627421529603953)
_C4 = (2.50334294179670453, 1.77013076977993053, 0.94617469575756001, 0.66904654355728916,
       0.1057855469152043038, 0.473087347878780009, 0.62583573544917613)
_C8 = (5.83141328139863895, 1.06466553211908514, 3.44991062209810801, 1.91366609903732278,
       1.23526615529554407, 0.91230451686981894, 0.1090412458987799555, 0.0090867704915649962938,
       0.456152258434909470, 0.478416524759330697, 0.53233276605954257, 0.72892666017482986)
SH_BAND_SLICES = ((0, 3, 1.0), (3, 8, 3.0), (8, 17, 10.0), (17, 34, 36.0))


def sh34_basis(d: Tensor) -> Tensor:
    """Bands l = 1, 2, 4, 8 of the real SH basis as polynomials in (x,y,z); [...,3] -> [...,34]."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    x2, y2, z2 = x**2, y**2, z**2
    xy, xz, yz = x * y, x * z, y * z
    a = x2 - y2  # x^2 - y^2
    p = 3 * x2 - y2  # Im-part helper of (x+iy)^3 / y
    q = x2 - 3 * y2  # Re-part helper of (x+iy)^3 / x
    z4 = z**4
    x4, y4 = x**4, y**4
    im5 = y4 - 10 * x2 * y2 + 5 * x4  # Im (x+iy)^5 / y
    re5 = x4 - 10 * x2 * y2 + 5 * y4  # Re (x+iy)^5 / x
    im7 = (x2 - 5 * y2) * 7 * x4 + (21 * x2 - y2) * y4  # Im (x+iy)^7 / y
    re7 = (x2 - 21 * y2) * x4 + (5 * x2 - y2) * 7 * y4  # Re (x+iy)^7 / x
    re4 = x2 * q - y2 * p  # Re (x+iy)^4
    t6 = 143 * z4 * z2 - 143 * z4 + 33 * z2 - 1
    t7 = 715 * z4 * z2 - 1001 * z4 + 385 * z2 - 35
    t5 = 39 * z4 - 26 * z2 + 3
    t4 = 65 * z4 - 26 * z2 + 1
    c = [
        _C1 * y, _C1 * z, _C1 * x,
        _C2[0] * xy, _C2[0] * yz, _C2[1] * (3 * z2 - 1), _C2[0] * xz, _C2[2] * a,
        _C4[0] * xy * a, _C4[1] * yz * p, _C4[2] * xy * (7 * z2 - 1), _C4[3] * yz * (7 * z2 - 3),
        _C4[4] * (35 * z4 - 30 * z2 + 3), _C4[3] * xz * (7 * z2 - 3), _C4[5] * a * (7 * z2 - 1),
        _C4[1] * xz * q, _C4[6] * re4,
        _C8[0] * xy * (x2 * x4 - 7 * x4 * y2 + 7 * x2 * y4 - y2 * y4),
        _C8[0] * yz * im7,
        _C8[1] * xy * (15 * z2 - 1) * (3 * x4 - 10 * x2 * y2 + 3 * y4),
        _C8[2] * yz * (5 * z2 - 1) * im5,
        _C8[3] * xy * t4 * a,
        _C8[4] * yz * t5 * p,
        _C8[5] * xy * t6,
        _C8[6] * yz * t7,
        _C8[7] * (6435 * z4 * z4 - 12012 * z4 * z2 + 6930 * z4 - 1260 * z2 + 35),
        _C8[6] * xz * t7,
        _C8[8] * t6 * a,
        _C8[4] * xz * t5 * q,
        _C8[9] * t4 * re4,
        _C8[2] * xz * (5 * z2 - 1) * re5,
        _C8[10] * (15 * z2 - 1) * (x2 * re5 - y2 * im5),
        _C8[0] * xz * re7,
        _C8[11] * (x2 * re7 - y2 * im7),
    ]
    return torch.stack(c, dim=-1)


def integrated_sh(d: Tensor, roughness: Tensor) -> Tensor:
    """d [...,3], roughness [...,1] -> [...,34]; no gradient flows through it (components.py:52;
    callers pass roughness.detach(), model.py:174)."""
    with torch.no_grad():
        out = sh34_basis(d).clone()
        r = roughness.detach()
        for lo, hi, k in SH_BAND_SLICES:
            out[..., lo:hi] = out[..., lo:hi] * torch.exp(-r * k)
    return out


# --------------------------------------------------------------------------------------
# F4, F6-F10: heads                                                      (field.py:139-207)
# --------------------------------------------------------------------------------------


def pred_normals(P, emb: Tensor) -> Tensor:
    n = F.normalize(head(P, "field_output_normals", emb), dim=-1)  # PredNormalsFieldHead
    return F.normalize(-n, dim=-1)  # field.py:142-143


def mid_color(P, fs: FieldSpec, sh: Tensor, emb: Tensor) -> Tensor:
    """get_mid / get_low / tail of get_inf_color: bottleneck -> Linear(34+W -> mid)+ReLU -> sigmoid RGB."""
    b = head(P, "field_output_bottleneck", emb)
    h = torch.relu(F.linear(torch.cat([sh, b], dim=-1), P["mlp_mid.layers.0.weight"], P["mlp_mid.layers.0.bias"]))
    return torch.sigmoid(head(P, "field_output_mid", h))


def inf_color(P, fs: FieldSpec, directions: Tensor, sqradius: Tensor) -> Tensor:
    """field.py:190-201: colour 'at infinity'; mean = 2d, Sigma = 0.6 r^2 (I - d d^T), NO contraction,
    SH inputs zeroed."""
    mean = 2 * directions
    cov_diag = 0.6 * sqradius * (1.0 - directions * directions)
    _, emb, _ = density_from_encoding(P, fs, ipe(fs, mean, cov_diag))
    sh0 = torch.zeros(*emb.shape[:-1], 34)
    return mid_color(P, fs, sh0, emb)


# --------------------------------------------------------------------------------------
# N5, N10, N11: weights and renderers
# --------------------------------------------------------------------------------------


def weights_from_density(sigma: Tensor, t0: Tensor, t1: Tensor) -> Tensor:
    """sigma [R,S,1], t0/t1 [R,S] -> weights [R,S,1] (RaySamples.get_weights)."""
    dd = (t1 - t0)[..., None] * sigma
    alpha = 1 - torch.exp(-dd)
    T = torch.cumsum(dd[..., :-1, :], dim=-2)
    T = torch.cat([torch.zeros(T.shape[0], 1, 1), T], dim=-2)
    return torch.nan_to_num(alpha * torch.exp(-T))


def composite_rgb(rgb: Tensor, w: Tensor, background: Optional[Tensor], training: bool) -> Tensor:
    """RGBRenderer.forward: background None == "random" (no blend); eval mode nan_to_num + clamp."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(w * rgb, dim=-2)
    if background is not None:
        comp = comp + background * (1.0 - torch.sum(w, dim=-2))
    if not training:
        comp = torch.clamp(comp, min=0.0, max=1.0)
    return comp


def median_depth(w: Tensor, t0: Tensor, t1: Tensor) -> Tensor:
    steps = (t0 + t1) / 2
    cw = torch.cumsum(w[..., 0], dim=-1)
    split = torch.ones(*w.shape[:-2], 1) * 0.5
    idx = torch.clamp(torch.searchsorted(cw, split, side="left"), 0, steps.shape[-1] - 1)
    return torch.gather(steps, dim=-1, index=idx)


def render_normals(n: Tensor, w: Tensor) -> Tensor:
    v = torch.sum(w * n, dim=-2)
    return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)


# --------------------------------------------------------------------------------------
# N7-N9 + F12: samplers
# --------------------------------------------------------------------------------------


def spacing_fns(kind: str, tan: float = 1.0):
    if kind == "uniform":
        return (lambda x: x), (lambda x: x)
    if kind == "reciprocal":  # components.py:32-33
        return (lambda x: x / (1 / tan + x)), (lambda x: x / tan / (1 - x))
    raise ValueError(kind)


def spaced_bins(kind: str, tan: float, nears: Tensor, fars: Tensor, num_samples: int, t_rand: Optional[Tensor]):
    """-> (spacing_bins [R or 1, S+1], euclidean_bins [R, S+1]); t_rand [R,S+1] = stratified jitter or None."""
    fn, fn_inv = spacing_fns(kind, tan)
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, ...]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    s_near, s_far = fn(nears), fn(fars)
    eucl = fn_inv(bins * s_far + (1 - bins) * s_near)
    return bins.expand(nears.shape[0], num_samples + 1), eucl


def pdf_bins(kind: str, tan: float, nears: Tensor, fars: Tensor, w: Tensor, spacing_bins: Tensor, num_samples: int,
             u_rand: Optional[Tensor], histogram_padding: float = 0.01, eps: float = 1e-5):
    """Inverse-CDF resampling (PDFSampler, include_original=False).  w [R,S_in,1], spacing_bins [R,S_in+1]
    -> (new spacing bins [R,S_out+1], euclidean [R,S_out+1]); u_rand [R,S_out+1] in [0,1) or None."""
    fn, fn_inv = spacing_fns(kind, tan)
    nb = num_samples + 1
    wp = w[..., 0] + histogram_padding
    wsum = torch.sum(wp, dim=-1, keepdim=True)
    pad = torch.relu(eps - wsum)
    wp = wp + pad / wp.shape[-1]
    wsum = wsum + pad
    pdf = wp / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    if u_rand is not None:
        u = u.expand(*cdf.shape[:-1], nb) + u_rand / nb
    else:
        u = (u + 1.0 / (2 * nb)).expand(*cdf.shape[:-1], nb)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, side="right")
    hi = spacing_bins.shape[-1] - 1
    below = torch.clamp(inds - 1, 0, hi)
    above = torch.clamp(inds, 0, hi)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(spacing_bins, -1, below), torch.gather(spacing_bins, -1, above)
    t = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    bins = (b0 + t * (b1 - b0)).detach()
    s_near, s_far = fn(nears), fn(fars)
    eucl = fn_inv(bins * s_far + (1 - bins) * s_near)
    return bins, eucl


# --------------------------------------------------------------------------------------
# one sampling level: field evaluation over [R,S] samples + compositing
# --------------------------------------------------------------------------------------


def field_level(P, fs: FieldSpec, origins, directions, pixel_area, eucl_bins, training: bool, want_normals: bool,
                mean_override: Optional[Tensor] = None):
    """Stages A/B/F/G of SURVEY §3.3 up to the per-sample colour.  Returns a dict of per-sample tensors.
    want_normals => analytic normals via autograd (training only; field.py:125-127,146-147).
    mean_override (tests): contracted Gaussian means [n,S,3] used instead of the computed ones (which carry no
    gradient: origins, directions and bins are constants of the graph) -- puts two pipelines on bit-identical
    positions; the covariance (which does carry gradient on the reflect levels) is still computed here."""
    t0, t1 = eucl_bins[..., :-1], eucl_bins[..., 1:]
    mean, cov = gaussian_blob(origins, directions, pixel_area, t0, t1)
    mean, cov = contract(mean, cov)
    if mean_override is not None:
        mean = mean_override.detach().reshape(mean.shape)
    if want_normals and training:
        mean = mean.detach().requires_grad_(True)  # field.py:126 (cov was computed before: constants)
    enc = ipe(fs, mean, torch.diagonal(cov, dim1=-2, dim2=-1))
    sigma, emb, raw = density_from_encoding(P, fs, enc)
    out = {"sigma": sigma, "emb": emb, "t0": t0, "t1": t1}
    pn = pred_normals(P, emb)
    out["pred_normals"] = pn
    if want_normals and training:
        g = torch.autograd.grad(raw, mean, grad_outputs=torch.ones_like(raw), retain_graph=True)[0]
        out["normals"] = -F.normalize(g, dim=-1)
    else:
        out["normals"] = pn
    d = directions[:, None, :].expand(*t0.shape, 3)
    out["n_dot_d"] = torch.sum(d * pn, dim=-1, keepdim=True)
    out["diff"] = torch.sigmoid(head(P, "field_output_diff", emb))
    out["tint"] = torch.sigmoid(head(P, "field_output_tint", emb))
    rough_raw = head(P, "field_output_roughness", emb)
    out["rough_raw"] = rough_raw
    sh = integrated_sh(d, F.softplus(rough_raw))
    out["mid"] = mid_color(P, fs, sh, emb)
    out["color"] = out["diff"] + out["tint"] * out["mid"]
    return out


def get_outputs(P: Dict[str, Tensor], fs: FieldSpec, ms: ModelSpec, origins: Tensor, directions: Tensor,
                pixel_area: Tensor, nears: Tensor, fars: Tensor, training: bool = False,
                jitter: Optional[Dict[str, Tensor]] = None, bins: Optional[Dict[str, Tensor]] = None,
                record_bins: Optional[Dict[str, Tensor]] = None,
                means: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """Restatement of ReflectSamplingNeRFModel.get_outputs (model.py:142-344).

    jitter (training only): {"coarse": [R,Sc+1], "fine": [R,Sf+1], "reflect_coarse": [M,Src+1],
    "reflect_fine": [M,Srf+1]} uniform [0,1) draws that replace the samplers' torch.rand calls.
    bins (tests): {"<level>_spacing": [n,S+1], "<level>_euclid": [n,S+1]} for level in coarse / fine / reflect_coarse /
    reflect_fine replaces that level's sampler output (the samplers' outputs are constants of the graph: PDFSampler
    detaches, model.py:182,317), so that two pipelines can be compared on identical sample positions.
    record_bins: a dict that receives every level's bins under the same keys.
    means (tests): {"coarse" | "fine" | "reflect_coarse" | "reflect_fine": [n,S,3]} contracted sample means to
    evaluate the field at (field_level's mean_override).
    """
    jitter = jitter or {}
    bins = bins or {}
    means = means or {}

    def level_bins(name, computed):
        sb, eb = computed
        if name + "_euclid" in bins:
            sb, eb = bins[name + "_spacing"], bins[name + "_euclid"]
        if record_bins is not None:
            record_bins[name + "_spacing"], record_bins[name + "_euclid"] = sb.detach().clone(), eb.detach().clone()
        return sb, eb

    white = torch.ones(3)
    jit = (lambda k: jitter[k]) if training else (lambda k: None)

    def jit_reflect(k, mask):
        """reflect-level jitter may be given per original ray [R,S+1]; the masked rows are used then."""
        t = jit(k)
        return t[mask] if (t is not None and t.shape[0] == mask.shape[0] and t.shape[0] != int(mask.sum())) else t

    # A. coarse primary (model.py:148-177)
    sbins_c, ebins_c = level_bins("coarse", spaced_bins("uniform", 1.0, nears, fars, ms.num_coarse, jit("coarse")))
    lc = field_level(P, fs, origins, directions, pixel_area, ebins_c, training, want_normals=True,
                     mean_override=means.get("coarse"))
    w_c = weights_from_density(lc["sigma"], lc["t0"], lc["t1"])
    acc_c = torch.sum(w_c, dim=-2)
    depth_c = median_depth(w_c, lc["t0"], lc["t1"])
    rgb_c = torch.clip(composite_rgb(lc["color"], w_c, white, training), 0.0, 1.0)

    # B. fine primary (model.py:182-211)
    sbins_f, ebins_f = level_bins("fine", pdf_bins("uniform", 1.0, nears, fars, w_c, sbins_c, ms.num_fine, jit("fine"),
                                                   ms.histogram_padding))
    lf = field_level(P, fs, origins, directions, pixel_area, ebins_f, training, want_normals=True,
                     mean_override=means.get("fine"))
    w_f = weights_from_density(lf["sigma"], lf["t0"], lf["t1"])
    acc_f = torch.sum(w_f, dim=-2)
    depth_f = median_depth(w_f, lf["t0"], lf["t1"])
    rgb_f = torch.clip(composite_rgb(lf["color"], w_f, white, training), 0.0, 1.0)

    # C. per-ray surface attributes (model.py:215-229)
    diff_f = composite_rgb(lf["diff"], w_f, white, training).detach()
    tint_f = composite_rgb(lf["tint"], w_f, None, training).detach()
    pn_f = render_normals(lf["pred_normals"], w_f).detach()
    n_dot_d = torch.sum(pn_f * directions, dim=-1, keepdim=True).detach()
    roughness = torch.sum(w_f * torch.sigmoid(lf["rough_raw"]), dim=-2)  # NOT detached (model.py:227)
    mask = torch.logical_and(acc_f > 1e-2, n_dot_d < 0).reshape(-1)

    # D. outputs + early-out (model.py:233-260)
    out = {
        "mid_rgb_coarse": rgb_c,
        "mid_rgb_fine": rgb_f,
        "mid_reflect_coarse": white.expand(rgb_f.shape) * (1.0 - acc_f),
        "mid_reflect_fine": white.expand(rgb_f.shape) * (1.0 - acc_f),
        "accumulation_coarse": acc_c.detach(),
        "accumulation_fine": acc_f.detach(),
        "depth_coarse": depth_c.detach(),
        "depth_fine": depth_f.detach(),
        "weights_coarse": w_c.detach(),
        "weights_fine": w_f.detach(),
        "pred_normals_coarse": lc["pred_normals"],
        "pred_normals_fine": lf["pred_normals"],
        "normals_coarse": lc["normals"].detach(),
        "normals_fine": lf["normals"].detach(),
        "n_dot_d_coarse": lc["n_dot_d"],
        "n_dot_d_fine": lf["n_dot_d"],
        "diff": diff_f,
        "tint": tint_f,
        "roughness": roughness,
        "mask": mask,
    }
    if not bool(mask.any()):
        return out

    # E. secondary rays (model.py:267-290)
    o2 = (origins[mask] + depth_f[mask] * directions[mask]).detach()
    d2 = F.normalize(directions[mask] - 2 * n_dot_d[mask] * pn_f[mask], dim=-1).detach()
    sqradius = 2 * torch.abs(n_dot_d[mask]) * roughness[mask] ** 2
    pa2 = torch.pi * sqradius
    near2 = torch.zeros_like(nears[mask])  # zeros_like(...) * self.near == 0 (model.py:287)
    far2 = torch.ones_like(fars[mask]) * ms.reflect_far
    background = inf_color(P, fs, d2, sqradius)

    # F. reflect coarse (model.py:292-313)
    sb_rc, eb_rc = level_bins("reflect_coarse", spaced_bins("reciprocal", ms.reflect_tan, near2, far2,
                                                            ms.num_reflect_coarse, jit_reflect("reflect_coarse", mask)))
    lrc = field_level(P, fs, o2, d2, pa2, eb_rc, training, want_normals=False, mean_override=means.get("reflect_coarse"))
    w_rc = weights_from_density(lrc["sigma"], lrc["t0"], lrc["t1"]).detach()
    comp_rc = composite_rgb(lrc["color"], w_rc, background, training)
    rc = out["mid_reflect_coarse"].clone()
    rc[mask] = torch.clip(diff_f[mask] + tint_f[mask] * comp_rc, 0.0, 1.0)
    out["mid_reflect_coarse"] = rc

    # G. reflect fine (model.py:317-342)
    sb_rf, eb_rf = level_bins("reflect_fine", pdf_bins("reciprocal", ms.reflect_tan, near2, far2, w_rc, sb_rc,
                                                       ms.num_reflect_fine, jit_reflect("reflect_fine", mask),
                                                       ms.histogram_padding))
    lrf = field_level(P, fs, o2, d2, pa2, eb_rf, training, want_normals=False, mean_override=means.get("reflect_fine"))
    w_rf = weights_from_density(lrf["sigma"], lrf["t0"], lrf["t1"]).detach()
    comp_rf = composite_rgb(lrf["color"], w_rf, background, training)
    rf = out["mid_reflect_fine"].clone()
    rf[mask] = torch.clip(diff_f[mask] + tint_f[mask] * comp_rf, 0.0, 1.0)
    out["mid_reflect_fine"] = rf
    out["depth_reflect_fine"] = median_depth(w_rf, lrf["t0"], lrf["t1"])
    return out


# --------------------------------------------------------------------------------------
# single-level entry used as the CPU baseline of bench.py (BASELINE.json configs[0], configs[1])
# --------------------------------------------------------------------------------------


def render_level(P, fs: FieldSpec, origins, directions, pixel_area, nears, fars, num_samples: int):
    """One eval-mode sampling level: uniform bins -> field -> weights -> RGB (white bg), accumulation, depth."""
    _, eb = spaced_bins("uniform", 1.0, nears, fars, num_samples, None)
    lv = field_level(P, fs, origins, directions, pixel_area, eb, training=False, want_normals=False)
    w = weights_from_density(lv["sigma"], lv["t0"], lv["t1"])
    rgb = torch.clip(composite_rgb(lv["color"], w, torch.ones(3), False), 0.0, 1.0)
    return {"rgb": rgb, "accumulation": torch.sum(w, dim=-2), "depth": median_depth(w, lv["t0"], lv["t1"]),
            "weights": w, "level": lv}



# --------------------------------------------------------------------------------------
# get_loss_dict                                                          (model.py:346-430)
# --------------------------------------------------------------------------------------
LOSS_COEFFICIENTS = {  # model.py:56-69 (the four normal / orientation values are what the pipeline restores after the
    # 50-step warm-up, pipeline.py:79-91); the four "low" terms are configured but never computed (model.py:348-430)
    "loss_low_coarse": 0.1, "loss_low_fine": 0.1, "loss_mid_coarse": 1.0, "loss_mid_fine": 1.0,
    "loss_reflect_low_coarse": 0.1, "loss_reflect_low_fine": 0.1, "loss_reflect_mid_coarse": 1.0,
    "loss_reflect_mid_fine": 1.0, "predicted_normal_loss_coarse": 3e-5, "predicted_normal_loss_fine": 3e-4,
    "orientation_loss_coarse": 1e-2, "orientation_loss_fine": 1e-1,
}


def loss_dict(out: Dict[str, Tensor], image: Tensor, coefficients: Optional[Dict[str, float]] = None) -> Dict[str, Tensor]:
    """The eight scaled loss terms.  RGBRenderer.blend_background_for_loss_computation with the white background
    (N10) leaves the prediction alone and blends the ground truth only when it carries alpha; MSELoss is the mean over
    all elements; the normal / orientation terms are SUMS weighted by the (detached) weights (model.py:403-407)."""
    coefficients = LOSS_COEFFICIENTS if coefficients is None else coefficients
    if image.shape[-1] == 4:
        image = image[..., :3] * image[..., 3:] + (1.0 - image[..., 3:])
    mse = lambda a, b: torch.mean((a - b) ** 2)  # noqa: E731
    terms = {
        "loss_mid_coarse": mse(image, out["mid_rgb_coarse"]),
        "loss_mid_fine": mse(image, out["mid_rgb_fine"]),
        "loss_reflect_mid_coarse": mse(image, out["mid_reflect_coarse"]),
        "loss_reflect_mid_fine": mse(image, out["mid_reflect_fine"]),
    }
    for lvl in ("coarse", "fine"):
        w = out[f"weights_{lvl}"]
        terms[f"predicted_normal_loss_{lvl}"] = torch.sum(
            w * torch.sum((out[f"normals_{lvl}"] - out[f"pred_normals_{lvl}"]) ** 2, dim=-1, keepdim=True))
        ndd = out[f"n_dot_d_{lvl}"]
        terms[f"orientation_loss_{lvl}"] = torch.sum(w * torch.max(torch.zeros_like(ndd), ndd) ** 2)
    return {k: v * coefficients[k] for k, v in terms.items() if k in coefficients}

def synthetic_rays(R: int, seed: int = 0):
    """SURVEY §8(d) synthetic inputs: camera shell of radius 4 looking at the origin."""
    g = torch.Generator().manual_seed(seed)
    o = F.normalize(torch.randn(R, 3, generator=g), dim=-1) * 4.0 + 0.05 * torch.randn(R, 3, generator=g)
    d = F.normalize(-o + 0.3 * torch.randn(R, 3, generator=g), dim=-1)
    pa = torch.full((R, 1), (1.0 / 800.0) ** 2)
    return o, d, pa
