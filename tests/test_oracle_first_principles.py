"""The oracle's restatements of the NERFSTUDIO pieces of the path (nerfstudio is not importable here
and the reference holds no fixtures for them, so the reference-generated golden vectors pin them only through `oracle/ns_shim`)
checked against FIRST PRINCIPLES instead -- formulations that share no code and no algebra with the restatement:

* conical frustum -> Gaussian (mip-NeRF eq. 7-8 in its numerically stable form): Monte-Carlo moments of points drawn uniformly
  from the frustum;
* volume-rendering weights: the front-to-back recurrence T_{i+1} = T_i (1 - alpha_i) in fp64;
* inverse-CDF resampling: the empirical distribution of the resampled bin edges against the padded histogram it is drawn from;
* reciprocal spacing: the pair of functions is an inverse pair, the bins run monotonically from near to far;
* median depth: the returned step is where the cumulative weight first reaches one half;
* contraction of a Gaussian (F2; pinned by the reference's own run as well): J Sigma J^T with J from autograd.

CPU only; test infrastructure (the oracle is never part of the product path).
"""
import math

import torch

from oracle import cpu_ref


def test_conical_frustum_gaussian_matches_monte_carlo_moments():
    g = torch.Generator().manual_seed(0)
    n = 1_000_000
    for (t0, t1, area, d) in ((2.0, 2.5, 1e-2, (0.0, 0.0, 1.0)), (0.5, 3.0, 4e-3, (0.6, 0.0, 0.8)), (4.0, 4.05, 1e-1, (1.0, 0.0, 0.0))):
        dv = torch.tensor(d, dtype=torch.float64)
        o = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
        radius = math.sqrt(area) / math.sqrt(math.pi)  # the cone's radius at distance 1 (pixel footprint of area `area`)
        u = torch.rand(n, 3, generator=g, dtype=torch.float64)
        t = (t0**3 + u[:, 0] * (t1**3 - t0**3)) ** (1.0 / 3.0)  # uniform in volume: density ~ t^2
        rho = radius * t * torch.sqrt(u[:, 1])                   # uniform in the disc of radius `radius * t`
        th = 2 * math.pi * u[:, 2]
        e1 = torch.linalg.cross(dv, torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64))
        e1 = e1 / e1.norm()
        e2 = torch.linalg.cross(dv, e1)
        pts = o + t[:, None] * dv + (rho * torch.cos(th))[:, None] * e1 + (rho * torch.sin(th))[:, None] * e2
        mc_mean = pts.mean(0)
        mc_cov = torch.cov(pts.T)
        mean, cov = cpu_ref.gaussian_blob(o[None].float(), dv[None].float(), torch.tensor([[area]]), torch.tensor([[t0]]),
                                          torch.tensor([[t1]]))
        mean, cov = mean[0, 0].double(), cov[0, 0].double()
        scale = float(mc_cov.diagonal().max())
        assert float((mean - mc_mean).abs().max()) <= 5.0 * math.sqrt(scale / n) + 1e-6, (t0, t1)  # five standard errors
        assert float((cov - mc_cov).abs().max()) <= 1e-2 * scale, (t0, t1, cov, mc_cov)


def test_weights_equal_the_front_to_back_recurrence():
    torch.manual_seed(1)
    R, S = 5, 40
    sigma = torch.rand(R, S, 1) * 8.0
    sigma[2] = 0.0          # empty ray
    sigma[3, 5] = 1e4       # an opaque sample
    edges = torch.cumsum(torch.rand(R, S + 1) * 0.2 + 1e-3, dim=-1) + 2.0
    t0, t1 = edges[:, :-1], edges[:, 1:]
    w = cpu_ref.weights_from_density(sigma, t0, t1)[..., 0].double()
    ref = torch.zeros(R, S, dtype=torch.float64)
    for r in range(R):
        T = 1.0
        for s in range(S):
            alpha = 1.0 - math.exp(-float(sigma[r, s, 0]) * float(t1[r, s] - t0[r, s]))
            ref[r, s] = T * alpha
            T *= 1.0 - alpha
        total = 1.0 - math.exp(-float(((t1[r] - t0[r]) * sigma[r, :, 0]).double().sum()))
        assert abs(float(w[r].sum()) - total) <= 2e-6
    assert float((w - ref).abs().max()) <= 2e-6
    assert float(w[2].abs().max()) == 0.0 and float(w[3, 6:].abs().max()) <= 1e-12


def test_pdf_resampling_follows_the_padded_histogram():
    torch.manual_seed(2)
    S_in, S_out, R = 12, 31, 20000
    w1 = torch.tensor([0.0, 0.0, 0.02, 0.3, 0.4, 0.05, 0.0, 0.0, 0.1, 0.0, 0.0, 0.0])
    w = w1[None, :, None].expand(R, S_in, 1)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    sb, _ = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S_in, None)
    u = torch.rand(R, S_out + 1)
    bins, eucl = cpu_ref.pdf_bins("uniform", 1.0, nears, fars, w, sb, S_out, u)
    assert bins.shape == (R, S_out + 1) and bool((bins[:, 1:] >= bins[:, :-1]).all())
    assert float(bins.min()) >= 0.0 and float(bins.max()) <= 1.0
    assert torch.allclose(eucl, 2.0 + 4.0 * bins, atol=1e-5)
    pdf = (w1 + 0.01) / (w1 + 0.01).sum()  # histogram_padding = 0.01 (PDFSampler's default)
    cdf = torch.cat([torch.zeros(1), torch.cumsum(pdf, 0)])
    edges = torch.linspace(0.0, 1.0, S_in + 1)
    emp = torch.stack([(bins <= e).float().mean() for e in edges])
    assert float((emp - cdf).abs().max()) <= 0.01, (emp, cdf)
    # without jitter: bin edge j sits at the (j + 1/2) / (S_out + 1) quantile of the same histogram
    bins0, _ = cpu_ref.pdf_bins("uniform", 1.0, nears[:1], fars[:1], w[:1], sb[:1], S_out, None)
    q = (torch.arange(S_out + 1) + 0.5) / (S_out + 1)
    k = torch.clamp(torch.searchsorted(cdf, q, right=True) - 1, 0, S_in - 1)
    expect = edges[k] + (q - cdf[k]) / (cdf[k + 1] - cdf[k]) * (edges[k + 1] - edges[k])
    assert float((bins0[0] - expect).abs().max()) <= 1e-5


def test_reciprocal_spacing_is_an_inverse_pair_and_bins_are_monotone():
    for tan in (1.0, 0.25, 3.0):
        fn, fn_inv = cpu_ref.spacing_fns("reciprocal", tan)
        x = torch.linspace(0.01, 50.0, 200, dtype=torch.float64)
        assert float((fn_inv(fn(x)) - x).abs().max()) <= 1e-9 * 50
        nears, fars = torch.tensor([[0.05], [2.0]]), torch.tensor([[4.0], [60.0]])
        t_rand = torch.rand(2, 17, generator=torch.Generator().manual_seed(3))
        for tr in (None, t_rand):
            sb, eb = cpu_ref.spaced_bins("reciprocal", tan, nears, fars, 16, tr)
            assert bool((eb[:, 1:] > eb[:, :-1]).all()) and bool((sb[:, 1:] >= sb[:, :-1]).all())
            if tr is None:
                assert torch.allclose(eb[:, 0], nears[:, 0], rtol=1e-5) and torch.allclose(eb[:, -1], fars[:, 0], rtol=1e-4)
            else:
                assert bool((eb[:, 0] >= nears[:, 0] * (1 - 1e-5)).all()) and bool((eb[:, -1] <= fars[:, 0] * (1 + 1e-4)).all())


def test_median_depth_is_where_the_cumulative_weight_reaches_one_half():
    torch.manual_seed(4)
    R, S = 64, 24
    w = torch.rand(R, S, 1) ** 4
    w = w / w.sum(dim=-2, keepdim=True) * torch.rand(R, 1, 1)  # accumulations in (0, 1)
    w[0] = 0.0
    edges = torch.cumsum(torch.rand(R, S + 1) * 0.3 + 0.01, dim=-1) + 2.0
    t0, t1 = edges[:, :-1], edges[:, 1:]
    dep = cpu_ref.median_depth(w, t0, t1)
    mid = (t0 + t1) / 2
    for r in range(R):
        c, k = 0.0, S - 1  # never reached: the last sample
        for s in range(S):
            c += float(w[r, s, 0])
            if c >= 0.5:
                k = s
                break
        assert float(dep[r, 0]) == float(mid[r, k]), r


def test_contraction_covariance_is_the_autograd_jacobian_sandwich():
    torch.manual_seed(5)
    means = torch.randn(8, 3, dtype=torch.float64) * 2.0
    means[0] = torch.tensor([0.2, 0.1, -0.3])  # inside the unit ball: identity
    A = torch.randn(8, 3, 3, dtype=torch.float64) * 0.1
    cov = A @ A.transpose(-1, -2)

    def f(x):
        n = x.norm()
        return x if float(n.detach()) <= 1 else (2 - 1 / n) * x / n

    mc, cc = cpu_ref.contract(means.float(), cov.float())
    for k in range(8):
        J = torch.autograd.functional.jacobian(f, means[k])
        ref = J @ cov[k] @ J.T
        assert float((mc[k].double() - f(means[k])).abs().max()) <= 1e-6
        assert float((cc[k].double() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))


def test_integrated_positional_encoding_layout_and_attenuation():
    """N2 in nerfstudio's convention (NeRFEncoding with covariances: the angles are 2 pi f x, the variances f^2 sigma^2 -- the
    2 pi is NOT squared into the variance; a convention of that library, so no first-principles identity pins it): layout
    [sin over (dim, freq) | cos as sin(. + pi / 2) | raw input], zero covariance = the plain encoding, attenuation exp(-f^2 s^2 / 2)."""
    fs = cpu_ref.FieldSpec(num_layers=8, width=256)
    torch.manual_seed(6)
    mean = torch.randn(4, 3, dtype=torch.float64) * 0.5
    var = torch.rand(4, 3, dtype=torch.float64) * 1e-3
    f = cpu_ref.frequencies(fs).double()
    F = f.numel()
    enc0 = cpu_ref.ipe(fs, mean, torch.zeros_like(var))
    assert enc0.shape == (4, 6 * F + 3)
    ang = (2 * math.pi * mean)[..., None] * f  # [4, 3, F]
    assert torch.allclose(enc0[:, :3 * F], torch.sin(ang).reshape(4, -1), atol=1e-9)
    assert torch.allclose(enc0[:, 3 * F:6 * F], torch.sin(ang + math.pi / 2).reshape(4, -1), atol=1e-9)
    assert torch.equal(enc0[:, 6 * F:], mean)
    assert torch.allclose(cpu_ref.ipe(fs, mean, None), enc0, atol=1e-12)
    enc = cpu_ref.ipe(fs, mean, var)
    att = torch.exp(-0.5 * var[..., None] * f**2).reshape(4, -1)
    assert torch.allclose(enc[:, :3 * F], enc0[:, :3 * F] * att, atol=1e-9)
    assert torch.allclose(enc[:, 3 * F:6 * F], enc0[:, 3 * F:6 * F] * att, atol=1e-9)
