"""GPU parity: the HIP path (through the C ABI of include/rsn.h) against the CPU oracle on identical seeded
inputs.  Tolerance (BASELINE.json north_star): 1e-4 absolute, fp32, on rendered RGB and accumulation; the
same bound is applied to every continuous output.  Discontinuous outputs (mask, median-depth bin) are compared
exactly except at samples that sit within 1e-5 of their threshold."""
import os

import pytest
import torch

import reflect_sampling_nerf_amd as pkg
from oracle import cpu_ref
from reflect_sampling_nerf_amd import ops
from reflect_sampling_nerf_amd._abi import RSN_SPACING_RECIPROCAL, RSN_SPACING_UNIFORM
from tests.helpers import load_golden, max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Per-sample unit normals (and n.d): normalising the small raw head output divides by |raw| (~0.05-0.3 at random
# init), and the resampled fine positions (equal to the oracle's to ~1e-6) enter through IPE features whose
# sensitivity reaches 2*pi*f*exp(-var f^2/2) ~ 3e3 per unit length.  Rendered quantities stay at TOL.
TOL_UNIT = 5e-4
# Analytic normals = normalised gradient of the raw density w.r.t. position: the gradient is dominated by the
# highest undamped IPE frequencies (each term carries 2*pi*f), so ulp-level differences of the sample position are
# amplified once more than for the predicted normals.  Bound: 2e-3 absolute on unit vectors (~0.1 degree).
TOL_GRAD_NORMAL = 2e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    pkg.load_library()
    return torch.device("cuda:0")


def make_field(layers, width, dev, seed=0, bias_shift=0.0):
    torch.manual_seed(seed)
    f = pkg.ReflectSamplingNeRFNerfField(base_mlp_num_layers=layers, base_mlp_layer_width=width)
    with torch.no_grad():
        f.field_output_density.net.bias += bias_shift
    P = {k: v.detach().clone() for k, v in f.state_dict().items()}
    return f.to(dev).eval(), P, cpu_ref.FieldSpec(num_layers=layers, width=width)


# ---------------------------------------------------------------------------------------------- samplers
@pytest.mark.parametrize("kind,tan,near,far", [("uniform", 1.0, 2.0, 6.0), ("reciprocal", 0.25, 0.0, 256.0)])
@pytest.mark.parametrize("S", [1, 24, 128])
def test_spaced_sampler(dev, kind, tan, near, far, S):
    R = 37
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    g = torch.Generator().manual_seed(S)
    for t_rand in (None, torch.rand(R, S + 1, generator=g)):
        sb_ref, eb_ref = cpu_ref.spaced_bins(kind, tan, nears, fars, S, t_rand)
        sp = RSN_SPACING_UNIFORM if kind == "uniform" else RSN_SPACING_RECIPROCAL
        sb, eb = ops.sample_spaced(R, None, S, sp, tan, nears.reshape(R).to(dev), fars.reshape(R).to(dev),
                                   None if t_rand is None else t_rand.to(dev))
        assert max_abs(sb.cpu(), sb_ref) <= 2e-7
        # euclidean bins reach 256 for the reciprocal spacing: relative tolerance
        rel = ((eb.cpu() - eb_ref).abs() / eb_ref.abs().clamp(min=1.0)).max()
        assert float(rel) <= 2e-6


@pytest.mark.parametrize("S_in,S_out", [(16, 16), (128, 128), (64, 33), (7, 130)])
def test_pdf_sampler(dev, S_in, S_out):
    R = 29
    g = torch.Generator().manual_seed(S_in * 1000 + S_out)
    w = torch.rand(R, S_in, 1, generator=g) ** 4
    w[0] = 0.0  # all-zero weights: histogram padding only
    w[1, : S_in // 2] = 0.0
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    sb_in, _ = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S_in, None)
    for u_rand in (None, torch.rand(R, S_out + 1, generator=g)):
        sb_ref, eb_ref = cpu_ref.pdf_bins("uniform", 1.0, nears, fars, w, sb_in, S_out, u_rand)
        sb, eb = ops.sample_pdf(R, None, S_in, S_out, RSN_SPACING_UNIFORM, 1.0, 0.01, nears.reshape(R).to(dev),
                                fars.reshape(R).to(dev), w[..., 0].contiguous().to(dev),
                                sb_in.contiguous().to(dev), None if u_rand is None else u_rand.to(dev))
        assert max_abs(sb.cpu(), sb_ref) <= 1e-5  # inverse CDF amplifies 1-ulp cdf differences by 1/pdf
        assert max_abs(eb.cpu(), eb_ref) <= 2e-5
        assert bool((sb[:, 1:] >= sb[:, :-1]).all()), "resampled bins must be sorted"


# ---------------------------------------------------------------------------------------------- compositing
@pytest.mark.parametrize("S", [1, 5, 64, 130, 192])
@pytest.mark.parametrize("background", [0, 1, 2])
def test_composite(dev, S, background):
    R = 23
    g = torch.Generator().manual_seed(S + 7 * background)
    sigma = torch.rand(R, S, 1, generator=g) * 8.0 * (torch.rand(R, S, 1, generator=g) > 0.5)
    sigma[0] = 0.0          # empty ray
    sigma[1] = 1e4          # saturates in the first sample
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, torch.rand(R, S + 1, generator=g))
    t0, t1 = eb[:, :-1], eb[:, 1:]
    color = torch.rand(R, S, 3, generator=g) * 2.0
    lv = {"diff": torch.rand(R, S, 3, generator=g), "tint": torch.rand(R, S, 3, generator=g),
          "pred_normals": torch.nn.functional.normalize(torch.randn(R, S, 3, generator=g), dim=-1),
          "roughness": torch.rand(R, S, generator=g)}
    bg = torch.rand(R, 3, generator=g)
    w_ref = cpu_ref.weights_from_density(sigma, t0, t1)
    bg_ref = {0: None, 1: torch.ones(3), 2: bg}[background]
    for training in (False, True):
        flags = 0 if training else ops.RSN_COMP_EVAL
        out = ops.composite(R, None, S, background, flags, sigma[..., 0].contiguous().to(dev), eb.contiguous().to(dev),
                            color.to(dev), bg_rgb=bg.to(dev), level={k: v.to(dev) for k, v in lv.items()},
                            surface=True)
        assert max_abs(out["weights"].cpu(), w_ref[..., 0]) <= 1e-6
        assert max_abs(out["rgb"].cpu(), cpu_ref.composite_rgb(color, w_ref, bg_ref, training)) <= 2e-6
        assert max_abs(out["accumulation"].cpu(), w_ref.sum(dim=-2)[..., 0]) <= 2e-6
        assert max_abs(out["diff"].cpu(), cpu_ref.composite_rgb(lv["diff"], w_ref, torch.ones(3), training)) <= 2e-6
        assert max_abs(out["tint"].cpu(), cpu_ref.composite_rgb(lv["tint"], w_ref, None, training)) <= 2e-6
        assert max_abs(out["normals"].cpu(), cpu_ref.render_normals(lv["pred_normals"], w_ref)) <= 2e-5
        assert max_abs(out["roughness"].cpu(), (w_ref * lv["roughness"][..., None]).sum(dim=-2)[..., 0]) <= 2e-6
        # median depth: exact unless the cumulative weight sits within 1e-5 of 0.5 at the chosen bin
        d_ref = cpu_ref.median_depth(w_ref, t0, t1)[..., 0]
        cw = torch.cumsum(w_ref[..., 0], dim=-1)
        near_half = ((cw - 0.5).abs() < 1e-5).any(dim=-1)
        bad = ((out["depth"].cpu() - d_ref).abs() > 1e-5) & ~near_half
        assert not bool(bad.any())


# ---------------------------------------------------------------------------------------------- the field kernel
@pytest.mark.parametrize("layers,width", [(8, 256), (4, 128), (8, 64), (6, 128), (2, 64)])
def test_field_level(dev, layers, width):
    R, S = 19, 24  # 456 points: ragged against the 128-point tiles
    fld, P, fs = make_field(layers, width, dev, seed=layers * 10 + width, bias_shift=1.0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=3)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, None)
    with torch.no_grad():
        ref = cpu_ref.field_level(P, fs, o, d, pa, eb, training=False, want_normals=False)
    lv = fld.evaluate_frustums(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb.contiguous().to(dev))
    torch.cuda.synchronize()
    pairs = {"sigma": ref["sigma"][..., 0], "color": ref["color"], "pred_normals": ref["pred_normals"],
             "n_dot_d": ref["n_dot_d"][..., 0], "diff": ref["diff"], "tint": ref["tint"],
             "roughness": torch.sigmoid(ref["rough_raw"])[..., 0]}
    for k, v in pairs.items():
        assert max_abs(lv[k].cpu(), v) <= TOL, k


def test_field_level_far_samples_are_contracted(dev):
    """reciprocal spacing out to t=256: exercises the |x|>1 contraction branch and huge IPE arguments."""
    R, S = 9, 64
    fld, P, fs = make_field(8, 128, dev, seed=5, bias_shift=-1.0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=4)
    nears, fars = torch.zeros(R, 1), torch.full((R, 1), 256.0)
    _, eb = cpu_ref.spaced_bins("reciprocal", 0.25, nears, fars, S, None)
    pa = pa * 400.0
    with torch.no_grad():
        ref = cpu_ref.field_level(P, fs, o, d, pa, eb, training=False, want_normals=False)
    lv = fld.evaluate_frustums(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb.contiguous().to(dev))
    assert max_abs(lv["sigma"].cpu(), ref["sigma"][..., 0]) <= TOL
    assert max_abs(lv["color"].cpu(), ref["color"]) <= TOL


def test_inf_color(dev):
    fld, P, fs = make_field(8, 128, dev, seed=9)
    g = torch.Generator().manual_seed(0)
    M = 77
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    sq = torch.rand(M, 1, generator=g) * 0.3
    with torch.no_grad():
        ref = cpu_ref.inf_color(P, fs, dirs, sq)
    out = fld.get_inf_color(dirs.to(dev), sq.to(dev))
    assert max_abs(out.cpu(), ref) <= TOL


def test_get_density_granular_api(dev):
    fld, P, fs = make_field(8, 128, dev, seed=11)
    g = torch.Generator().manual_seed(1)
    mean = torch.randn(5, 13, 3, generator=g)
    A = torch.randn(5, 13, 3, 3, generator=g) * 0.02
    cov = A @ A.transpose(-1, -2)
    with torch.no_grad():
        enc = cpu_ref.ipe(fs, mean, torch.diagonal(cov, dim1=-2, dim2=-1))
        sig_ref, emb_ref, _ = cpu_ref.density_from_encoding(P, fs, enc)
    sig, emb = fld.get_density(mean.to(dev), cov.to(dev))
    assert sig.shape == (5, 13, 1) and emb.shape == (5, 13, 128)
    assert max_abs(sig.cpu(), sig_ref) <= TOL
    assert max_abs(emb.cpu(), emb_ref) <= TOL


# ---------------------------------------------------------------------------------------------- full get_outputs
def _run_model(dev, layers, width, samples, R, seed, bias_shift, near=2.0, far=6.0):
    torch.manual_seed(seed)
    cfg = pkg.ReflectSamplingNeRFModelConfig(
        num_coarse_samples=samples[0], num_importance_samples=samples[1], num_reflect_coarse_samples=samples[2],
        num_reflect_importance_samples=samples[3], base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += bias_shift
    P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
    model.to(dev).eval()
    o, d, pa = cpu_ref.synthetic_rays(R, seed=seed + 50)
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev),
                       fars=fars.to(dev))
    out = model(rb)
    torch.cuda.synchronize()
    fs = cpu_ref.FieldSpec(num_layers=layers, width=width)
    ms = cpu_ref.ModelSpec(*samples)
    with torch.no_grad():
        ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=False)
    return out, ref


@pytest.mark.parametrize("layers,width,samples,R,bias", [
    (8, 256, (32, 32, 16, 16), 70, 2.0),
    (8, 128, (128, 128, 64, 64), 33, 1.0),
    (4, 128, (64, 48, 24, 40), 130, 1.5),
    (8, 64, (16, 16, 8, 8), 257, 2.0),
])
def test_get_outputs_matches_oracle(dev, layers, width, samples, R, bias):
    out, ref = _run_model(dev, layers, width, samples, R, seed=layers + width, bias_shift=bias)
    assert set(out.keys()) == set(ref.keys())
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "accumulation_coarse", "accumulation_fine", "weights_coarse",
              "weights_fine", "diff", "tint", "roughness"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert max_abs(out[k].cpu(), ref[k]) <= TOL, k
    for k in ("pred_normals_coarse", "pred_normals_fine", "normals_coarse", "normals_fine", "n_dot_d_coarse",
              "n_dot_d_fine"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert max_abs(out[k].cpu(), ref[k]) <= TOL_UNIT, k
    # mask: exact except for rays sitting on a threshold
    m_gpu, m_ref = out["mask"].cpu(), ref["mask"]
    assert m_gpu.dtype == torch.bool
    flips = int((m_gpu != m_ref).sum())
    assert flips == 0, f"{flips} mask flips"
    for k in ("mid_reflect_coarse", "mid_reflect_fine"):
        assert max_abs(out[k].cpu(), ref[k]) <= TOL, k
    # median depths: same bin unless the cumulative weight is within 1e-5 of 0.5
    for lvl in ("coarse", "fine"):
        cw = torch.cumsum(ref[f"weights_{lvl}"][..., 0], dim=-1)
        near_half = ((cw - 0.5).abs() < 1e-5).any(dim=-1, keepdim=True)
        bad = ((out[f"depth_{lvl}"].cpu() - ref[f"depth_{lvl}"]).abs() > 1e-4) & ~near_half
        assert not bool(bad.any()), lvl
    assert "depth_reflect_fine" in out and out["depth_reflect_fine"].shape == ref["depth_reflect_fine"].shape


def test_get_outputs_empty_mask_takes_early_out_shape(dev):
    out, ref = _run_model(dev, 6, 64, (8, 8, 8, 8), 16, seed=4, bias_shift=-12.0)
    assert int(out["mask"].sum()) == 0 and "depth_reflect_fine" not in out and "depth_reflect_fine" not in ref
    for k in ("mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_fine"):
        assert max_abs(out[k].cpu(), ref[k]) <= TOL


@pytest.mark.parametrize("name", ["eval_l8_w64_near0", "eval_l8_w256", "eval_l4_w128", "eval_l6_w64_nomask",
                                  "eval_trained_l8_w64", "eval_trained_l8_w256",
                                  # widths that are not one of the kernels' 64 / 128 / 256 (base_mlp_layer_width is a free
                                  # constructor knob of the reference Field, field.py:41): run zero-padded (param_width)
                                  "eval_l8_w32", "eval_l4_w32", "eval_l6_w32_nomask", "eval_l8_w200"])
def test_get_outputs_on_reference_golden_rays(dev, name):
    """Ties the GPU path to the REFERENCE run directly: the reference's own parameters (state_dict loaded by name), its
    rays, and ITS outputs (tests/golden/*.npz, written by oracle/make_golden.py from the reference's modules) -- at the
    BASELINE network (8 x 256), the configs[0] network (4 x 128), near plane 0 and the no-mask early-out
    (model.py:259-260) -- and on TRAINED weights (eval_trained_*: the reference trained on the procedural scene by
    oracle/make_golden_trained.py until it left the initialisation regime: peaked densities, surface-concentrated
    weights, a reflection mask decided by learned normals; config.py:32 trains for 100 000 iterations, that is the
    regime the method runs in).  Rendered outputs within 1e-4, mask exact, early-out keys identical."""
    meta, g = load_golden(name)
    s = meta["samples"]
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=s[0], num_importance_samples=s[1],
                                            num_reflect_coarse_samples=s[2], num_reflect_importance_samples=s[3],
                                            base_mlp_num_layers=meta["layers"], base_mlp_layer_width=meta["width"])
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(g["param"])  # the reference's own state_dict loads by name
    model.to(dev).eval()
    i = g["in"]
    rb = pkg.RayBundle(**{k: i[k].to(dev) for k in ("origins", "directions", "pixel_area", "nears", "fars")})
    out = model(rb)
    ref = g["out"]  # produced by the reference's own modules
    trained = "trained" in name
    assert sorted(out.keys()) == sorted(ref.keys()) == meta["keys"]
    assert torch.equal(out["mask"].cpu().to(torch.uint8), ref["mask"]) and int(out["mask"].sum()) == meta["M"]
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "accumulation_coarse", "accumulation_fine", "mid_reflect_coarse",
              "mid_reflect_fine", "diff", "tint", "roughness", "weights_coarse", "weights_fine"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        if trained and "reflect" in k:
            # TRAINED weights, reflected rays: the secondary ray starts at the MEDIAN depth of the primary ray (a bin index:
            # discontinuous) and runs 256 units through a field whose IPE arguments reach 8e5 radians -- fp32 itself is
            # 2e-2 (8 x 256) / 8e-4 (8 x 64) away from an fp64 evaluation of the same model there, while the two fp32
            # pipelines sit 1.3e-4 from each other (profiles/r04_fp64_arbiter_*.json).  Measured against the reference
            # (profiles/r04_trained_fixture_report.json): <= 2.3e-4, 2-5 of 64 rays over 1e-4.  Bound: 1e-3, and nine
            # rays in ten within the north-star's 1e-4.
            per_ray = (out[k].cpu() - ref[k]).abs().reshape(ref[k].shape[0], -1).max(dim=1).values
            assert float(per_ray.max()) <= 1e-3 and float((per_ray > TOL).float().mean()) <= 0.1, k
            continue
        assert max_abs(out[k].cpu(), ref[k]) <= TOL, k  # the primary render keeps the north-star's 1e-4 on trained weights too
    # per-sample unit normals on trained weights: the raw normal head of a trained surface sample is normalised from a small
    # vector at ulp-different fine positions (measured 1.7e-3 at 8 x 64, 5.3e-4 at 8 x 256 against the reference)
    tol_unit = 3e-3 if trained else TOL_UNIT
    for k in ("pred_normals_coarse", "pred_normals_fine", "normals_coarse", "normals_fine", "n_dot_d_coarse",
              "n_dot_d_fine"):
        assert max_abs(out[k].cpu(), ref[k]) <= tol_unit, k
    for lvl in ("coarse", "fine"):  # median depths: same bin unless the cumulative weight is within 1e-5 of 0.5
        cw = torch.cumsum(ref[f"weights_{lvl}"][..., 0], dim=-1)
        near_half = ((cw - 0.5).abs() < 1e-5).any(dim=-1, keepdim=True)
        bad = ((out[f"depth_{lvl}"].cpu() - ref[f"depth_{lvl}"]).abs() > 1e-4) & ~near_half
        assert not bool(bad.any()), lvl
    if meta["M"] > 0:
        assert out["depth_reflect_fine"].shape == ref["depth_reflect_fine"].shape


def test_train_mode_outputs_on_reference_golden_rays_padded_width(dev):
    """Training-mode get_outputs of the REFERENCE (tests/golden/train_l8_w32.npz: logged stratified jitter, autograd normals,
    model.py:159-160 -> field.py:146-147) at width 32 -- not one of the kernels' 64 / 128 / 256: the HIP path runs it
    zero-padded to 64 (rsn_field_desc.param_width) -- on the reference's logged bins: every output incl. the analytic normals;
    and the granular Field API keeps the parameter width (get_density's embedding is [..., 32])."""
    meta, g = load_golden("train_l8_w32")
    s = meta["samples"]
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=s[0], num_importance_samples=s[1],
                                            num_reflect_coarse_samples=s[2], num_reflect_importance_samples=s[3],
                                            base_mlp_num_layers=meta["layers"], base_mlp_layer_width=meta["width"])
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(g["param"])
    model.to(dev).train()
    assert (model.field.param_width, model.field.width) == (32, 64)
    i = g["in"]
    rb = pkg.RayBundle(**{k: i[k].to(dev) for k in ("origins", "directions", "pixel_area", "nears", "fars")})
    out = model._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in g["jitter"].items()}, bins=g["bins"])
    ref = g["out"]
    assert sorted(out.keys()) == sorted(ref.keys())
    assert torch.equal(out["mask"].cpu().to(torch.uint8), ref["mask"])
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
              "accumulation_fine", "weights_coarse", "weights_fine", "diff", "tint", "roughness"):
        assert max_abs(out[k].detach().cpu(), ref[k]) <= TOL, k
    for k in ("pred_normals_coarse", "pred_normals_fine", "n_dot_d_coarse", "n_dot_d_fine"):
        assert max_abs(out[k].detach().cpu(), ref[k]) <= TOL_UNIT, k
    for lvl in ("coarse", "fine"):
        e = (out[f"normals_{lvl}"].cpu() - ref[f"normals_{lvl}"]).abs()
        assert float(e.mean()) <= 1e-3 and float(e.flatten().quantile(0.99)) <= 5e-3, lvl
    model.eval()
    mean = torch.randn(5, 7, 3, generator=torch.Generator().manual_seed(1)).to(dev) * 0.5
    sigma, emb = model.field.get_density(mean)
    assert emb.shape == (5, 7, 32) and sigma.shape == (5, 7, 1)
    P = {k: v for k, v in g["param"].items()}
    fs = cpu_ref.FieldSpec(num_layers=meta["layers"], width=meta["width"])
    enc = cpu_ref.ipe(fs, mean.cpu(), None)
    sig_ref, emb_ref, _ = cpu_ref.density_from_encoding(P, fs, enc)
    assert max_abs(emb.cpu(), emb_ref) <= 1e-4 and max_abs(sigma.cpu(), sig_ref) <= 1e-4
    pn = model.field.get_pred_normals(emb)
    assert max_abs(pn.cpu(), cpu_ref.pred_normals(P, emb_ref)) <= TOL_UNIT


# ---------------------------------------------------------------------------------------------- full-size properties
def test_full_size_properties(dev):
    """BASELINE config 2 size (4096 x 128, 8 x 256): size-independent properties instead of an oracle run."""
    R, S = 4096, 128
    fld, _, _ = make_field(8, 256, dev, seed=0, bias_shift=1.0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    lv = fld.evaluate_frustums(o, d, pa, eb)
    c = ops.composite(R, None, S, 1, ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB, lv["sigma"], eb, lv["color"])
    torch.cuda.synchronize()
    w = c["weights"]
    assert bool(torch.isfinite(lv["color"]).all()) and bool(torch.isfinite(lv["sigma"]).all())
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
    assert max_abs(w.sum(-1), c["accumulation"]) <= 1e-5
    assert float(c["rgb"].min()) >= 0.0 and float(c["rgb"].max()) <= 1.0
    assert float((lv["pred_normals"].norm(dim=-1) - 1).abs().max()) <= 1e-5
    # determinism: a second launch is bit-identical
    lv2 = fld.evaluate_frustums(o, d, pa, eb)
    assert torch.equal(lv["color"], lv2["color"]) and torch.equal(lv["sigma"], lv2["sigma"])
    # tile independence: evaluating a ragged sub-batch gives the same per-sample values
    sub = fld.evaluate_frustums(o[:777], d[:777], pa[:777], eb[:777].contiguous())
    assert torch.equal(sub["color"], lv["color"][:777]) and torch.equal(sub["sigma"], lv["sigma"][:777])
    # a slice of it against the oracle
    Rs = 8
    P = {k: v.detach().cpu() for k, v in fld.state_dict().items()}
    with torch.no_grad():
        ref = cpu_ref.field_level(P, cpu_ref.FieldSpec(), o[:Rs].cpu(), d[:Rs].cpu(), pa[:Rs].cpu()[:, None],
                                  eb[:Rs].cpu(), training=False, want_normals=False)
    assert max_abs(lv["color"][:Rs].cpu(), ref["color"]) <= TOL
    assert max_abs(lv["sigma"][:Rs].cpu(), ref["sigma"][..., 0]) <= TOL


# ---------------------------------------------------------------------------------------------- training-mode forward
@pytest.mark.parametrize("layers,width", [(8, 128), (4, 64), (8, 256)])
def test_train_level_normals_and_saved_activations(dev, layers, width):
    """Training-mode level: analytic normals (Field.get_normals: autograd of the raw density w.r.t. the contracted
    mean) and the activations saved for the backward pass, against the oracle with the same jittered bins."""
    R, S = 11, 20
    fld, P, fs = make_field(layers, width, dev, seed=3 * layers + width, bias_shift=1.0)
    fld.train()
    o, d, pa = cpu_ref.synthetic_rays(R, seed=8)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    g = torch.Generator().manual_seed(2)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, torch.rand(R, S + 1, generator=g))
    ref = cpu_ref.field_level(P, fs, o, d, pa, eb, training=True, want_normals=True)
    lv = fld.evaluate_frustums_train(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb.contiguous().to(dev))
    torch.cuda.synchronize()
    assert max_abs(lv["sigma"].cpu(), ref["sigma"][..., 0].detach()) <= TOL
    assert max_abs(lv["color"].cpu(), ref["color"].detach()) <= TOL
    assert max_abs(lv["saved"]["act"][-1].cpu().reshape(R, S, width), ref["emb"].detach()) <= TOL
    # analytic normals: unit vectors; compare where the raw gradient is not tiny (normalisation amplifies)
    n_gpu, n_ref = lv["normals"].cpu(), ref["normals"].detach()
    assert max_abs(n_gpu, n_ref) <= TOL_GRAD_NORMAL
    assert float((n_gpu - n_ref).abs().mean()) <= 5e-5
    assert float((n_gpu.norm(dim=-1) - 1).abs().max()) <= 1e-5


@pytest.mark.parametrize("layers,width", [(8, 128), (8, 256)])
def test_granular_get_density_with_density_grad_and_get_normals(dev, layers, width):
    """The reference model's own call sequence in training mode (model.py:153,160): get_density(mean, cov, True) followed by
    get_normals() -- density, embedding and -normalize(d raw_density / d mean) of the CONTRACTED means handed in (field.py:125-127,
    146-147), against autograd through the oracle on the same Gaussians; the other MMA modes say so instead of guessing."""
    fld, P, fs = make_field(layers, width, dev, seed=5 * layers + width, bias_shift=1.0)
    fld.train()
    R, S = 9, 14
    o, d, pa = cpu_ref.synthetic_rays(R, seed=11)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, torch.rand(R, S + 1, generator=torch.Generator().manual_seed(4)))
    mean, cov = cpu_ref.contract(*cpu_ref.gaussian_blob(o, d, pa, eb[:, :-1], eb[:, 1:]))
    mean_r = mean.clone().requires_grad_(True)
    enc = cpu_ref.ipe(fs, mean_r, torch.diagonal(cov, dim1=-2, dim2=-1))
    sig_ref, emb_ref, raw = cpu_ref.density_from_encoding(P, fs, enc)
    (g,) = torch.autograd.grad(raw.sum(), mean_r)
    n_ref = -g / g.norm(dim=-1, keepdim=True)
    with pytest.raises(RuntimeError, match="requires_density_grad"):
        fld.get_normals()
    sig, emb = fld.get_density(mean.to(dev), cov.to(dev), requires_density_grad=True)
    n = fld.get_normals()
    assert sig.shape == (R, S, 1) and emb.shape == (R, S, width) and n.shape == (R, S, 3)
    assert max_abs(sig.cpu(), sig_ref.detach()) <= TOL and max_abs(emb.cpu(), emb_ref.detach()) <= TOL
    assert max_abs(n.cpu(), n_ref) <= TOL_GRAD_NORMAL and float((n.cpu() - n_ref).abs().mean()) <= 5e-5
    assert float((n.norm(dim=-1) - 1).abs().max()) <= 1e-5
    sig_e, _ = fld.get_density(mean.to(dev), cov.to(dev))  # without the flag: the plain forward, and no stale normals
    assert torch.equal(sig_e, sig)
    with pytest.raises(RuntimeError):
        fld.get_normals()
    fld.set_mma_mode("bf16x6")
    with pytest.raises(NotImplementedError, match="exact-fp32"):
        fld.get_density(mean.to(dev), cov.to(dev), requires_density_grad=True)


# ---------------------------------------------------------------------------------------------- training: forward + backward
def _loss_from_outputs(out, tgt):
    """A loss touching every gradient-carrying output, shaped like get_loss_dict (model.py:395-407)."""
    loss = 0.0
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine"):
        loss = loss + torch.mean((out[k] - tgt[k]) ** 2)
    for lvl, c1, c2 in (("coarse", 3e-3, 1e-2), ("fine", 3e-3, 1e-1)):
        w = out[f"weights_{lvl}"].detach()
        loss = loss + c1 * torch.sum(w * torch.sum((out[f"normals_{lvl}"].detach() - out[f"pred_normals_{lvl}"]) ** 2,
                                                   dim=-1, keepdim=True))
        loss = loss + c2 * torch.sum(w * torch.clamp(out[f"n_dot_d_{lvl}"], min=0.0) ** 2)
    return loss


def _coarse_loss(out, tgt):
    w = out["weights_coarse"].detach()
    return (torch.mean((out["mid_rgb_coarse"] - tgt["mid_rgb_coarse"]) ** 2)
            + 3e-3 * torch.sum(w * torch.sum((out["normals_coarse"].detach() - out["pred_normals_coarse"]) ** 2, dim=-1,
                                             keepdim=True))
            + 1e-2 * torch.sum(w * torch.clamp(out["n_dot_d_coarse"], min=0.0) ** 2))


@pytest.mark.parametrize("layers,width,samples,R,bias", [
    (8, 64, (16, 16, 8, 8), 48, 2.0),
    (8, 128, (32, 24, 16, 12), 37, 1.5),
    (4, 64, (16, 16, 8, 8), 40, 2.0),
])
@pytest.mark.parametrize("mma", ["f32", "bf16x6"])
def test_train_forward_backward_matches_oracle_autograd(dev, layers, width, samples, R, bias, mma):
    """Training-mode get_outputs (stratified jitter, analytic normals) and its backward against autograd through the
    oracle, with the samplers' uniform draws shared.

    Two losses.  (a) coarse-level terms only: the coarse samples are bit-identical in both pipelines, so every
    parameter gradient must agree tightly (2e-4 of the tensor's largest entry).  (b) the full loss (all four colour
    terms, both normal losses, both orientation losses): the resampled fine/reflect positions agree to ~1e-6 only
    (CDF scan order), the IPE amplifies that by up to ~3e3 and a small fraction of near-zero ReLU units flips, which
    changes a gradient discontinuously -- so (b) checks direction (cosine >= 0.999) and relative L2 (<= 3e-2) per
    tensor, and the head / mid-MLP gradients (which see no flips below them) tightly.  (c) closes that hole: the full
    loss with the oracle evaluated at the sample positions the kernels used (bins + contracted means) -- EVERY
    parameter gradient within 5e-5 of the tensor's largest entry."""
    seed = layers * 7 + width
    torch.manual_seed(seed)
    cfg = pkg.ReflectSamplingNeRFModelConfig(
        num_coarse_samples=samples[0], num_importance_samples=samples[1], num_reflect_coarse_samples=samples[2],
        num_reflect_importance_samples=samples[3], base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += bias
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.field.state_dict().items()}
    model.to(dev).train()
    model.field.set_mma_mode(mma)  # bf16x6: the fp32-emulating split-bf16 sweeps must meet the same bounds
    o, d, pa = cpu_ref.synthetic_rays(R, seed=seed + 50)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    fs, ms = cpu_ref.FieldSpec(num_layers=layers, width=width), cpu_ref.ModelSpec(*samples)
    g = torch.Generator().manual_seed(seed + 1)
    jit = {"coarse": torch.rand(R, samples[0] + 1, generator=g), "fine": torch.rand(R, samples[1] + 1, generator=g),
           "reflect_coarse": torch.rand(R, samples[2] + 1, generator=g),
           "reflect_fine": torch.rand(R, samples[3] + 1, generator=g)}
    tgt = {k: torch.rand(R, 3, generator=g) for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse",
                                                      "mid_reflect_fine")}
    tgt_dev = {k: v.to(dev) for k, v in tgt.items()}
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev),
                       fars=fars.to(dev))

    def run(loss_fn, share_normals, share_bins=False):
        for p in P.values():
            p.grad = None
        model.zero_grad(set_to_none=True)
        rec = {}
        ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit, record_bins=rec)
        loss_fn(ref, tgt).backward()
        M = int(ref["mask"].sum())
        assert M > 0
        jit_gpu = dict(jit, reflect_coarse=jit["reflect_coarse"][ref["mask"]],
                       reflect_fine=jit["reflect_fine"][ref["mask"]])
        out = model._get_outputs_train(rb, jitter=jit_gpu, bins=rec if share_bins else None)
        checked = dict(out)
        if share_normals:  # the analytic normals are a detached loss target: give both sides the same constant
            checked["normals_coarse"] = ref["normals_coarse"].detach().to(dev)
            checked["normals_fine"] = ref["normals_fine"].detach().to(dev)
        loss_fn(checked, tgt_dev).backward()
        torch.cuda.synchronize()
        return ref, out

    # ---- forward values + (a) coarse-only loss: tight
    ref, out = run(_coarse_loss, share_normals=True)
    assert set(out.keys()) == set(ref.keys())
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
              "accumulation_fine", "weights_coarse", "weights_fine", "diff", "tint", "roughness"):
        assert max_abs(out[k].detach().cpu(), ref[k].detach()) <= TOL, k
    for k in ("pred_normals_coarse", "pred_normals_fine", "n_dot_d_coarse", "n_dot_d_fine"):
        assert max_abs(out[k].detach().cpu(), ref[k].detach()) <= TOL_UNIT, k
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    assert out["depth_reflect_fine"].shape == ref["depth_reflect_fine"].shape
    for lvl in ("coarse", "fine"):  # analytic normals: ill-conditioned where the raw gradient is tiny
        e = (out[f"normals_{lvl}"].cpu() - ref[f"normals_{lvl}"]).abs()
        assert float(e.mean()) <= 1e-3 and float(e.flatten().quantile(0.99)) <= 5e-3, lvl
    for name, p in model.field.named_parameters():
        gr = P[name].grad
        if gr is None or float(gr.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) <= 1e-12, name
            continue
        scale = float(gr.abs().max())
        err = float((p.grad.cpu() - gr).abs().max())
        assert err <= 2e-4 * scale + 1e-9, f"coarse loss, {name}: abs err {err:.3e}, scale {scale:.3e}"

    # ---- (b) full loss
    run(_loss_from_outputs, share_normals=True)
    for name, p in model.field.named_parameters():
        gr = P[name].grad
        if "field_output_low" in name:
            assert p.grad is None and gr is None
            continue
        assert p.grad is not None, name
        a, b = p.grad.cpu().flatten().double(), gr.flatten().double()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        assert cos >= 0.999 and rel <= 3e-2, f"full loss, {name}: cos {cos:.6f} rel-L2 {rel:.3e}"
        if not name.startswith("mlp_base") or name.startswith(f"mlp_base.layers.{layers - 1}."):
            assert rel <= 2e-3, f"full loss, {name}: rel-L2 {rel:.3e}"

    # ---- (c) full loss on IDENTICAL sample positions.  The HIP pass runs first; the oracle is then evaluated at the
    #      bins AND the contracted sample means the kernels used (the raw-coordinate columns of the saved encoding).
    #      Bins alone are not enough: torch's vectorised CPU sqrt is not correctly rounded (~0.7 % of inputs are 1 ulp
    #      off), the contracted mean then differs by an ulp, the undamped IPE frequencies amplify that to ~1e-3 and a
    #      handful of near-zero ReLU units flip.  On identical positions every parameter gradient must agree tightly.
    for p in P.values():
        p.grad = None
    model.zero_grad(set_to_none=True)
    model._keep_train_state = True
    jit_gpu = dict(jit, reflect_coarse=jit["reflect_coarse"][ref["mask"]], reflect_fine=jit["reflect_fine"][ref["mask"]])
    out = model._get_outputs_train(rb, jitter=jit_gpu)
    st = model._train_state
    model._keep_train_state, model._train_state = False, None
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    M = int(ref["mask"].sum())
    hip_bins, hip_means = {}, {}
    for name, sbk, ebk, lvk, n, S in (("coarse", "sb_c", "eb_c", "lc", R, samples[0]), ("fine", "sb_f", "eb_f", "lf", R, samples[1]),
                                      ("reflect_coarse", "sb_rc", "eb_rc", "lrc", M, samples[2]),
                                      ("reflect_fine", "sb_rf", "eb_rf", "lrf", M, samples[3])):
        hip_bins[name + "_spacing"], hip_bins[name + "_euclid"] = st[sbk].cpu(), st[ebk].cpu()
        hip_means[name] = st[lvk]["saved"]["enc"][:, 96:99].reshape(n, S, 3).cpu()
    ref2 = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit, bins=hip_bins, means=hip_means)
    _loss_from_outputs(ref2, tgt).backward()
    checked = dict(out)
    checked["normals_coarse"] = ref2["normals_coarse"].detach().to(dev)
    checked["normals_fine"] = ref2["normals_fine"].detach().to(dev)
    _loss_from_outputs(checked, tgt_dev).backward()
    torch.cuda.synchronize()
    worst = (0.0, "")
    for name, p in model.field.named_parameters():
        gr = P[name].grad
        if "field_output_low" in name:
            assert p.grad is None and gr is None
            continue
        scale = float(gr.abs().max())
        err = float((p.grad.cpu() - gr).abs().max())
        worst = max(worst, (err / scale, name))
    print(f"identical positions, full loss ({layers}x{width}, {mma}): worst gradient error / tensor max {worst[0]:.2e} ({worst[1]})")
    # measured on MI355X: 6e-7 .. 4e-6 over the three shapes and both MMA modes; bound 5e-5 (north-star asks 2e-4)
    assert worst[0] <= 5e-5, f"full loss, identical positions: {worst[1]} off by {worst[0]:.3e} of its largest entry"


def test_training_trajectory_and_psnr_match_oracle(dev):
    """End-to-end: 24 optimisation steps (fused RAdam with the reference's decay, 50-step warm-up schedule, shared rays
    and jitter) of the HIP path and of the CPU oracle from the same initial weights, run IN-PROCESS (tools.train_parity.run).
    Loss trajectories agree to 1e-4 relative, the PSNR of the rendered held-out rays agrees within 0.1 dB (north-star
    bound).  The long run of the same harness -- 400 lockstep steps, then the HIP path alone until the render exceeds
    20 dB -- is recorded in profiles/r02_train_parity.json."""
    from tools.train_parity import run

    res = run(steps=24, rays=128, samples=(16, 16, 8, 8), verbose=False)
    assert res["max_rel_loss_diff_first10"] <= 1e-4
    assert abs(res["loss_last"][0] - res["loss_last"][1]) <= 1e-3 * abs(res["loss_last"][0])
    assert abs(res["psnr_delta_db"]) <= 0.1
    assert res["psnr_hip_vs_oracle_render"] >= 50.0


# ---------------------------------------------------------------------------------------------- split-bf16 MMA modes
@pytest.mark.parametrize("mode,tol", [("bf16x6", 1e-4), ("bf16x3", 2e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("layers,width", [(8, 256), (4, 128), (8, 64)])
def test_field_level_split_bf16_modes(dev, mode, tol, layers, width):
    """RSN_MMA_BF16X6 (fp32 emulation, must meet the fp32 tolerance) and RSN_MMA_BF16X3 (opt-in reduced precision)."""
    R, S = 19, 24
    fld, P, fs = make_field(layers, width, dev, seed=layers * 10 + width, bias_shift=1.0)
    fld.set_mma_mode(mode)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=3)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, None)
    with torch.no_grad():
        ref = cpu_ref.field_level(P, fs, o, d, pa, eb, training=False, want_normals=False)
    lv = fld.evaluate_frustums(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb.contiguous().to(dev))
    torch.cuda.synchronize()
    assert max_abs(lv["sigma"].cpu(), ref["sigma"][..., 0]) <= tol
    assert max_abs(lv["color"].cpu(), ref["color"]) <= tol
    assert max_abs(lv["diff"].cpu(), ref["diff"]) <= tol
    if mode != "bf16":  # unit normals of a small raw vector amplify the 8-bit operand rounding: not bounded for bf16
        assert max_abs(lv["pred_normals"].cpu(), ref["pred_normals"]) <= max(tol, TOL_UNIT) * (10 if mode == "bf16x3" else 1)


def test_get_outputs_bf16x6_meets_fp32_tolerance(dev):
    torch.manual_seed(12)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=32, num_importance_samples=32,
                                            num_reflect_coarse_samples=16, num_reflect_importance_samples=16,
                                            base_mlp_num_layers=8, base_mlp_layer_width=256)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
    model.to(dev).eval()
    model.field.set_mma_mode("bf16x6")
    R = 70
    o, d, pa = cpu_ref.synthetic_rays(R, seed=62)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev),
                       fars=fars.to(dev))
    out = model(rb)
    with torch.no_grad():
        ref = cpu_ref.get_outputs(P, cpu_ref.FieldSpec(), cpu_ref.ModelSpec(32, 32, 16, 16), o, d, pa, nears, fars)
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
              "accumulation_fine", "weights_fine", "diff", "tint", "roughness"):
        assert max_abs(out[k].cpu(), ref[k]) <= TOL, k
    assert torch.equal(out["mask"].cpu(), ref["mask"])


def test_get_outputs_bf16_ring_kernel_all_modes(dev):
    """The bf16 ring kernel (width 256) behind a whole eval get_outputs: frustum levels, the reflected levels with
    the device-side ray count (n_dev), get_inf_color (no view-direction inputs), ragged tiles (R = 70, 333 rays: last
    256-point tile partly empty) -- against the oracle within the bf16 tolerance, and the Gaussian / embedding entry
    against the exact-fp32 kernel."""
    for R, seed in ((70, 62), (333, 63)):
        torch.manual_seed(12)
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=32, num_importance_samples=24,
                                                num_reflect_coarse_samples=16, num_reflect_importance_samples=8,
                                                base_mlp_num_layers=8, base_mlp_layer_width=256)
        model = cfg.setup(scene_box=None, num_train_data=1)
        with torch.no_grad():
            model.field.field_output_density.net.bias += 2.0
        P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
        model.to(dev).eval()
        model.field.set_mma_mode("bf16")
        o, d, pa = cpu_ref.synthetic_rays(R, seed=seed)
        nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
        rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev),
                           fars=fars.to(dev))
        out = model(rb)
        with torch.no_grad():
            ref = cpu_ref.get_outputs(P, cpu_ref.FieldSpec(), cpu_ref.ModelSpec(32, 24, 16, 8), o, d, pa, nears, fars)
        assert int(ref["mask"].sum()) > 0
        same = out["mask"].cpu() == ref["mask"]
        assert float(same.float().mean()) >= 0.97  # a ray on the mask threshold may flip at bf16 precision
        for k in ("mid_rgb_coarse", "mid_rgb_fine", "accumulation_coarse", "accumulation_fine", "diff", "tint", "roughness"):
            assert bool(torch.isfinite(out[k]).all()), k
            assert max_abs(out[k].cpu(), ref[k]) <= 3e-2, k
        for k in ("mid_reflect_coarse", "mid_reflect_fine"):  # on the rays both pipelines reflect (or both do not)
            assert max_abs(out[k].cpu()[same], ref[k][same]) <= 5e-2, k
    # explicit Gaussians + embedding output through the ring kernel vs the exact-fp32 kernel
    fld = model.field
    g = torch.Generator().manual_seed(4)
    means = (torch.randn(300, 3, generator=g) * 0.8).to(dev)
    covd = (torch.rand(300, 3, generator=g) * 1e-4).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1).to(dev)
    lb = fld.evaluate_gaussians(means, covd, dirs, want_embedding=True)
    fld.set_mma_mode("f32")
    lf = fld.evaluate_gaussians(means, covd, dirs, want_embedding=True)
    assert max_abs(lb["color"], lf["color"]) <= 3e-2 and max_abs(lb["pred_normals"], lf["pred_normals"]) <= 5e-2
    assert float(((lb["embedding"] - lf["embedding"]).abs() / (1.0 + lf["embedding"].abs())).max()) <= 3e-2
    assert float(((lb["sigma"] - lf["sigma"]).abs() / (1.0 + lf["sigma"].abs())).max()) <= 3e-2


# ---------------------------------------------------------------------------------------------- granular Field API
def test_field_spatial_distortion_in_the_granular_api(dev):
    """field.py:49,92-94: a Field built with a `spatial_distortion` applies it to the Gaussians in get_blob (here: the reference's own
    contraction wrapped as a distortion, so the result is checkable against the golden-pinned `contract`); the fused level kernels,
    which form their Gaussians in-kernel, refuse such a field instead of ignoring the knob."""
    from types import SimpleNamespace

    plain, _, _ = make_field(8, 128, dev, seed=21)
    seen = {}

    class Contract360:
        def __call__(self, g):
            seen["type"] = type(g).__name__
            m, c = plain.contract(g.mean, g.cov)
            return type(g)(mean=m, cov=c)

    torch.manual_seed(21)
    fld = pkg.ReflectSamplingNeRFNerfField(base_mlp_num_layers=8, base_mlp_layer_width=128, spatial_distortion=Contract360()).to(dev)
    R, S = 5, 6
    o, d, pa = cpu_ref.synthetic_rays(R, seed=5)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, None)
    t0, t1 = eb[:, :-1], eb[:, 1:]
    fr = SimpleNamespace(origins=o[:, None, :].to(dev), directions=d[:, None, :].to(dev), starts=t0[..., None].to(dev),
                         ends=t1[..., None].to(dev), pixel_area=pa[:, None, :].to(dev))
    mean, cov = fld.get_blob(SimpleNamespace(frustums=fr))
    assert seen["type"] == "Gaussians"
    mean_ref, cov_ref = cpu_ref.contract(*cpu_ref.gaussian_blob(o, d, pa, t0, t1))
    assert max_abs(mean.cpu(), mean_ref) <= 2e-6 and max_abs(cov.cpu(), cov_ref) <= 2e-6
    with pytest.raises(NotImplementedError, match="granular API"):
        fld.evaluate_frustums(o.to(dev), d.to(dev), pa.reshape(-1).to(dev), eb.to(dev))


def test_granular_field_api(dev):
    """The Field methods the reference Model calls one by one: get_blob, contract, get_density,
    get_pred_normals, get_diff, get_tint, get_roughness, get_mid, get_low, get_reflection -- against the oracle, and
    contract against the golden vectors captured from the reference's own Field.contract."""
    from types import SimpleNamespace

    fld, P, fs = make_field(8, 128, dev, seed=21)
    R, S = 7, 9
    o, d, pa = cpu_ref.synthetic_rays(R, seed=5)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, None)
    t0, t1 = eb[:, :-1], eb[:, 1:]
    fr = SimpleNamespace(origins=o[:, None, :].to(dev), directions=d[:, None, :].to(dev), starts=t0[..., None].to(dev),
                         ends=t1[..., None].to(dev), pixel_area=pa[:, None, :].to(dev))
    mean, cov = fld.get_blob(SimpleNamespace(frustums=fr))
    mean_ref, cov_ref = cpu_ref.gaussian_blob(o, d, pa, t0, t1)
    assert mean.shape == (R, S, 3) and cov.shape == (R, S, 3, 3)
    assert max_abs(mean.cpu(), mean_ref) <= 1e-6 and max_abs(cov.cpu(), cov_ref) <= 1e-8
    # contract: golden from the reference's own method (points inside and far outside the unit ball)
    _, g = load_golden("units")
    c = g["contract"]
    mc, cc = fld.contract(c["mean"].to(dev), c["cov"].to(dev))
    assert max_abs(mc.cpu(), c["out_mean"]) <= 2e-6 and max_abs(cc.cpu(), c["out_cov"]) <= 2e-6
    # density + head getters
    mean_c, cov_c = cpu_ref.contract(mean_ref, cov_ref)
    with torch.no_grad():
        enc = cpu_ref.ipe(fs, mean_c, torch.diagonal(cov_c, dim1=-2, dim2=-1))
        sig_ref, emb_ref, _ = cpu_ref.density_from_encoding(P, fs, enc)
    sig, emb = fld.get_density(mean_c.to(dev), cov_c.to(dev))
    assert max_abs(sig.cpu(), sig_ref) <= TOL and max_abs(emb.cpu(), emb_ref) <= TOL
    with torch.no_grad():
        pn_ref = cpu_ref.pred_normals(P, emb_ref)
        diff_ref = torch.sigmoid(cpu_ref.head(P, "field_output_diff", emb_ref))
        tint_ref = torch.sigmoid(cpu_ref.head(P, "field_output_tint", emb_ref))
        rr = cpu_ref.head(P, "field_output_roughness", emb_ref)
        dirs = d[:, None, :].expand(R, S, 3)
        rough = torch.rand(R, S, 1) * 1.5
        mid_ref = cpu_ref.mid_color(P, fs, cpu_ref.integrated_sh(dirs, rough), emb_ref)
        low_ref = cpu_ref.mid_color(P, fs, torch.zeros(R, S, 34), emb_ref)
    emb_dev = emb_ref.to(dev)
    assert max_abs(fld.get_pred_normals(emb_dev).cpu(), pn_ref) <= TOL_UNIT
    assert max_abs(fld.get_diff(emb_dev).cpu(), diff_ref) <= TOL
    assert max_abs(fld.get_tint(emb_dev).cpu(), tint_ref) <= TOL
    assert max_abs(fld.get_roughness(emb_dev).cpu(), torch.sigmoid(rr)) <= TOL
    assert max_abs(fld.get_roughness(emb_dev, torch.nn.Softplus()).cpu(), torch.nn.functional.softplus(rr)) <= TOL
    # large head outputs (|raw| >> 17, where sigmoid saturates in fp32): the activation sees the RAW head value
    big = emb_ref * 1.0e4
    with torch.no_grad():
        rr_big = cpu_ref.head(P, "field_output_roughness", big)
    assert float(rr_big.abs().max()) > 30.0
    got = fld.get_roughness(big.to(dev), torch.nn.Softplus()).cpu()
    want = torch.nn.functional.softplus(rr_big)
    assert bool(torch.isfinite(got).all()) and float(((got - want).abs() / (1.0 + want.abs())).max()) <= 1e-5
    assert max_abs(fld.get_mid(dirs.to(dev), rough.to(dev), emb_dev).cpu(), mid_ref) <= TOL
    assert max_abs(fld.get_low(emb_dev).cpu(), low_ref) <= TOL
    refl, ndd = fld.get_reflection(dirs.to(dev), pn_ref.to(dev))
    ndd_ref = torch.sum(dirs * pn_ref, dim=-1, keepdim=True)
    refl_ref = torch.nn.functional.normalize(dirs - 2 * ndd_ref * pn_ref, dim=-1)
    assert max_abs(ndd.cpu(), ndd_ref) <= 1e-6 and max_abs(refl.cpu(), refl_ref) <= 1e-6


def test_camera_ray_bundle_chunked_eval(dev):
    """SURVEY 8(f) row 4, the eval-image path (config.py:41, model.py:432-482): a 12 x 21 image rendered by
    Model.get_outputs_for_camera_ray_bundle in chunks of eval_num_rays_per_chunk = 100 rays (252 = 100 + 100 + 52: the
    chunk size does not divide H*W) against the CPU oracle run on all 252 rays at once, then
    get_image_metrics_and_images against an independent plain-torch CPU computation on the same outputs."""
    import math

    torch.manual_seed(3)
    samples = (16, 16, 8, 8)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=16, num_importance_samples=16,
                                            num_reflect_coarse_samples=8, num_reflect_importance_samples=8,
                                            base_mlp_num_layers=4, base_mlp_layer_width=64, eval_num_rays_per_chunk=100)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 1.5
    P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
    model.to(dev).eval()
    H, Wd = 12, 21
    n = H * Wd
    assert n % cfg.eval_num_rays_per_chunk != 0
    o, d, pa = cpu_ref.synthetic_rays(n, seed=9)
    rb = pkg.RayBundle(origins=o.reshape(H, Wd, 3).to(dev), directions=d.reshape(H, Wd, 3).to(dev),
                       pixel_area=pa.reshape(H, Wd, 1).to(dev))  # no nears/fars: the collider fills them
    img = model.get_outputs_for_camera_ray_bundle(rb)
    # -- against the oracle on the whole image (eval: the collider's near plane is reset to 0, N12)
    near, far = 0.0, cfg.collider_params["far_plane"]
    with torch.no_grad():
        ref = cpu_ref.get_outputs(P, cpu_ref.FieldSpec(num_layers=4, width=64), cpu_ref.ModelSpec(*samples), o, d, pa,
                                  torch.full((n, 1), near), torch.full((n, 1), far), training=False)
    assert torch.equal(img["mask"].reshape(-1).cpu(), ref["mask"]) and 0 < int(ref["mask"].sum()) < n
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
              "accumulation_fine", "diff", "tint", "roughness"):
        assert img[k].shape[:2] == (H, Wd), k
        assert max_abs(img[k].reshape(n, -1).cpu(), ref[k]) <= TOL, k
    for lvl in ("coarse", "fine"):  # median depth: same bin unless the cumulative weight sits on 0.5
        cw = torch.cumsum(ref[f"weights_{lvl}"][..., 0], dim=-1)
        near_half = ((cw - 0.5).abs() < 1e-5).any(dim=-1, keepdim=True)
        bad = ((img[f"depth_{lvl}"].reshape(n, 1).cpu() - ref[f"depth_{lvl}"]).abs() > 1e-4) & ~near_half
        assert not bool(bad.any()), lvl
    assert "depth_reflect_fine" not in img  # [M,1] is not an image
    # -- the eval-image hook against plain torch on the CPU (the reference's own raises KeyError at model.py:438)
    gt = torch.rand(H, Wd, 4, generator=torch.Generator().manual_seed(5))  # RGBA: blended on white like the reference
    metrics, images = model.get_image_metrics_and_images(img, {"image": gt.to(dev)})
    c = {k: v.cpu() for k, v in img.items()}
    gt3 = gt[..., :3] * gt[..., 3:] + (1.0 - gt[..., 3:])
    rgb_c, rgb_f = c["mid_rgb_coarse"].clip(0, 1), c["mid_reflect_fine"].clip(0, 1)
    psnr = lambda a, b: 10.0 * math.log10(1.0 / float(((a.double() - b.double()) ** 2).mean()))  # noqa: E731
    assert set(metrics) == {"psnr", "coarse_psnr", "fine_psnr"}
    assert abs(metrics["fine_psnr"] - psnr(gt3, rgb_f)) <= 1e-3 and metrics["psnr"] == metrics["fine_psnr"]
    assert abs(metrics["coarse_psnr"] - psnr(gt3, rgb_c)) <= 1e-3
    assert max_abs(images["img"].cpu(), torch.cat([gt3, rgb_c, rgb_f], dim=1)) <= 1e-6

    def depth_panel(dep, acc):
        x = ((dep - cfg.collider_params["near_plane"]) / (far - cfg.collider_params["near_plane"])).clamp(0, 1)
        return (x * acc + (1 - acc)).expand(H, Wd, 3)

    assert max_abs(images["accumulation"].cpu(), torch.cat([c["accumulation_coarse"].expand(H, Wd, 3),
                                                           c["accumulation_fine"].expand(H, Wd, 3)], dim=1)) <= 1e-6
    assert max_abs(images["depth"].cpu(), torch.cat([depth_panel(c["depth_coarse"], c["accumulation_coarse"]),
                                                    depth_panel(c["depth_fine"], c["accumulation_fine"])], dim=1)) <= 1e-6


def test_chunked_eval_image_issues_no_device_to_host_read(dev):
    """The eval-image path (reference config.py:41: chunks of 1024 rays through Model.forward) enqueues EVERY chunk
    before anything is read back: eval-mode get_outputs keeps the reflected-ray count on the device and its one
    M-shaped output (depth_reflect_fine, model.py:341) lazy.  torch's sync-debug mode turns any implicit
    device-to-host synchronisation into an error.  Afterwards the lazy entry of a single forward still has the
    reference's shape and the dict lists the reference's keys, and the chunked image equals the one-chunk image."""
    torch.manual_seed(4)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=16, num_importance_samples=16,
                                            num_reflect_coarse_samples=8, num_reflect_importance_samples=8,
                                            base_mlp_num_layers=4, base_mlp_layer_width=64, eval_num_rays_per_chunk=64)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 1.5
    model.to(dev).eval()
    H, Wd = 10, 33
    n = H * Wd
    o, d, pa = cpu_ref.synthetic_rays(n, seed=12)
    rb = pkg.RayBundle(origins=o.reshape(H, Wd, 3).to(dev), directions=d.reshape(H, Wd, 3).to(dev),
                       pixel_area=pa.reshape(H, Wd, 1).to(dev))
    flat = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(n, 1).to(dev))
    model(flat)  # warm-up: one-time uploads (packed weights)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        img = model.get_outputs_for_camera_ray_bundle(rb)  # 6 chunks (330 = 5 x 64 + 10)
        single = model(flat)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    M = int(single["mask"].sum())
    assert 0 < M < n
    assert "depth_reflect_fine" in single.keys() and single["depth_reflect_fine"].shape == (M, 1)
    assert len(single) == 21 and "depth_reflect_fine" not in img
    for k in ("mid_rgb_fine", "mid_reflect_fine", "accumulation_fine", "depth_fine", "diff", "tint", "roughness"):
        assert img[k].shape[:2] == (H, Wd)
        assert torch.equal(img[k].reshape(n, -1), single[k].reshape(n, -1)), k  # rays are independent: bit-identical
    assert torch.equal(img["mask"].reshape(n), single["mask"])


@pytest.mark.parametrize("mode", ["f32", "bf16x6", "bf16"])
@pytest.mark.parametrize("layers,width", [(8, 256), (4, 128), (6, 64)])
def test_single_launch_packing_equals_per_segment_packing(dev, layers, width, mode):
    """rsn_pack_weights_table (every segment in one launch, job table in device memory) against rsn_pack_weights (one
    launch per segment): bit-identical packed buffers, also when the table is re-used for changed parameter values."""
    import ctypes as C

    from reflect_sampling_nerf_amd._abi import check, load_library, ptr

    lib = load_library()
    fld, _, _ = make_field(layers, width, dev, seed=5)
    fld.set_mma_mode(mode)
    fld.packed_weights()  # allocates the buffer; regions no launch writes (split-bf16 copies in f32 mode) get zeros
    fld._packed.zero_()
    fld._packed_key = None
    fast = fld.packed_weights().clone()  # the Field packs through the table path
    desc, ps = fld.field_desc(), fld._param_struct()
    nbytes = lib.rsn_packed_weights_bytes(C.byref(desc))
    slow = torch.zeros(nbytes // 4, device=dev)
    check(lib.rsn_pack_weights(C.byref(desc), C.byref(ps), ptr(slow), nbytes, ops._stream()))
    torch.cuda.synchronize()
    assert torch.equal(fast.view(torch.int32), slow.view(torch.int32))
    with torch.no_grad():  # new values, same pointers: the table is NOT rebuilt
        for p in fld.parameters():
            p.mul_(1.5).add_(0.01)
    key_before = fld._pack_table_key
    fast2 = fld.packed_weights()
    assert fld._pack_table_key == key_before
    check(lib.rsn_pack_weights(C.byref(desc), C.byref(ps), ptr(slow), nbytes, ops._stream()))
    torch.cuda.synchronize()
    assert torch.equal(fast2.view(torch.int32), slow.view(torch.int32)) and not torch.equal(fast2, fast)


# ---------------------------------------------------------------------------------------------- edge shapes
@pytest.mark.parametrize("R,samples", [(1, (1, 1, 1, 1)), (3, (2, 5, 1, 3)), (130, (33, 7, 9, 2))])
def test_edge_shapes_single_ray_single_sample(dev, R, samples):
    out, ref = _run_model(dev, 4, 64, samples, R, seed=R + samples[0], bias_shift=2.0)
    assert set(out.keys()) == set(ref.keys())
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_fine",
              "weights_coarse", "weights_fine", "diff", "tint", "roughness"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert max_abs(out[k].cpu(), ref[k]) <= TOL, k
    assert torch.equal(out["mask"].cpu(), ref["mask"])


def test_large_batch_and_noncontiguous_inputs(dev):
    """65536 rays x 64 samples (4.2 M points: beyond 2^31 bytes of per-sample outputs across buffers) with strided
    input views; checks finiteness, determinism against a contiguous copy, and a slice against the oracle."""
    R, S = 65536, 64
    fld, P, fs = make_field(4, 128, dev, seed=1, bias_shift=1.0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=2)
    big = torch.zeros(R, 8, device=dev)
    big[:, 0:3], big[:, 4:7] = o.to(dev), d.to(dev)
    o_nc, d_nc = big[:, 0:3], big[:, 4:7]  # non-contiguous views
    assert not o_nc.is_contiguous()
    model_cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S,
                                                  num_reflect_coarse_samples=8, num_reflect_importance_samples=8,
                                                  base_mlp_num_layers=4, base_mlp_layer_width=128)
    torch.manual_seed(1)
    model = model_cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(P)
    model.to(dev).eval()
    rb = pkg.RayBundle(origins=o_nc, directions=d_nc, pixel_area=pa.to(dev))
    out = model(rb)  # the collider fills nears/fars (eval: near plane reset to 0)
    assert all(bool(torch.isfinite(out[k]).all()) for k in ("mid_rgb_fine", "mid_reflect_fine", "weights_fine"))
    rb2 = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev))
    out2 = model(rb2)
    assert torch.equal(out["mid_rgb_fine"], out2["mid_rgb_fine"]) and torch.equal(out["mask"], out2["mask"])
    Rs = 16
    with torch.no_grad():
        ref = cpu_ref.get_outputs(P, fs, cpu_ref.ModelSpec(S, S, 8, 8), o[-Rs:], d[-Rs:], pa[-Rs:],
                                  torch.zeros(Rs, 1), torch.full((Rs, 1), 6.0), training=False)
    assert max_abs(out["mid_rgb_fine"][-Rs:].cpu(), ref["mid_rgb_fine"]) <= TOL
    assert max_abs(out["accumulation_fine"][-Rs:].cpu(), ref["accumulation_fine"]) <= TOL


# ---------------------------------------------------------------------------------------------- §8(f): loss + optimiser
def test_fused_loss_matches_torch_formulas(dev):
    """get_loss_dict (model.py:346-430) through rsn_loss_forward_backward: values and gradients against the plain
    torch formulas, including a zeroed (warm-up) coefficient."""
    from reflect_sampling_nerf_amd.train_ops import LOSS_TERMS, fused_loss_dict

    g = torch.Generator().manual_seed(5)
    R, Sc, Sf = 37, 11, 19
    mk = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    leaves = {"mid_rgb_coarse": mk(R, 3), "mid_rgb_fine": mk(R, 3), "mid_reflect_coarse": mk(R, 3),
              "mid_reflect_fine": mk(R, 3), "pred_normals_coarse": mk(R, Sc, 3) - 0.5,
              "pred_normals_fine": mk(R, Sf, 3) - 0.5, "n_dot_d_coarse": mk(R, Sc, 1) - 0.5,
              "n_dot_d_fine": mk(R, Sf, 1) - 0.5}
    consts = {"weights_coarse": mk(R, Sc, 1), "weights_fine": mk(R, Sf, 1), "normals_coarse": mk(R, Sc, 3) - 0.5,
              "normals_fine": mk(R, Sf, 3) - 0.5}
    image = mk(R, 3)
    coef = {k: c for k, c in zip(LOSS_TERMS, (1.0, 1.0, 1.0, 1.0, 3e-5, 0.0, 1e-2, 1e-1))}
    ref_in = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
    o = {**ref_in, **consts}
    ref = {
        "loss_mid_coarse": torch.nn.functional.mse_loss(image, o["mid_rgb_coarse"]),
        "loss_mid_fine": torch.nn.functional.mse_loss(image, o["mid_rgb_fine"]),
        "loss_reflect_mid_coarse": torch.nn.functional.mse_loss(image, o["mid_reflect_coarse"]),
        "loss_reflect_mid_fine": torch.nn.functional.mse_loss(image, o["mid_reflect_fine"]),
        "predicted_normal_loss_coarse": torch.sum(o["weights_coarse"] * torch.sum((o["normals_coarse"] - o["pred_normals_coarse"]) ** 2, dim=-1, keepdim=True)),
        "predicted_normal_loss_fine": torch.sum(o["weights_fine"] * torch.sum((o["normals_fine"] - o["pred_normals_fine"]) ** 2, dim=-1, keepdim=True)),
        "orientation_loss_coarse": torch.sum(o["weights_coarse"] * torch.max(torch.zeros_like(o["n_dot_d_coarse"]), o["n_dot_d_coarse"]) ** 2),
        "orientation_loss_fine": torch.sum(o["weights_fine"] * torch.max(torch.zeros_like(o["n_dot_d_fine"]), o["n_dot_d_fine"]) ** 2),
    }
    ref = {k: v * coef[k] for k, v in ref.items()}
    sum(ref.values()).backward()
    gpu_in = {k: v.clone().to(dev).requires_grad_(True) for k, v in leaves.items()}
    out = fused_loss_dict({**gpu_in, **{k: v.to(dev) for k, v in consts.items()}}, image.to(dev), coef)
    assert list(out.keys()) == list(LOSS_TERMS)
    for k in LOSS_TERMS:
        assert abs(float(out[k]) - float(ref[k])) <= 1e-5 * max(1.0, abs(float(ref[k]))), k
    sum(out.values()).backward()
    for k in leaves:
        assert max_abs(gpu_in[k].grad.cpu(), ref_in[k].grad) <= 1e-6, k


def test_fused_radam_matches_torch_optim(dev):
    """rsn_radam_step against torch.optim.RAdam (lr 1e-3, eps 1e-15: reference config.py:50-53) over 12 steps, i.e.
    across the rho_t > 5 switch (step 6), with a parameter that never receives a gradient."""
    g = torch.Generator().manual_seed(3)
    shapes = [(17, 9), (33,), (256, 99), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes] + [torch.nn.Parameter(torch.ones(5))]
    gpu_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    opt_ref = torch.optim.RAdam(ref_p, lr=1e-3, eps=1e-15)
    opt_gpu = pkg.FusedRAdam(gpu_p, lr=1e-3, eps=1e-15)
    for step in range(12):
        for pr, pg in zip(ref_p[:-1], gpu_p[:-1]):
            gr = torch.randn(*pr.shape, generator=g) * (0.1 + step)
            pr.grad, pg.grad = gr.clone(), gr.clone().to(dev)
        v_before = gpu_p[0]._version
        opt_ref.step()
        opt_gpu.step()
        assert gpu_p[0]._version > v_before  # consumers keyed on Tensor._version (packed weights) see the update
    for pr, pg in zip(ref_p, gpu_p):
        assert max_abs(pg.detach().cpu(), pr.detach()) <= 2e-6
    assert pkg.exponential_decay_lr(0) == 1e-3 and abs(pkg.exponential_decay_lr(50000) - 1e-4) < 1e-12
    assert abs(pkg.exponential_decay_lr(25000) - (1e-3 * 1e-4) ** 0.5) < 1e-12
    # checkpoint / resume in torch.optim.RAdam's own state_dict format, both ways: a fresh FusedRAdam takes torch's state, a fresh
    # torch.optim.RAdam takes FusedRAdam's, and three more steps on the same gradients keep all four trajectories together
    ref2 = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    gpu2 = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    opt_ref2 = torch.optim.RAdam(ref2, lr=1e-3, eps=1e-15)
    opt_ref2.load_state_dict(opt_gpu.state_dict())
    opt_gpu2 = pkg.FusedRAdam(gpu2, lr=5e-2, eps=1e-3)  # hyper-parameters come from the checkpoint
    opt_gpu2.load_state_dict(opt_ref.state_dict())
    assert opt_gpu2.step_count == 12 and opt_gpu2.lr == 1e-3 and opt_gpu2.eps == 1e-15
    for step in range(12, 15):
        for k in range(len(shapes)):
            gr = torch.randn(*shapes[k], generator=g) * (0.1 + step)
            for plist in (ref_p, ref2):
                plist[k].grad = gr.clone()
            for plist in (gpu_p, gpu2):
                plist[k].grad = gr.clone().to(dev)
        for o_ in (opt_ref, opt_ref2, opt_gpu, opt_gpu2):
            o_.step()
    for k in range(len(ref_p)):
        for other in (ref2[k].detach(), gpu_p[k].detach().cpu(), gpu2[k].detach().cpu()):
            assert max_abs(other, ref_p[k].detach()) <= 3e-6, k


@pytest.mark.parametrize("n_out,k_in,ld_dy,ld_x", [
    (256, 256, 256, 256),  # trunk layer: vector loads of X and dY
    (256, 104, 256, 104),  # encoded inputs (column map in the training graph)
    (128, 40, 128, 40),    # mlp_mid, SH part: dwordx2 X loads
    (16, 256, 16, 256),    # heads block: one row block, waves split the points
    (3, 128, 4, 128),      # RGB head
    (250, 99, 251, 99),    # odd sizes: scalar-load fallback for both operands
    (37, 130, 38, 132),    # X vector path impossible (130 % 8): scalar loads, odd row count
])
def test_weight_grad_segments_and_shapes(dev, n_out, k_in, ld_dy, ld_x):
    """rsn_weight_grad_multi: dW = sum over all segments of dY^T X, db = column sums; against fp64 matmul.
    Segment lengths cover empty segments, lengths below one pipeline stage and non-multiples of the stage."""
    from reflect_sampling_nerf_amd.train_graph import _wgrad_multi

    g = torch.Generator().manual_seed(n_out * 1000 + k_in)
    lens = [1000, 0, 37, 5003, 3]
    segs, ref_w, ref_b = [], torch.zeros(n_out, k_in, dtype=torch.float64), torch.zeros(n_out, dtype=torch.float64)
    for n in lens:
        dy = torch.randn(n, ld_dy, generator=g)
        x = torch.randn(n, ld_x, generator=g)
        ref_w += dy[:, :n_out].double().t() @ x[:, :k_in].double()
        ref_b += dy[:, :n_out].double().sum(0)
        segs.append((dy.to(dev), x.to(dev)))
    dw = torch.zeros(n_out, k_in + 5, device=dev)  # leading dimension > k_in, accumulate at a column offset
    db = torch.zeros(n_out, device=dev)
    _wgrad_multi(segs, n_out, k_in, dw, 5, db)
    _wgrad_multi(segs[:1], n_out, k_in, dw, 5, None)  # accumulation, no bias
    ref_w1 = segs[0][0][:, :n_out].double().cpu().t() @ segs[0][1][:, :k_in].double().cpu()
    scale = float(ref_w.abs().max())
    assert float((dw[:, 5:].double().cpu() - ref_w - ref_w1).abs().max()) <= 2e-5 * scale
    assert float(dw[:, :5].abs().max()) == 0.0
    assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())


@pytest.mark.parametrize("mode", ["bf16", "bf16x6"])
@pytest.mark.parametrize("n_out,k_in", [(256, 256), (256, 104), (128, 40), (16, 256), (250, 99)])
def test_weight_grad_mma_modes(dev, monkeypatch, n_out, k_in, mode):
    """rsn_weight_grad_multi_mode.  bf16 (the opt-in reduced-precision training mode): operands rounded to bf16 inside the
    kernel, fp32 accumulation -> equals the fp64 product of the bf16-ROUNDED operands to fp32 accumulation accuracy.
    bf16x6: operands split into bf16 triples, 6 products -> meets the exact kernel's bound against the fp64 product.
    The bias sums stay exact fp32, and shapes the vector loads cannot take fall back to the exact kernel."""
    from reflect_sampling_nerf_amd import train_graph

    g = torch.Generator().manual_seed(n_out + k_in)
    lens = [1000, 0, 37, 5003, 3]
    segs = []
    ref_w, ref_b = torch.zeros(n_out, k_in, dtype=torch.float64), torch.zeros(n_out, dtype=torch.float64)
    exact_w = torch.zeros(n_out, k_in, dtype=torch.float64)
    for n in lens:
        dy, x = torch.randn(n, n_out, generator=g), torch.randn(n, k_in, generator=g)
        ref_w += dy.bfloat16().double().t() @ x.bfloat16().double()
        exact_w += dy.double().t() @ x.double()
        ref_b += dy.double().sum(0)
        segs.append((dy.to(dev), x.to(dev)))
    dw, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
    monkeypatch.setattr(train_graph, "_WGRAD_MODE", {"bf16": 3, "bf16x6": 1}[mode])
    train_graph._wgrad_multi(segs, n_out, k_in, dw, 0, db)
    got = dw.double().cpu()
    if mode == "bf16x6":
        assert float((got - exact_w).abs().max()) <= 2e-5 * float(exact_w.abs().max())
        assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())
        return
    vector_path = n_out > 32 and k_in % (8 if k_in > 128 else 4 if k_in > 64 else 2) == 0
    target = ref_w if vector_path else exact_w
    assert float((got - target).abs().max()) <= 2e-5 * float(target.abs().max())
    assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())
    assert float((got - exact_w).abs().max()) <= 2e-2 * float(exact_w.abs().max())  # bf16 rounding of both operands


@pytest.mark.parametrize("n_out,k_in,x_bf16,dy_bf16", [(256, 256, True, True), (256, 104, False, True), (128, 40, False, True),
                                                       (128, 256, True, True), (16, 256, True, False), (3, 128, True, False),
                                                       (64, 64, True, True), (256, 128, True, True)])
def test_weight_grad_bf16_rows(dev, monkeypatch, n_out, k_in, x_bf16, dy_bf16):
    """rsn_weight_grad_multi_dev with operand rows that ARE bf16 in memory (the reduced-precision training mode keeps
    its wide buffers as bf16: rsn_field_saved / rsn_field_grads_out): every operand combination the training step
    produces -- wide layers (both bf16), encoded / SH inputs (fp32 X, bf16 dY), heads and RGB head (bf16 X, narrow fp32 dY)
    -- over several segments, one of them cut by a DEVICE-side count.  Equals the fp64 product of the bf16 values to fp32
    accumulation accuracy; bias sums are sums of the stored (bf16) rows."""
    from reflect_sampling_nerf_amd import train_graph

    g = torch.Generator().manual_seed(7 * n_out + k_in)
    lens = [1000, 37, 5003, 640]
    count = torch.tensor([9], dtype=torch.int32, device=dev)  # the last segment holds 9 x 64 = 576 of its 640 rows
    segs = []
    ref_w, ref_b = torch.zeros(n_out, k_in, dtype=torch.float64), torch.zeros(n_out, dtype=torch.float64)
    for si, n in enumerate(lens):
        dy, x = torch.randn(n, n_out, generator=g), torch.randn(n, k_in, generator=g)
        ld_dy = n_out + (n_out & 1) if n_out > 32 else (16 if n_out > 4 else 4)
        dyp = torch.zeros(n, ld_dy)
        dyp[:, :n_out] = dy
        dyd = dyp.to(dev).bfloat16() if dy_bf16 else dyp.to(dev)
        xd = x.to(dev).bfloat16() if x_bf16 else x.to(dev)
        live = 576 if si == 3 else n
        # (rows that arrive as fp32 are rounded to bf16 by the kernel as it packs them: the mode's arithmetic)
        ref_w += dyd[:live, :n_out].bfloat16().double().cpu().t() @ xd[:live].bfloat16().double().cpu()
        ref_b += dyd[:live, :n_out].double().cpu().sum(0)
        segs.append((dyd, xd, (count, 64)) if si == 3 else (dyd, xd))
    dw, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
    monkeypatch.setattr(train_graph, "_WGRAD_MODE", 3)
    train_graph._wgrad_multi(segs, n_out, k_in, dw, 0, db)
    assert float((dw.double().cpu() - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max())
    assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])  # exact fp32, split-bf16 (fp32-equivalent)
@pytest.mark.parametrize("n_jobs,n_out,k_in", [(8, 256, 256), (3, 256, 256), (2, 256, 104), (5, 64, 64)])
def test_weight_grad_jobs_equal_separate_launches(dev, monkeypatch, mode, n_jobs, n_out, k_in):
    """rsn_weight_grad_jobs: several reductions of one shape over the same segments (one cut by a DEVICE-side count) in ONE
    launch -- the workgroups are dealt to the jobs, each flushes one tile -- against the fp64 products, the bound of the
    single-job kernel; the destination column offset / column map / missing bias of a job are its own
    (reference: autograd of the trunk's Linear layers, reflect_sampling_nerf_field.py:54-60)."""
    from reflect_sampling_nerf_amd import train_graph

    g = torch.Generator().manual_seed(31 * n_jobs + k_in + mode)
    lens = [700, 33, 2051, 640]
    count = torch.tensor([7], dtype=torch.int32, device=dev)  # the last segment holds 7 x 64 = 448 of its 640 rows
    cmap = torch.randperm(k_in, generator=g).to(torch.int32).to(dev) if k_in == 104 else None
    jobs, refs = [], []
    for jb in range(n_jobs):
        segs = []
        ref_w, ref_b = torch.zeros(n_out, k_in, dtype=torch.float64), torch.zeros(n_out, dtype=torch.float64)
        for si, n in enumerate(lens):
            dy, x = torch.randn(n, n_out, generator=g), torch.randn(n, k_in, generator=g)
            live = 448 if si == 3 else n
            ref_w += dy[:live].double().t() @ x[:live].double()
            ref_b += dy[:live].double().sum(0)
            segs.append((dy.to(dev), x.to(dev), (count, 64)) if si == 3 else (dy.to(dev), x.to(dev)))
        c0 = 5 if jb == 1 else 0                      # job 1 lands at a column offset of a wider matrix ...
        dw = torch.zeros(n_out, k_in + c0, device=dev)
        db = None if jb == 1 else torch.zeros(n_out, device=dev)   # ... and has no bias
        jobs.append((segs, dw, c0, db, cmap))
        refs.append((ref_w, ref_b, c0))
    monkeypatch.setattr(train_graph, "_WGRAD_MODE", mode)
    train_graph._wgrad_jobs(jobs, n_out, k_in)
    for (segs, dw, c0, db, _), (ref_w, ref_b, _) in zip(jobs, refs):
        got = dw[:, c0:].double().cpu()
        if cmap is not None:
            exp = torch.zeros_like(ref_w)
            exp[:, cmap.cpu().long()] = ref_w
        else:
            exp = ref_w
        assert float((got - exp).abs().max()) <= 2e-5 * float(ref_w.abs().max())
        if c0:
            assert float(dw[:, :c0].abs().max()) == 0.0
        if db is not None:
            assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())


@pytest.mark.parametrize("path", ["f32", "bf16", "bf16x6", "jobs"])
@pytest.mark.parametrize("n_out,k_in", [(3, 128), (16, 256), (250, 99), (250, 256), (37, 130)])
def test_weight_grad_flush_stays_inside_its_rows(dev, monkeypatch, path, n_out, k_in):
    """The weight-gradient flush adds a 32-row (64-row) register tile into dW by buffer atomics; rows >= n_out of the tile
    hold clamped duplicates of live rows and MUST be dropped.  In the training step dW is a view into the flat gradient
    buffer (train_graph._GradAcc), so a missed drop would land in the next parameter's gradient.  Here dW is the first
    n_out rows of a larger zero buffer: the rows behind it must stay exactly 0 (reference: autograd writes a Linear's
    weight gradient into that parameter only, reflect_sampling_nerf_field.py:54-86)."""
    from reflect_sampling_nerf_amd import train_graph

    g = torch.Generator().manual_seed(5 * n_out + k_in)
    lens = [1000, 37, 2051]
    segs, ref_w = [], torch.zeros(n_out, k_in, dtype=torch.float64)
    ld_dy = n_out + (n_out & 1) if n_out > 32 else (16 if n_out > 4 else 4)
    for n in lens:
        dy, x = torch.randn(n, ld_dy, generator=g) + 0.5, torch.randn(n, k_in, generator=g) + 0.5
        ref_w += dy[:, :n_out].double().t() @ x.double()
        segs.append((dy.to(dev), x.to(dev)))
    rows_total = n_out + 70  # more than one register tile of rows behind the live ones
    big = torch.zeros(rows_total, k_in, device=dev)
    bias_big = torch.zeros(rows_total, device=dev)
    dw, db = big[:n_out], bias_big[:n_out]
    monkeypatch.setattr(train_graph, "_WGRAD_MODE", {"f32": 0, "bf16": 3, "bf16x6": 1, "jobs": 0}[path])
    if path == "jobs":
        big2 = torch.zeros(rows_total, k_in, device=dev)
        train_graph._wgrad_jobs([(segs, dw, 0, db), (segs, big2[:n_out], 0, None)], n_out, k_in)
        assert float(big2[n_out:].abs().max()) == 0.0
        assert float((big2[:n_out].double().cpu() - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max())
    else:
        train_graph._wgrad_multi(segs, n_out, k_in, dw, 0, db)
    tol = 2e-2 if path == "bf16" else 2e-5
    assert float((dw.double().cpu() - ref_w).abs().max()) <= tol * float(ref_w.abs().max())
    assert float(big[n_out:].abs().max()) == 0.0, "the flush wrote behind the n_out live rows of dW"
    assert float(bias_big[n_out:].abs().max()) == 0.0



def test_standalone_sh34_encoding_matches_reference_golden(dev):
    """IntegratedSHEncoding called as a module (rsn_sh34_encode) against the output of the reference's own
    IntegratedSHEncoding.forward (tests/golden/units.npz, oracle/make_golden.py) and against the oracle."""
    _, g = load_golden("units")
    enc = pkg.ReflectSamplingNeRFNerfField().direction_encoding
    dirs, rough = g["sh"]["dirs"], g["sh"]["roughness"]
    out = enc(dirs.to(dev), rough.to(dev)).cpu()
    assert out.shape == (257, 34)
    assert max_abs(out, g["sh"]["out"]) <= 1e-5
    assert max_abs(out, cpu_ref.integrated_sh(dirs, rough)) <= 1e-5
    # leading batch dims, no roughness (zero attenuation exponent)
    d2 = torch.nn.functional.normalize(torch.randn(3, 5, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
    o2 = enc(d2.to(dev)).cpu()
    assert o2.shape == (3, 5, 34)
    assert max_abs(o2, cpu_ref.integrated_sh(d2, torch.zeros(3, 5, 1))) <= 1e-5
    # the unattenuated basis under the reference's method name (components.py:52-129)
    assert torch.equal(enc.pytorch_fwd(d2.to(dev)).cpu(), o2) and max_abs(o2, cpu_ref.sh34_basis(d2)) <= 1e-5


def test_standalone_ipe_encoding_matches_oracle(dev):
    """NeRFEncoding called as a module (rsn_ipe_encode): reference column order [sin 48 | cos 48 | raw 3]."""
    fs = cpu_ref.FieldSpec()
    enc = pkg.ReflectSamplingNeRFNerfField().position_encoding
    gen = torch.Generator().manual_seed(3)
    mean = (torch.rand(4, 33, 3, generator=gen) - 0.5) * 4.0  # contracted means live in the radius-2 ball
    A = torch.randn(4, 33, 3, 3, generator=gen) * 0.02
    cov = A @ A.transpose(-1, -2)
    out = enc(mean.to(dev), covs=cov.to(dev)).cpu()
    ref = cpu_ref.ipe(fs, mean, torch.diagonal(cov, dim1=-2, dim2=-1))
    assert out.shape == (4, 33, 99)
    # sin of arguments up to 2*pi*2*2^16: fp32 argument rounding is shared, the device sin is <= 1.5 ulp
    assert max_abs(out, ref) <= 2e-6
    out_nc = enc(mean.to(dev)).cpu()
    assert max_abs(out_nc, cpu_ref.ipe(fs, mean, None)) <= 2e-6


def test_empty_ray_bundle(dev):
    """R = 0: every output exists with a zero-length leading dimension; nothing is launched on an empty batch."""
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=8, num_importance_samples=8, num_reflect_coarse_samples=4,
                                            num_reflect_importance_samples=4, base_mlp_num_layers=4,
                                            base_mlp_layer_width=64)
    model = cfg.setup(scene_box=None, num_train_data=1).to(dev).eval()
    z = torch.zeros(0, 3, device=dev)
    out = model(pkg.RayBundle(origins=z, directions=z, pixel_area=torch.zeros(0, 1, device=dev)))
    assert tuple(out["mid_rgb_fine"].shape) == (0, 3) and tuple(out["mask"].shape) == (0,)
    assert tuple(out["weights_fine"].shape) == (0, 8, 1) and tuple(out["pred_normals_coarse"].shape) == (0, 8, 3)
    assert "depth_reflect_fine" not in out


def test_composite_non_finite_and_saturated_density(dev):
    """RaySamples.get_weights (N5) ends in nan_to_num: infinite / NaN / huge densities must give the oracle's weights
    (no NaN leaves the compositor), also when a bin has zero width."""
    R, S = 6, 9
    g = torch.Generator().manual_seed(11)
    eb = torch.sort(torch.rand(R, S + 1, generator=g) * 4.0 + 2.0, dim=-1).values
    eb[1, 3] = eb[1, 4]  # zero-width bin
    sigma = torch.rand(R, S, generator=g) * 3.0
    sigma[0, 2] = float("inf")
    sigma[2, 0] = 1e30
    sigma[3, 5] = float("nan")
    sigma[4, :] = 0.0
    color = torch.rand(R, S, 3, generator=g)
    out = ops.composite(R, None, S, 1, ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB, sigma.to(dev), eb.to(dev),
                        color.to(dev))
    w_ref = cpu_ref.weights_from_density(sigma[..., None], eb[:, :-1], eb[:, 1:])[..., 0]
    w = out["weights"].cpu()
    assert bool(torch.isfinite(w).all())
    assert max_abs(w, w_ref) <= 1e-6
    rgb_ref = cpu_ref.composite_rgb(color, w_ref[..., None], torch.ones(3), training=False)
    assert max_abs(out["rgb"].cpu(), torch.clip(rgb_ref, 0.0, 1.0)) <= 1e-5


@pytest.mark.parametrize("kind,tan,near,far", [("uniform", 1.0, 2.0, 6.0), ("reciprocal", 0.25, 0.0, 256.0)])
def test_pdf_sampler_degenerate_histograms(dev, kind, tan, near, far):
    """Inverse-CDF resampling (N9) on histograms that exercise its guards: one-hot weights (flat CDF runs: 0/0 -> 0),
    weights far below the 1e-5 padding threshold, a single dominant bin at either end, equal weights."""
    R, S_in, S_out = 8, 12, 40
    g = torch.Generator().manual_seed(17)
    w = torch.zeros(R, S_in, 1)
    w[0, 5] = 1.0
    w[1] = 1e-12
    w[2, 0] = 1.0
    w[3, S_in - 1] = 1.0
    w[4] = 1.0 / S_in
    w[5] = torch.rand(S_in, 1, generator=g) * 1e-7
    w[6, 3], w[6, 9] = 0.5, 0.5
    w[7] = torch.rand(S_in, 1, generator=g)
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    sb_in, _ = cpu_ref.spaced_bins(kind, tan, nears, fars, S_in, None)
    code = RSN_SPACING_UNIFORM if kind == "uniform" else RSN_SPACING_RECIPROCAL
    for u_rand in (None, torch.rand(R, S_out + 1, generator=g)):
        sb_ref, eb_ref = cpu_ref.pdf_bins(kind, tan, nears, fars, w, sb_in, S_out, u_rand)
        sb, eb = ops.sample_pdf(R, None, S_in, S_out, code, tan, 0.01, nears.reshape(R).to(dev), fars.reshape(R).to(dev),
                                w[..., 0].contiguous().to(dev), sb_in.contiguous().to(dev),
                                None if u_rand is None else u_rand.to(dev))
        assert bool(torch.isfinite(sb).all()) and bool(torch.isfinite(eb).all())
        assert max_abs(sb.cpu(), sb_ref) <= 1e-5
        # euclidean bins of the reciprocal spacing reach 256: relative bound
        assert float(((eb.cpu() - eb_ref).abs() / (1.0 + eb_ref.abs())).max()) <= 2e-5


@pytest.mark.parametrize("name", ["trainstep_l8_w64", "trainstep_l8_w256", "trainstep_l4_w128",
                                  "trainstep_trained_l8_w64", "trainstep_trained_l8_w256",  # trained_*: TRAINED weights
                                  "trainstep_l6_w48", "trainstep_l8_w200"])  # widths the kernels run zero-padded
@pytest.mark.parametrize("inject_bins,mma", [(True, "f32"), (False, "f32"), (True, "bf16x6")])
def test_train_step_against_reference_fixture(dev, name, inject_bins, mma):
    """One whole training step of the REFERENCE itself (tests/golden/trainstep_*.npz: get_outputs in train mode, its
    own get_loss_dict, backward; generated by oracle/make_golden.py) against the HIP path with the reference's
    parameters, rays, logged jitter and target image: outputs, the eight loss terms, every parameter gradient.

    inject_bins=True: the sampler outputs the reference logged (fine / reflect bins, model.py:182,292,317) replace the
    HIP samplers' results, so both pipelines evaluate IDENTICAL sample positions -> EVERY parameter gradient must be
    within 2e-4 of the tensor's largest entry.  inject_bins=False is the free-running pipeline (its own PDF
    resampling, ~1e-6 from the reference's): tight everywhere above the skip layer, direction + 5 % below it (a few
    ulp-wide resampled bins flip near-zero ReLU units of the two encoding-consuming layers).
    mma="bf16x6": the fp32-equivalent split-bf16 mode (forward, sweeps, weight gradients) meets the SAME bounds on the
    reference's logged bins."""
    meta, g = load_golden(name)
    s = meta["samples"]
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=s[0], num_importance_samples=s[1],
                                            num_reflect_coarse_samples=s[2], num_reflect_importance_samples=s[3],
                                            base_mlp_num_layers=meta["layers"], base_mlp_layer_width=meta["width"])
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(g["param"])
    model.to(dev).train()
    model.field.set_mma_mode(mma)
    assert dict(model.config.loss_coefficients) == pytest.approx(meta["loss_coefficients"])
    i = g["in"]
    rb = pkg.RayBundle(origins=i["origins"].to(dev), directions=i["directions"].to(dev),
                       pixel_area=i["pixel_area"].to(dev), nears=i["nears"].to(dev), fars=i["fars"].to(dev))
    out = model._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in g["jitter"].items()},
                                   bins=g["bins"] if inject_bins else None)
    ref = g["out"]
    assert sorted(out.keys()) == sorted(ref.keys())
    assert torch.equal(out["mask"].cpu().to(torch.uint8), ref["mask"])
    trained_free = "trained" in name and not inject_bins  # trained weights on the pipeline's OWN resampled positions
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
              "accumulation_fine", "weights_coarse", "weights_fine", "diff", "tint", "roughness"):
        # (reflected rays of a trained field on free-running positions: 1e-3, see test_get_outputs_on_reference_golden_rays;
        # measured 3.6e-4 at 8 x 64, profiles/r04_trained_fixture_report.json)
        assert max_abs(out[k].detach().cpu(), ref[k]) <= (1e-3 if trained_free and "reflect" in k else TOL), k
    for k in ("pred_normals_coarse", "pred_normals_fine", "n_dot_d_coarse", "n_dot_d_fine"):
        assert max_abs(out[k].detach().cpu(), ref[k]) <= (3e-3 if "trained" in name else TOL_UNIT), k
    for lvl in ("coarse", "fine"):
        e = (out[f"normals_{lvl}"].cpu() - ref[f"normals_{lvl}"]).abs()
        assert float(e.mean()) <= (3e-3 if "trained" in name else 1e-3) and float(e.flatten().quantile(0.99)) <= (3e-2 if "trained" in name else 5e-3), lvl
    # the analytic normals are a detached loss target in both pipelines (checked above within their conditioning):
    # give the loss the reference's constant so that the gradient comparison sees the same target
    checked = dict(out)
    checked["normals_coarse"], checked["normals_fine"] = ref["normals_coarse"].to(dev), ref["normals_fine"].to(dev)
    losses = model.get_loss_dict(checked, {"image": i["image"].to(dev)})
    assert sorted(losses) == sorted(g["loss"])
    for k, v in g["loss"].items():
        assert abs(float(losses[k].detach()) - float(v)) <= (5e-4 if trained_free else 5e-5) * max(abs(float(v)), 1e-3), k
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    skip = 4 if meta["layers"] > 5 else -1
    report = []
    for name_p, p in model.field.named_parameters():
        if name_p not in g["grad"]:
            assert p.grad is None, name_p
            continue
        gr = g["grad"][name_p]
        a, b = p.grad.cpu().flatten().double(), gr.flatten().double()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        err_max = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-300)
        report.append((name_p, err_max, rel))
        if inject_bins:
            # trained weights: 1e-3 (measured 4.5e-4 at 8 x 64, 2.5e-6 at 8 x 256: profiles/r04_trained_fixture_report.json;
            # the trained field is ill-conditioned in fp32 -- 2e-2 from its own fp64 evaluation, profiles/r04_fp64_arbiter_*)
            bound = 1e-3 if "trained" in name else 2e-4
            enc_layer = name_p.startswith("mlp_base.layers.") and int(name_p.split(".")[2]) in (0, skip)
            if err_max > bound and enc_layer:
                # The two layers that consume the encoding directly: identical bins still leave the contracted means an ulp apart
                # (torch's vectorised CPU sqrt is not correctly rounded), the undamped IPE frequencies turn that into ~1e-3 of a
                # feature, and a unit whose pre-activation sits at zero for some sample flips its ReLU -- which changes THAT
                # unit's row of this layer's weight / bias gradient and nothing else (seen at unit 173 of trainstep_l8_w200:
                # every other row and every other tensor agree to 2e-5).  At most one unit in a hundred may do so.
                rows = (p.grad.cpu().double() - gr.double()).abs().reshape(gr.shape[0], -1).max(dim=1).values / float(gr.abs().max())
                n_off = int((rows > bound).sum())
                assert n_off <= max(1, gr.shape[0] // 100), f"{name_p}: {n_off} rows over {bound:.0e} (identical bins)"
                continue
            assert err_max <= bound, f"{name_p}: max abs err / tensor max {err_max:.3e} (identical bins)"
            continue
        below = name_p.startswith("mlp_base.layers.") and (skip < 0 or int(name_p.split(".")[2]) <= skip)
        if "trained" in name:  # free-running on trained weights: direction and size everywhere (measured cos 0.99997, rel 1.5e-2)
            assert cos >= 0.999 and rel <= 5e-2, f"{name_p}: cos {cos:.6f} rel-L2 {rel:.3e}"
            continue
        if below:
            assert cos >= 0.999 and rel <= 5e-2, f"{name_p}: cos {cos:.6f} rel-L2 {rel:.3e}"
        else:
            assert rel <= 5e-4, f"{name_p}: rel-L2 {rel:.3e}"
    print(f"{name} inject_bins={inject_bins} mma={mma}: worst gradient error / tensor max "
          f"{max(r[1] for r in report):.2e} ({max(report, key=lambda r: r[1])[0]})")


@pytest.mark.parametrize("name", ["trainstep_l8_w256", "trainstep_trained_l8_w256"])
def test_train_step_bf16_against_reference_fixture(dev, name):
    """The reduced-precision training mode (bf16 MFMA operands, fp32 accumulation; at width 256 the LDS-ring kernels of
    rsn_field_bf16_train.hip: forward with the in-kernel analytic-normal sweep, backward sweep, weight gradients over bf16 rows)
    against one whole training step of the REFERENCE itself in fp32 (tests/golden/trainstep_*.npz: random-init and TRAINED
    weights of the BASELINE network), on the reference's logged bins: rendered outputs within the bf16 tolerance (3e-2), the eight
    loss terms within 5 %, every parameter gradient in direction (cosine >= 0.99) and size (rel-L2 <= 1e-1).  The reference
    trains under autocast itself (config.py:33); this is the precision class its shipped configuration runs in."""
    meta, g = load_golden(name)
    s = meta["samples"]
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=s[0], num_importance_samples=s[1],
                                            num_reflect_coarse_samples=s[2], num_reflect_importance_samples=s[3],
                                            base_mlp_num_layers=meta["layers"], base_mlp_layer_width=meta["width"])
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(g["param"])
    model.to(dev).train()
    model.field.set_mma_mode("bf16")
    assert model.field.train_layout()["enc_cols"] == 128  # the ring kernels' layout: this test runs rsn_field_bf16_train.hip
    i = g["in"]
    rb = pkg.RayBundle(origins=i["origins"].to(dev), directions=i["directions"].to(dev),
                       pixel_area=i["pixel_area"].to(dev), nears=i["nears"].to(dev), fars=i["fars"].to(dev))
    out = model._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in g["jitter"].items()}, bins=g["bins"])
    ref = g["out"]
    assert sorted(out.keys()) == sorted(ref.keys())
    flips = int((out["mask"].cpu().to(torch.uint8) != ref["mask"]).sum())
    assert flips <= max(1, meta["R"] // 8), flips  # a threshold decision on a bf16-evaluated field
    same = (out["mask"].cpu().to(torch.uint8) == ref["mask"])
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "accumulation_coarse", "accumulation_fine", "diff", "tint", "roughness"):
        assert max_abs(out[k].detach().cpu(), ref[k]) <= 3e-2, k
    for k in ("mid_reflect_coarse", "mid_reflect_fine"):
        assert max_abs(out[k].detach().cpu()[same], ref[k][same]) <= 6e-2, k
    for lvl in ("coarse", "fine"):  # analytic normals from a bf16 sweep: direction
        a, b = out[f"normals_{lvl}"].cpu(), ref[f"normals_{lvl}"]
        w = ref[f"weights_{lvl}"][..., 0]
        cosn = (a * b).sum(-1)
        assert float((cosn * w).sum() / w.sum()) >= 0.97, lvl  # weight-averaged cosine (where the loss looks)
    if flips:
        return  # a flipped reflection mask changes which rays the reflect losses see: the scalar comparisons below need equal masks
    checked = dict(out)
    checked["normals_coarse"], checked["normals_fine"] = ref["normals_coarse"].to(dev), ref["normals_fine"].to(dev)
    losses = model.get_loss_dict(checked, {"image": i["image"].to(dev)})
    for k, v in g["loss"].items():
        assert abs(float(losses[k].detach()) - float(v)) <= 5e-2 * max(abs(float(v)), 1e-3), k
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    for name_p, p in model.field.named_parameters():
        if name_p not in g["grad"]:
            assert p.grad is None, name_p
            continue
        a, b = p.grad.cpu().flatten().double(), g["grad"][name_p].flatten().double()
        assert bool(torch.isfinite(a).all()), name_p
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        assert cos >= 0.99 and rel <= 1e-1, f"{name_p}: cos {cos:.5f} rel-L2 {rel:.3e}"


# ---------------------------------------------------------------------------------------------- the step without host reads
def _train_setup(dev, R, samples, layers=8, width=128, bias_shift=2.0, seed=0):
    torch.manual_seed(seed)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=samples[0], num_importance_samples=samples[1],
                                            num_reflect_coarse_samples=samples[2], num_reflect_importance_samples=samples[3],
                                            base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += bias_shift
    model.to(dev).train()
    o, d, pa = cpu_ref.synthetic_rays(R, seed=seed)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    batch = {"image": torch.rand(R, 3, generator=torch.Generator().manual_seed(seed + 1)).to(dev)}
    return model, rb, batch


def test_training_step_issues_no_device_to_host_read(dev):
    """The reflected-ray count M (reference model.py:229,259: a host-side `mask.any()` / boolean gather) stays on the device
    for the whole optimisation step -- forward, loss, backward, weight gradients (their segment lengths are device-side),
    RAdam.  torch's sync debug mode turns any implicit device-to-host synchronisation into an error."""
    from reflect_sampling_nerf_amd.parallel import train_step

    model, rb, batch = _train_setup(dev, 192, (24, 24, 16, 16))
    opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=1e-3, eps=1e-15)
    train_step(model, rb, batch, opt, None, 100)  # warm-up: one-time uploads (coefficient tensor, column maps, pack table)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for k in range(3):
            loss = train_step(model, rb, batch, opt, None, 101 + k)
            outputs = model(rb)  # a bare forward too: the lazy [M, 1] entry must not be touched by building the dict
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bool(torch.isfinite(loss))
    M = model._last_num_reflected  # reading it is the (explicit) host read
    assert 0 < M <= 192
    assert outputs["depth_reflect_fine"].shape == (M, 1) and "depth_reflect_fine" in outputs.keys()


@pytest.mark.parametrize("mma", ["f32", "bf16x6"])
def test_weight_grad_groups_two_equals_one(dev, mma):
    """Opt-in `model.weight_grad_groups = 2` (train_graph backward): the reflect branch's weight gradients are reduced right behind
    its backward sweeps and its saved rows / sweep outputs are released before the primary levels' sweep outputs are allocated.
    Same sample draws, same reductions in two launches per layer instead of one: gradients equal to the order of the atomics,
    peak memory lower, still no device-to-host read in the step."""
    from reflect_sampling_nerf_amd.parallel import train_step

    R = 704
    grads, peak = {}, {}
    for groups in (1, 2):
        model, rb, batch = _train_setup(dev, R, (16, 16, 16, 16), layers=8, width=256, bias_shift=1.0)
        model.field.set_mma_mode(mma)
        model.weight_grad_groups = groups
        opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=0.0, eps=1e-15)
        torch.manual_seed(5)
        train_step(model, rb, batch, opt, None, 100)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.set_sync_debug_mode("error")
        try:
            torch.manual_seed(77)
            train_step(model, rb, batch, opt, None, 110)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        peak[groups] = torch.cuda.max_memory_allocated()
        grads[groups] = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
        assert 0 < model._last_num_reflected < R
    assert sorted(grads[1]) == sorted(grads[2])
    for n, g0 in grads[1].items():
        assert float((grads[2][n] - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-12, n
    assert peak[2] < 0.9 * peak[1], peak


def test_reflect_capacity_auto_equals_full_size_buffers(dev):
    """Opt-in `model.reflect_capacity = "auto"` (train_graph.reflect_capacity): the two reflect levels -- a third of the step's
    memory -- are sized for 1.25 x the PREVIOUS step's reflected-ray count (copied to pinned memory asynchronously, looked at one
    step late: still no device-to-host read in the step) instead of for all R rays.  On a steady batch the step is the default
    step: same outputs, same gradients (to the order of the weight-gradient atomics), fewer bytes; a step whose count jumps past
    its capacity is detected one step late, counted, warned about, and the capacity falls back to R."""
    import warnings

    from reflect_sampling_nerf_amd.parallel import train_step

    R = 704
    grads, peak = {}, {}
    for mode in (None, "auto"):
        model, rb, batch = _train_setup(dev, R, (16, 16, 16, 16), layers=8, width=128, bias_shift=1.0)
        model.reflect_capacity = mode
        opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=0.0, eps=1e-15)  # lr 0: every step sees the same weights
        torch.manual_seed(5)
        for k in range(3):  # step 0 sizes for R; from step 2 on the capacity follows step k - 1's count
            train_step(model, rb, batch, opt, None, 100 + k)
            torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.set_sync_debug_mode("error")
        try:
            torch.manual_seed(77)
            train_step(model, rb, batch, opt, None, 110)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        peak[mode] = torch.cuda.max_memory_allocated()
        grads[mode] = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
        M = model._last_num_reflected
        assert 0 < M < 0.7 * R, M
        if mode == "auto":
            assert getattr(model, "reflect_overflows", 0) == 0
            assert abs(model._reflect_cap_state["m_seen"] - M) <= 0.1 * R  # the same rays every step (jitter moves a few)
            assert model._reflect_cap_state["m_seen"] * 1.25 + 192 < R  # so the levels really were sized below R
    assert sorted(grads[None]) == sorted(grads["auto"])
    for n, g0 in grads[None].items():
        assert float((grads["auto"][n] - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-12, n
    assert peak["auto"] < peak[None]
    # overflow: a model that reflected few rays is handed a batch that reflects many
    model, rb, batch = _train_setup(dev, R, (16, 16, 16, 16), layers=8, width=128, bias_shift=1.0)
    model.reflect_capacity = "auto"
    opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=0.0, eps=1e-15)
    train_step(model, rb, batch, opt, None, 100)
    torch.cuda.synchronize()
    model._reflect_cap_state["pending"].clear()
    model._reflect_cap_state["m_seen"] = 1  # as if the previous steps had reflected a single ray: capacity 192 < M
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        train_step(model, rb, batch, opt, None, 101)   # truncated step
        torch.cuda.synchronize()
        train_step(model, rb, batch, opt, None, 102)   # detects it, falls back
        torch.cuda.synchronize()
    M = model._last_num_reflected
    if M > 192:  # (capacity of m_seen = 1: 1.25 + 128 rounded up to 64 = 192)
        assert model.reflect_overflows == 1 and model._reflect_cap_state["disabled"]
        assert any("reflect_capacity" in str(x.message) for x in w)


def test_train_step_without_reflected_rays(dev):
    """M == 0 in TRAINING mode (the reference returns early, model.py:259-260): every reflect launch runs on a device-side
    count of zero, backward included.  Reflect colours are white * (1 - acc_fine) (model.py:240-241), every parameter
    still receives a finite gradient (the default reflect colour carries the live accumulation), nothing is NaN."""
    model, rb, batch = _train_setup(dev, 160, (16, 16, 8, 8), bias_shift=-12.0)
    out = model(rb)
    assert int(out["mask"].sum()) == 0 and model._last_num_reflected == 0
    assert "depth_reflect_fine" not in out
    white = 1.0 - out["accumulation_fine"]
    for k in ("mid_reflect_coarse", "mid_reflect_fine"):
        assert max_abs(out[k].detach().cpu(), white.expand(-1, 3).cpu()) <= 1e-6, k
    sum(model.get_loss_dict(out, batch).values()).backward()
    torch.cuda.synchronize()
    for name, p in model.field.named_parameters():
        if "field_output_low" in name:
            assert p.grad is None
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    # and against the oracle on shared draws: the same early-out (no reflected ray on either side), the same outputs
    P = {k: v.detach().cpu() for k, v in model.field.state_dict().items()}
    fs, ms = cpu_ref.FieldSpec(num_layers=8, width=128), cpu_ref.ModelSpec(16, 16, 8, 8)
    g = torch.Generator().manual_seed(9)
    jit = {"coarse": torch.rand(160, 17, generator=g), "fine": torch.rand(160, 17, generator=g)}
    model.zero_grad(set_to_none=True)
    out = model._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in jit.items()})
    ref = cpu_ref.get_outputs(P, fs, ms, rb.origins.cpu(), rb.directions.cpu(), rb.pixel_area.cpu(), rb.nears.cpu(),
                              rb.fars.cpu(), training=True, jitter=jit)  # (autograd normals: no no_grad around it)
    assert int(ref["mask"].sum()) == 0 and "depth_reflect_fine" not in ref
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_fine"):
        assert max_abs(out[k].detach().cpu(), ref[k].detach()) <= TOL, k


def test_loss_fused_into_compositing_equals_per_sample_loss(dev):
    """get_loss_dict on the training graph's own outputs takes the per-ray reductions the compositing kernel made
    (sum_s w |n - n_pred|^2, sum_s w max(0, n.d)^2: reference model.py:403-407) and forms d/d pred_normals, d/d n_dot_d
    inside the field's backward kernel; on a plain dict of the same tensors it takes the per-sample path
    (rsn_loss_forward_backward).  Same eight terms, same parameter gradients; and the per-ray sums equal torch's."""
    model, rb, batch = _train_setup(dev, 200, (24, 40, 16, 16), layers=8, width=128)
    g = torch.Generator().manual_seed(3)
    jit = {k: torch.rand(200, s + 1, generator=g).to(dev) for k, s in (("coarse", 24), ("fine", 40))}

    def run(fused):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)  # the reflect levels' draws
        out = model._get_outputs_train(rb, jitter=dict(jit))
        assert out.fused is not None
        for lvl in ("coarse", "fine"):
            w, n, pn = out[f"weights_{lvl}"][..., 0], out[f"normals_{lvl}"], out[f"pred_normals_{lvl}"].detach()
            ndd = out[f"n_dot_d_{lvl}"][..., 0].detach()
            assert max_abs((w * ((n - pn) ** 2).sum(-1)).sum(-1), out.fused[f"pn_loss_ray_{lvl}"].detach()) <= 1e-5
            assert max_abs((w * ndd.clamp(min=0) ** 2).sum(-1), out.fused[f"ori_loss_ray_{lvl}"].detach()) <= 1e-5
        losses = model.get_loss_dict(out if fused else dict(out), batch)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in losses.items()},
                {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None})

    l_f, g_f = run(True)
    l_s, g_s = run(False)
    l_f2, _ = run(True)
    # the eight reported values of the fused path are reduced in a fixed order (one workgroup, no float atomics): identical
    # draws -> bit-identical forward -> BIT-identical loss values run to run
    assert l_f2 == l_f
    assert sorted(l_f) == sorted(l_s)
    for k in l_s:
        assert abs(l_f[k] - l_s[k]) <= 2e-6 * max(abs(l_s[k]), 1e-3), k
    assert sorted(g_f) == sorted(g_s)
    for n in g_s:
        scale = float(g_s[n].abs().max()) + 1e-30
        assert float((g_f[n] - g_s[n]).abs().max()) <= 1e-5 * scale, n


def test_device_counted_step_equals_host_counted_step(dev, monkeypatch):
    """Production path (no draws injected: every reflect launch and the weight-gradient segments take M from device
    memory) against the test path (draws injected: M read on the host, reflect draws gathered per reflected ray), with
    torch.rand pinned to one constant so that both see the same draws: identical outputs, and parameter gradients equal
    up to the order of the fp32 atomics of the weight-gradient flush."""
    R, samples = 224, (24, 40, 16, 24)
    model, rb, batch = _train_setup(dev, R, samples, layers=8, width=64)
    const = 0.37

    def fake_rand(*shape, **kw):
        kw.pop("generator", None)
        return torch.full(tuple(shape), const, **kw)

    def run(inject):
        model.zero_grad(set_to_none=True)
        jit = None
        if inject:
            jit = {k: torch.full((R, s + 1), const, device=dev) for k, s in
                   zip(("coarse", "fine", "reflect_coarse", "reflect_fine"), samples)}
        out = model._get_outputs_train(rb, jitter=jit)
        sum(model.get_loss_dict(out, batch).values()).backward()
        torch.cuda.synchronize()
        return out, {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}

    monkeypatch.setattr(torch, "rand", fake_rand)
    out_d, g_d = run(False)
    out_h, g_h = run(True)
    monkeypatch.undo()
    M = int(out_h["mask"].sum())
    assert 0 < M < R
    assert sorted(out_d.materialise().keys()) == sorted(out_h.keys())
    for k in out_h:
        assert torch.equal(out_d[k], out_h[k]), k
    assert sorted(g_d) == sorted(g_h)
    for n in g_h:
        scale = float(g_h[n].abs().max()) + 1e-30
        assert float((g_d[n] - g_h[n]).abs().max()) <= 2e-6 * scale, n


def test_headline_workload_properties(dev):
    """The headline workload itself (bench.py default, BASELINE metric: 4096 rays x (128 coarse + 128 fine) + reflect
    64 + 64, 8 x 256, training step): size-independent properties of the full-size step."""
    from reflect_sampling_nerf_amd.parallel import train_step

    R = 4096
    model, rb, batch = _train_setup(dev, R, (128, 128, 64, 64), layers=8, width=256)
    params = model.get_param_groups()["fields"]
    opt = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)
    out = model(rb)
    M = int(out["mask"].sum())
    assert 0 < M < R and model._last_num_reflected == M
    for lvl in ("coarse", "fine"):
        w = out[f"weights_{lvl}"][..., 0]
        assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
        assert max_abs(w.sum(-1, keepdim=True), out[f"accumulation_{lvl}"]) <= 1e-5
        assert float((out[f"normals_{lvl}"].norm(dim=-1) - 1).abs().max()) <= 1e-4
        assert float((out[f"pred_normals_{lvl}"].norm(dim=-1) - 1).abs().max()) <= 1e-4
        rgb = out[f"mid_rgb_{lvl}"]
        assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    # rays that are not reflected keep white * (1 - acc_fine) (model.py:240-241); reflected ones are clipped colours
    keep = ~out["mask"]
    white = (1.0 - out["accumulation_fine"]).expand(-1, 3)
    for k in ("mid_reflect_coarse", "mid_reflect_fine"):
        assert max_abs(out[k][keep], white[keep]) <= 1e-6
        assert float(out[k].min()) >= 0.0 and float(out[k].max()) <= 1.0 + 1e-6
    assert out["depth_reflect_fine"].shape == (M, 1)
    loss = sum(model.get_loss_dict(out, batch).values())
    loss.backward()
    for name, p in model.field.named_parameters():
        if "field_output_low" in name:
            assert p.grad is None
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0.0, name
    # linearity of the backward pass in the upstream gradient: 2 x loss -> 2 x every gradient (weight-gradient atomics
    # reorder the sums: 2e-5 of the tensor maximum)
    g1 = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    torch.manual_seed(7)
    o1 = model(rb)
    torch.manual_seed(7)
    o2 = model(rb)
    for k in ("mid_rgb_fine", "mid_reflect_fine", "weights_fine"):
        assert torch.equal(o1[k], o2[k]), k  # same draws -> bit-identical forward
    (2.0 * sum(model.get_loss_dict(o2, batch).values())).backward()
    del o1
    torch.manual_seed(7)
    g2 = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    sum(model.get_loss_dict(model(rb), batch).values()).backward()
    for n, p in model.field.named_parameters():
        if p.grad is not None:
            scale = float(p.grad.abs().max()) + 1e-30
            assert float((g2[n] - 2.0 * p.grad).abs().max()) <= 4e-5 * scale, n
    # a few optimiser steps: finite, every trained tensor moves, the loss of the SAME batch goes down
    losses = [float(train_step(model, rb, batch, opt, None, 100 + k)) for k in range(4)]
    assert all(x == x and abs(x) < float("inf") for x in losses) and losses[-1] < losses[0]
    assert all(bool(torch.isfinite(p).all()) for p in params) and g1


# ---------------------------------------------------------------------------------------------- BASELINE configurations
def test_baseline_config0_level_matches_oracle(dev):
    """BASELINE configs[0] exactly: 1024 rays x 64 samples, 4-layer 128-wide trunk, random-init weights, one level of
    the Field.get_outputs plumbing (forward + composite) against the CPU oracle."""
    R, S = 1024, 64
    fld, P, fs = make_field(4, 128, dev, seed=0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=0)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    with torch.no_grad():
        ref = cpu_ref.render_level(P, fs, o, d, pa, nears, fars, S)
    sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears.reshape(R).to(dev), fars.reshape(R).to(dev),
                               None)
    lv = fld.evaluate_frustums(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb)
    c = ops.composite(R, None, S, 1, ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB, lv["sigma"], eb, lv["color"])
    assert max_abs(c["rgb"].cpu(), ref["rgb"]) <= TOL
    assert max_abs(c["accumulation"].cpu(), ref["accumulation"][..., 0]) <= TOL
    assert max_abs(c["weights"].cpu(), ref["weights"][..., 0]) <= 1e-5


def test_baseline_config2_training_step_properties(dev):
    """BASELINE configs[2] size (4096 rays x (64 coarse + 128 fine) + reflect 64 + 64, 8 x 256, forward + backward):
    size-independent properties of whole training steps instead of an oracle run."""
    from reflect_sampling_nerf_amd.parallel import train_step

    R = 4096
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=64, num_importance_samples=128)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    o, d, pa = cpu_ref.synthetic_rays(R, seed=0)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    batch = {"image": torch.rand(R, 3, generator=torch.Generator().manual_seed(1)).to(dev)}
    params = model.get_param_groups()["fields"]
    opt = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15)
    # one forward/backward by hand: every parameter except the never-evaluated field_output_low gets a finite gradient
    out = model(rb)
    assert int(out["mask"].sum()) > 0, "the reflect branch must be exercised"
    loss = sum(model.get_loss_dict(out, batch).values())
    loss.backward()
    for name, p in model.field.named_parameters():
        if "field_output_low" in name:
            assert p.grad is None
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0.0, name
    w = out["weights_fine"][..., 0]
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
    assert float((out["normals_fine"].norm(dim=-1) - 1).abs().max()) <= 1e-4
    # the same step twice from the same state: equal up to the order of the fp32 atomics in the weight-gradient flush
    g1 = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    torch.manual_seed(123)
    l1 = float(sum(model.get_loss_dict(model(rb), batch).values()).detach())
    torch.manual_seed(123)
    out2 = model(rb)
    loss2 = sum(model.get_loss_dict(out2, batch).values())
    # (the loss terms are sums of ~2,000 block partials added by fp32 atomics: their order moves the total by ~1e-6)
    assert abs(float(loss2.detach()) - l1) <= 1e-5 * abs(l1)
    # a few optimiser steps move every trained tensor and keep everything finite
    before = [p.detach().clone() for p in params]
    losses = [float(train_step(model, rb, batch, opt, None, 100 + k)) for k in range(3)]
    assert all(x == x and abs(x) < float("inf") for x in losses)
    moved = [not torch.equal(a, b.detach()) for a, b in zip(before, params)]
    assert sum(moved) >= len(params) - 2 and all(bool(torch.isfinite(p).all()) for p in params)
    assert g1  # gradients existed


def test_reduced_precision_training_bf16_sweeps(dev):
    """Opt-in reduced-precision training (the reference trains under autocast, config.py:33): with mma mode "bf16" the
    training forward and the backward sweeps run on plain bf16 MFMA operands with fp32 accumulation; the wide saved
    activations and layer gradients are STORED as bf16 and the weight-gradient kernel reads them as such (fp32
    accumulation).  Against the exact-fp32 HIP step on the same rays and jitter: rendered outputs
    within the bf16 tolerance, every parameter gradient in direction (cosine >= 0.99) and size (rel-L2 <= 8e-2)."""
    R, samples = 96, (16, 16, 8, 8)
    grads, outs = {}, {}
    g = torch.Generator().manual_seed(3)
    jit = {"coarse": torch.rand(R, 17, generator=g), "fine": torch.rand(R, 17, generator=g),
           "reflect_coarse": torch.rand(R, 9, generator=g), "reflect_fine": torch.rand(R, 9, generator=g)}
    image = torch.rand(R, 3, generator=g).to(dev)
    for mode in ("f32", "bf16"):
        torch.manual_seed(11)
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=16, num_importance_samples=16,
                                                num_reflect_coarse_samples=8, num_reflect_importance_samples=8)
        model = cfg.setup(scene_box=None, num_train_data=1)
        with torch.no_grad():
            model.field.field_output_density.net.bias += 2.0
        model.to(dev).train()
        model.field.set_mma_mode(mode)
        o, d, pa = cpu_ref.synthetic_rays(R, seed=21)
        rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev),
                           nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
        out = model._get_outputs_train(rb, jitter=jit)
        sum(model.get_loss_dict(out, {"image": image}).values()).backward()
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None}
        outs[mode] = {k: v.detach().clone() for k, v in out.items() if v.dtype.is_floating_point}
    for k in ("mid_rgb_coarse", "mid_rgb_fine", "accumulation_fine"):
        assert max_abs(outs["bf16"][k], outs["f32"][k]) <= 3e-2, k
    assert sorted(grads["bf16"]) == sorted(grads["f32"])
    for n, gr in grads["f32"].items():
        a, b = grads["bf16"][n].flatten().double(), gr.flatten().double()
        assert bool(torch.isfinite(a).all()), n
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        assert cos >= 0.99 and rel <= 8e-2, f"{n}: cos {cos:.5f} rel-L2 {rel:.3e}"


def test_chunked_train_step_equals_whole_batch_step(dev):
    """parallel.train_step(..., ray_chunk=n): gradient accumulation over ray chunks is the whole-batch step (losses, every
    parameter gradient, the updated parameters), with the live activation memory of one chunk."""
    from reflect_sampling_nerf_amd.parallel import train_step

    R = 320
    real_rand = torch.rand
    results = []
    try:
        # the samplers' stratified jitter: a constant, so that chunked and whole-batch runs see the same sample positions
        torch.rand = lambda *shape, **kw: torch.full(shape, 0.37, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
        for chunk in (None, 96):  # 96 does not divide 320: a ragged last chunk
            torch.manual_seed(0)
            cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=16, num_importance_samples=24,
                                                    num_reflect_coarse_samples=8, num_reflect_importance_samples=8,
                                                    base_mlp_num_layers=8, base_mlp_layer_width=64)
            model = cfg.setup(scene_box=None, num_train_data=1)
            with torch.no_grad():
                model.field.field_output_density.net.bias += 2.0
            model.to(dev).train()
            o, d, pa = cpu_ref.synthetic_rays(R, seed=3)
            rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev),
                               nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
            batch = {"image": real_rand(R, 3, generator=torch.Generator().manual_seed(1)).to(dev)}
            params = model.get_param_groups()["fields"]
            opt = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            loss = float(train_step(model, rb, batch, opt, None, 100, ray_chunk=chunk))
            torch.cuda.synchronize()
            results.append((loss, {n: p.grad.clone() for n, p in model.field.named_parameters() if p.grad is not None},
                            [p.detach().clone() for p in params], torch.cuda.max_memory_allocated() - base))
    finally:
        torch.rand = real_rand
    (l0, g0, p0, m0), (l1, g1, p1, m1) = results
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    assert sorted(g0) == sorted(g1)
    for n in g0:
        scale = float(g0[n].abs().max())
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * scale + 1e-10, n
    for a, b in zip(p0, p1):
        assert float((a - b).abs().max()) <= 1e-6
    assert m1 < 0.6 * m0, f"chunked peak {m1} vs whole {m0} bytes"  # 96-ray chunks of a 320-ray batch


def test_baseline_config3_bf16_properties(dev):
    """BASELINE configs[3] size (16384 rays x 192 samples, bf16 MFMA hidden GEMMs): properties + a slice against the
    exact-fp32 kernel within the bf16 tolerance."""
    R, S = 16384, 192
    fld, _, _ = make_field(8, 256, dev, seed=0, bias_shift=1.0)
    o, d, pa = cpu_ref.synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    fld.set_mma_mode("bf16")
    lv = fld.evaluate_frustums(o, d, pa, eb)
    c = ops.composite(R, None, S, 1, ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB, lv["sigma"], eb, lv["color"])
    lv2 = fld.evaluate_frustums(o, d, pa, eb)
    assert torch.equal(lv["color"], lv2["color"]) and torch.equal(lv["sigma"], lv2["sigma"])  # deterministic
    assert bool(torch.isfinite(lv["color"]).all()) and bool(torch.isfinite(lv["sigma"]).all())
    w = c["weights"]
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
    assert float(c["rgb"].min()) >= 0.0 and float(c["rgb"].max()) <= 1.0
    fld.set_mma_mode("f32")
    ref = fld.evaluate_frustums(o[:512], d[:512], pa[:512], eb[:512].contiguous())
    assert max_abs(lv["color"][:512], ref["color"]) <= 3e-2
    rel = (lv["sigma"][:512] - ref["sigma"]).abs() / (1.0 + ref["sigma"].abs())
    assert float(rel.max()) <= 3e-2


def test_reducer_on_rccl_single_rank(dev):
    """The reducer's production path -- backend "nccl" (RCCL), all-reduce on the side stream, events, device-resident
    flags -- on the one GPU a test box has: a one-rank group (world_size 1; two ranks cannot share a device under RCCL).
    The average over one rank is the identity: every gradient survives bit for bit, a parameter without gradient stays
    None, and steady state issues no device->host read.  The N > 1 arithmetic is covered on gloo (test_parallel_cpu.py,
    tools/ddp_equiv.py)."""
    import torch.distributed as dist
    from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        g = torch.Generator().manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(n, generator=g).to(dev)) for n in (7, 256 * 256, 3, 1000)]
        red = FlatGradAllReduce(params)
        red.run_single_rank = True
        for step in range(4):
            grads = [torch.randn(p.shape, generator=g).to(dev) for p in params]
            for p, gr in zip(params, grads):
                p.grad = gr.clone()
            params[2].grad = None if step == 0 else params[2].grad  # unused on "every rank" in the first step: dropped
            red()
            torch.cuda.synchronize()
            if step == 0:
                assert all(q is not params[2] for q in red.params) and params[2].grad is None
            if step == 1:
                # the dropped parameter has a gradient now: flagged inside the reduced buffer, NOT applied this step (another
                # rank might not have it: replicas stay identical), revived by every rank together one step later
                assert all(q is not params[2] for q in red.params) and params[2].grad is None
            for i, (p, gr) in enumerate(zip(params, grads)):
                if i == 2 and step <= 1:
                    continue
                assert torch.equal(p.grad, gr)
        # revived in step 2 (one more flag all-reduce: the second and last device->host read); steady state (step 3) issues none
        assert any(q is params[2] for q in red.params)
        assert red.host_syncs == 2
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("need_input", [False, True])
def test_plain_bf16_ring_sweeps_repeat_bitwise(dev, need_input):
    """The plain-bf16 ring training kernels (rsn_field_bf16_train.hip): forward + backward sweep of one level four times on the same
    inputs, every saved row and every gradient row bit for bit (no atomics in these kernels: any difference is a hazard -- see the
    split-bf16 test below for the one round 4 found)."""
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd import train_graph

    torch.manual_seed(0)
    R, S = 37, 32
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S, num_reflect_coarse_samples=16,
                                            num_reflect_importance_samples=16)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    f = model.field
    f.set_mma_mode("bf16")
    o, d, pa = cpu_ref.synthetic_rays(R, seed=1)
    o, d, pa = o.to(dev), d.to(dev), pa.to(dev).reshape(-1)
    bins = (2.0 + 4.0 * torch.linspace(0, 1, S + 1)).repeat(R, 1).to(dev).contiguous()
    gen = torch.Generator().manual_seed(3)
    gin = {"sigma": torch.randn(R, S, generator=gen).to(dev), "color": torch.randn(R, S, 3, generator=gen).to(dev),
           "pred_normals": torch.randn(R, S, 3, generator=gen).to(dev), "n_dot_d": torch.randn(R, S, generator=gen).to(dev),
           "roughness": torch.randn(R, S, generator=gen).to(dev)}
    first = None
    for _ in range(4):
        lv = f.evaluate_frustums_train(o, d, pa, bins, want_normals=not need_input)
        go = train_graph._field_backward(f, (o, d, pa), bins, lv, gin, need_input)
        torch.cuda.synchronize()
        cur = {**{"saved." + k: v for k, v in lv["saved"].items()}, **{"gout." + k: v for k, v in go.items()},
               **{k: lv[k] for k in ("sigma", "color", "pred_normals", "diff", "tint", "roughness")}}
        if first is None:
            first = {k: v.clone() for k, v in cur.items()}
        else:
            for k, v in cur.items():
                assert torch.equal(v.view(torch.uint8), first[k].view(torch.uint8)), f"not repeatable: {k}"


@pytest.mark.gpu
@pytest.mark.parametrize("need_input", [False, True])
def test_split_bf16_ring_rows_match_the_exact_kernels_and_repeat_bitwise(dev, need_input):
    """The split-bf16 (bf16x6) training kernels on the LDS weight ring (rsn_field_x6_train.hip) against the exact-fp32 kernels, buffer
    by buffer on the same inputs: every saved forward row and every layer-gradient row of the backward sweep (the operands of the
    weight gradients; both modes keep them fp32 in natural feature order) within 2e-6 of the tensor's largest value, and the backward
    sweep repeated four times on the same inputs bit for bit.  (Round 4: a wide buffer store with a register soffset let the next VALU
    overwrite its data -- intermittently, rows 12..15 of a 16-row tile, one element per K-step: the repetition and the row-level
    comparison are what showed it)"""
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd import train_graph

    torch.manual_seed(0)
    R, S = 37, 32   # 1184 points: nine full 128-point tiles and one of 32 (two live waves)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S, num_reflect_coarse_samples=16,
                                            num_reflect_importance_samples=16)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    f = model.field
    o, d, pa = cpu_ref.synthetic_rays(R, seed=1)
    o, d, pa = o.to(dev), d.to(dev), pa.to(dev).reshape(-1)
    bins = (2.0 + 4.0 * torch.linspace(0, 1, S + 1)).repeat(R, 1).to(dev).contiguous()
    gen = torch.Generator().manual_seed(3)
    gin = {"sigma": torch.randn(R, S, generator=gen).to(dev), "color": torch.randn(R, S, 3, generator=gen).to(dev),
           "pred_normals": torch.randn(R, S, 3, generator=gen).to(dev), "n_dot_d": torch.randn(R, S, generator=gen).to(dev),
           "roughness": torch.randn(R, S, generator=gen).to(dev)}
    res = {}
    for mma in ("f32", "bf16x6"):
        f.set_mma_mode(mma)
        lv = f.evaluate_frustums_train(o, d, pa, bins, want_normals=not need_input)
        runs = [train_graph._field_backward(f, (o, d, pa), bins, lv, gin, need_input) for _ in range(4 if mma == "bf16x6" else 1)]
        torch.cuda.synchronize()
        res[mma] = (lv, runs)
    (la, (ga,)), (lb, gbs) = res["f32"], res["bf16x6"]

    def close(name, a, b):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 2e-6 * max(float(a.abs().max()), 1e-3), name

    for k in ("sigma", "color", "pred_normals", "diff", "tint", "roughness", "raw_density") + (() if need_input else ("normals",)):
        if k == "normals":  # unit vectors through a division by a small gradient norm: the suite's unit-vector bound
            assert max_abs(la[k].cpu(), lb[k].cpu()) <= TOL_UNIT, k
        else:
            close(k, la[k], lb[k])
    for k in ("bott", "hid", "heads"):
        close("saved." + k, la["saved"][k], lb["saved"][k])
    for l in range(8):
        close(f"saved.act[{l}]", la["saved"]["act"][l], lb["saved"]["act"][l])
    gb = gbs[0]
    for k in ("dz_rgb", "da_mid", "d_bott", "dz_heads") + (("d_input",) if need_input else ()):
        close("gout." + k, ga[k], gb[k])
    for l in range(8):
        close(f"gout.dy[{l}]", ga["dy"][l], gb["dy"][l])
    for other in gbs[1:]:
        for k, v in gb.items():
            assert torch.equal(v, other[k]), f"backward sweep not repeatable: {k}"
