"""The CPU oracle (oracle/cpu_ref.py) against golden vectors captured from the REFERENCE's own
modules (oracle/make_golden.py).  CPU only.  Tolerances: the oracle is a re-ordering-free
restatement in the same torch ops, so fp32 agreement is expected at ~1e-6; discontinuous
outputs (mask, median depth index) must match exactly on these fixtures."""
import pytest
import torch

from oracle import cpu_ref
from tests.helpers import field_spec_from_meta, load_golden, max_abs, model_spec_from_meta

CASES = ["eval_l8_w32", "train_l8_w32", "eval_l4_w32", "eval_l6_w32_nomask", "eval_l8_w64_near0",
         # shapes the HIP kernels run (tests/test_gpu_parity.py compares the HIP path with the same files directly)
         "eval_l8_w256", "eval_l4_w128", "eval_l6_w64_nomask",
         # TRAINED weights: the reference trained by oracle/make_golden_trained.py (its own get_outputs / get_loss_dict /
         # RAdam on the procedural scene) until it left the initialisation regime, then recorded
         "eval_trained_l8_w64", "eval_trained_l8_w256",
         # a width that is not one of the kernels' 64 / 128 / 256 (the HIP path runs it zero-padded)
         "eval_l8_w200"]
TRAIN_STEP_CASES = ["trainstep_l8_w64", "trainstep_l8_w256", "trainstep_l4_w128",
                    "trainstep_trained_l8_w64", "trainstep_trained_l8_w256", "trainstep_l6_w48", "trainstep_l8_w200"]


@pytest.mark.parametrize("name", CASES)
def test_get_outputs_matches_reference(name):
    meta, g = load_golden(name)
    fs, ms = field_spec_from_meta(meta), model_spec_from_meta(meta)
    i = g["in"]
    torch.manual_seed(0)
    rec = {}
    out = cpu_ref.get_outputs(g["param"], fs, ms, i["origins"], i["directions"], i["pixel_area"], i["nears"],
                              i["fars"], training=meta["training"], jitter=g.get("jitter"), record_bins=rec)
    ref = g["out"]
    # the sampler outputs the reference logged (forward hooks on its four samplers): same bin edges
    assert sorted(rec) == sorted(g["bins"])
    for k, v in g["bins"].items():
        if k.endswith("spacing") or "reflect" not in k:
            assert max_abs(rec[k], v) <= 2e-6, k
        else:  # euclidean bins of the reciprocal spacing reach 256: relative bound
            assert float(((rec[k] - v).abs() / (1.0 + v.abs())).max()) <= 2e-5, k
    assert sorted(out.keys()) == sorted(ref.keys()) == meta["keys"]
    assert torch.equal(out["mask"].to(torch.uint8), ref["mask"])
    assert int(out["mask"].sum()) == meta["M"]
    for k, v in ref.items():
        if k == "mask":
            continue
        assert tuple(out[k].shape) == tuple(v.shape), k
        if k.startswith("depth"):  # median depths: up to 256 along a reflected ray (reciprocal spacing): relative bound
            err = float(((out[k].detach() - v).abs() / (1.0 + v.abs())).max())
            assert err <= 1e-5, f"{name}:{k}: max rel err {err}"
            continue
        err = max_abs(out[k].detach(), v)
        assert err <= 2e-6, f"{name}:{k}: max abs err {err}"


def test_nomask_case_takes_early_out():
    meta, g = load_golden("eval_l6_w32_nomask")
    assert meta["M"] == 0 and "depth_reflect_fine" not in g["out"]
    meta, g = load_golden("eval_l6_w64_nomask")
    assert meta["M"] == 0 and "depth_reflect_fine" not in g["out"]


def test_units_contract():
    _, g = load_golden("units")
    c = g["contract"]
    m, cov = cpu_ref.contract(c["mean"], c["cov"])
    assert max_abs(m, c["out_mean"]) <= 1e-6
    assert max_abs(cov, c["out_cov"]) <= 1e-6
    assert float(m.norm(dim=-1).max()) < 2.0  # contraction maps into the radius-2 ball
    assert float(torch.diagonal(cov, dim1=-2, dim2=-1).min()) >= 0.0


def test_units_sh34():
    _, g = load_golden("units")
    s = g["sh"]
    out = cpu_ref.integrated_sh(s["dirs"], s["roughness"])
    assert out.shape[-1] == 34
    assert max_abs(out, s["out"]) <= 2e-6


def test_units_reciprocal_sampler():
    _, g = load_golden("units")
    r = g["recip"]
    R = r["eval_starts"].shape[0]
    nears, fars = torch.zeros(R, 1), torch.full((R, 1), 256.0)
    _, eb = cpu_ref.spaced_bins("reciprocal", 0.25, nears, fars, 16, None)
    assert max_abs(eb[:, :-1], r["eval_starts"]) <= 1e-5 and max_abs(eb[:, 1:], r["eval_ends"]) <= 2e-5
    sb, eb = cpu_ref.spaced_bins("reciprocal", 0.25, nears, fars, 16, r["train_rand"])
    assert max_abs(sb[:, :-1], r["train_spacing_starts"]) <= 1e-6
    assert max_abs(eb[:, :-1], r["train_starts"]) <= 1e-5 and max_abs(eb[:, 1:], r["train_ends"]) <= 2e-5


def test_units_inf_color_and_heads():
    meta, g = load_golden("units")
    fs = cpu_ref.FieldSpec(num_layers=meta["layers"], width=meta["width"])
    P = g["param"]
    out = cpu_ref.inf_color(P, fs, g["sh"]["dirs"], g["inf"]["sqradius"])
    assert max_abs(out, g["inf"]["out"]) <= 2e-6
    h = g["heads"]
    pn = cpu_ref.pred_normals(P, h["emb"])
    assert max_abs(pn, h["pred_normals"]) <= 1e-6
    ndd = torch.sum(g["sh"]["dirs"] * pn, dim=-1, keepdim=True)
    assert max_abs(ndd, h["n_dot_d"]) <= 1e-6
    rs = torch.sigmoid(cpu_ref.head(P, "field_output_roughness", h["emb"]))
    assert max_abs(rs, h["roughness_sigmoid"]) <= 1e-6


def test_param_count_matches_reference_default():
    fs = cpu_ref.FieldSpec()
    P = cpu_ref.init_params(fs)
    assert sum(v.numel() for v in P.values()) == 618513


@pytest.mark.parametrize("name", TRAIN_STEP_CASES)
@pytest.mark.parametrize("inject_bins", [False, True])
def test_train_step_matches_reference(name, inject_bins):
    """One whole training step of the REFERENCE (get_outputs in train mode -> its own get_loss_dict -> backward,
    tests/golden/trainstep_*.npz): the oracle's outputs, its restated loss terms and autograd's gradients through the
    oracle must reproduce the reference's -- free-running from the logged jitter, and on the reference's logged bins."""
    meta, g = load_golden(name)
    fs, ms = field_spec_from_meta(meta), model_spec_from_meta(meta)
    i = g["in"]
    P = {k: v.clone().requires_grad_(True) for k, v in g["param"].items()}
    out = cpu_ref.get_outputs(P, fs, ms, i["origins"], i["directions"], i["pixel_area"], i["nears"], i["fars"],
                              training=True, jitter=g["jitter"], bins=g["bins"] if inject_bins else None)
    assert torch.equal(out["mask"].to(torch.uint8), g["out"]["mask"]) and int(out["mask"].sum()) == meta["M"] > 0
    for k, v in g["out"].items():
        if k != "mask":
            assert max_abs(out[k].detach(), v) <= (2e-6 if not k.startswith("depth") else 1e-5), k
    assert meta["loss_coefficients"] == pytest.approx(cpu_ref.LOSS_COEFFICIENTS)
    losses = cpu_ref.loss_dict(out, i["image"], meta["loss_coefficients"])
    assert sorted(losses) == sorted(g["loss"])
    for k, v in g["loss"].items():
        assert abs(float(losses[k].detach()) - float(v)) <= 2e-6 * max(1.0, abs(float(v))), k
    sum(losses.values()).backward()
    assert sorted(k for k, p in P.items() if p.grad is not None) == sorted(g["grad"])
    for k, gr in g["grad"].items():
        scale = float(gr.abs().max())
        assert max_abs(P[k].grad, gr) <= 2e-5 * scale + 1e-10, k
