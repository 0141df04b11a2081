#!/usr/bin/env python3
"""fp64 arbiter on TRAINED weights (GPU box): when the HIP render and the fp32 CPU oracle differ, which one is off?

Round 3 recorded a largest pixel difference of 1.3e-4 between the HIP render and the oracle render of a trained model (above
north_star's 1e-4) with nothing to say which side carries the error.  This tool renders held-out rays of the procedural scene
with the trained parameters of a reference-generated fixture (tests/golden/params_trained_*.npz) three ways

    hip     the HIP path, fp32 (eval-mode get_outputs)
    o32     oracle/cpu_ref.py in fp32 (the parity oracle: the reference's op sequence in eager PyTorch)
    o64     the SAME oracle code evaluated in fp64 (torch default dtype float64, parameters and rays cast) -- the arbiter

and reports (a) end to end: max / mean |hip - o64|, |o32 - o64|, |hip - o32| of every rendered output, mask flips, PSNR;
(b) PER STAGE ON IDENTICAL INPUTS -- the local error of each stage of the primary path, every pipeline fed the HIP path's own
stage inputs (sample positions, densities, weights): the coarse field, the coarse compositing, the PDF resampling, the fine
field, the fine compositing; (c) the same stage table at the worst pixel (largest |hip - o32| of mid_rgb_fine).

Usage (GPU box):  python tools/fp64_arbiter.py --fixture trainstep_trained_l8_w256 --rays 4096 --json profiles/r04_fp64_arbiter.json
"""
import argparse
import contextlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.train_parity import psnr, scene_rays


@contextlib.contextmanager
def default_dtype(dt):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def err(a, ref):
    d = (a.double() - ref.double()).abs()
    return {"max": float(d.max()), "mean": float(d.mean())}


def run(fixture="trainstep_trained_l8_w256", rays=4096, json_path="", verbose=True):
    import reflect_sampling_nerf_amd as pkg
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd import ops
    from tests.helpers import load_golden

    meta, g = load_golden(fixture)
    S, layers, width = meta["samples"], meta["layers"], meta["width"]
    dev = torch.device("cuda:0")
    fs, ms = cpu_ref.FieldSpec(num_layers=layers, width=width), cpu_ref.ModelSpec(*S)
    P32 = {k: v.clone() for k, v in g["param"].items()}
    P64 = {k: v.double() for k, v in P32.items()}
    o, d, pa, gt = scene_rays(rays, torch.Generator().manual_seed(4242))
    nears, fars = torch.full((rays, 1), 2.0), torch.full((rays, 1), 6.0)

    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S[0], num_importance_samples=S[1],
                                            num_reflect_coarse_samples=S[2], num_reflect_importance_samples=S[3],
                                            base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.field.load_state_dict(P32)
    model.to(dev).eval()
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev), fars=fars.to(dev))
    with torch.no_grad():
        hip = {k: v.cpu() for k, v in model(rb).items()}
        o32 = cpu_ref.get_outputs(P32, fs, ms, o, d, pa, nears, fars, training=False)
        with default_dtype(torch.float64):
            o64 = cpu_ref.get_outputs(P64, fs, ms, o.double(), d.double(), pa.double(), nears.double(), fars.double(),
                                      training=False)
    keys = ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
            "accumulation_fine", "weights_fine", "diff", "tint", "roughness")
    # reflect outputs: compare on rays whose mask agrees in all three (a flipped mask swaps the whole branch)
    agree = (hip["mask"] == o32["mask"]) & (o32["mask"] == o64["mask"])
    end = {}
    for k in keys:
        sel = agree if "reflect" in k else torch.ones_like(agree)
        end[k] = {"hip_vs_fp64": err(hip[k][sel], o64[k][sel]), "oracle32_vs_fp64": err(o32[k][sel], o64[k][sel]),
                  "hip_vs_oracle32": err(hip[k][sel], o32[k][sel])}
    res = {"fixture": fixture, "field": f"{layers}x{width}", "samples": S, "rays": rays,
           "trained_steps": meta.get("trained_steps"), "psnr_vs_ground_truth_db": {
               "hip": psnr(hip["mid_rgb_fine"], gt), "oracle32": psnr(o32["mid_rgb_fine"], gt),
               "fp64": psnr(o64["mid_rgb_fine"].float(), gt)},
           "psnr_between_renders_db": {"hip_vs_fp64": psnr(hip["mid_rgb_fine"].double(), o64["mid_rgb_fine"]),
                                       "oracle32_vs_fp64": psnr(o32["mid_rgb_fine"].double(), o64["mid_rgb_fine"]),
                                       "hip_vs_oracle32": psnr(hip["mid_rgb_fine"], o32["mid_rgb_fine"])},
           "mask_flips": {"hip_vs_fp64": int((hip["mask"] != o64["mask"]).sum()),
                          "oracle32_vs_fp64": int((o32["mask"] != o64["mask"]).sum()),
                          "hip_vs_oracle32": int((hip["mask"] != o32["mask"]).sum())},
           "reflected_rays": int(o64["mask"].sum()), "end_to_end": end}

    # ---- per stage, on the HIP path's own stage inputs
    fld = model.field
    od, dd, pad = o.to(dev), d.to(dev), pa.reshape(rays).to(dev)
    nd, fd = nears.reshape(rays).to(dev), fars.reshape(rays).to(dev)
    uni = model.sampler_uniform.spec
    EVAL, CLIP = ops.RSN_COMP_EVAL, ops.RSN_COMP_CLIP_RGB
    with torch.no_grad():
        sb_c, eb_c = ops.sample_spaced(rays, None, S[0], uni.spacing, uni.tan, nd, fd, None)
        lc = fld.evaluate_frustums(od, dd, pad, eb_c)
        cc = ops.composite(rays, None, S[0], 1, EVAL | CLIP, lc["sigma"], eb_c, lc["color"])
        sb_f, eb_f = ops.sample_pdf(rays, None, S[0], S[1], uni.spacing, uni.tan, model.sampler_pdf.histogram_padding, nd, fd,
                                    cc["weights"], sb_c, None)
        lf = fld.evaluate_frustums(od, dd, pad, eb_f)
        cf = ops.composite(rays, None, S[1], 1, EVAL | CLIP, lf["sigma"], eb_f, lf["color"])
    torch.cuda.synchronize()
    c = lambda t: t.detach().cpu()  # noqa: E731
    white = torch.ones(3)

    def field_stage(eb, lv):
        eb = c(eb)
        with torch.no_grad():
            f32 = cpu_ref.field_level(P32, fs, o, d, pa, eb, False, False)
            with default_dtype(torch.float64):
                f64 = cpu_ref.field_level(P64, fs, o.double(), d.double(), pa.double(), eb.double(), False, False)
        out = {}
        for name, hv, k in (("sigma", c(lv["sigma"]).unsqueeze(-1), "sigma"), ("color", c(lv["color"]), "color")):
            out[name] = {"hip_vs_fp64": (hv.double() - f64[k]).abs(), "oracle32_vs_fp64": (f32[k].double() - f64[k]).abs()}
        # densities span orders of magnitude on a trained model: the relative error is the meaningful one
        out["sigma_rel"] = {k: v / (1.0 + f64["sigma"].abs()) for k, v in out["sigma"].items()}
        return out

    def comp_stage(eb, lv, cp):
        eb, sig, col = c(eb), c(lv["sigma"]).unsqueeze(-1), c(lv["color"])
        t0, t1 = eb[:, :-1], eb[:, 1:]
        with torch.no_grad():
            w32 = cpu_ref.weights_from_density(sig, t0, t1)
            rgb32 = torch.clip(cpu_ref.composite_rgb(col, w32, white, False), 0.0, 1.0)
            with default_dtype(torch.float64):
                w64 = cpu_ref.weights_from_density(sig.double(), t0.double(), t1.double())
                rgb64 = torch.clip(cpu_ref.composite_rgb(col.double(), w64, white.double(), False), 0.0, 1.0)
        return {"weights": {"hip_vs_fp64": (c(cp["weights"]).unsqueeze(-1).double() - w64).abs(),
                            "oracle32_vs_fp64": (w32.double() - w64).abs()},
                "rgb": {"hip_vs_fp64": (c(cp["rgb"]).double() - rgb64).abs(), "oracle32_vs_fp64": (rgb32.double() - rgb64).abs()}}

    def pdf_stage():
        w, sb = c(cc["weights"]).unsqueeze(-1), c(sb_c)
        with torch.no_grad():
            s32, e32 = cpu_ref.pdf_bins("uniform", 1.0, nears, fars, w, sb, S[1], None, ms.histogram_padding)
            with default_dtype(torch.float64):
                s64, e64 = cpu_ref.pdf_bins("uniform", 1.0, nears.double(), fars.double(), w.double(), sb.double(), S[1], None,
                                            ms.histogram_padding)
        return {"euclid_bins": {"hip_vs_fp64": (c(eb_f).double() - e64).abs(), "oracle32_vs_fp64": (e32.double() - e64).abs()}}

    stages = {"1_coarse_field": field_stage(eb_c, lc), "2_coarse_composite": comp_stage(eb_c, lc, cc), "3_pdf_resampling": pdf_stage(),
              "4_fine_field": field_stage(eb_f, lf), "5_fine_composite": comp_stage(eb_f, lf, cf)}
    worst = int((hip["mid_rgb_fine"] - o32["mid_rgb_fine"]).abs().max(dim=-1).values.argmax())
    res["worst_pixel"] = {"ray": worst, "hip": hip["mid_rgb_fine"][worst].tolist(), "oracle32": o32["mid_rgb_fine"][worst].tolist(),
                          "fp64": o64["mid_rgb_fine"][worst].tolist(),
                          "abs_err_hip_vs_fp64": float((hip["mid_rgb_fine"][worst].double() - o64["mid_rgb_fine"][worst]).abs().max()),
                          "abs_err_oracle32_vs_fp64": float((o32["mid_rgb_fine"][worst].double() - o64["mid_rgb_fine"][worst]).abs().max()),
                          "abs_err_hip_vs_oracle32": float((hip["mid_rgb_fine"][worst] - o32["mid_rgb_fine"][worst]).abs().max())}
    table, table_worst = {}, {}
    for sname, st in stages.items():
        table[sname] = {q: {who: {"max": float(v.max()), "mean": float(v.mean())} for who, v in pair.items()} for q, pair in st.items()}
        table_worst[sname] = {q: {who: float(v[worst].max()) for who, v in pair.items()} for q, pair in st.items()}
    res["per_stage_local_error_on_identical_inputs"] = table
    res["per_stage_at_worst_pixel"] = table_worst
    e = end["mid_rgb_fine"]
    res["verdict"] = ("mid_rgb_fine, %d rays on trained weights: max |hip - fp64| %.2e, max |oracle32 - fp64| %.2e, max |hip - "
                      "oracle32| %.2e -> the %s carries the larger end-to-end error" %
                      (rays, e["hip_vs_fp64"]["max"], e["oracle32_vs_fp64"]["max"], e["hip_vs_oracle32"]["max"],
                       "fp32 ORACLE" if e["oracle32_vs_fp64"]["max"] >= e["hip_vs_fp64"]["max"] else "HIP path"))
    if verbose:
        print(json.dumps(res, indent=1))
    if json_path:
        os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
        with open(json_path, "w") as f:
            json.dump(res, f, indent=1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="trainstep_trained_l8_w256")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("RSN_CPU_THREADS", "16")))
    run(a.fixture, a.rays, a.json)


if __name__ == "__main__":
    main()
