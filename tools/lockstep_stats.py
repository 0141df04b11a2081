#!/usr/bin/env python3
"""Lockstep-training statistics on a TRAINED model (GPU box): is the PSNR of the HIP pipeline after n lockstep steps
distinguishable from the CPU oracle's, given how chaotic the trained regime is?

Round 3 had ONE lockstep window (tools/train_parity.py) and the sign of its PSNR delta was negative twice.  One window
says nothing: an oracle whose parameters were perturbed by 1e-6 (one fp32 rounding) drifts from the oracle by 0.8 dB within
40 steps (tools/chaos_probe.py).  This tool takes K windows with different batch seeds from the SAME trained state
(tests/golden/params_trained_*.npz: the REFERENCE trained by oracle/make_golden_trained.py) and, per window, trains three
pipelines on identical batches, jitter and learning rate:

    hip     the HIP path (FusedRAdam)
    oracle  oracle/cpu_ref.py + autograd + torch.optim.RAdam
    pert    the same oracle started from parameters multiplied by (1 + 1e-6 * N(0, 1))   -- the noise floor

then renders held-out rays with each and reports PSNR(hip) - PSNR(oracle) and PSNR(pert) - PSNR(oracle) against the
scene's analytic ground truth: mean, standard deviation and standard error over the K windows.  north_star: "PSNR within
0.1 dB of reference" -- met if |mean delta| <= 0.1 dB, or if the delta is inside the noise floor's spread.

Usage (GPU box):  python tools/lockstep_stats.py --fixture trainstep_trained_l8_w64 --windows 6 --steps 40 --json out.json
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.train_parity import loss_terms, psnr, scene_rays


def run(fixture="trainstep_trained_l8_w64", windows=6, steps=40, rays=256, eval_rays=1024, perturb=1e-6, lr_step=3000,
        json_path="", verbose=True, modes=("f32",)):
    import reflect_sampling_nerf_amd as pkg
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd.train_ops import exponential_decay_lr
    from tests.helpers import load_golden

    meta, g = load_golden(fixture)
    S, layers, width = meta["samples"], meta["layers"], meta["width"]
    dev = torch.device("cuda:0")
    fs, ms = cpu_ref.FieldSpec(num_layers=layers, width=width), cpu_ref.ModelSpec(*S)
    P0 = {k: v.clone() for k, v in g["param"].items()}
    eo, ed, epa, ergb = scene_rays(eval_rays, torch.Generator().manual_seed(99))
    near = lambda n: torch.full((n, 1), 2.0)  # noqa: E731
    far = lambda n: torch.full((n, 1), 6.0)  # noqa: E731
    coeff = dict(cpu_ref.LOSS_COEFFICIENTS)

    def oracle_psnr(P):
        with torch.no_grad():
            oc = cpu_ref.get_outputs({k: v.detach() for k, v in P.items()}, fs, ms, eo, ed, epa, near(eval_rays),
                                     far(eval_rays), training=False)
        return psnr(oc["mid_rgb_fine"], ergb), psnr(oc["mid_reflect_fine"], ergb)

    results = []
    t_start = time.time()
    for w in range(windows):
        gen = torch.Generator().manual_seed(1000 + 17 * w)
        # --- the three pipelines start from the same trained parameters (pert: one fp32 rounding away)
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S[0], num_importance_samples=S[1],
                                                num_reflect_coarse_samples=S[2], num_reflect_importance_samples=S[3],
                                                base_mlp_num_layers=layers, base_mlp_layer_width=width)
        models = {}
        for md in modes:  # one HIP pipeline per MMA mode (f32 exact | bf16x6 fp32-equivalent | bf16 reduced precision), same batches
            mm = cfg.setup(scene_box=None, num_train_data=1)
            mm.field.load_state_dict(P0)
            mm.to(dev).train()
            mm.field.set_mma_mode(md)
            models[md] = mm
        model = models[modes[0]]
        names = [n for n, _ in model.field.named_parameters()]
        Po = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        pg = torch.Generator().manual_seed(5000 + w)
        Pp = {k: (v * (1.0 + perturb * torch.randn(v.shape, generator=pg))).clone().requires_grad_(True) for k, v in P0.items()}
        opts_h = {md: pkg.FusedRAdam(mm.get_param_groups()["fields"], lr=1e-3, eps=1e-15) for md, mm in models.items()}
        opt_h = opts_h[modes[0]]
        opt_o = torch.optim.RAdam([Po[n] for n in names], lr=1e-3, eps=1e-15)
        opt_p = torch.optim.RAdam([Pp[n] for n in names], lr=1e-3, eps=1e-15)
        erb = pkg.RayBundle(origins=eo.to(dev), directions=ed.to(dev), pixel_area=epa.to(dev), nears=near(eval_rays).to(dev),
                            fars=far(eval_rays).to(dev))
        flips = 0
        rel = []
        for step in range(steps):
            o, d, pa, rgb = scene_rays(rays, gen)
            jit = {"coarse": torch.rand(rays, S[0] + 1, generator=gen), "fine": torch.rand(rays, S[1] + 1, generator=gen),
                   "reflect_coarse": torch.rand(rays, S[2] + 1, generator=gen),
                   "reflect_fine": torch.rand(rays, S[3] + 1, generator=gen)}
            lr = exponential_decay_lr(lr_step + step, 1e-3, 1e-4, 50000)
            losses = []
            for P, opt in ((Po, opt_o), (Pp, opt_p)):
                for grp in opt.param_groups:
                    grp["lr"] = lr
                opt.zero_grad(set_to_none=True)
                ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, near(rays), far(rays), training=True, jitter=jit)
                lc = sum(v * coeff[k] for k, v in loss_terms(ref, rgb).items())
                lc.backward()
                opt.step()
                losses.append(float(lc))
                if P is Po:
                    mask_o = ref["mask"]
            opt_h.lr = lr
            opt_h.zero_grad(set_to_none=True)
            rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=near(rays).to(dev),
                               fars=far(rays).to(dev))
            out = model._get_outputs_train(rb, jitter=jit)
            lg = sum(model.get_loss_dict(out, {"image": rgb.to(dev)}).values())
            lg.backward()
            opt_h.step()
            for md in modes[1:]:
                opts_h[md].lr = lr
                opts_h[md].zero_grad(set_to_none=True)
                om = models[md]._get_outputs_train(rb, jitter=jit)
                sum(models[md].get_loss_dict(om, {"image": rgb.to(dev)}).values()).backward()
                opts_h[md].step()
            flips += int((out["mask"].cpu() != mask_o).sum())
            rel.append((abs(float(lg) - losses[0]) / abs(losses[0]), abs(losses[1] - losses[0]) / abs(losses[0])))
        model.eval()
        with torch.no_grad():
            og = model(erb)
        ph = (psnr(og["mid_rgb_fine"].cpu(), ergb), psnr(og["mid_reflect_fine"].cpu(), ergb))
        po, pp = oracle_psnr(Po), oracle_psnr(Pp)
        extra = {}
        for md in modes[1:]:
            models[md].eval()
            with torch.no_grad():
                ogm = models[md](erb)
            extra["delta_%s_db" % md] = psnr(ogm["mid_rgb_fine"].cpu(), ergb) - po[0]
            extra["delta_%s_reflect_db" % md] = psnr(ogm["mid_reflect_fine"].cpu(), ergb) - po[1]
        rec = {**extra, "window": w, "psnr_hip": ph[0], "psnr_oracle": po[0], "psnr_pert": pp[0],
               "delta_hip_db": ph[0] - po[0], "delta_pert_db": pp[0] - po[0],
               "delta_hip_reflect_db": ph[1] - po[1], "delta_pert_reflect_db": pp[1] - po[1],
               "mask_flips_hip_vs_oracle": flips, "rel_loss_diff_first_step": rel[0], "rel_loss_diff_last_step": rel[-1],
               "max_rel_loss_diff_hip": max(r[0] for r in rel), "max_rel_loss_diff_pert": max(r[1] for r in rel)}
        results.append(rec)
        if verbose:
            print(json.dumps(rec), flush=True)
        if json_path:
            os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
            with open(json_path, "w") as f:
                json.dump({"partial": True, "windows": results}, f)

    def stats(key):
        xs = [r[key] for r in results]
        m = sum(xs) / len(xs)
        sd = math.sqrt(sum((x - m) ** 2 for x in xs) / max(len(xs) - 1, 1))
        return {"mean": m, "std": sd, "stderr": sd / math.sqrt(len(xs)), "min": min(xs), "max": max(xs)}

    with torch.no_grad():
        start = cpu_ref.get_outputs(P0, fs, ms, eo, ed, epa, near(eval_rays), far(eval_rays), training=False)
    summary = {"fixture": fixture, "field": f"{layers}x{width}", "samples": S, "rays_per_step": rays, "steps_per_window": steps,
               "windows": windows, "perturbation": perturb, "eval_rays": eval_rays,
               "psnr_of_the_trained_start_db": psnr(start["mid_rgb_fine"], ergb),
               "delta_hip_db": stats("delta_hip_db"), "delta_pert_db": stats("delta_pert_db"),
               "delta_hip_reflect_db": stats("delta_hip_reflect_db"), "delta_pert_reflect_db": stats("delta_pert_reflect_db"),
               "hip_mode": modes[0], **{"delta_%s_db" % md: stats("delta_%s_db" % md) for md in modes[1:]},
               **{"delta_%s_reflect_db" % md: stats("delta_%s_reflect_db" % md) for md in modes[1:]},
               "seconds": time.time() - t_start}
    summary["verdict"] = (
        "PSNR(hip) - PSNR(oracle) = %+.3f +- %.3f dB (standard error, %d windows); noise floor PSNR(oracle perturbed by "
        "%.0e) - PSNR(oracle) = %+.3f +- %.3f dB (std of one window %.3f dB vs %.3f dB)" %
        (summary["delta_hip_db"]["mean"], summary["delta_hip_db"]["stderr"], windows, perturb,
         summary["delta_pert_db"]["mean"], summary["delta_pert_db"]["stderr"], summary["delta_hip_db"]["std"],
         summary["delta_pert_db"]["std"]))
    if verbose:
        print(json.dumps(summary))
    if json_path:
        with open(json_path, "w") as f:
            json.dump({"summary": summary, "windows": results}, f, indent=1)
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="trainstep_trained_l8_w64")
    ap.add_argument("--windows", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--eval-rays", type=int, default=1024)
    ap.add_argument("--perturb", type=float, default=1e-6)
    ap.add_argument("--json", default="")
    ap.add_argument("--mma", default="f32", help="comma-separated MMA modes of the HIP pipelines trained in lockstep (the first one is "
                    "`hip` in the records): f32,bf16x6,bf16")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("RSN_CPU_THREADS", "16")))
    run(a.fixture, a.windows, a.steps, a.rays, a.eval_rays, a.perturb, json_path=a.json, modes=tuple(a.mma.split(",")))


if __name__ == "__main__":
    main()
