#!/usr/bin/env python3
"""Trains the BASELINE network (8 x 256) from the same initialisation on the same batches once per MMA mode (GPU box) and records the
held-out PSNR along the way: does training on the split-bf16 ring kernels (fp32-equivalent) / the plain-bf16 ring kernels (reduced
precision) converge like the exact-fp32 path?  The procedural scene of tools/train_parity.py; forward + the reference's eight loss
terms (with its 50-step warm-up) + backward + FusedRAdam with the reference's exponential decay, through parallel.train_step.

Usage: python tools/mode_convergence.py [--steps 3000] [--rays 1024] [--mma f32,bf16x6,bf16] [--json profiles/....json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.train_parity import psnr, scene_rays


def main():
    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd.parallel import train_step

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--eval-rays", type=int, default=4096)
    ap.add_argument("--eval-every", type=int, default=500)
    ap.add_argument("--samples", type=int, nargs=4, default=[32, 32, 16, 16])
    ap.add_argument("--mma", default="f32,bf16x6,bf16")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S = a.samples
    eo, ed, epa, ergb = scene_rays(a.eval_rays, torch.Generator().manual_seed(99))
    near = lambda n: torch.full((n, 1), 2.0)  # noqa: E731
    far = lambda n: torch.full((n, 1), 6.0)  # noqa: E731
    erb = pkg.RayBundle(origins=eo.to(dev), directions=ed.to(dev), pixel_area=epa.to(dev), nears=near(a.eval_rays).to(dev),
                        fars=far(a.eval_rays).to(dev))
    out = {"field": "8x256", "samples": S, "rays_per_step": a.rays, "steps": a.steps, "modes": {}}
    for mode in a.mma.split(","):
        torch.manual_seed(0)
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S[0], num_importance_samples=S[1],
                                                num_reflect_coarse_samples=S[2], num_reflect_importance_samples=S[3])
        model = cfg.setup(scene_box=None, num_train_data=1)
        model.to(dev).train()
        model.field.set_mma_mode(mode)
        opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)
        gen = torch.Generator().manual_seed(7)
        torch.manual_seed(1234)  # the model's own jitter draws: the same sequence in every mode
        curve, t0 = [], time.time()
        for step in range(a.steps + 1):
            if step % a.eval_every == 0:
                model.eval()
                with torch.no_grad():
                    og = model(erb)
                model.train()
                curve.append({"step": step, "psnr_mid_rgb_fine": psnr(og["mid_rgb_fine"].cpu(), ergb),
                              "psnr_mid_reflect_fine": psnr(og["mid_reflect_fine"].cpu(), ergb)})
                print(mode, json.dumps(curve[-1]), flush=True)
            if step == a.steps:
                break
            o, d, pa, rgb = scene_rays(a.rays, gen)
            rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=near(a.rays).to(dev),
                               fars=far(a.rays).to(dev))
            train_step(model, rb, {"image": rgb.to(dev)}, opt, None, step)
        torch.cuda.synchronize()
        out["modes"][mode] = {"curve": curve, "seconds": time.time() - t0}
    ref = out["modes"].get("f32")
    if ref:
        for mode, r in out["modes"].items():
            r["final_psnr_minus_f32_db"] = r["curve"][-1]["psnr_mid_rgb_fine"] - ref["curve"][-1]["psnr_mid_rgb_fine"]
    print(json.dumps({m: (r["curve"][-1], r.get("final_psnr_minus_f32_db")) for m, r in out["modes"].items()}))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
