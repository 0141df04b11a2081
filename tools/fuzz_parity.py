"""Randomised differential run (GPU box): eval get_outputs of the HIP path vs the CPU oracle over random network shapes,
sample counts, ray batches and degenerate inputs.  Prints the worst error per case; exits non-zero on a violation.

    python tools/fuzz_parity.py [--cases 30] [--seed 0]
"""
import argparse
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mma", default="f32", help="f32 | bf16x6 (the fp32-emulating mode must meet the same bounds)")
    ap.add_argument("--train", action="store_true", help="training mode: forward with shared jitter + backward of the "
                    "reference loss; gradients compared in direction (cosine) and None-pattern")
    args = ap.parse_args()
    import reflect_sampling_nerf_amd as pkg
    from oracle import cpu_ref

    pkg.load_library()
    dev = torch.device("cuda:0")
    rng = random.Random(args.seed)
    bad = 0
    for case in range(args.cases):
        layers = rng.choice([2, 3, 4, 6, 7, 8])  # 5 is invalid in the reference too (skip index 4 = last layer: shape error)
        width = rng.choice([64, 64, 128, 256])
        samples = tuple(rng.choice([1, 2, 3, 5, 8, 13, 24, 33, 40]) for _ in range(4))
        R = rng.choice([1, 2, 7, 31, 64, 97, 130])
        bias = rng.choice([-12.0, 0.0, 1.0, 2.0, 4.0])
        kind = rng.choice(["plain", "plain", "unnormalised_dirs", "huge_pixel_area", "tiny_pixel_area", "near_eq_far",
                           "inside_unit_ball"])
        torch.manual_seed(case)
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=samples[0], num_importance_samples=samples[1],
                                                num_reflect_coarse_samples=samples[2],
                                                num_reflect_importance_samples=samples[3], base_mlp_num_layers=layers,
                                                base_mlp_layer_width=width)
        model = cfg.setup(scene_box=None, num_train_data=1)
        with torch.no_grad():
            model.field.field_output_density.net.bias += bias
        P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
        model.to(dev).eval()
        model.field.set_mma_mode(args.mma)
        o, d, pa = cpu_ref.synthetic_rays(R, seed=1000 + case)
        nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
        if kind == "unnormalised_dirs":
            d = d * (0.25 + 3.0 * torch.rand(R, 1))
        elif kind == "huge_pixel_area":
            pa = pa * 1e4
        elif kind == "tiny_pixel_area":
            pa = pa * 1e-6
        elif kind == "near_eq_far":
            fars = nears + 1e-3
        elif kind == "inside_unit_ball":
            o = o * 0.1
            nears, fars = torch.full((R, 1), 0.05), torch.full((R, 1), 1.5)
        fs, ms = cpu_ref.FieldSpec(num_layers=layers, width=width), cpu_ref.ModelSpec(*samples)
        rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev),
                           fars=fars.to(dev))
        grad_note = ""
        if args.train:
            g = torch.Generator().manual_seed(case)
            jit = {"coarse": torch.rand(R, samples[0] + 1, generator=g), "fine": torch.rand(R, samples[1] + 1, generator=g),
                   "reflect_coarse": torch.rand(R, samples[2] + 1, generator=g),
                   "reflect_fine": torch.rand(R, samples[3] + 1, generator=g)}
            image = torch.rand(R, 3, generator=g)
            Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            ref = cpu_ref.get_outputs(Pg, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit)
            sum(cpu_ref.loss_dict(ref, image).values()).backward()
            model.train()
            mk = ref["mask"]
            jg = dict(jit, reflect_coarse=jit["reflect_coarse"][mk], reflect_fine=jit["reflect_fine"][mk])
            out = model._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in jg.items()})
            chk = dict(out)
            chk["normals_coarse"], chk["normals_fine"] = ref["normals_coarse"].detach().to(dev), ref["normals_fine"].detach().to(dev)
            sum(model.get_loss_dict(chk, {"image": image.to(dev)}).values()).backward()
            worst_cos, pat = 1.0, True
            for name, p in model.field.named_parameters():
                gr = Pg[name].grad
                if gr is None or float(gr.abs().max()) == 0.0:
                    pat &= p.grad is None or float(p.grad.abs().max()) <= 1e-12
                    continue
                if p.grad is None or not bool(torch.isfinite(p.grad).all()):
                    pat = False
                    continue
                a, b = p.grad.cpu().flatten().double(), gr.flatten().double()
                worst_cos = min(worst_cos, float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)))
            grad_note = " grad-pattern=%s min-cos=%.5f" % (pat, worst_cos)
            ref = {k: v.detach() for k, v in ref.items()}
            out = {k: v.detach() for k, v in out.items()}
            if not pat or worst_cos < 0.99:
                grad_note += " GRAD-VIOLATION"
        else:
            with torch.no_grad():
                ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=False)
            out = model._get_outputs_eval(rb) if hasattr(model, "_get_outputs_eval") else model(rb)
        worst, worst_key = 0.0, ""
        ok = set(out.keys()) == set(ref.keys())
        flips = int((out["mask"].cpu() != ref["mask"]).sum())
        if ok and flips == 0:
            for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_coarse",
                      "accumulation_fine", "weights_coarse", "weights_fine", "diff", "tint", "roughness"):
                e = float((out[k].cpu() - ref[k]).abs().max()) if out[k].numel() else 0.0
                if not (e == e):
                    e = float("inf")
                if e > worst:
                    worst, worst_key = e, k
        status = "ok" if (ok and flips == 0 and worst <= 1e-4 and "GRAD-VIOLATION" not in grad_note) else "VIOLATION"
        bad += status != "ok"
        print("case %2d L=%d W=%3d S=%-16s R=%3d bias=%5.1f %-18s M=%3d keys=%s flips=%d worst %.2e %s  %s" %
              (case, layers, width, samples, R, bias, kind, int(ref["mask"].sum()), ok, flips, worst, worst_key + grad_note, status),
              flush=True)
    print("violations:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
