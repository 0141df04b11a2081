#!/usr/bin/env python3
"""CPU only: how fast do two runs of the CPU ORACLE ITSELF drift apart from a trained state?

tools/train_parity.py --save-state writes the state at the start of its tail window (parameters, RAdam moments, step,
the batch generator).  This probe continues training from it twice with the oracle -- once as is, once with every
parameter perturbed by a relative 1e-6 (an fp32 rounding's worth) -- on identical batches and jitter, and reports the
relative loss difference per step and the PSNR of both on the held-out rays.  It calibrates what "the HIP path and the
oracle drift apart by x after n steps" means: anything the oracle does to itself is the conditioning of the training
dynamics (reflection-mask thresholds, ReLU kinks, RAdam's normalised updates), not a kernel difference.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import cpu_ref
from tools.train_parity import loss_terms, psnr, scene_rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("state")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--eps", type=float, default=1e-6)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    z = np.load(args.state)
    meta = json.loads(bytes(z["meta"]).decode())
    R, S = meta["rays"], meta["samples"]
    fs, ms = cpu_ref.FieldSpec(num_layers=meta["layers"], width=meta["width"]), cpu_ref.ModelSpec(*S)
    names = [k[6:] for k in z.files if k.startswith("param/")]
    step0 = int(z["step"][0])
    from reflect_sampling_nerf_amd.train_ops import exponential_decay_lr

    def make(eps):
        g = torch.Generator().manual_seed(123)
        P = {}
        for n in names:
            p = torch.from_numpy(z["param/" + n]).clone()
            if eps:
                p = p * (1.0 + eps * torch.randn(p.shape, generator=g))
            P[n] = p.requires_grad_(True)
        opt = torch.optim.RAdam([P[n] for n in names], lr=1e-3, eps=1e-15)
        for n in names:
            opt.state[P[n]] = {"step": torch.tensor(float(step0)), "exp_avg": torch.from_numpy(z["m1/" + n]).clone(),
                               "exp_avg_sq": torch.from_numpy(z["m2/" + n]).clone()}
        return P, opt

    runs = [make(0.0), make(args.eps)]
    gen = torch.Generator()
    gen.set_state(torch.from_numpy(z["gen_state"]))
    eo, ed, epa, ergb = scene_rays(1024, torch.Generator().manual_seed(99))
    near = lambda n: torch.full((n, 1), 2.0)  # noqa: E731
    far = lambda n: torch.full((n, 1), 6.0)  # noqa: E731
    coeff = dict(cpu_ref.LOSS_COEFFICIENTS)
    hist = []
    for k in range(args.steps):
        step = step0 + k
        o, d, pa, rgb = scene_rays(R, gen)
        jit = {"coarse": torch.rand(R, S[0] + 1, generator=gen), "fine": torch.rand(R, S[1] + 1, generator=gen),
               "reflect_coarse": torch.rand(R, S[2] + 1, generator=gen), "reflect_fine": torch.rand(R, S[3] + 1, generator=gen)}
        losses, masks = [], []
        for P, opt in runs:
            for grp in opt.param_groups:
                grp["lr"] = exponential_decay_lr(step, 1e-3, 1e-4, 50000)
            opt.zero_grad(set_to_none=True)
            out = cpu_ref.get_outputs(P, fs, ms, o, d, pa, near(R), far(R), training=True, jitter=jit)
            loss = sum(v * coeff[kk] for kk, v in loss_terms(out, rgb).items())
            loss.backward()
            opt.step()
            losses.append(float(loss)); masks.append(out["mask"])
        hist.append((step, losses[0], losses[1], abs(losses[0] - losses[1]) / abs(losses[0]), int((masks[0] != masks[1]).sum())))
        if k % 5 == 0 or k == args.steps - 1:
            print("step %d loss %.6f %.6f rel %.2e mask flips %d" % hist[-1], flush=True)
    ps = []
    for P, _ in runs:
        with torch.no_grad():
            oc = cpu_ref.get_outputs({k: v.detach() for k, v in P.items()}, fs, ms, eo, ed, epa, near(1024), far(1024), training=False)
        ps.append(psnr(oc["mid_rgb_fine"], ergb))
    res = {"steps": args.steps, "relative_perturbation": args.eps, "psnr": ps, "psnr_delta_db": ps[1] - ps[0],
           "max_rel_loss_diff": max(h[3] for h in hist), "rel_loss_diff_by_step": [(h[0], h[3]) for h in hist[::5]],
           "mask_flips_total": sum(h[4] for h in hist)}
    print(json.dumps(res))
    if args.json:
        json.dump({"result": res, "history": hist}, open(args.json, "w"))


if __name__ == "__main__":
    main()
