#!/usr/bin/env python3
"""End-to-end training parity on a procedural scene (GPU box): the HIP path vs the CPU oracle (autograd),
same initial weights, same rays, same stratified jitter, same optimiser (RAdam lr 1e-3 eps 1e-15 with the exponential
decay of the reference's config.py:50-53) and the 50-step loss warm-up (pipeline.py:79-91).

Lockstep windows: in steps [0, oracle_steps) and again in the LAST `oracle_tail` steps both pipelines step on identical
batches (same rays, jitter, learning rate); in between the HIP path trains alone (the CPU oracle needs seconds per
step).  At the start of the tail window the oracle takes over the HIP path's parameters AND optimiser state, so the
second comparison -- loss trajectories and the PSNR of both on held-out rays, north-star bound 0.1 dB -- is made on a
model that has actually learned the scene (PSNR vs the analytic ground truth > 20 dB), not on two untrained ones.

Scene: a diffuse sphere of radius 0.8 lit by a directional light, white background, cameras on a radius-4 shell.
Importable: tests call `run(...)` in-process (no subprocess from a process that holds the GPU).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def scene_rays(n, gen):
    o = F.normalize(torch.randn(n, 3, generator=gen), dim=-1) * 4.0
    target = (torch.rand(n, 3, generator=gen) - 0.5) * 1.6
    d = F.normalize(target - o, dim=-1)
    # analytic sphere (radius 0.8 at the origin), Lambert + ambient, white background
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - 0.8**2
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp(min=0))
    p = o + t[:, None] * d
    nrm = F.normalize(p, dim=-1)
    light = F.normalize(torch.tensor([0.5, 0.8, 0.3]), dim=0)
    shade = 0.25 + 0.75 * (nrm @ light).clamp(min=0)
    base = torch.tensor([0.85, 0.35, 0.25])
    rgb = torch.where(hit[:, None], shade[:, None] * base[None, :], torch.ones(n, 3))
    pa = torch.full((n, 1), (1.0 / 200.0) ** 2)
    return o, d, pa, rgb


def psnr(a, b):
    return float(10.0 * torch.log10(1.0 / torch.mean((a - b) ** 2)))


def loss_terms(out, image):
    return {
        "loss_mid_coarse": F.mse_loss(image, out["mid_rgb_coarse"]), "loss_mid_fine": F.mse_loss(image, out["mid_rgb_fine"]),
        "loss_reflect_mid_coarse": F.mse_loss(image, out["mid_reflect_coarse"]),
        "loss_reflect_mid_fine": F.mse_loss(image, out["mid_reflect_fine"]),
        "predicted_normal_loss_coarse": torch.sum(out["weights_coarse"] * torch.sum((out["normals_coarse"] - out["pred_normals_coarse"]) ** 2, dim=-1, keepdim=True)),
        "predicted_normal_loss_fine": torch.sum(out["weights_fine"] * torch.sum((out["normals_fine"] - out["pred_normals_fine"]) ** 2, dim=-1, keepdim=True)),
        "orientation_loss_coarse": torch.sum(out["weights_coarse"] * torch.clamp(out["n_dot_d_coarse"], min=0.0) ** 2),
        "orientation_loss_fine": torch.sum(out["weights_fine"] * torch.clamp(out["n_dot_d_fine"], min=0.0) ** 2),
    }


def run(steps=120, oracle_steps=None, rays=256, width=64, layers=8, samples=(32, 32, 16, 16), eval_rays=1024,
        eval_every=0, json_path="", verbose=True, oracle_tail=0, tail_eval_every=0, save_state=""):
    import reflect_sampling_nerf_amd as pkg
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd.parallel import apply_loss_warmup
    from reflect_sampling_nerf_amd.train_ops import exponential_decay_lr

    oracle_steps = steps if oracle_steps is None else min(oracle_steps, steps)
    tail_start = steps - oracle_tail if (oracle_tail and steps - oracle_tail > oracle_steps) else None
    dev = torch.device("cuda:0")
    R, S = rays, list(samples)
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S[0], num_importance_samples=S[1],
                                            num_reflect_coarse_samples=S[2], num_reflect_importance_samples=S[3],
                                            base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.field.state_dict().items()}
    model.to(dev)
    fs = cpu_ref.FieldSpec(num_layers=layers, width=width)
    ms = cpu_ref.ModelSpec(*S)
    names = [n for n, _ in model.field.named_parameters()]
    opt_g = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)
    opt_c = torch.optim.RAdam([P[n] for n in names], lr=1e-3, eps=1e-15)
    gen = torch.Generator().manual_seed(7)
    eo, ed, epa, ergb = scene_rays(eval_rays, torch.Generator().manual_seed(99))  # held-out rays
    near = lambda n: torch.full((n, 1), 2.0)  # noqa: E731
    far = lambda n: torch.full((n, 1), 6.0)  # noqa: E731
    erb = pkg.RayBundle(origins=eo.to(dev), directions=ed.to(dev), pixel_area=epa.to(dev), nears=near(eval_rays).to(dev),
                        fars=far(eval_rays).to(dev))

    def eval_hip():
        model.eval()
        with torch.no_grad():
            og = model(erb)
        model.train()
        return og

    hist, checkpoints = [], []
    mask_flips = 0
    t_g = t_c = 0.0
    model.train()
    for step in range(steps):
        lockstep = step < oracle_steps or (tail_start is not None and step >= tail_start)
        if tail_start is not None and step == tail_start:
            # the oracle resumes from the HIP path's state: parameters, RAdam moments, step count
            torch.cuda.synchronize()
            with torch.no_grad():
                for n, p_hip in model.field.named_parameters():
                    P[n].copy_(p_hip.detach().cpu())
            opt_c = torch.optim.RAdam([P[n] for n in names], lr=1e-3, eps=1e-15)
            for p_c, m1, m2 in zip([P[n] for n in names], opt_g.exp_avg, opt_g.exp_avg_sq):
                opt_c.state[p_c] = {"step": torch.tensor(float(opt_g.step_count)), "exp_avg": m1.detach().cpu().clone(),
                                    "exp_avg_sq": m2.detach().cpu().clone()}
            og = eval_hip()
            with torch.no_grad():  # the two renders of the SAME trained weights (SURVEY 8(d) PSNR (i))
                oc = cpu_ref.get_outputs({k: v.detach() for k, v in P.items()}, fs, ms, eo, ed, epa, near(eval_rays),
                                         far(eval_rays), training=False)
            checkpoints.append({"step": step, "psnr_hip": psnr(og["mid_rgb_fine"].cpu(), ergb),
                                "psnr_oracle_same_weights": psnr(oc["mid_rgb_fine"], ergb),
                                "psnr_hip_vs_oracle_render_same_weights": psnr(og["mid_rgb_fine"].cpu(), oc["mid_rgb_fine"]),
                                "max_abs_render_diff_same_weights": float((og["mid_rgb_fine"].cpu() - oc["mid_rgb_fine"]).abs().max()),
                                "mask_flips_same_weights": int((og["mask"].cpu() != oc["mask"]).sum()),
                                "note": "oracle resumes from the HIP state here"})
            if save_state:  # for tools/chaos_probe.py (CPU): parameters, RAdam moments, step, the batch generator's state
                import numpy as np
                os.makedirs(os.path.dirname(os.path.abspath(save_state)), exist_ok=True)
                arrs = {"param/" + n: P[n].detach().numpy() for n in names}
                for n, m1, m2 in zip(names, opt_g.exp_avg, opt_g.exp_avg_sq):
                    arrs["m1/" + n], arrs["m2/" + n] = m1.detach().cpu().numpy(), m2.detach().cpu().numpy()
                arrs["step"] = np.array([opt_g.step_count]); arrs["gen_state"] = gen.get_state().numpy()
                arrs["meta"] = np.frombuffer(json.dumps({"rays": R, "samples": S, "layers": layers, "width": width}).encode(), dtype=np.uint8)
                np.savez_compressed(save_state, **arrs)
            if verbose:
                print("checkpoint", json.dumps(checkpoints[-1]), flush=True)
        o, d, pa, rgb = scene_rays(R, gen)
        jit = {"coarse": torch.rand(R, S[0] + 1, generator=gen), "fine": torch.rand(R, S[1] + 1, generator=gen),
               "reflect_coarse": torch.rand(R, S[2] + 1, generator=gen), "reflect_fine": torch.rand(R, S[3] + 1, generator=gen)}
        apply_loss_warmup(model, step)
        coeff = dict(model.config.loss_coefficients)
        lc = None
        if lockstep:  # --- CPU oracle (torch.optim.RAdam with the same learning-rate schedule)
            t0 = time.perf_counter()
            for grp in opt_c.param_groups:
                grp["lr"] = exponential_decay_lr(step, 1e-3, 1e-4, 50000)
            opt_c.zero_grad(set_to_none=True)
            ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, near(R), far(R), training=True, jitter=jit)
            lc = sum(v * coeff[k] for k, v in loss_terms(ref, rgb).items())
            lc.backward()
            opt_c.step()
            t_c += time.perf_counter() - t0
        # --- HIP path
        t0 = time.perf_counter()
        opt_g.zero_grad(set_to_none=True)
        rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=near(R).to(dev), fars=far(R).to(dev))
        if lockstep:  # reflect-level draws are per ORIGINAL ray: each pipeline uses the rows of its own reflected rays
            out = model._get_outputs_train(rb, jitter=jit)
            if not torch.equal(out["mask"].cpu(), ref["mask"]):
                mask_flips += int((out["mask"].cpu() != ref["mask"]).sum())
        else:
            out = model._get_outputs_train(rb, jitter={"coarse": jit["coarse"], "fine": jit["fine"]})
        lg = sum(model.get_loss_dict(out, {"image": rgb.to(dev)}).values())
        lg.backward()
        opt_g.step()
        if lockstep or step % 50 == 0 or step == steps - 1:
            torch.cuda.synchronize()
            hist.append((step, None if lc is None else float(lc), float(lg)))
        t_g += time.perf_counter() - t0
        if verbose and (step % 50 == 0 or step == steps - 1):
            print(f"step {step:5d} loss hip {float(lg):.6f}" + (f" cpu {float(lc):.6f} rel {abs(float(lc)-float(lg))/abs(float(lc)):.2e}" if lc is not None else ""), flush=True)
        at_end_of_lockstep = step == oracle_steps - 1 or (tail_start is not None and step == steps - 1)
        in_tail = tail_start is not None and step >= tail_start
        if at_end_of_lockstep or (eval_every and step % eval_every == eval_every - 1) or step == steps - 1 or \
                (in_tail and tail_eval_every and (step - tail_start) % tail_eval_every == tail_eval_every - 1):
            og = eval_hip()
            cp = {"step": step + 1, "psnr_hip": psnr(og["mid_rgb_fine"].cpu(), ergb),
                  "psnr_hip_reflect_fine": psnr(og["mid_reflect_fine"].cpu(), ergb)}
            if at_end_of_lockstep or in_tail:
                with torch.no_grad():
                    oc = cpu_ref.get_outputs({k: v.detach() for k, v in P.items()}, fs, ms, eo, ed, epa, near(eval_rays),
                                             far(eval_rays), training=False)
                cp["psnr_oracle"] = psnr(oc["mid_rgb_fine"], ergb)
                cp["psnr_hip_vs_oracle_render"] = psnr(og["mid_rgb_fine"].cpu(), oc["mid_rgb_fine"])
                cp["psnr_delta_db"] = cp["psnr_hip"] - cp["psnr_oracle"]
            checkpoints.append(cp)
            if verbose:
                print("checkpoint", json.dumps(cp), flush=True)
            if json_path:  # partial results survive a time limit
                os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
                with open(json_path, "w") as f:
                    json.dump({"partial": True, "checkpoints": checkpoints, "history": hist}, f)
    lock = [h for h in hist if h[1] is not None]
    at_lock = [c for c in checkpoints if "psnr_oracle" in c][-1]  # the last lockstep window's end
    tail = [h for h in lock if tail_start is not None and h[0] >= tail_start]
    tail_cps = [c for c in checkpoints if tail_start is not None and c["step"] > tail_start and "psnr_oracle" in c]
    res = {"steps": steps, "oracle_steps": oracle_steps, "oracle_tail": oracle_tail, "rays": R, "samples": S,
           "field": f"{layers}x{width}",
           "max_rel_loss_diff_tail": max((abs(a - b) / abs(a) for _, a, b in tail), default=None),
           "tail_mean_psnr_hip": (sum(c["psnr_hip"] for c in tail_cps) / len(tail_cps)) if tail_cps else None,
           "tail_mean_psnr_oracle": (sum(c["psnr_oracle"] for c in tail_cps) / len(tail_cps)) if tail_cps else None,
           "psnr_hip": checkpoints[-1]["psnr_hip"], "psnr_hip_at_lockstep_end": at_lock["psnr_hip"],
           "psnr_oracle": at_lock["psnr_oracle"], "psnr_delta_db": at_lock["psnr_delta_db"],
           "psnr_hip_vs_oracle_render": at_lock["psnr_hip_vs_oracle_render"],
           "loss_first": lock[0][1:], "loss_last": lock[-1][1:], "mask_flips_in_lockstep": mask_flips,
           "max_rel_loss_diff_first10": max(abs(a - b) / abs(a) for _, a, b in lock[:10]),
           "max_rel_loss_diff_lockstep": max(abs(a - b) / abs(a) for _, a, b in lock),
           "sec_per_step_cpu": t_c / max(len(lock), 1), "sec_per_step_hip": t_g / steps, "checkpoints": checkpoints}
    if verbose:
        print(json.dumps(res))
    if json_path:
        os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
        with open(json_path, "w") as f:
            json.dump({"result": res, "history": hist}, f)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--oracle-steps", type=int, default=None, help="lockstep steps with the CPU oracle (default: all)")
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--samples", type=int, nargs=4, default=[32, 32, 16, 16])
    ap.add_argument("--oracle-tail", type=int, default=0, help="final steps in lockstep with the oracle resumed from the HIP state")
    ap.add_argument("--tail-eval-every", type=int, default=0, help="evaluate both pipelines every n steps of the tail window")
    ap.add_argument("--save-state", default="", help="npz with the state at the start of the tail window (tools/chaos_probe.py)")
    ap.add_argument("--eval-every", type=int, default=0)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    run(steps=args.steps, oracle_steps=args.oracle_steps, rays=args.rays, width=args.width, layers=args.layers,
        samples=tuple(args.samples), eval_every=args.eval_every, json_path=args.json, oracle_tail=args.oracle_tail,
        tail_eval_every=args.tail_eval_every, save_state=args.save_state)


if __name__ == "__main__":
    main()
