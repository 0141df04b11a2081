#!/usr/bin/env python3
"""Generate the TRAINED-weights fixtures (tests/golden/*_trained_*.npz) by TRAINING the reference's own model.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).

The fixtures of oracle/make_golden.py are all `nn.Linear`-default weights.  The method's operating regime is a trained
field (reference config.py:32: 100 000 iterations), where densities are peaked, weights concentrate on a surface, the
reflection mask is decided by learned normals and the PDF resampler works on sharp histograms.  This script therefore
trains the REFERENCE (its own `get_outputs`, model.py:142-344, its own `get_loss_dict`, model.py:346-430, `torch.optim.RAdam`
with the learning-rate decay of config.py:50-53 and the 50-step loss warm-up of pipeline.py:79-91) on the procedural scene
of tools/train_parity.py on the CPU until it has left the initialisation regime, and then records with the trained
parameters

  eval_trained_<tag>.npz       eval-mode `get_outputs` on held-out rays (all output keys, the four samplers' bins)
  trainstep_trained_<tag>.npz  one whole training step (outputs, the eight scaled loss terms, every parameter gradient,
                               jitter draws, bins) against the scene's ground-truth colours
  params_trained_<tag>.npz     the trained state_dict, shared by both

Usage:  python oracle/make_golden_trained.py [--tag l8_w64 --layers 8 --width 64 --steps 1500 ...]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import make_golden as mg  # noqa: E402  (puts the shim + the reference on sys.path, imports the reference)
from oracle.make_golden import BinLog, RandLog, RayBundle  # noqa: E402
from tools.train_parity import psnr, scene_rays  # noqa: E402

WARMUP_KEYS = ("predicted_normal_loss_coarse", "predicted_normal_loss_fine", "orientation_loss_coarse",
               "orientation_loss_fine")


def lr_at(step, lr0=1e-3, lr1=1e-4, max_steps=50000):
    """ExponentialDecayScheduler of the reference's config (config.py:50-53): log-linear from lr0 to lr1."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return float(np.exp(np.log(lr0) * (1 - t) + np.log(lr1) * t))


def bundle(o, d, pa, near=2.0, far=6.0):
    n = o.shape[0]
    nears, fars = torch.full((n, 1), near), torch.full((n, 1), far)
    return RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=pa.clone(), nears=nears.clone(),
                     fars=fars.clone()), nears, fars


def train_reference(model, steps, rays, seed, log_every=100):
    """The reference's training iteration, minus nerfstudio's Trainer: rays of the scene -> get_outputs (train mode) ->
    get_loss_dict -> backward -> RAdam."""
    params = list(model.field.parameters())
    opt = torch.optim.RAdam(params, lr=1e-3, eps=1e-15)
    gen = torch.Generator().manual_seed(seed)
    full = dict(model.config.loss_coefficients)
    sink = io.StringIO()
    model.train(True)
    t0 = time.time()
    hist = []
    for step in range(steps):
        o, d, pa, rgb = scene_rays(rays, gen)
        for k in WARMUP_KEYS:  # pipeline.py:79-91
            model.config.loss_coefficients[k] = 0.0 if step < 50 else full[k]
        for g in opt.param_groups:
            g["lr"] = lr_at(step)
        opt.zero_grad(set_to_none=True)
        rb, _, _ = bundle(o, d, pa)
        with contextlib.redirect_stdout(sink):  # the reference prints debug lines
            out = model.get_outputs(rb)
            loss = sum(model.get_loss_dict(out, {"image": rgb}).values())
        loss.backward()
        opt.step()
        sink.seek(0), sink.truncate(0)
        if step % log_every == 0 or step == steps - 1:
            hist.append((step, float(loss), int(out["mask"].sum())))
            print(f"  step {step:5d} loss {float(loss):.6f} M {int(out['mask'].sum())}/{rays} "
                  f"({time.time() - t0:.0f} s)", flush=True)
    for k in WARMUP_KEYS:
        model.config.loss_coefficients[k] = full[k]
    return hist


def record(model, tag, R_eval, R_step, seed, train_meta):
    param_file = f"params_trained_{tag}"
    path = os.path.join(mg.OUT_DIR, param_file + ".npz")
    if os.path.exists(path):
        os.remove(path)
    sink = io.StringIO()
    # ---- eval-mode get_outputs on held-out rays
    o, d, pa, rgb = scene_rays(R_eval, torch.Generator().manual_seed(seed + 1000))
    rb, nears, fars = bundle(o, d, pa)
    model.train(False)
    binlog = BinLog(model)
    with contextlib.redirect_stdout(sink), torch.no_grad():
        out = model.get_outputs(rb)
    binlog.close()
    arrays = {}
    mg.save_params(arrays, model, param_file)
    for k, v in binlog.bins.items():
        arrays["bins/" + k] = v.numpy()
    for k, v in [("origins", o), ("directions", d), ("pixel_area", pa), ("nears", nears), ("fars", fars), ("image", rgb)]:
        arrays["in/" + k] = v.numpy()
    for k, v in out.items():
        a = v.detach().numpy()
        arrays["out/" + k] = a.astype(np.uint8) if a.dtype == np.bool_ else a.astype(np.float32)
    meta = dict(name=f"eval_trained_{tag}", R=R_eval, training=False, near=2.0, far=6.0, M=int(out["mask"].sum()),
                keys=sorted(out.keys()), param_file=param_file, density_bias_shift=0.0,
                psnr_mid_rgb_fine=psnr(out["mid_rgb_fine"], rgb), psnr_mid_reflect_fine=psnr(out["mid_reflect_fine"], rgb),
                generator="oracle/make_golden_trained.py over /root/reference + oracle/ns_shim", torch=torch.__version__,
                **train_meta)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    p = os.path.join(mg.OUT_DIR, meta["name"] + ".npz")
    np.savez_compressed(p, **arrays)
    print(f"{meta['name']}: R={R_eval} M={meta['M']} PSNR fine {meta['psnr_mid_rgb_fine']:.2f} dB reflect "
          f"{meta['psnr_mid_reflect_fine']:.2f} dB -> {os.path.getsize(p)/1024:.0f} KiB")
    # ---- one whole training step on the trained parameters
    o, d, pa, rgb = scene_rays(R_step, torch.Generator().manual_seed(seed + 2000))
    rb, nears, fars = bundle(o, d, pa)
    model.train(True)
    for p_ in model.field.parameters():
        p_.grad = None
    torch.manual_seed(seed + 7)
    binlog = BinLog(model)
    with RandLog() as log, contextlib.redirect_stdout(sink):
        out = model.get_outputs(rb)
        loss_dict = model.get_loss_dict(out, {"image": rgb})
        total = sum(loss_dict.values())
        total.backward()
    binlog.close()
    arrays = {}
    mg.save_params(arrays, model, param_file)
    for k, v in binlog.bins.items():
        arrays["bins/" + k] = v.numpy()
    n_grad = 0
    for k, p_ in model.field.named_parameters():
        if p_.grad is not None:
            arrays["grad/" + k] = p_.grad.detach().numpy().astype(np.float32)
            n_grad += 1
    for k, v in [("origins", o), ("directions", d), ("pixel_area", pa), ("nears", nears), ("fars", fars), ("image", rgb)]:
        arrays["in/" + k] = v.numpy()
    for k, v in out.items():
        a = v.detach().numpy()
        arrays["out/" + k] = a.astype(np.uint8) if a.dtype == np.bool_ else a.astype(np.float32)
    for k, v in loss_dict.items():
        arrays["loss/" + k] = np.float32(v.detach().item()).reshape(1)
    assert len(log.draws) in (2, 4), len(log.draws)
    for n, t in zip(["coarse", "fine", "reflect_coarse", "reflect_fine"], log.draws):
        arrays["jitter/" + n] = t.numpy()
    meta = dict(name=f"trainstep_trained_{tag}", R=R_step, training=True, near=2.0, far=6.0, M=int(out["mask"].sum()),
                keys=sorted(out.keys()), loss_coefficients={k: float(v) for k, v in model.config.loss_coefficients.items()},
                param_file=param_file, density_bias_shift=0.0,
                generator="oracle/make_golden_trained.py over /root/reference + oracle/ns_shim "
                          "(get_outputs + get_loss_dict + backward)", torch=torch.__version__, **train_meta)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    p = os.path.join(mg.OUT_DIR, meta["name"] + ".npz")
    np.savez_compressed(p, **arrays)
    print(f"{meta['name']}: R={R_step} M={meta['M']} losses={len(loss_dict)} grads={n_grad} total={float(total):.6f} "
          f"-> {os.path.getsize(p)/1024:.0f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="l8_w64")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--samples", type=int, nargs=4, default=[32, 32, 16, 16])
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--eval-rays", type=int, default=64)
    ap.add_argument("--step-rays", type=int, default=32)
    ap.add_argument("--seed", type=int, default=30)
    ap.add_argument("--threads", type=int, default=4)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    os.makedirs(mg.OUT_DIR, exist_ok=True)
    model = mg.build_model(tuple(args.samples), args.layers, args.width, args.seed, 0.0)
    print(f"training the reference: {args.layers} x {args.width}, {args.rays} rays x {args.samples}, {args.steps} steps")
    hist = train_reference(model, args.steps, args.rays, args.seed + 1)
    train_meta = dict(layers=args.layers, width=args.width, samples=list(args.samples), seed=args.seed,
                      trained_steps=args.steps, trained_rays=args.rays,
                      scene="tools/train_parity.py:scene_rays (Lambert sphere r=0.8, white background)",
                      loss_first=hist[0][1], loss_last=hist[-1][1])
    record(model, args.tag, args.eval_rays, args.step_rays, args.seed, train_meta)


if __name__ == "__main__":
    main()
