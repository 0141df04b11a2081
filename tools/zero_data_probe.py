#!/usr/bin/env python3
"""How much of a field kernel's time is the clock the chip holds under load (MI355X_MICROARCH.md, DVFS give-back
item 1): the SAME binary and launch, once with the synthetic random-init weights and once with every field parameter
set to zero (all MFMA operands past layer 0 are then zeros; the instruction stream has no data-dependent branch).
Prints kernel time by HIP events on the launch stream for each eval mode.  Usage (GPU box): python tools/zero_data_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reflect_sampling_nerf_amd as pkg  # noqa: E402
from reflect_sampling_nerf_amd import ops  # noqa: E402
from reflect_sampling_nerf_amd._abi import RSN_SPACING_UNIFORM  # noqa: E402
from reflect_sampling_nerf_amd.synthetic import synthetic_rays  # noqa: E402


def level_ms(R, S, mma, zero, steps=30, warmup=10):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S)
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.to(dev).eval()
    fld = model.field
    if zero:
        with torch.no_grad():
            for p in fld.parameters():
                p.zero_()
    fld.set_mma_mode(mma)
    o, d, pa = synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    fld.packed_weights()
    sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    for _ in range(warmup):
        fld.evaluate_frustums(o, d, pa, eb, full=True)
    timer = ops.KernelTimer()
    ops.TIMER = timer
    try:
        for _ in range(steps):
            fld.evaluate_frustums(o, d, pa, eb, full=True)
        torch.cuda.synchronize()
    finally:
        ops.TIMER = None
    t = timer.totals()["field_forward_eval"]
    return t["ms"] / t["calls"]


def main():
    print("# field kernel (eval level), ms per launch by HIP events: random-init weights vs all-zero weights, same binary")
    for name, R, S, mma in (("configs[1] f32", 4096, 128, "f32"), ("4096 x 128 bf16x6", 4096, 128, "bf16x6"),
                            ("configs[3] bf16 ring16", 16384, 192, "bf16")):
        for rep in range(2):
            a = level_ms(R, S, mma, zero=False)
            b = level_ms(R, S, mma, zero=True)
            print("%-24s random %.3f ms   zeros %.3f ms   ratio %.3f" % (name, a, b, a / b), flush=True)


if __name__ == "__main__":
    main()
