"""Per-phase shader-clock accounting of rsn_field_kernel (debug build with -DRSN_PHASE_TIMERS, compiled on the spot
into a temporary directory; the shipped librsn_hip.so never carries the timers).

    python tools/phase_report.py [--mma f32] [--rays 4096] [--samples 128]
"""
import argparse
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

PHASES = ["encode (inputs, IPE, stash)", "init_acc (bias -> accumulators)", "gemm enc0 / enc_skip", "store_act (ReLU, LDS)",
          "gemm trunk layers 1..L-1", "gemm bottleneck+heads", "heads epilogue + SH-34", "gemm mid (SH part)",
          "gemm mid (x part)", "gemm rgb", "final epilogue (colour out)", "tile loop head",
          "normals sweep: gemms (training)", "normals sweep: masks, zero_acc, masked stores", "normals: encode chain + output"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mma", default="f32")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--define", action="append", default=[], help="extra -D macros for kernel experiments")
    ap.add_argument("--train", action="store_true", help="the training forward (saved activations, analytic normals)")
    args = ap.parse_args()
    import torch

    from tools._variant import build_variant

    lib = build_variant(["RSN_PHASE_TIMERS", *args.define])
    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd import _abi, ops

    handle = pkg.load_library(lib)
    dbg = handle.rsn_debug_phase_cycles
    dbg.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    dbg.restype = ctypes.c_int
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    R, S = args.rays, args.samples
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S)
    model = cfg.setup(scene_box=None, num_train_data=1).to(dev)
    model = model.train() if args.train else model.eval()
    fld = model.field
    fld.set_mma_mode(args.mma)
    o, d, pa = synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears = torch.full((R,), 2.0, device=dev)
    fars = torch.full((R,), 6.0, device=dev)
    sb, eb = ops.sample_spaced(R, None, S, _abi.RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    run = (lambda: fld.evaluate_frustums_train(o, d, pa, eb, want_normals=True)) if args.train else \
        (lambda: fld.evaluate_frustums(o, d, pa, eb, full=True))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)()
    assert dbg(None, 1) == 0
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    assert dbg(buf, 0) == 0
    cyc = [int(v) for v in buf]
    waves = cyc[15]
    tot = sum(cyc[:15])
    rep = {"mma": args.mma, "rays": R, "samples": S, "kernel_ms_with_timers": e0.elapsed_time(e1) / n,
           "timed_waves": waves, "cycles_per_timed_wave": tot / max(waves, 1), "phases": {}}
    print("kernel (with timers) %.3f ms; %d timed waves, %.0f cycles each" % (rep["kernel_ms_with_timers"], waves,
                                                                             rep["cycles_per_timed_wave"]))
    for name, c in zip(PHASES, cyc[:15]):
        rep["phases"][name] = {"cycles_per_wave": c / max(waves, 1), "frac": c / max(tot, 1)}
        print("  %-34s %10.0f cycles/wave  %5.1f %%" % (name, c / max(waves, 1), 100.0 * c / max(tot, 1)))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()
