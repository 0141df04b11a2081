"""Times rsn_weight_grad for the layer shapes of one training step and prints MFMA efficiency per shape.

    python tools/wgrad_report.py [--out profiles/rNN_wgrad.json]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--define", action="append", default=[], help="extra -D macros: times an experimental build")
    ap.add_argument("--quick", action="store_true", help="only the two large point counts")
    ap.add_argument("--mode", default="f32", choices=["f32", "bf16x6", "bf16"], help="MMA mode of the reduction")
    args = ap.parse_args()
    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd import train_graph
    from reflect_sampling_nerf_amd.train_graph import _wgrad

    train_graph._WGRAD_MODE = {"f32": 0, "bf16x6": 1, "bf16": 3}[args.mode]

    if args.define:
        from tools._variant import build_variant

        pkg.load_library(build_variant(args.define))
    else:
        pkg.load_library()
    dev = torch.device("cuda", 0)
    shapes = [(256, 256), (256, 104), (128, 256), (128, 40), (16, 256), (3, 128)]
    rows = []
    for n in ((262144, 524288) if args.quick else (262144, 524288, 212992, 4096)):
        for n_out, k_in in shapes:
            dy = torch.randn(n, max(n_out, 4), device=dev)
            x = torch.randn(n, k_in, device=dev)
            dw = torch.zeros(n_out, k_in, device=dev)
            db = torch.zeros(n_out, device=dev)
            for _ in range(3):
                _wgrad(dy, n_out, x, k_in, dw, 0, db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                _wgrad(dy, n_out, x, k_in, dw, 0, db)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            flop = 2.0 * n * n_out * k_in
            ref = dy[:, :n_out].double().t() @ x.double()
            err = float((dw.double() / (reps + 3) - ref).abs().max() / ref.abs().max())
            rows.append({"n_points": n, "n_out": n_out, "k_in": k_in, "us": us, "tflops": flop / us / 1e6,
                         "frac_of_fp32_mfma_peak": flop / (us * 1e-6) / PEAK, "rel_err": err})
            print("N=%7d  %3d x %3d  %8.1f us  %6.1f TF  %5.1f %% of peak  err %.1e" %
                  (n, n_out, k_in, us, rows[-1]["tflops"], 100 * rows[-1]["frac_of_fp32_mfma_peak"], err), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
