"""Times the HBM-side kernels (samplers, compositing forward/backward) at BASELINE config 2 / 3 sizes and prints the
achieved GB/s against their algorithmic bytes.

    python tools/render_kernels_report.py [--out profiles/rNN_render_kernels.json]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd import _abi, ops
    from reflect_sampling_nerf_amd.train_graph import _composite_backward

    pkg.load_library()
    dev = torch.device("cuda", 0)
    rows = []
    for R, S in ((4096, 128), (16384, 192), (65536, 128)):
        g = torch.Generator(device="cpu").manual_seed(0)
        nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
        sigma = (torch.rand(R, S, generator=g) * 4.0).to(dev)
        color = torch.rand(R, S, 3, generator=g).to(dev)
        sb, eb = ops.sample_spaced(R, None, S, _abi.RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
        comp = ops.composite(R, None, S, 1, 0, sigma, eb, color)
        w = comp["weights"]
        cases = {
            # name: (callable, algorithmic bytes)
            "sample_spaced": (lambda: ops.sample_spaced(R, None, S, _abi.RSN_SPACING_UNIFORM, 1.0, nears, fars, None),
                              R * 8 + 2 * R * (S + 1) * 4),
            "sample_pdf": (lambda: ops.sample_pdf(R, None, S, S, _abi.RSN_SPACING_UNIFORM, 1.0, 0.01, nears, fars, w, sb, None),
                           R * S * 4 + R * (S + 1) * 4 + 2 * R * (S + 1) * 4),
            "composite (rgb, acc, depth, weights)": (lambda: ops.composite(R, None, S, 1, ops.RSN_COMP_EVAL, sigma, eb, color),
                                                    R * S * (4 + 12) + R * (S + 1) * 4 + R * S * 4 + R * 20),
        }
        lv = {"sigma": sigma, "color": color}
        g_rgb = torch.rand(R, 3, generator=g).to(dev)
        cases["composite backward (g_sigma, g_color)"] = (
            lambda: _composite_backward(R, S, 1, ops.RSN_COMP_CLIP_RGB, 0, lv, eb, w, g_rgb),
            R * S * (4 + 12 + 4) + R * (S + 1) * 4 + R * S * (4 + 12) + R * 12)
        for name, (fn, nbytes) in cases.items():
            us = timed(fn)
            rows.append({"rays": R, "samples": S, "kernel": name, "us": us, "algorithmic_bytes": nbytes,
                         "GBps": nbytes / us / 1e3, "frac_of_8TBps": nbytes / us / 1e3 / 8000.0})
            print("R=%6d S=%3d  %-40s %8.1f us  %7.1f MB  %7.0f GB/s (%4.1f %% of 8 TB/s)" %
                  (R, S, name, us, nbytes / 1e6, rows[-1]["GBps"], 100 * rows[-1]["frac_of_8TBps"]), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
