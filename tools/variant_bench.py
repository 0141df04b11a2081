"""Times one eval field-kernel launch of an experimental build (extra -D macros, tools/_variant.py) and checks a slice
of its output against the exact-fp32 kernel of the same build.

    python tools/variant_bench.py --mma bf16x6 --define RSN_X6_REGSPLIT
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--mma", default="f32")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    args = ap.parse_args()
    import reflect_sampling_nerf_amd as pkg

    if args.define:
        from tools._variant import build_variant

        pkg.load_library(build_variant(args.define))
    else:
        pkg.load_library()
    from reflect_sampling_nerf_amd import _abi, ops
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    R, S = args.rays, args.samples
    model = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S).setup(
        scene_box=None, num_train_data=1).to(dev).eval()
    fld = model.field
    o, d, pa = synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    sb, eb = ops.sample_spaced(R, None, S, _abi.RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    ref = fld.evaluate_frustums(o[:256], d[:256], pa[:256], eb[:256].contiguous())
    fld.set_mma_mode(args.mma)
    for _ in range(3):
        lv = fld.evaluate_frustums(o, d, pa, eb, full=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        lv = fld.evaluate_frustums(o, d, pa, eb, full=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    err = float((lv["color"][:256] - ref["color"]).abs().max())
    print("%s %s: %.3f ms per launch, %.0f rays/s, max |colour - f32| on 256 rays %.2e" %
          (args.define, args.mma, ms, R / ms * 1e3, err))


if __name__ == "__main__":
    main()
