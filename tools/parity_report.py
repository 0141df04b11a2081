#!/usr/bin/env python3
"""Prints max-abs error per output key of the HIP get_outputs vs the CPU oracle (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_parity import _run_model
from tests.helpers import max_abs
dev = torch.device("cuda:0")
for layers, width, samples, R, bias in [(8, 256, (32, 32, 16, 16), 70, 2.0), (8, 128, (128, 128, 64, 64), 33, 1.0),
                                        (4, 128, (64, 48, 24, 40), 130, 1.5), (8, 64, (16, 16, 8, 8), 257, 2.0)]:
    out, ref = _run_model(dev, layers, width, samples, R, seed=layers + width, bias_shift=bias)
    print(f"--- L={layers} W={width} S={samples} R={R} M={int(ref['mask'].sum())}")
    for k in sorted(ref.keys()):
        if k == "mask":
            print(f"  mask flips: {int((out[k].cpu() != ref[k]).sum())}")
        else:
            print(f"  {k:24s} {max_abs(out[k].cpu(), ref[k]):.3e}")
