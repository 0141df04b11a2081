#!/usr/bin/env python3
"""Bitwise comparison of two tools/level_dump.py files.  Usage: python tools/level_dump_cmp.py a.pt b.pt"""
import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
bad = 0
for k in a:
    x, y = a[k], b[k]
    if x.dtype in (torch.int32, torch.int64, torch.uint8):
        eq = torch.equal(x, y)
        n = int((x != y).sum())
        print(f"{k:40s} int equal {eq} ({n} differ of {x.numel()})")
        bad += (not eq)
    else:
        e = (x.double() - y.double()).abs()
        eq = torch.equal(x.view(torch.uint8), y.view(torch.uint8))
        print(f"{k:40s} bit-equal {eq}  max|diff| {float(e.max()):.3e}  max|a| {float(x.abs().max()):.3e}")
        bad += (not eq)
print("tensors not bit-equal:", bad)
