#!/usr/bin/env python3
"""Repeats the split-bf16 (bf16x6) ring backward sweep of one level six times on the same inputs (GPU box) and compares every
layer-gradient row with the exact-fp32 kernels' rows: the script that showed round 4's store hazard (rows 12..15 of a 16-row tile, one
element per K-step, intermittently: DESIGN 4.7).  The suite's test_split_bf16_ring_rows_match_the_exact_kernels_and_repeat_bitwise is
its regression form.  Usage: python tools/x6_bwd_repeat.py"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import train_graph
from oracle import cpu_ref
dev = torch.device("cuda:0")
torch.manual_seed(0)
R, S = 40, 32
cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S, num_reflect_coarse_samples=16, num_reflect_importance_samples=16)
model = cfg.setup(scene_box=None, num_train_data=1)
with torch.no_grad():
    model.field.field_output_density.net.bias += 2.0
model.to(dev).train()
f = model.field
o, d, pa = cpu_ref.synthetic_rays(R, seed=1)
o, d, pa = o.to(dev), d.to(dev), pa.to(dev).reshape(-1)
bins = (2.0 + 4.0 * torch.linspace(0, 1, S + 1)).repeat(R, 1).to(dev).contiguous()
gen = torch.Generator().manual_seed(3)
gin = {"sigma": torch.randn(R, S, generator=gen).to(dev), "color": torch.randn(R, S, 3, generator=gen).to(dev),
       "pred_normals": torch.randn(R, S, 3, generator=gen).to(dev), "n_dot_d": torch.randn(R, S, generator=gen).to(dev),
       "roughness": torch.randn(R, S, generator=gen).to(dev)}
f.set_mma_mode("f32")
lv = f.evaluate_frustums_train(o, d, pa, bins, want_normals=True)
ref = train_graph._field_backward(f, (o, d, pa), bins, lv, gin, False)
f.set_mma_mode("bf16x6")
lv = f.evaluate_frustums_train(o, d, pa, bins, want_normals=True)
bits = lv["saved"]["relu_bits"].clone()
runs = []
for it in range(6):
    go = train_graph._field_backward(f, (o, d, pa), bins, lv, gin, False)
    torch.cuda.synchronize()
    runs.append({k: v.clone() for k, v in go.items()})
    assert torch.equal(bits, lv["saved"]["relu_bits"])
for it, go in enumerate(runs):
    tot = 0
    desc = []
    for l in range(8):
        e = (go["dy"][l].double() - ref["dy"][l].double()).abs().cpu()
        bad = torch.nonzero(e > 1e-5)
        tot += bad.shape[0]
        if bad.shape[0]:
            desc.append((l, bad.shape[0], sorted(set((bad[:, 0] % 16).tolist())), sorted(set(bad[:, 1].tolist()))[:8]))
    e = (go["da_mid"].double() - ref["da_mid"].double()).abs().cpu()
    bad = torch.nonzero(e > 1e-5)
    print("run", it, "bad dy elements", tot, desc, "| da_mid bad", bad.shape[0], sorted(set((bad[:, 0] % 16).tolist())), sorted(set(bad[:, 1].tolist()))[:8])
import struct
go = runs[-1]
for l in (7, 4, 1):
    a, b = ref["dy"][l].cpu(), go["dy"][l].cpu()
    e = (a.double() - b.double()).abs()
    bad = torch.nonzero(e > 1e-5)
    print("layer", l)
    for r_, c_ in bad[:10].tolist():
        bv = float(b[r_, c_]); av = float(a[r_, c_])
        print(f"   row {r_} (tile {r_//128} wave {(r_%128)//16} m {r_%16}) col {c_}: f32 {av:.6g} x6 {bv:.6g} bits {struct.unpack('<I', struct.pack('<f', bv))[0]:#010x}  neighbours x6 {b[r_, c_-1]:.4g} {b[r_, c_+1]:.4g} f32 {a[r_, c_-1]:.4g} {a[r_, c_+1]:.4g}")
