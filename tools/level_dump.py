#!/usr/bin/env python3
"""Runs one training-mode level on three shapes (37 x 32 with / without normals, 5 x 13) with whatever library RSN_LIBRARY names
(default: the product) and saves every output and saved buffer; tools/level_dump_cmp.py compares two such files bit for bit.  Used to
show that the exact-fp32 ring probe (tools/probes/rsn_field_f32_ring.hip) reproduces the slab kernel (profiles/r04_f32_ring.txt).
Usage: [RSN_LIBRARY=<.so>] python tools/level_dump.py out.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reflect_sampling_nerf_amd as pkg
from oracle import cpu_ref
dev = torch.device("cuda:0")
torch.manual_seed(0)
out_path = sys.argv[1]
res = {}
for (R, S, wn) in ((37, 32, True), (37, 32, False), (5, 13, True)):
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S, num_reflect_coarse_samples=16, num_reflect_importance_samples=16)
    torch.manual_seed(0)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    f = model.field
    o, d, pa = cpu_ref.synthetic_rays(R, seed=1)
    o, d, pa = o.to(dev), d.to(dev), pa.to(dev).reshape(-1)
    bins = (2.0 + 4.0 * torch.linspace(0, 1, S + 1)).repeat(R, 1).to(dev).contiguous()
    lv = f.evaluate_frustums_train(o, d, pa, bins, want_normals=wn)
    torch.cuda.synchronize()
    for k, v in lv.items():
        if k == "saved":
            for kk, vv in v.items():
                res[f"{R}x{S}/{wn}/saved.{kk}"] = vv.cpu()
        else:
            res[f"{R}x{S}/{wn}/{k}"] = v.cpu()
torch.save(res, out_path)
print("saved", len(res), "tensors to", out_path)
