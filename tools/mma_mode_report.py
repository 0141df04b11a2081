#!/usr/bin/env python3
"""GPU: max-abs error of the eval get_outputs vs the CPU oracle for each MMA mode (f32, bf16x6, bf16x3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reflect_sampling_nerf_amd as pkg
from oracle import cpu_ref
from tests.helpers import max_abs
dev = torch.device("cuda:0")
torch.manual_seed(12)
cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=64, num_importance_samples=64, num_reflect_coarse_samples=32,
                                        num_reflect_importance_samples=32)
model = cfg.setup(scene_box=None, num_train_data=1)
with torch.no_grad():
    model.field.field_output_density.net.bias += 2.0
P = {k: v.detach().clone() for k, v in model.field.state_dict().items()}
model.to(dev).eval()
R = 96
o, d, pa = cpu_ref.synthetic_rays(R, seed=62)
nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev), fars=fars.to(dev))
with torch.no_grad():
    ref = cpu_ref.get_outputs(P, cpu_ref.FieldSpec(), cpu_ref.ModelSpec(64, 64, 32, 32), o, d, pa, nears, fars)
    ref64 = None
for mode in ("f32", "bf16x6", "bf16x3"):
    model.field.set_mma_mode(mode)
    out = model(rb)
    keys = ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine", "accumulation_fine", "weights_fine",
            "diff", "tint", "roughness", "pred_normals_fine", "n_dot_d_fine")
    print(mode, "mask flips", int((out["mask"].cpu() != ref["mask"]).sum()), " ".join(f"{k}={max_abs(out[k].cpu(), ref[k]):.2e}" for k in keys))
