#!/usr/bin/env python3
"""Diagnostics (GPU box): training-mode forward/backward of the HIP path vs oracle autograd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reflect_sampling_nerf_amd as pkg
from oracle import cpu_ref
from tests.test_gpu_parity import _loss_from_outputs
from tests.helpers import max_abs

dev = torch.device("cuda:0")
for layers, width, samples, R, bias in [(8, 64, (16, 16, 8, 8), 48, 2.0), (8, 128, (32, 24, 16, 12), 37, 1.5),
                                        (4, 64, (16, 16, 8, 8), 40, 2.0)]:
    seed = layers * 7 + width
    torch.manual_seed(seed)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=samples[0], num_importance_samples=samples[1],
                                            num_reflect_coarse_samples=samples[2], num_reflect_importance_samples=samples[3],
                                            base_mlp_num_layers=layers, base_mlp_layer_width=width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += bias
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.field.state_dict().items()}
    model.to(dev).train()
    o, d, pa = cpu_ref.synthetic_rays(R, seed=seed + 50)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    fs, ms = cpu_ref.FieldSpec(num_layers=layers, width=width), cpu_ref.ModelSpec(*samples)
    g = torch.Generator().manual_seed(seed + 1)
    jit = {"coarse": torch.rand(R, samples[0] + 1, generator=g), "fine": torch.rand(R, samples[1] + 1, generator=g),
           "reflect_coarse": torch.rand(R, samples[2] + 1, generator=g), "reflect_fine": torch.rand(R, samples[3] + 1, generator=g)}
    tgt = {k: torch.rand(R, 3, generator=g) for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_coarse", "mid_reflect_fine")}
    ref = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit)
    _loss_from_outputs(ref, tgt).backward()
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev), nears=nears.to(dev), fars=fars.to(dev))
    jg = dict(jit, reflect_coarse=jit["reflect_coarse"][ref["mask"]], reflect_fine=jit["reflect_fine"][ref["mask"]])
    out = model._get_outputs_train(rb, jitter=jg)
    print(f"=== L={layers} W={width} S={samples} R={R} M={int(ref['mask'].sum())}")
    for k in sorted(ref.keys()):
        if k == "mask":
            continue
        e = (out[k].detach().cpu().float() - ref[k].detach().float()).abs()
        print(f"  {k:22s} max {float(e.max()):.3e} mean {float(e.mean()):.3e} q99 {float(e.flatten().quantile(0.99)):.3e}")
    if os.environ.get("SAME_NORMAL_TARGETS", "1") == "1":  # isolate the backward from the normals' conditioning
        out = dict(out)
        out["normals_coarse"] = ref["normals_coarse"].detach().to(dev)
        out["normals_fine"] = ref["normals_fine"].detach().to(dev)
    if os.environ.get("LOSS_TERMS") == "coarse":
        def _l(o_, t_):
            w = o_["weights_coarse"].detach()
            return (torch.mean((o_["mid_rgb_coarse"] - t_["mid_rgb_coarse"]) ** 2)
                    + 3e-3 * torch.sum(w * torch.sum((o_["normals_coarse"].detach() - o_["pred_normals_coarse"]) ** 2, dim=-1, keepdim=True))
                    + 1e-2 * torch.sum(w * torch.clamp(o_["n_dot_d_coarse"], min=0.0) ** 2))
        for p_ in P.values():
            p_.grad = None
        ref2 = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit)
        _l(ref2, tgt).backward()
        loss = _l(out, {k: v.to(dev) for k, v in tgt.items()})
    else:
        loss = _loss_from_outputs(out, {k: v.to(dev) for k, v in tgt.items()})
    loss.backward()
    for name, p in model.field.named_parameters():
        gr = P[name].grad
        if gr is None or p.grad is None:
            print(f"  grad {name:40s} ref {gr is not None} hip {p.grad is not None}")
            continue
        scale = float(gr.abs().max()) + 1e-20
        err = float((p.grad.cpu() - gr).abs().max())
        print(f"  grad {name:40s} scale {scale:.3e} max-err/scale {err/scale:.3e}")
