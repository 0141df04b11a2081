#!/usr/bin/env python3
"""GPU diagnostics: per-layer pre-activation gradients dY_l of the backward sweep vs autograd (single level)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import reflect_sampling_nerf_amd as pkg
from oracle import cpu_ref
from reflect_sampling_nerf_amd import train_graph as tg

dev = torch.device("cuda:0")
for layers, width in [(8, 64), (4, 64), (6, 64), (8, 128)]:
    torch.manual_seed(layers * 10 + width)
    fld = pkg.ReflectSamplingNeRFNerfField(base_mlp_num_layers=layers, base_mlp_layer_width=width)
    P = {k: v.detach().clone() for k, v in fld.state_dict().items()}
    fld.to(dev).train()
    fs = cpu_ref.FieldSpec(num_layers=layers, width=width)
    R, S = 16, 16
    o, d, pa = cpu_ref.synthetic_rays(R, seed=3)
    nears, fars = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    _, eb = cpu_ref.spaced_bins("uniform", 1.0, nears, fars, S, None)
    # oracle forward with retained pre-activations
    t0, t1 = eb[..., :-1], eb[..., 1:]
    mean, cov = cpu_ref.gaussian_blob(o, d, pa, t0, t1)
    mean, cov = cpu_ref.contract(mean, cov)
    enc = cpu_ref.ipe(fs, mean, torch.diagonal(cov, dim1=-2, dim2=-1))
    x, pre = enc, []
    for i in range(layers):
        if i in fs.skip and 0 < i < layers - 1:
            x = torch.cat([enc, x], dim=-1)
        z = F.linear(x, P[f"mlp_base.layers.{i}.weight"], P[f"mlp_base.layers.{i}.bias"]).requires_grad_(True)
        z.retain_grad(); pre.append(z)
        x = torch.relu(z)
    emb = x
    g = torch.Generator().manual_seed(1)
    g_col = torch.randn(R, S, 3, generator=g)
    g_sig = torch.randn(R, S, generator=g)
    raw = cpu_ref.head(P, "field_output_density", emb)
    sigma = F.softplus(raw + fs.density_bias)
    diff = torch.sigmoid(cpu_ref.head(P, "field_output_diff", emb))
    tint = torch.sigmoid(cpu_ref.head(P, "field_output_tint", emb))
    rr = cpu_ref.head(P, "field_output_roughness", emb)
    sh = cpu_ref.integrated_sh(d[:, None, :].expand(R, S, 3), F.softplus(rr))
    mid = cpu_ref.mid_color(P, fs, sh, emb)
    color = diff + tint * mid
    g_pn = torch.randn(R, S, 3, generator=g); g_nd = torch.randn(R, S, generator=g); g_rg = torch.randn(R, S, generator=g)
    pn = cpu_ref.pred_normals(P, emb)
    ndd = torch.sum(d[:, None, :].expand(R, S, 3) * pn, dim=-1)
    loss = (color * g_col).sum() + (sigma[..., 0] * g_sig).sum() + (pn * g_pn).sum() + (ndd * g_nd).sum() \
        + (torch.sigmoid(rr)[..., 0] * g_rg).sum()
    # pre[i] are leaves created by requires_grad_(True) on non-leaf? make graph: use autograd.grad wrt list
    grads = torch.autograd.grad(loss, pre, allow_unused=True)
    # HIP
    lv = fld.evaluate_frustums_train(o.to(dev), d.to(dev), pa.reshape(R).to(dev), eb.contiguous().to(dev), want_normals=True)
    for need_input in (False, True):
        gout = tg._field_backward(fld, (o.to(dev), d.to(dev), pa.reshape(R).to(dev)), eb.contiguous().to(dev), lv,
                                  {"sigma": g_sig.to(dev), "color": g_col.to(dev), "pred_normals": g_pn.to(dev),
                                   "n_dot_d": g_nd.to(dev), "roughness": g_rg.to(dev)}, need_input=need_input)
        torch.cuda.synchronize()
        print(f"=== L={layers} W={width} need_input={need_input}")
        for l in range(layers):
            a = gout["dy"][l].cpu().reshape(R, S, width); b = grads[l]
            sc = float(b.abs().max()) + 1e-20
            print(f"  dY[{l}] scale {sc:.3e} max-err/scale {float((a-b).abs().max())/sc:.3e}")
