"""Writes the eval outputs of one sampling level and every parameter gradient of one training step (two weight scales, fixed
seeds) to a .pt file; run it from two trees (the working tree and `git archive <commit>` built beside it) and compare the
files: kernel changes that must not change the arithmetic (address paths, register allocation) give bit-identical eval tensors.
    python tools/bitcmp.py out.pt"""
import os, sys, torch, hashlib
sys.path.insert(0, os.getcwd())
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import ops
from reflect_sampling_nerf_amd._abi import RSN_SPACING_UNIFORM
from reflect_sampling_nerf_amd.synthetic import synthetic_rays
pkg.load_library()
dev = torch.device("cuda:0")
R, S = 1024, 64
out = {}
for scale in (1.0, 3.0):
    torch.manual_seed(0)
    model = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S).setup(scene_box=None, num_train_data=1).to(dev)
    with torch.no_grad():
        for p in model.field.parameters():
            p.mul_(scale)
    o, d, pa = synthetic_rays(R, seed=1)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    model.eval()
    fld = model.field
    sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
    lv = fld.evaluate_frustums(o, d, pa, eb, full=True)
    for k in sorted(lv):
        if torch.is_tensor(lv[k]):
            out["eval%g_%s" % (scale, k)] = lv[k].float().cpu()
    # one training step's gradients
    model.train()
    from reflect_sampling_nerf_amd import parallel
    torch.manual_seed(1)
    img = torch.rand(R, 3, device=dev)
    model.zero_grad(set_to_none=True)
    class RB: pass
    from reflect_sampling_nerf_amd.nerfstudio_compat import RayBundle
    rb = RayBundle(origins=o, directions=d, pixel_area=pa.reshape(R, 1), nears=nears.reshape(R, 1), fars=fars.reshape(R, 1))
    torch.manual_seed(2)
    outs = model.get_outputs(rb)
    loss = sum(model.get_loss_dict(outs, {"image": img}).values())
    loss.backward()
    for n, p in model.field.named_parameters():
        if p.grad is not None:
            out["grad%g_%s" % (scale, n)] = p.grad.float().cpu()
torch.save(out, sys.argv[1])
print("saved", len(out))
