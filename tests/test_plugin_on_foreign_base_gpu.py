"""The drop-in claim on the GPU with FOREIGN base classes: when nerfstudio is importable, this package's Model and Field derive from
nerfstudio's Model / Field and take nerfstudio's RayBundle (nerfstudio_compat.py).  nerfstudio cannot be installed here, so a child
interpreter puts the oracle's class-shaped shim (oracle/ns_shim: test infrastructure, RayBundle with stride-0 broadcasting, the base
Model's forward -> collider -> get_outputs, the chunked camera-ray-bundle path) on the path and runs eval, the chunked image path,
the loss and one training step through the HIP library; the parent runs the same inputs on the package's own stand-ins.  Same kernels,
same seeds: outputs, image and loss must be bit-identical (the updated parameters agree to the order of the weight-gradient atomics) -- what
is tested is the plumbing around the kernels.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import hashlib, json, sys
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import nerfstudio_compat as nc
from reflect_sampling_nerf_amd.parallel import train_step

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=16, num_importance_samples=16, num_reflect_coarse_samples=8,
                                        num_reflect_importance_samples=8, base_mlp_num_layers=8, base_mlp_layer_width=128,
                                        eval_num_rays_per_chunk=64)
model = cfg.setup(scene_box=None, num_train_data=1)
with torch.no_grad():
    model.field.field_output_density.net.bias += 2.0
model.to(dev).eval()
g = torch.Generator().manual_seed(1)
R = 150
o = torch.randn(R, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, -4.0])
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
pa = torch.full((R, 1), (1.0 / 800) ** 2)
rb = nc.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev))  # nears / fars come from the collider
out = model(rb)
img = model.get_outputs_for_camera_ray_bundle(nc.RayBundle(origins=o.to(dev).reshape(10, 15, 3), directions=d.to(dev).reshape(10, 15, 3),
                                                           pixel_area=pa.to(dev).reshape(10, 15, 1)))
batch = {"image": torch.rand(R, 3, generator=g).to(dev)}
model.train()
opt = pkg.FusedRAdam(model.get_param_groups()["fields"], lr=1e-3, eps=1e-15)
torch.manual_seed(7)
loss = float(train_step(model, rb, batch, opt, None, 100))
torch.cuda.synchronize()


def digest(t):
    return hashlib.sha256(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


print(json.dumps({
    "have_nerfstudio": nc.HAVE_NERFSTUDIO, "model_base": type(model).__mro__[1].__module__, "ray_bundle": type(rb).__module__,
    "keys": sorted(k for k in out.keys()), "eval": {k: digest(out[k]) for k in ("mid_rgb_coarse", "mid_rgb_fine", "mid_reflect_fine", "accumulation_fine")},
    "image_shape": list(img["mid_rgb_fine"].shape), "image": digest(img["mid_rgb_fine"]), "loss": loss,
    "params": [float(v) for v in (lambda q: (q.sum(), q.norm(), q.abs().max()))(
        torch.cat([p.detach().double().reshape(-1) for p in model.get_param_groups()["fields"]]))],
}))
'''


def _run(with_shim: bool):
    paths = [REPO] + ([os.path.join(REPO, "oracle", "ns_shim")] if with_shim else [])
    pre = "import sys, types\n" + "".join("sys.path.insert(0, %r)\n" % p for p in paths)
    if with_shim:  # the one module of the plugin surface the oracle's shim keeps elsewhere (its collider lives beside the base Model)
        pre += ("from nerfstudio.models import base_model as _bm\n"
                "_m = types.ModuleType('nerfstudio.model_components.scene_colliders'); _m.NearFarCollider = _bm.NearFarCollider\n"
                "sys.modules[_m.__name__] = _m\n"
                # ... and nerfstudio's RayBundle.get_row_major_sliced_ray_bundle (the shim restates only what the REFERENCE calls)
                "from nerfstudio.cameras import rays as _r\n"
                "def _sl(self, a, b):\n"
                "    kw = {n: v.reshape(-1, v.shape[-1])[a:b] for n, v in self._tensor_items()}\n"
                "    return _r.RayBundle(metadata=self.metadata, **kw)\n"
                "_r.RayBundle.get_row_major_sliced_ray_bundle = _sl\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", pre + BODY], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_model_on_foreign_base_classes_equals_the_stand_in_path():
    assert torch.cuda.is_available()
    own, foreign = _run(False), _run(True)
    assert own["have_nerfstudio"] is False and foreign["have_nerfstudio"] is True
    assert foreign["model_base"] == "nerfstudio.models.base_model" and foreign["ray_bundle"] == "nerfstudio.cameras.rays"
    assert own["keys"] == foreign["keys"] and "mid_reflect_fine" in own["keys"]
    assert own["eval"] == foreign["eval"]                      # eval get_outputs through the foreign forward / collider
    assert own["image_shape"] == [10, 15, 3] == foreign["image_shape"] and own["image"] == foreign["image"]  # chunked image path
    assert own["loss"] == foreign["loss"]  # one whole training step: the loss bit for bit, the updated parameters to the order of the
    for a, b in zip(own["params"], foreign["params"]):  # weight-gradient atomics (sum, norm, max of all parameters)
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (own["params"], foreign["params"])
