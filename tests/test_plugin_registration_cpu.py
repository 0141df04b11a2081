"""reflect_sampling_nerf_amd/reflect_sampling_nerf_config.py (the `nerfstudio.method_configs` entry point, mirror of the reference's
config.py:27-63) can only be imported where nerfstudio is: it never ran in this image.  This test runs it -- and the
"nerfstudio is importable" branch of nerfstudio_compat.py -- in a child interpreter on top of test-only stand-ins: the oracle's
class-shaped nerfstudio shim (oracle/ns_shim: RayBundle, Field, Model / ModelConfig) plus inert dataclasses for the trainer-side names
the config module touches (TrainerConfig, pipeline / datamanager / dataparser / optimiser / scheduler configs, MethodSpecification).
What is checked: the module executes, registers the method under the reference's name with the reference's trainer settings
(config.py:28-61, values quoted below), points at THIS package's Model, the Model builds on the foreign base class, and the
pipeline's loss warm-up hook is the reference's (pipeline.py:79-91).  It is a rehearsal of the plumbing, not nerfstudio itself.
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import dataclasses, json, sys, types
sys.path.insert(0, REPO); sys.path.insert(0, SHIM)
import nerfstudio
from nerfstudio.models import base_model


def module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent not in sys.modules:
        module(parent)
    setattr(sys.modules[parent], leaf, m)
    return m


def cfg(name, **fields):  # an inert config dataclass that records what it is given
    ns = {"__annotations__": {k: object for k in fields}}
    ns.update(fields)
    return dataclasses.make_dataclass(name, [(k, object, dataclasses.field(default=v)) for k, v in fields.items()])


class VanillaPipeline:
    def __init__(self, model=None):
        self.model = model

    def get_train_loss_dict(self, step):
        return ("super", step)


class VanillaDataManager:
    pass


module("nerfstudio.model_components.scene_colliders", NearFarCollider=base_model.NearFarCollider)
module("nerfstudio.configs.base_config", ViewerConfig=cfg("ViewerConfig", num_rays_per_chunk=None))
module("nerfstudio.data.datamanagers.base_datamanager", VanillaDataManager=VanillaDataManager,
       VanillaDataManagerConfig=cfg("VanillaDataManagerConfig", _target=None, dataparser=None, train_num_rays_per_batch=None,
                                    eval_num_rays_per_batch=None))
module("nerfstudio.data.dataparsers.blender_dataparser", BlenderDataParserConfig=cfg("BlenderDataParserConfig"))
module("nerfstudio.engine.optimizers", AdamOptimizerConfig=cfg("AdamOptimizerConfig", lr=None, eps=None),
       RAdamOptimizerConfig=cfg("RAdamOptimizerConfig", lr=None, eps=None))
module("nerfstudio.engine.schedulers", ExponentialDecaySchedulerConfig=cfg("ExponentialDecaySchedulerConfig", lr_final=None, max_steps=None))
module("nerfstudio.engine.trainer", TrainerConfig=cfg("TrainerConfig", method_name=None, pipeline=None, optimizers=None,
                                                       max_num_iterations=None, steps_per_eval_batch=None, steps_per_save=None,
                                                       mixed_precision=None, viewer=None, vis=None))
module("nerfstudio.pipelines.base_pipeline", VanillaPipeline=VanillaPipeline,
       VanillaPipelineConfig=cfg("VanillaPipelineConfig", _target=None, datamanager=None, model=None))
module("nerfstudio.plugins.types", MethodSpecification=cfg("MethodSpecification", config=None, description=None))

import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import nerfstudio_compat as nc
from reflect_sampling_nerf_amd import reflect_sampling_nerf_config as rc

spec = rc.reflect_sampling_nerf
t = spec.config
mc = t.pipeline.model
model = mc.setup(scene_box=None, num_train_data=3)
pipe = t.pipeline._target(model=model)
model.config.loss_coefficients = dict(model.config.loss_coefficients)
before = dict(model.config.loss_coefficients)
r0 = pipe.get_train_loss_dict(0)
warm = dict(model.config.loss_coefficients)
r1 = pipe.get_train_loss_dict(50)
after = dict(model.config.loss_coefficients)
print(json.dumps({
    "have_nerfstudio": nc.HAVE_NERFSTUDIO, "model_base": type(model).__mro__[1].__module__, "field_base": type(model.field).__mro__[1].__module__,
    "method_name": t.method_name, "max_num_iterations": t.max_num_iterations, "steps_per_eval_batch": t.steps_per_eval_batch,
    "steps_per_save": t.steps_per_save, "mixed_precision": t.mixed_precision, "vis": t.vis, "viewer_chunk": t.viewer.num_rays_per_chunk,
    "rays": [t.pipeline.datamanager.train_num_rays_per_batch, t.pipeline.datamanager.eval_num_rays_per_batch, mc.eval_num_rays_per_chunk],
    "dataparser": type(t.pipeline.datamanager.dataparser).__name__, "groups": sorted(t.optimizers),
    "fields_opt": [type(t.optimizers["fields"]["optimizer"]).__name__, t.optimizers["fields"]["optimizer"].lr, t.optimizers["fields"]["optimizer"].eps,
                   t.optimizers["fields"]["scheduler"].lr_final, t.optimizers["fields"]["scheduler"].max_steps],
    "model_target": mc._target.__module__ + "." + mc._target.__name__, "model_class": type(model).__name__,
    "param_groups": sorted(model.get_param_groups()), "n_params": sum(p.numel() for p in model.get_param_groups()["fields"]),
    "collider": type(model.collider).__module__, "super_called": [list(r0), list(r1)],
    "warmup_zeroed": sorted(k for k in before if warm[k] == 0.0 and before[k] != 0.0), "restored": after == before,
    "num_train_data": model.num_train_data,
}))
'''


def test_method_registration_runs_on_top_of_a_nerfstudio_stand_in():
    code = "REPO = %r\nSHIM = %r\n" % (REPO, os.path.join(REPO, "oracle", "ns_shim")) + CHILD
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    # the "nerfstudio is importable" branch: this package's Model / Field derive from the foreign base classes
    assert r["have_nerfstudio"] is True
    assert r["model_base"] == "nerfstudio.models.base_model" and r["field_base"] == "nerfstudio.fields.base_field"
    assert r["collider"] == "nerfstudio.models.base_model" and r["num_train_data"] == 3
    # reference config.py:28-61 (quoted): method name, 100000 iterations, eval batch every 100 steps, save every 1000, 1 << 10 rays per
    # batch / chunk, blender dataparser, RAdam(lr 1e-3, eps 1e-15) with exponential decay to 1e-4 over 50000 steps on "fields", viewer
    assert r["method_name"] == "reflect-sampling-nerf" and r["max_num_iterations"] == 100000
    assert r["steps_per_eval_batch"] == 100 and r["steps_per_save"] == 1000 and r["vis"] == "viewer"
    assert r["rays"] == [1024, 1024, 1024] and r["viewer_chunk"] == 1024 and r["dataparser"] == "BlenderDataParserConfig"
    assert r["groups"] == ["camera_opt", "fields", "proposal_networks"]
    assert r["fields_opt"] == ["RAdamOptimizerConfig", 1e-3, 1e-15, 1e-4, 50000]
    assert r["mixed_precision"] is False  # the one deliberate departure: the HIP path computes in fp32 (reference: fp16 autocast)
    # this package's Model behind the reference's method name; one parameter group, as the reference returns (model.py:134-139)
    assert r["model_target"] == "reflect_sampling_nerf_amd.reflect_sampling_nerf_model.ReflectSamplingNeRFModel"
    assert r["model_class"] == "ReflectSamplingNeRFModel" and r["param_groups"] == ["fields"] and r["n_params"] > 500000
    # pipeline.py:79-91: the normal / orientation coefficients are zero for the first 50 steps, then the configured ones
    assert r["super_called"] == [["super", 0], ["super", 50]]
    assert r["warmup_zeroed"] == ["orientation_loss_coarse", "orientation_loss_fine", "predicted_normal_loss_coarse", "predicted_normal_loss_fine"]
    assert r["restored"] is True
