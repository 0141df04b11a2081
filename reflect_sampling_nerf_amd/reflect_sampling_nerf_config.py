"""Nerfstudio method registration for the MI355X path (mirror of the reference's
reflect_sampling_nerf_config.py:27-63; entry-point group `nerfstudio.method_configs`, method name
`reflect-sampling-nerf`, see pyproject.toml).

Importable only where `nerfstudio` is installed (it is not in the build image nor on the GPU box, and there is no
network): everything here is plain Nerfstudio configuration -- trainer, datamanager, dataparser, optimiser groups --
which is out of the hot-path scope; only the `_target` classes differ from the reference.
"""
from __future__ import annotations

from .nerfstudio_compat import HAVE_NERFSTUDIO

if not HAVE_NERFSTUDIO:  # pragma: no cover - depends on the environment
    raise ImportError("reflect_sampling_nerf_amd.reflect_sampling_nerf_config needs nerfstudio (ns-train); the hot path "
                      "itself (ReflectSamplingNeRFModel / Field) works without it")

from dataclasses import dataclass, field  # noqa: E402
from typing import Type  # noqa: E402

from nerfstudio.configs.base_config import ViewerConfig  # noqa: E402
from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager, VanillaDataManagerConfig  # noqa: E402
from nerfstudio.data.dataparsers.blender_dataparser import BlenderDataParserConfig  # noqa: E402
from nerfstudio.engine.optimizers import AdamOptimizerConfig, RAdamOptimizerConfig  # noqa: E402
from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig  # noqa: E402
from nerfstudio.engine.trainer import TrainerConfig  # noqa: E402
from nerfstudio.pipelines.base_pipeline import VanillaPipeline, VanillaPipelineConfig  # noqa: E402
from nerfstudio.plugins.types import MethodSpecification  # noqa: E402

from .parallel import apply_loss_warmup  # noqa: E402
from .reflect_sampling_nerf_model import ReflectSamplingNeRFModelConfig  # noqa: E402


@dataclass
class ReflectSamplingNeRFDataManagerConfig(VanillaDataManagerConfig):
    _target: Type = field(default_factory=lambda: VanillaDataManager)


class ReflectSamplingNeRFPipeline(VanillaPipeline):
    """reference pipeline.py:79-91: the four normal/orientation coefficients are zero for the first 50 steps.
    (DDP wrapping is Nerfstudio's own; its gradient average equals parallel.FlatGradAllReduce.)"""

    def get_train_loss_dict(self, step: int):
        apply_loss_warmup(self.model, step)
        return super().get_train_loss_dict(step)


@dataclass
class ReflectSamplingNeRFPipelineConfig(VanillaPipelineConfig):
    _target: Type = field(default_factory=lambda: ReflectSamplingNeRFPipeline)


def _group(optimizer_cls, lr_final: float, max_steps: int):
    """One optimiser group: lr 1e-3 / eps 1e-15 with exponential decay to lr_final (reference config.py:44-59)."""
    return {"optimizer": optimizer_cls(lr=1e-3, eps=1e-15),
            "scheduler": ExponentialDecaySchedulerConfig(lr_final=lr_final, max_steps=max_steps)}


def method_config() -> TrainerConfig:
    """The reference's trainer settings (config.py:28-61) around the MI355X Model; built in steps rather than as one
    literal so that each departure from the reference is visible: mixed precision off, this package's targets."""
    rays = 1 << 10
    data = ReflectSamplingNeRFDataManagerConfig(dataparser=BlenderDataParserConfig(), train_num_rays_per_batch=rays,
                                                eval_num_rays_per_batch=rays)
    pipeline = ReflectSamplingNeRFPipelineConfig(datamanager=data,
                                                 model=ReflectSamplingNeRFModelConfig(eval_num_rays_per_chunk=rays))
    groups = {
        "fields": _group(RAdamOptimizerConfig, 1e-4, 50000),  # the only group get_param_groups() returns
        "proposal_networks": _group(AdamOptimizerConfig, 1e-4, 200000),
        "camera_opt": _group(AdamOptimizerConfig, 1e-4, 5000),
    }
    return TrainerConfig(method_name="reflect-sampling-nerf", pipeline=pipeline, optimizers=groups,
                         max_num_iterations=100000, steps_per_eval_batch=100, steps_per_save=1000,
                         mixed_precision=False,  # the HIP path computes in fp32 (the reference enables fp16 autocast)
                         viewer=ViewerConfig(num_rays_per_chunk=rays), vis="viewer")


reflect_sampling_nerf = MethodSpecification(
    config=method_config(),
    description="reflect-sampling-nerf on MI355X (HIP kernels behind the Nerfstudio Model/Field surface).",
)
