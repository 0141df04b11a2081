"""reflect_sampling_nerf_amd -- MI355X-native implementation of the reflect-sampling-nerf hot path.

Importing the package does not touch the GPU; the HIP library (librsn_hip.so, built by
`__graft_entry__.build()`) is loaded on first use and its absence is a hard error.
"""
from ._abi import EXPORTED_SYMBOLS, RsnError, load_library  # noqa: F401
from .nerfstudio_compat import HAVE_NERFSTUDIO, RayBundle  # noqa: F401
from .reflect_sampling_nerf_components import (  # noqa: F401
    IntegratedSHEncoding,
    NeRFEncoding,
    PDFSampler,
    ReciprocalSampler,
    UniformSampler,
)
from .reflect_sampling_nerf_field import ReflectSamplingNeRFNerfField  # noqa: F401
from .reflect_sampling_nerf_model import ReflectSamplingNeRFModel, ReflectSamplingNeRFModelConfig  # noqa: F401

from .train_ops import FusedRAdam, exponential_decay_lr  # noqa: F401,E402

__all__ = [
    "ReflectSamplingNeRFModel", "ReflectSamplingNeRFModelConfig", "ReflectSamplingNeRFNerfField", "RayBundle",
    "ReciprocalSampler", "IntegratedSHEncoding", "NeRFEncoding", "UniformSampler", "PDFSampler", "load_library",
    "RsnError",
]
