"""Oracle shim: only the enum the config default names (reflect_sampling_nerf_model.py:22,73)."""
from enum import Enum


class TemporalDistortionKind(Enum):
    DNERF = "dnerf"
