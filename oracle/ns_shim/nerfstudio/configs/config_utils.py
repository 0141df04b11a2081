"""Oracle shim: to_immutable_dict (reflect_sampling_nerf_model.py:19,56,73)."""
from dataclasses import field
from typing import Any, Dict


def to_immutable_dict(d: Dict[str, Any]):
    """Dataclass default that yields a fresh dict per instance."""
    return field(default_factory=lambda: dict(d))
