"""Import shim for `nerfacc` (test infrastructure only).

reflect_sampling_nerf_components.py:7 imports OccGridEstimator and never uses it.
"""


class OccGridEstimator:  # pragma: no cover - never instantiated by the reference
    def __init__(self, *args, **kwargs):
        raise RuntimeError("nerfacc is not available; the reference never instantiates this")
