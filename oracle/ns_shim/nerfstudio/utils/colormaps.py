"""Oracle shim: names only; used by the (broken) eval-image path, model.py:440-455."""


def apply_colormap(*args, **kwargs):  # pragma: no cover
    raise RuntimeError("colormaps are not available in the oracle shim")


def apply_depth_colormap(*args, **kwargs):  # pragma: no cover
    raise RuntimeError("colormaps are not available in the oracle shim")
