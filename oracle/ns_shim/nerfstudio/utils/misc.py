"""Oracle shim: misc.scale_dict (reflect_sampling_nerf_model.py:429)."""
from typing import Any, Dict


def scale_dict(dictionary: Dict[Any, Any], coefficients: Dict[str, float]) -> Dict[Any, Any]:
    """Scale dictionary entries in place by the coefficient of the same key, when there is one."""
    for key in dictionary:
        if key in coefficients:
            dictionary[key] *= coefficients[key]
    return dictionary
