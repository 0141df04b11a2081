"""Shared helpers for the tests (fixture loading, tolerant comparisons)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    groups = {}
    for k in z.files:
        if k == "meta":
            continue
        grp, key = k.split("/", 1)
        groups.setdefault(grp, {})[key] = torch.from_numpy(z[k])
    if meta.get("param_file"):  # parameters shared by several cases live in their own file
        pz = np.load(os.path.join(GOLDEN, meta["param_file"] + ".npz"), allow_pickle=False)
        groups["param"] = {k: torch.from_numpy(pz[k]) for k in pz.files}
    return meta, groups


def field_spec_from_meta(meta):
    from oracle.cpu_ref import FieldSpec

    return FieldSpec(num_layers=meta["layers"], width=meta["width"])


def model_spec_from_meta(meta):
    from oracle.cpu_ref import ModelSpec

    s = meta["samples"]
    return ModelSpec(num_coarse=s[0], num_fine=s[1], num_reflect_coarse=s[2], num_reflect_fine=s[3])


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
