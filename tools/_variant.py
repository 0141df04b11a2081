"""Builds an experimental variant of librsn_hip.so (extra -D macros) into a temporary directory; tools only."""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def build_variant(defines):
    from reflect_sampling_nerf_amd import _build

    tmp = tempfile.mkdtemp(prefix="rsn_variant_")
    lib = os.path.join(tmp, "librsn_variant.so")
    cmd = ["hipcc", *_build.FLAGS, *["-D" + d for d in defines], "-I", os.path.join(REPO, "include"), "-I", _build.CSRC,
           *[os.path.join(_build.CSRC, s) for s in _build.SOURCES], "-o", lib]
    subprocess.run(cmd, check=True)
    return lib
