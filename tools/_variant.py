"""Builds an experimental variant of librsn_hip.so (extra -D macros) beside the product library; tools only.
The variant lands in build/variants/ (git-ignored; it travels to the GPU box with the snapshot when built here)."""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def build_variant(defines, extra_sources=()):
    from reflect_sampling_nerf_amd import _build

    flags = ["-DRSN_DIAG_BUILD"] + ["-D" + d for d in defines]  # marks the library as never-the-product (rsn_common.h)
    tag = hashlib.sha256(" ".join([*flags, *extra_sources]).encode()).hexdigest()[:10]
    out = os.path.join(REPO, "build", "variants")
    os.makedirs(out, exist_ok=True)
    return _build.build_library(extra_flags=flags, lib_path=os.path.join(out, "librsn_variant_%s.so" % tag),
                                extra_sources=[os.path.join(REPO, s) for s in extra_sources])
