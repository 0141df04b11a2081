"""Builds librsn_hip.so (the C-ABI library of include/rsn.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).

Every .hip source is its own translation unit (no relocatable device code: the kernels share only headers), compiled
to build/<name>.o in parallel and re-compiled only when it, a header or the flags changed; one link step makes the .so.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(REPO, "build", "obj")
LIB_PATH = os.path.join(PKG_DIR, "librsn_hip.so")
SOURCES = ["rsn_pack.hip", "rsn_field.hip", "rsn_field_split.hip", "rsn_field_bf16.hip", "rsn_field_bf16_train.hip", "rsn_field_x6_train.hip", "rsn_field_bwd.hip", "rsn_wgrad.hip", "rsn_render.hip", "rsn_train_ops.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17"]
# Per-file flags on top of FLAGS.  -amdgpu-mfma-vgpr-form: MFMA accumulators in architected VGPRs instead of AGPRs.  The
# field kernels' layer epilogues read every accumulator (ReLU, mask bits, LDS hand-off, saved rows) and re-load it with
# the next bias: from AGPRs that is a v_accvgpr_read / _write per value (1,035 + 290 static copies in
# rsn_field_kernel<8,true,0>, 512 registers, 132 B of scratch); in VGPR form 67 + 66, 357 registers, no scratch, and the
# step's forward / backward sweeps run 1.4 % / 1.5 % faster (profiles/r03_vgpr_form.txt).  NOT for rsn_wgrad.hip: its
# 256-accumulator kernels need the AGPR half of the file for them (605 spills in VGPR form).
_MFMA_VGPR = ("-mllvm", "-amdgpu-mfma-vgpr-form")
# -pragma-unroll-threshold: the split-bf16 ring GEMMs are fully unrolled loops of up to 27 groups x 16 pieces (static register indices
# for the activations); hipcc prices them above its default 16 K budget, unrolls them late and partially, and the activation /
# accumulator arrays then live in scratch (832 B per lane).  With the budget raised: no scratch in the backward kernels.
_UNROLL_BUDGET = ("-mllvm", "-pragma-unroll-threshold=131072")
SOURCE_FLAGS = {"rsn_field.hip": _MFMA_VGPR, "rsn_field_bwd.hip": _MFMA_VGPR, "rsn_field_x6_train.hip": _UNROLL_BUDGET, "rsn_field_f32_ring.hip": _UNROLL_BUDGET}


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(REPO, "include", "rsn.h")]


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "rsn.h"), os.path.abspath(__file__)]  # this file: the flags
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(hipcc, src, obj, flags, verbose):
    cmd = [hipcc, *flags, "-I", os.path.join(REPO, "include"), "-I", CSRC, "-c", src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (os.path.basename(src), res.stdout, res.stderr))
    os.replace(obj + ".tmp", obj)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), lib_path: str = LIB_PATH,
                  extra_sources=(), source_flags=None) -> str:
    """Compile the HIP sources into librsn_hip.so; returns its path.  `force` recompiles every object.
    extra_flags / lib_path / extra_sources (absolute paths of further .hip files) / source_flags (per-file flags replacing
    SOURCE_FLAGS): diagnostic variants (tools/) build a second library beside the product one."""
    if source_flags is None:
        source_flags = SOURCE_FLAGS
    if not force and not extra_flags and lib_path == LIB_PATH and source_flags is SOURCE_FLAGS and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build librsn_hip.so")
    flags = [*FLAGS, *extra_flags]
    tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:10]
    obj_dir = os.path.join(OBJ_DIR, tag)
    os.makedirs(obj_dir, exist_ok=True)
    hdr = _headers()
    jobs, objs = [], []
    for s in [*SOURCES, *extra_sources]:
        src = s if os.path.isabs(s) else os.path.join(CSRC, s)
        sflags = [*flags, *source_flags.get(os.path.basename(s), ())]
        stag = "" if len(sflags) == len(flags) else "." + hashlib.sha256(" ".join(sflags).encode()).hexdigest()[:8]
        obj = os.path.join(obj_dir, os.path.basename(s).replace(".hip", stag + ".o"))
        stamp = obj + ".sha"
        want = _digest([src, *hdr], " ".join(sflags))
        have = open(stamp).read() if os.path.exists(stamp) and os.path.exists(obj) else ""
        objs.append(obj)
        if force or have != want:
            jobs.append((src, obj, stamp, want, sflags))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        futs = {ex.submit(_compile_one, hipcc, src, obj, sflags, verbose): (stamp, want) for src, obj, stamp, want, sflags in jobs}
        for f in concurrent.futures.as_completed(futs):
            f.result()
            stamp, want = futs[f]
            with open(stamp, "w") as fh:
                fh.write(want)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    import sys

    print(build_library(force="--force" in sys.argv, verbose=True))
