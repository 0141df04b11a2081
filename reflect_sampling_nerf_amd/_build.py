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
SOURCES = ["rsn_pack.hip", "rsn_field.hip", "rsn_field_bf16.hip", "rsn_field_bwd.hip", "rsn_wgrad.hip", "rsn_render.hip", "rsn_train_ops.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17"]


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
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "rsn.h")]
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
                  extra_sources=()) -> str:
    """Compile the HIP sources into librsn_hip.so; returns its path.  `force` recompiles every object.
    extra_flags / lib_path / extra_sources (absolute paths of further .hip files): diagnostic variants (tools/) build a
    second library beside the product one."""
    if not force and not extra_flags and lib_path == LIB_PATH and not _stale():
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
        obj = os.path.join(obj_dir, os.path.basename(s).replace(".hip", ".o"))
        stamp = obj + ".sha"
        want = _digest([src, *hdr], " ".join(flags))
        have = open(stamp).read() if os.path.exists(stamp) and os.path.exists(obj) else ""
        objs.append(obj)
        if force or have != want:
            jobs.append((src, obj, stamp, want))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        futs = {ex.submit(_compile_one, hipcc, src, obj, flags, verbose): (stamp, want) for src, obj, stamp, want in jobs}
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
