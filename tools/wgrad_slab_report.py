"""The slab-order weight-gradient PROBE (tools/probes/rsn_wgrad_slab.hip: operands in the field kernels' register order,
transposed through LDS; not part of librsn_hip.so) against the product kernel (row-major operands): correctness against an
fp64 product and time per launch.  Results and why it was not adopted: profiles/r03_wgrad_slab.txt.
    python tools/wgrad_slab_report.py [--define WS_DIAG_NO_FILL ...]"""
import ctypes as C
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import _abi, ops, train_graph
from reflect_sampling_nerf_amd._abi import check, ptr

from tools._variant import build_variant

defs = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = pkg.load_library(build_variant(defs, extra_sources=["tools/probes/rsn_wgrad_slab.hip"]))
lib.rsn_weight_grad_slab.restype = C.c_int
lib.rsn_weight_grad_slab.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)


def to_slab(m):
    n, f = m.shape
    npad, fpad = (n + 31) // 32 * 32, (f + 7) // 8 * 8
    p = torch.zeros(npad, fpad, device=m.device, dtype=m.dtype)
    p[:n, :f] = m
    return p.reshape(npad // 32, 32, fpad // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(npad // 32, fpad // 8, 64, 4)


def wgrad_slab(segs, n_out, k_in, dw, db, counts=None):
    ns = len(segs)
    npts = (C.c_int64 * ns)(*[s[2] for s in segs])
    dys = (C.c_void_p * ns)(*[s[0].data_ptr() for s in segs])
    xs = (C.c_void_p * ns)(*[s[1].data_ptr() for s in segs])
    ndev = (C.c_void_p * ns)(*[None if (counts is None or counts[i] is None) else counts[i][0].data_ptr() for i in range(ns)])
    per = (C.c_int32 * ns)(*[1 if (counts is None or counts[i] is None) else counts[i][1] for i in range(ns)])
    check(lib.rsn_weight_grad_slab(ns, npts, ndev, per, dys, n_out, xs, k_in, None, ptr(dw), dw.stride(0), ptr(db), ops._stream()))


torch.manual_seed(0)
train_graph._WGRAD_MODE = 0
for n_out, k_in, n in ((256, 256, 524288), (256, 104, 524288), (128, 256, 524288), (128, 40, 524288), (16, 256, 524288),
                       (8, 128, 524288), (256, 256, 160000 + 7)):
    dy, x = torch.randn(n, n_out, device=dev), torch.randn(n, k_in, device=dev)
    ref = (dy.double().t() @ x.double()).cpu()
    refb = dy.double().sum(0).cpu()
    sd, sx = to_slab(dy), to_slab(x)
    dw, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
    wgrad_slab([(sd, sx, n)], n_out, k_in, dw, db)
    e = float((dw.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    eb = float((db.double().cpu() - refb).abs().max()) / float(refb.abs().max())
    def timeit(f):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    t_slab = timeit(lambda: wgrad_slab([(sd, sx, n)], n_out, k_in, dw, db))
    ld = n_out if n_out > 32 else 16
    dyp = torch.zeros(n, max(ld, n_out), device=dev); dyp[:, :n_out] = dy
    t_row = timeit(lambda: train_graph._wgrad_multi([(dyp, x)], n_out, k_in, dw, 0, db))
    fl = 2.0 * n * n_out * k_in
    print("%3d x %3d over %6d points: slab %7.1f us (%5.1f TF)  row-major %7.1f us (%5.1f TF)   max err / max %.1e, bias %.1e" %
          (n_out, k_in, n, t_slab, fl / t_slab / 1e6, t_row, fl / t_row / 1e6, e, eb))
