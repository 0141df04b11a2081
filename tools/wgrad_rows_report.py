"""Weight-gradient kernel of the reduced-precision training mode: fp32 rows (rounded in-kernel) against bf16 rows.
python tools/wgrad_rows_report.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import train_graph
pkg.load_library()
dev = torch.device("cuda", 0)
train_graph._WGRAD_MODE = 3
n = 524288
for n_out, k_in in ((256, 256), (128, 256), (256, 104)):
    dy, x = torch.randn(n, n_out, device=dev), torch.randn(n, k_in, device=dev)
    for name, d, xx in (("fp32 rows", dy, x), ("bf16 rows", dy.bfloat16(), x.bfloat16()), ("bf16 dY, fp32 X", dy.bfloat16(), x)):
        dw, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
        for _ in range(3):
            train_graph._wgrad_multi([(d, xx)], n_out, k_in, dw, 0, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            train_graph._wgrad_multi([(d, xx)], n_out, k_in, dw, 0, db)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gb = n * (n_out * d.element_size() + k_in * xx.element_size()) / 1e9
        print("%3d x %3d over %d points, %-16s %7.1f us  %5.2f GB  %5.2f TB/s" % (n_out, k_in, n, name, us, gb, gb / us * 1e3))
