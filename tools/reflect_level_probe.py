"""Why do the reflect levels of the training step run ~50 % slower per point than the primary levels?  Times
evaluate_frustums_train (no normals) over 2500 x 64 samples under four conditions."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import _abi, ops
from reflect_sampling_nerf_amd.synthetic import synthetic_rays
pkg.load_library()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1).to(dev).train()
fld = model.field
def run(name, R, S, spacing, tan, near, far, n_dev=None, Rmax=None, want_normals=False):
    Rm = Rmax or R
    o, d, pa = synthetic_rays(Rm, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(Rm).to(dev)
    nears, fars = torch.full((Rm,), near, device=dev), torch.full((Rm,), far, device=dev)
    sb, eb = ops.sample_spaced(Rm, None, S, spacing, tan, nears, fars, None)
    nd = None if n_dev is None else torch.tensor([n_dev], dtype=torch.int32, device=dev)
    f = lambda: fld.evaluate_frustums_train(o, d, pa, eb, n_dev=nd, want_normals=want_normals)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    pts = (n_dev or R) * S
    print("%-66s %7.3f ms  %6.2f ns/point" % (name, ms, ms * 1e6 / pts))
U, RC = _abi.RSN_SPACING_UNIFORM, _abi.RSN_SPACING_RECIPROCAL
run("4096 x 128 uniform [2,6], with normals (primary level)", 4096, 128, U, 1.0, 2.0, 6.0, want_normals=True)
run("4096 x 128 uniform [2,6], no normals", 4096, 128, U, 1.0, 2.0, 6.0)
run("2500 x 64 uniform [2,6], no normals", 2500, 64, U, 1.0, 2.0, 6.0)
run("2500 x 64 reciprocal [0,256], no normals", 2500, 64, RC, 0.25, 0.0, 256.0)
run("2500 of 4096 x 64 (device count) uniform [2,6]", 2500, 64, U, 1.0, 2.0, 6.0, n_dev=2500, Rmax=4096)
run("2500 of 4096 x 64 (device count) reciprocal [0,256]", 2500, 64, RC, 0.25, 0.0, 256.0, n_dev=2500, Rmax=4096)
run("8192 x 64 reciprocal [0,256], no normals", 8192, 64, RC, 0.25, 0.0, 256.0)

# ---- the same launch on the REAL reflect rays / bins of a training step (bench.py's model: density bias +2)
print("-- real reflect level of a training step")
torch.manual_seed(0)
model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1)
with torch.no_grad():
    model.field.field_output_density.net.bias += 2.0
model.to(dev).train()
fld = model.field
R = 4096
o, d, pa = synthetic_rays(R, seed=0)
rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(R, 1).to(dev),
                   nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
model._keep_train_state = True
out = model(rb)
st = model._train_state
M = st["M"]
o2, d2, pa2 = st["rays2"]
eb = st["eb_rc"]
print("M =", M, "pixel_area2 quantiles", torch.quantile(pa2[:M], torch.tensor([0.01, 0.5, 0.99], device=dev)).tolist())
def timeit(name, f, pts):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-66s %7.3f ms  %6.2f ns/point" % (name, ms, ms * 1e6 / pts))
oM, dM, pM, eM = o2[:M].contiguous(), d2[:M].contiguous(), pa2[:M].contiguous(), eb.contiguous()
timeit("real reflect-coarse level (M x 64)", lambda: fld.evaluate_frustums_train(oM, dM, pM, eM, want_normals=False), M * 64)
small = torch.full_like(pM, 1.5625e-6)
timeit("  the same rays and bins, pixel_area = (1/800)^2", lambda: fld.evaluate_frustums_train(oM, dM, small, eM, want_normals=False), M * 64)
o, d, pa = synthetic_rays(M, seed=0)
timeit("  synthetic camera rays, the reflect bins", lambda: fld.evaluate_frustums_train(o.to(dev), d.to(dev), pa.reshape(M).to(dev), eM, want_normals=False), M * 64)
