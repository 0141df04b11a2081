#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference, which
never travels to the GPU box -- only the small .npz fixtures do).

How: `reflect_sampling_nerf.reflect_sampling_nerf_{model,field,components}` are imported
unmodified from /root/reference on top of the import shim in oracle/ns_shim/ (the shim
provides the names the reference imports from nerfstudio / jaxtyping / nerfacc /
torchmetrics; see oracle/README.md).  Nothing of the reference's text is copied: the
fixtures hold inputs, parameters and outputs only.

What is pinned: rows F1-F13 of SURVEY.md §8(a) (reference-owned arithmetic) exactly;
rows N1-N12 (nerfstudio-owned) only up to the shim's restatement -- PARITY UNPINNED there.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("RSN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ns_shim"))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, REPO)

from nerfstudio.cameras.rays import RayBundle  # noqa: E402  (shim)
from nerfstudio.field_components.encodings import NeRFEncoding  # noqa: E402  (shim)
from reflect_sampling_nerf.reflect_sampling_nerf_components import (  # noqa: E402  (REFERENCE)
    IntegratedSHEncoding,
    ReciprocalSampler,
)
from reflect_sampling_nerf.reflect_sampling_nerf_field import ReflectSamplingNeRFNerfField  # noqa: E402
from reflect_sampling_nerf.reflect_sampling_nerf_model import (  # noqa: E402  (REFERENCE)
    ReflectSamplingNeRFModelConfig,
)

from oracle.cpu_ref import synthetic_rays  # noqa: E402

OUT_DIR = os.path.join(REPO, "tests", "golden")


class RandLog:
    """Records every torch.rand draw (the samplers' stratified jitter) in call order."""

    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def logged(*a, **k):
            t = self._orig(*a, **k)
            self.draws.append(t.clone())
            return t

        torch.rand = logged
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


class BinLog:
    """Forward hooks on the model's four samplers: records the (spacing, euclidean) bin edges [n, S+1] each one returns
    (reference call sites model.py:148,182,292,317), so that another pipeline can be run on identical sample positions."""

    NAMES = {"sampler_uniform": "coarse", "sampler_pdf": "fine", "sampler_reciprocal": "reflect_coarse",
             "sampler_reflect_pdf": "reflect_fine"}

    def __init__(self, model):
        self.bins = {}
        self.handles = []
        for attr, name in self.NAMES.items():
            self.handles.append(getattr(model, attr).register_forward_hook(self._hook(name)))

    def _hook(self, name):
        def hook(_module, _inputs, rs):
            n, S = rs.frustums.starts.shape[0], rs.frustums.starts.shape[1]
            edges = lambda a, b: torch.cat([a[..., 0], b[..., -1:, 0]], dim=-1).expand(n, S + 1).detach().clone()  # noqa: E731
            self.bins[name + "_euclid"] = edges(rs.frustums.starts, rs.frustums.ends)
            self.bins[name + "_spacing"] = edges(rs.spacing_starts, rs.spacing_ends)
        return hook

    def close(self):
        for h in self.handles:
            h.remove()


def save_params(arrays, model, param_file):
    """Parameters go into the case file, or -- for the 2.5 MB networks several cases share -- once into
    tests/golden/<param_file>.npz (the case's meta names it)."""
    P = {k: v.detach().numpy().astype(np.float32) for k, v in model.field.state_dict().items()}
    if param_file is None:
        for k, v in P.items():
            arrays["param/" + k] = v
        return
    path = os.path.join(OUT_DIR, param_file + ".npz")
    if os.path.exists(path):
        old = np.load(path)
        assert all(np.array_equal(old[k], v) for k, v in P.items()), f"{param_file}: cases do not share parameters"
    else:
        np.savez_compressed(path, **P)


def build_model(samples, layers, width, seed, density_bias_shift):
    torch.manual_seed(seed)
    cfg = ReflectSamplingNeRFModelConfig(
        num_coarse_samples=samples[0],
        num_importance_samples=samples[1],
        num_reflect_coarse_samples=samples[2],
        num_reflect_importance_samples=samples[3],
    )
    model = cfg.setup(scene_box=None, num_train_data=1)
    # The Model builds the Field with default sizes (model.py:103-106); the sizes BASELINE's
    # configs turn are constructor knobs of the reference Field (field.py:40-41).
    model.field = ReflectSamplingNeRFNerfField(
        position_encoding=NeRFEncoding(in_dim=3, num_frequencies=16, min_freq_exp=0.0, max_freq_exp=16.0,
                                       include_input=True),
        direction_encoding=IntegratedSHEncoding(),
        base_mlp_num_layers=layers,
        base_mlp_layer_width=width,
    )
    with torch.no_grad():
        model.field.field_output_density.net.bias += density_bias_shift
    return model


def run_case(name, R, samples, layers, width, training, seed, density_bias_shift, near=2.0, far=6.0, param_file=None,
             ray_seed=None):
    model = build_model(samples, layers, width, seed, density_bias_shift)
    o, d, pa = synthetic_rays(R, seed=(seed if ray_seed is None else ray_seed) + 100)
    nears = torch.full((R, 1), near)
    fars = torch.full((R, 1), far)
    bundle = RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=pa.clone(), nears=nears.clone(),
                       fars=fars.clone())
    model.train(training)
    torch.manual_seed(seed + 7)
    sink = io.StringIO()
    binlog = BinLog(model)
    with RandLog() as log, contextlib.redirect_stdout(sink):  # the reference prints debug lines
        if training:
            out = model.get_outputs(bundle)
        else:
            with torch.no_grad():
                out = model.get_outputs(bundle)
    binlog.close()
    arrays = {}
    save_params(arrays, model, param_file)
    for k, v in binlog.bins.items():
        arrays["bins/" + k] = v.numpy()
    for k, v in [("origins", o), ("directions", d), ("pixel_area", pa), ("nears", nears), ("fars", fars)]:
        arrays["in/" + k] = v.numpy()
    for k, v in out.items():
        a = v.detach().numpy()
        arrays["out/" + k] = a.astype(np.uint8) if a.dtype == np.bool_ else a.astype(np.float32)
    jitter_names = ["coarse", "fine", "reflect_coarse", "reflect_fine"]
    if training:
        assert len(log.draws) in (2, 4), len(log.draws)
        for n, t in zip(jitter_names, log.draws):
            arrays["jitter/" + n] = t.numpy()
    meta = dict(name=name, R=R, samples=list(samples), layers=layers, width=width, training=training, seed=seed,
                density_bias_shift=density_bias_shift, near=near, far=far,
                M=int(out["mask"].sum()), keys=sorted(out.keys()), param_file=param_file,
                generator="oracle/make_golden.py over /root/reference + oracle/ns_shim",
                torch=torch.__version__)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: R={R} M={meta['M']} keys={len(out)} -> {os.path.getsize(path)/1024:.0f} KiB")


def run_train_step_case(name, R, samples, layers, width, seed, density_bias_shift, near=2.0, far=6.0, param_file=None,
                        ray_seed=None):
    """One whole training step of the reference: get_outputs (train mode, logged jitter) -> get_loss_dict
    (model.py:346-430, post-warm-up coefficients of the config) -> backward.  Stores the outputs, the eight scaled loss
    terms and the gradient of their sum w.r.t. every Field parameter."""
    model = build_model(samples, layers, width, seed, density_bias_shift)
    o, d, pa = synthetic_rays(R, seed=(seed if ray_seed is None else ray_seed) + 100)
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    bundle = RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=pa.clone(), nears=nears.clone(),
                       fars=fars.clone())
    image = torch.rand(R, 3, generator=torch.Generator().manual_seed(seed + 200))
    model.train(True)
    torch.manual_seed(seed + 7)
    sink = io.StringIO()
    binlog = BinLog(model)
    with RandLog() as log, contextlib.redirect_stdout(sink):
        out = model.get_outputs(bundle)
        loss_dict = model.get_loss_dict(out, {"image": image})
        total = sum(loss_dict.values())
        total.backward()
    binlog.close()
    arrays = {}
    save_params(arrays, model, param_file)
    for k, v in binlog.bins.items():
        arrays["bins/" + k] = v.numpy()
    n_grad = 0
    for k, p in model.field.named_parameters():
        if p.grad is not None:
            arrays["grad/" + k] = p.grad.detach().numpy().astype(np.float32)
            n_grad += 1
    for k, v in [("origins", o), ("directions", d), ("pixel_area", pa), ("nears", nears), ("fars", fars), ("image", image)]:
        arrays["in/" + k] = v.numpy()
    for k, v in out.items():
        a = v.detach().numpy()
        arrays["out/" + k] = a.astype(np.uint8) if a.dtype == np.bool_ else a.astype(np.float32)
    for k, v in loss_dict.items():
        arrays["loss/" + k] = np.float32(v.detach().item()).reshape(1)
    assert len(log.draws) in (2, 4), len(log.draws)
    for n, t in zip(["coarse", "fine", "reflect_coarse", "reflect_fine"], log.draws):
        arrays["jitter/" + n] = t.numpy()
    meta = dict(name=name, R=R, samples=list(samples), layers=layers, width=width, training=True, seed=seed,
                density_bias_shift=density_bias_shift, near=near, far=far, M=int(out["mask"].sum()),
                keys=sorted(out.keys()), loss_coefficients={k: float(v) for k, v in model.config.loss_coefficients.items()},
                param_file=param_file,
                generator="oracle/make_golden.py over /root/reference + oracle/ns_shim (get_outputs + get_loss_dict + backward)",
                torch=torch.__version__)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: R={R} M={meta['M']} losses={len(loss_dict)} grads={n_grad} total={float(total):.6f} "
          f"-> {os.path.getsize(path)/1024:.0f} KiB")


def run_units(seed=3):
    """Direct goldens for reference-owned units: contract, IntegratedSHEncoding, ReciprocalSampler,
    get_inf_color, get_pred_normals, get_reflection."""
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    field = ReflectSamplingNeRFNerfField(
        position_encoding=NeRFEncoding(in_dim=3, num_frequencies=16, min_freq_exp=0.0, max_freq_exp=16.0,
                                       include_input=True),
        direction_encoding=IntegratedSHEncoding(),
        base_mlp_num_layers=6, base_mlp_layer_width=32)
    field.eval()
    arrays = {}
    for k, v in field.state_dict().items():
        arrays["param/" + k] = v.detach().numpy().astype(np.float32)
    # contract: means spanning inside (|x|<1) and far outside the unit ball
    mean = torch.randn(64, 5, 3, generator=g) * torch.tensor([0.3, 1.0, 3.0, 30.0, 200.0])[None, :, None]
    A = torch.randn(64, 5, 3, 3, generator=g) * 0.05
    cov = A @ A.transpose(-1, -2)
    with torch.no_grad():
        mc, cc = field.contract(mean.clone(), cov.clone())
    arrays.update({"contract/mean": mean.numpy(), "contract/cov": cov.numpy(), "contract/out_mean": mc.numpy(),
                   "contract/out_cov": cc.numpy()})
    # SH-34
    dirs = torch.nn.functional.normalize(torch.randn(257, 3, generator=g), dim=-1)
    rough = torch.rand(257, 1, generator=g) * 2.0
    with torch.no_grad():
        sh = IntegratedSHEncoding()(dirs, rough)
    arrays.update({"sh/dirs": dirs.numpy(), "sh/roughness": rough.numpy(), "sh/out": sh.numpy()})
    # reciprocal sampler, eval and train
    R = 8
    o, d, pa = synthetic_rays(R, seed=11)
    nears, fars = torch.zeros(R, 1), torch.full((R, 1), 256.0)
    samp = ReciprocalSampler(num_samples=16, tan=0.25)
    samp.eval()
    rs = samp(RayBundle(origins=o, directions=d, pixel_area=pa, nears=nears, fars=fars))
    arrays["recip/eval_starts"] = rs.frustums.starts[..., 0].numpy()
    arrays["recip/eval_ends"] = rs.frustums.ends[..., 0].numpy()
    samp.train()
    with RandLog() as log:
        rs = samp(RayBundle(origins=o, directions=d, pixel_area=pa, nears=nears, fars=fars))
    arrays["recip/train_rand"] = log.draws[0].numpy()
    arrays["recip/train_starts"] = rs.frustums.starts[..., 0].numpy()
    arrays["recip/train_ends"] = rs.frustums.ends[..., 0].numpy()
    arrays["recip/train_spacing_starts"] = rs.spacing_starts[..., 0].expand(R, 16).numpy()
    # inf colour, pred normals, reflection
    sq = torch.rand(257, 1, generator=g) * 0.2
    emb = torch.relu(torch.randn(257, 32, generator=g))
    with torch.no_grad():
        inf = field.get_inf_color(dirs, sq)
        pn = field.get_pred_normals(emb)
        refl, ndd = field.get_reflection(dirs, pn)
        rough_sig = field.get_roughness(emb)
    arrays.update({"inf/sqradius": sq.numpy(), "inf/out": inf.numpy(), "heads/emb": emb.numpy(),
                   "heads/pred_normals": pn.numpy(), "heads/n_dot_d": ndd.numpy(), "heads/reflections": refl.numpy(),
                   "heads/roughness_sigmoid": rough_sig.numpy()})
    meta = dict(name="units", layers=6, width=32, generator="oracle/make_golden.py", torch=torch.__version__)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT_DIR, "units.npz")
    np.savez_compressed(path, **arrays)
    print(f"units -> {os.path.getsize(path)/1024:.0f} KiB")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.set_num_threads(4)
    # eval, 8-layer trunk with the layer-4 skip, density bias raised so the reflect branch runs
    run_case("eval_l8_w32", R=48, samples=(16, 16, 8, 8), layers=8, width=32, training=False, seed=0,
             density_bias_shift=2.0)
    # train mode: stratified jitter + autograd normals
    run_case("train_l8_w32", R=40, samples=(16, 16, 8, 8), layers=8, width=32, training=True, seed=1,
             density_bias_shift=2.0)
    # 4-layer trunk (BASELINE configs[0] shape, narrow), default bias, ragged sample counts
    run_case("eval_l4_w32", R=33, samples=(24, 12, 10, 6), layers=4, width=32, training=False, seed=2,
             density_bias_shift=1.0)
    # empty-mask early-out (model.py:259-260): rays that never hit density
    run_case("eval_l6_w32_nomask", R=16, samples=(8, 8, 8, 8), layers=6, width=32, training=False, seed=4,
             density_bias_shift=-12.0)
    # eval near plane 0 (collider reset in eval), wide far
    run_case("eval_l8_w64_near0", R=24, samples=(32, 32, 16, 16), layers=8, width=64, training=False, seed=5,
             density_bias_shift=1.5, near=0.0, far=6.0)
    # one whole training step (forward + the reference's own get_loss_dict + backward) at a width the HIP path runs
    run_train_step_case("trainstep_l8_w64", R=32, samples=(16, 16, 8, 8), layers=8, width=64, seed=6,
                        density_bias_shift=2.0)
    # ---- shapes the HIP kernels run (widths 64 / 128 / 256), compared with the reference's outputs DIRECTLY on the GPU
    # the BASELINE network (8 x 256, field.py:40-41 defaults): eval and one whole training step on the same parameters
    run_case("eval_l8_w256", R=16, samples=(16, 16, 8, 8), layers=8, width=256, training=False, seed=10,
             density_bias_shift=2.0, param_file="params_l8_w256_seed10")
    run_train_step_case("trainstep_l8_w256", R=16, samples=(16, 16, 8, 8), layers=8, width=256, seed=10,
                        density_bias_shift=2.0, param_file="params_l8_w256_seed10", ray_seed=20)
    # 4-layer 128-wide trunk (BASELINE configs[0] network): eval with ragged sample counts, and a training step
    run_case("eval_l4_w128", R=24, samples=(24, 12, 10, 6), layers=4, width=128, training=False, seed=11,
             density_bias_shift=1.5, param_file="params_l4_w128_seed11")
    run_train_step_case("trainstep_l4_w128", R=24, samples=(16, 16, 8, 8), layers=4, width=128, seed=11,
                        density_bias_shift=1.5, param_file="params_l4_w128_seed11", ray_seed=21)
    # the no-mask early-out (model.py:259-260) at a width the kernels accept
    run_case("eval_l6_w64_nomask", R=16, samples=(8, 8, 8, 8), layers=6, width=64, training=False, seed=4,
             density_bias_shift=-12.0)
    # ---- widths that are NOT one of the kernels' 64 / 128 / 256: the HIP path runs them zero-padded (rsn_field_desc.param_width)
    run_train_step_case("trainstep_l6_w48", R=24, samples=(16, 16, 8, 8), layers=6, width=48, seed=12, density_bias_shift=2.0)
    run_case("eval_l8_w200", R=16, samples=(16, 16, 8, 8), layers=8, width=200, training=False, seed=13,
             density_bias_shift=2.0, param_file="params_l8_w200_seed13")
    run_train_step_case("trainstep_l8_w200", R=12, samples=(16, 16, 8, 8), layers=8, width=200, seed=13,
                        density_bias_shift=2.0, param_file="params_l8_w200_seed13", ray_seed=23)
    run_units()


if __name__ == "__main__":
    main()
