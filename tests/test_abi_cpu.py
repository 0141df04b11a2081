"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/rsn.h declares, and the host mirror keeps the reference's names (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import _abi
from reflect_sampling_nerf_amd._build import LIB_PATH, build_library
from tests.helpers import load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build_library()
    return pkg.load_library()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(REPO, "include", "rsn.h")).read()
    declared = set(re.findall(r"\b(rsn_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_abi.EXPORTED_SYMBOLS), declared ^ set(_abi.EXPORTED_SYMBOLS)
    raw = C.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in rsn.h but not exported"
    assert lib.rsn_abi_version() == _abi.RSN_ABI_VERSION


def test_packed_size_and_argument_errors(lib):
    model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1)
    desc = model.field.field_desc()
    nbytes = lib.rsn_packed_weights_bytes(C.byref(desc))
    # every nn.Linear of the path appears as fp32 in forward order, fp32 transposed (dX sweeps) and as 3-way bf16
    # splits of both (1.5x each), zero padded to MFMA tiles; field_output_low does not appear
    n_used = sum(p.numel() for n, p in model.field.named_parameters() if "field_output_low" not in n)
    assert nbytes // 4 >= 4.8 * n_used and nbytes // 4 < 5.4 * n_used
    bad = _abi.FieldDesc()
    bad.num_layers, bad.width, bad.skip_layer, bad.mid_width = 8, 100, 4, 128
    assert lib.rsn_packed_weights_bytes(C.byref(bad)) == 0
    assert b"width=100" in lib.rsn_last_error()
    bad.width, bad.num_layers, bad.skip_layer = 128, 5, 4  # the reference MLP raises for skip == last layer
    assert lib.rsn_packed_weights_bytes(C.byref(bad)) == 0
    # NULL / bad-size arguments are rejected before any launch
    assert lib.rsn_sample_spaced(4, None, 0, 0, 1.0, None, None, None, None, None, None) == -1
    assert lib.rsn_composite(4, None, 8, 7, 0, None, None) == -1


def test_state_dict_names_match_reference():
    meta, g = load_golden("eval_l8_w32")
    ref_names = set(g["param"].keys())
    model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1)
    assert set(model.field.state_dict().keys()) == ref_names
    assert sum(p.numel() for p in model.get_param_groups()["fields"]) == 618513
    assert set(model.get_param_groups().keys()) == {"fields"}


def test_same_seed_same_init_as_reference_order():
    """Modules are created in the reference's order, so the oracle's init (seeded nn.Linear-equivalent draws in
    that order) and ours have identical shapes in identical order."""
    from oracle import cpu_ref

    P = cpu_ref.init_params(cpu_ref.FieldSpec())
    model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1)
    sd = model.field.state_dict()
    assert [tuple(v.shape) for v in P.values()] == [tuple(sd[k].shape) for k in P.keys()]


def test_config_fields_and_train_mode_is_loud():
    cfg = pkg.ReflectSamplingNeRFModelConfig()
    assert (cfg.num_coarse_samples, cfg.num_importance_samples, cfg.num_reflect_coarse_samples,
            cfg.num_reflect_importance_samples) == (128, 128, 64, 64)
    assert cfg.loss_coefficients["orientation_loss_fine"] == 1e-1 and len(cfg.loss_coefficients) == 12
    model = cfg.setup(scene_box=None, num_train_data=1)
    assert model.far == 256 and model.near == 1.0 / 16
    o = torch.zeros(4, 3)
    rb = pkg.RayBundle(origins=o, directions=o, pixel_area=torch.ones(4, 1), nears=torch.ones(4, 1),
                       fars=torch.ones(4, 1))
    model.eval()
    with pytest.raises(pkg.RsnError):  # CPU tensors: there is no CPU fallback
        model.get_outputs(rb)


def test_bench_input_generator_equals_the_oracles():
    """bench.py draws its rays from the package (the product path never imports oracle/); same rays as the tests'."""
    from oracle import cpu_ref
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    for a, b in zip(synthetic_rays(37, seed=5), cpu_ref.synthetic_rays(37, seed=5)):
        assert torch.equal(a, b)
    import ast

    tree = ast.parse(open(os.path.join(REPO, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            mods = [node.module or ""] if isinstance(node, ast.ImportFrom) else (
                [a.name for a in node.names] if isinstance(node, ast.Import) else [])
            if any(m.split(".")[0] == "oracle" for m in mods):
                assert fn.name.startswith("cpu_baseline"), \
                    f"bench.py:{fn.name} imports oracle/ (only the cpu_baseline legs may)"


def test_slot_maps_are_permutations_of_the_reference_columns():
    """Host tables of the training graph: the kernel's slot order of the 99 encoded inputs / 34 SH inputs must hit every
    reference column exactly once (padding slots map to -1)."""
    from reflect_sampling_nerf_amd.train_graph import ENC_SLOTS, SH_SLOTS, enc_slot_columns, sh_slot_columns

    enc = enc_slot_columns()
    assert len(enc) == ENC_SLOTS == 104 and sorted(c for c in enc if c >= 0) == list(range(99))
    sh = sh_slot_columns()
    assert len(sh) == SH_SLOTS == 40 and sorted(c for c in sh if c >= 0) == list(range(34))
    # the library's own statement of the layout (rsn_train_saved_layout, ABI 15): the default kernels' slot order equals the host
    # tables above; the plain-bf16 ring kernels (width 256) keep bf16 rows of 128 / 64 slots -- again every reference column once
    torch = pytest.importorskip("torch")
    for mode, width, want in (("f32", 256, (104, 40, torch.float32)), ("bf16", 128, (104, 40, torch.float32)),
                              ("bf16", 256, (128, 64, torch.bfloat16))):
        f = pkg.ReflectSamplingNeRFNerfField(base_mlp_num_layers=8, base_mlp_layer_width=width)
        f.set_mma_mode(mode)
        lay = f.train_layout()
        assert (lay["enc_cols"], lay["sh_cols"], lay["narrow_dtype"]) == want
        assert sorted(c for c in lay["enc_map"] if c >= 0) == list(range(99)) and len(lay["enc_map"]) == want[0]
        assert sorted(c for c in lay["sh_map"] if c >= 0) == list(range(34)) and len(lay["sh_map"]) == want[1]
        if want[0] == 104:
            assert lay["enc_map"] == enc and lay["sh_map"] == sh


def test_config_defaults_match_the_reference_run():
    """Loss coefficients and sample counts of the default ModelConfig against what the reference's own config held when
    the training-step fixture was generated (meta of tests/golden/trainstep_l8_w64.npz)."""
    meta, _ = load_golden("trainstep_l8_w64")
    cfg = pkg.ReflectSamplingNeRFModelConfig()
    assert dict(cfg.loss_coefficients) == pytest.approx(meta["loss_coefficients"])
    assert (cfg.num_coarse_samples, cfg.num_importance_samples, cfg.num_reflect_coarse_samples,
            cfg.num_reflect_importance_samples) == (128, 128, 64, 64)  # model.py:46-54


def test_exponential_decay_lr_schedule():
    """config.py:50-53: lr 1e-3 decaying log-linearly to 1e-4 at step 50 000, constant afterwards."""
    from reflect_sampling_nerf_amd.train_ops import exponential_decay_lr

    assert exponential_decay_lr(0) == pytest.approx(1e-3)
    assert exponential_decay_lr(25000) == pytest.approx((1e-3 * 1e-4) ** 0.5)
    assert exponential_decay_lr(50000) == pytest.approx(1e-4)
    assert exponential_decay_lr(10 ** 6) == pytest.approx(1e-4)


def test_bench_flop_accounting_matches_survey():
    """bench.py's algorithmic MAC counts per field point against SURVEY 8(d): 615,296 (8 x 256) and 100,736 (4 x 128)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    m = bench.algorithmic_macs(8, 256)
    assert m["forward"] == 615296 == m["wgrad"] and m["normals"] == 509440
    assert m["backward"] == 3 * 128 + 128 * 256 + (256 + 11) * 256 + 7 * 256 * 256
    assert m["backward_input"] == m["backward"] + 2 * 99 * 256
    assert bench.algorithmic_macs(4, 128)["forward"] == 100736


def test_diagnostic_macros_cannot_enter_the_product_library(tmp_path):
    """Timing ablations (RSN_RING_NO_*, RSN_R16_*, RSN_DIAG_NO_SAVED_ROWS: wrong results by construction) compile only under
    -DRSN_DIAG_BUILD, which the product flags never carry, and a diagnostic library reports RSN_ABI_DIAG_FLAG in
    rsn_abi_version() (the loader refuses it at the product path)."""
    import shutil
    import subprocess

    from reflect_sampling_nerf_amd import _build

    assert not any("RSN_DIAG_BUILD" in f or f.startswith("-DRSN_R") for f in _build.FLAGS)
    lib = _abi.load_library()
    assert lib.rsn_abi_version() & _abi.RSN_ABI_DIAG_FLAG == 0
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "probe.hip"
    src.write_text('#include "rsn_common.h"\nint main() { return 0; }\n')
    base = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I", os.path.join(REPO, "include"), "-I",
            _build.CSRC, str(src)]
    assert subprocess.run(base, capture_output=True).returncode == 0
    bad = subprocess.run(base + ["-DRSN_R16_NO_MFMA"], capture_output=True, text=True)
    assert bad.returncode != 0 and "RSN_DIAG_BUILD" in bad.stderr
    assert subprocess.run(base + ["-DRSN_R16_NO_MFMA", "-DRSN_DIAG_BUILD"], capture_output=True).returncode == 0
    # the LDS-staged weight-gradient probe kernels (csrc/rsn_wgrad_staged_probe.h) are diagnostic builds only as well
    for macro in ("-DWG_X6_STAGED", "-DWG_F32_STAGED"):
        bad = subprocess.run(base + [macro], capture_output=True, text=True)
        assert bad.returncode != 0 and "RSN_DIAG_BUILD" in bad.stderr, macro
    blob = open(os.path.join(REPO, "reflect_sampling_nerf_amd", "librsn_hip.so"), "rb").read()
    assert b"rsn_wgrad_x6s_kernel" not in blob and b"rsn_wgrad_f32s_kernel" not in blob


def test_field_constructor_knobs():
    """reference field.py:38-47: `spatial_distortion` is accepted and kept (applied in get_blob: GPU test); the knobs the fused
    kernels do not cover are refused loudly instead of being ignored."""
    import pytest

    from reflect_sampling_nerf_amd.nerfstudio_compat import Gaussians

    fn = lambda g: g  # noqa: E731
    fld = pkg.ReflectSamplingNeRFNerfField(base_mlp_num_layers=4, base_mlp_layer_width=64, spatial_distortion=fn)
    assert fld.spatial_distortion is fn
    with pytest.raises(NotImplementedError, match="granular API"):
        fld._no_distortion()
    g = Gaussians(mean=torch.zeros(2, 3), cov=torch.zeros(2, 3, 3))
    assert g.mean.shape == (2, 3) and g.cov.shape == (2, 3, 3)
    with pytest.raises(NotImplementedError):
        pkg.ReflectSamplingNeRFNerfField(head_mlp_num_layers=2)
    with pytest.raises(NotImplementedError):
        pkg.ReflectSamplingNeRFNerfField(skip_connections=(2, 4))


def test_product_library_reads_no_environment_switches():
    """The product librsn_hip.so takes no run-time A/B switches from the environment (RSN_BF16_PER_WAVE_STREAM,
    RSN_RING_STAGGER, RSN_RING_WAVES, RSN_RING_SHAPE32 were read by getenv in the product path): those names, and getenv
    itself, exist in diagnostic builds only (-DRSN_DIAG_BUILD); the 32x32x16 A/B ring kernel is not in the code object."""
    import subprocess

    path = _abi.library_path() if hasattr(_abi, "library_path") else os.path.join(REPO, "reflect_sampling_nerf_amd", "librsn_hip.so")
    blob = open(path, "rb").read()
    for name in (b"RSN_RING_STAGGER", b"RSN_RING_WAVES", b"RSN_RING_SHAPE32", b"RSN_BF16_PER_WAVE_STREAM", b"RSN_RING_", b"RSN_BF16_"):
        assert name not in blob, name
    nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True)
    assert nm.returncode == 0 and "getenv" not in nm.stdout
    assert b"rsn_field_bf16_ring_kernel" not in blob and b"rsn_field_bf16_ring16_kernel" in blob


def test_fused_radam_state_dict_is_torch_radams():
    """FusedRAdam.state_dict / load_state_dict speak torch.optim.RAdam's format (the reference's optimiser: config.py:50-53 through
    nerfstudio's RAdamOptimizerConfig), so that checkpoints cross over; the state transfer itself needs no GPU."""
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g)),
          torch.nn.Parameter(torch.ones(2))]
    ref = torch.optim.RAdam(ps, lr=1e-3, eps=1e-15)
    for _ in range(3):
        ps[0].grad, ps[1].grad = torch.randn(4, 3, generator=g), torch.randn(5, generator=g)  # ps[2]: never a gradient
        ref.step()
    mine = pkg.FusedRAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1.0, eps=1.0)
    mine.load_state_dict(ref.state_dict())
    assert mine.step_count == 3 and mine.lr == 1e-3 and mine.eps == 1e-15 and tuple(mine.betas) == (0.9, 0.999)
    sd = ref.state_dict()["state"]
    assert torch.equal(mine.exp_avg[0], sd[0]["exp_avg"]) and torch.equal(mine.exp_avg_sq[1], sd[1]["exp_avg_sq"])
    assert float(mine.exp_avg[2].abs().max()) == 0.0
    back = torch.optim.RAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1.0)
    back.load_state_dict(mine.state_dict())
    assert back.param_groups[0]["lr"] == 1e-3 and float(back.state_dict()["state"][0]["step"]) == 3.0
    with pytest.raises(ValueError):
        pkg.FusedRAdam(ps[:2]).load_state_dict(ref.state_dict())


def test_reference_pipeline_checkpoint_keys_load_strictly():
    """a pipeline checkpoint of the reference carries `_model.field.*` (names / shapes: test_state_dict_names_match_reference),
    `_model.device_indicator_param` and the weights of the torchmetrics modules its Model owns (`_model.lpips.net.*`, model.py:131-133).
    Loaded through a parent module with strict key checking -- as nerfstudio's pipeline does -- the metric entries are dropped, everything
    else must match."""
    cfg = pkg.ReflectSamplingNeRFModelConfig(base_mlp_num_layers=4, base_mlp_layer_width=64)

    class Pipe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(1)
            self._model = cfg.setup(scene_box=None, num_train_data=1)

    src, dst = Pipe(), Pipe()
    with torch.no_grad():
        for p in src.parameters():
            p.add_(1.0)
    ckpt = dict(src.state_dict())
    ckpt["_model.lpips.net.net.slice1.0.weight"] = torch.zeros(64, 3, 11, 11)  # what the reference's checkpoint adds
    ckpt["_model.lpips.net.lin0.model.1.weight"] = torch.zeros(1, 64, 1, 1)
    res = dst.load_state_dict(ckpt, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for (n, a), (_, b) in zip(src.named_parameters(), dst.named_parameters()):
        assert torch.equal(a, b), n
    ckpt["_model.field.not_a_parameter"] = torch.zeros(1)  # anything else unexpected is still an error
    with pytest.raises(RuntimeError, match="not_a_parameter"):
        dst.load_state_dict(ckpt, strict=True)
    direct = dict(src._model.state_dict())
    direct["lpips.net.lin0.model.1.weight"] = torch.zeros(1, 64, 1, 1)
    assert not dst._model.load_state_dict(direct, strict=True).unexpected_keys
