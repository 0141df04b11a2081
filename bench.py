#!/usr/bin/env python3
"""bench.py -- rays/s of the reflect-sampling-nerf training step on MI355X (BASELINE.json `metric`:
"rays/sec (train step) at 4096 rays x 128 samples, 1/2/4/8 MI355X").

Headline workload, the SAME at every N (so the driver's 1->2->4->8 curve compares like with like):
    one optimisation step per rank on 4096 rays -- training-mode get_outputs with the reference's default sample
    counts (128 coarse + 128 fine + reflect 64 + 64, model.py:46-54; stratified jitter, analytic normals), the eight
    loss terms of get_loss_dict, backward (dX sweeps + weight gradients), ONE flat gradient all-reduce over RCCL when
    N > 1 (parallel.FlatGradAllReduce == DDP with find_unused_parameters=True, pipeline.py:72-77), fused RAdam --
    8-layer 256-wide field, exact fp32, synthetic camera-shell rays, random-init weights, inputs
    resident in HBM.  Weak scaling: every rank draws its own 4096 rays; `value` = N * 4096 * steps / max-over-ranks
    wall time between two barrier + synchronize brackets.

One JSON line is printed by rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel of the step (the training forward rsn_field_kernel<8,true,0>), MFMA-bound:
                  algorithmic FLOP of its launches / their summed duration, measured live with HIP events on the
                  launch stream inside the timed region (ops.KernelTimer);
  train_step   -- algorithmic FLOP per step (formula + measured reflect-ray count M) and the same per-kernel
                  figures for the backward sweep and the weight-gradient kernels;
  cpu_baseline -- the CPU oracle (oracle/cpu_ref.py, a port of the reference's PyTorch op sequence: get_outputs in
                  training mode + loss + autograd backward) timed on the host cores on a bounded sample;
  eval_level   -- (N = 1) BASELINE configs[1]: 4096 rays x 128 samples fused forward + composite of one level, with
                  its own roofline and CPU baseline (the round-1 headline, kept for continuity).
A stalled or failed step exits NON-ZERO (watchdog: rc 3, exception: rc 4 after printing the error to stderr).

Launching: `python bench.py --gpus N` is enough at every N.  With N > 1 and no WORLD_SIZE in the environment the
process starts N fresh children of itself (one rank per GPU over RCCL) before it touches any GPU, relays rank 0's JSON
line and exits with the worst child code (`self_launch`).  Under `python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N` (WORLD_SIZE set) it is one of the ranks.  RSN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with
gloo as the backend: the 2-rank rehearsal a one-GPU box allows.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
import traceback

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense
HBM_PEAK_TBPS = 8.0  # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s measured for a float4 copy)
ENC_DIM, SH_DIM, MID_W = 99, 34, 128


def algorithmic_macs(layers: int, width: int):
    """MAC per field point.  forward: trunk + bottleneck + mlp_mid + heads (615,296 at 8 x 256);
    normals: the dX-only sweep through the trunk that Field.get_normals' autograd.grad performs (= trunk MACs);
    backward: dX through every GEMM whose input carries gradient (the SH part of mlp_mid does not; the encoded input
    only where pixel_area does: reflect levels and get_inf_color); wgrad: dW = dY^T X of every weight (= forward)."""
    W, L = width, layers
    skip = L > 5  # skip_connections=(4,) is live only when layer 4 is not the last
    trunk = ENC_DIM * W + (L - 1) * W * W + (ENC_DIM * W if skip else 0)
    fwd = trunk + W * W + (SH_DIM + W) * MID_W + 11 * W + 3 * MID_W
    bwd = 3 * MID_W + MID_W * W + (W + 11) * W + (L - 1) * W * W
    return {"forward": fwd, "normals": trunk, "backward": bwd,
            "backward_input": bwd + ENC_DIM * W * (2 if skip else 1), "wgrad": fwd}


def collective_config(dist, backend, world):
    """The `config` entries that describe the data-parallel layout of a run (the driver checks `rccl_ranks` against N)."""
    return {
        "parallelism": "dp%d" % world,
        "collective": None if dist is None else "all-reduce(sum)/N of one flat fp32 gradient buffer per step",
        "rccl_ranks": int(dist.get_world_size()) if (dist is not None and backend == "nccl") else 0,
        "backend": backend,
    }


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising HIP in the launcher (torch.cuda.device_count() falls through to
    hipGetDeviceCount on a build without amdsmi): the visible-devices environment if set, else the KFD topology (nodes
    with a gfx target are GPUs).  None if neither is available -- the ranks then report a shortfall themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as fh:
                props = dict(line.split(None, 1) for line in fh if " " in line)
            if int(props.get("gfx_target_version", "0")) != 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128, help="samples per primary level (coarse and fine)")
    ap.add_argument("--coarse", type=int, default=0, help="coarse samples if different from --samples (configs[2]: 64)")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--workload", default="train", choices=["train", "level", "get_outputs", "eval_image", "selftest"],
                    help="train = the BASELINE metric (default): full optimisation step; level = BASELINE configs[1], "
                         "fused forward + composite of one sampling level (eval); get_outputs = full eval get_outputs; "
                         "selftest = launch plumbing only (process group over gloo on the CPU, no GPU work)")
    ap.add_argument("--mma", default="f32", choices=["f32", "bf16x6", "bf16x3", "bf16"],
                    help="matrix-core arithmetic of the field kernels: f32 = exact fp32 MFMA (default); bf16x6 = fp32 "
                         "emulation by 3-way bf16 splits (fp32-equivalent); bf16x3 / bf16 = reduced precision (eval)")
    ap.add_argument("--ray-chunk", type=int, default=0,
                    help="train: walk the batch in chunks of this many rays (gradient accumulation, exact; bounds the "
                         "activation memory). 0 = the whole batch at once")
    ap.add_argument("--reflect-capacity", default="", choices=["", "auto"],
                    help="train: size the reflect levels' buffers from the previous step's reflected-ray count (opt-in: "
                         "train_graph.reflect_capacity) instead of for all rays")
    ap.add_argument("--wgrad-groups", type=int, default=1, choices=[1, 2],
                    help="2: reduce the reflect branch's weight gradients before the primary levels' backward sweeps and release its "
                         "buffers (opt-in memory bound, model.weight_grad_groups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (eval_level, configs[2])")
    ap.add_argument("--cpu-rays", type=int, default=0, help="rays of the bounded CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def host_threads():
    # host cores this process may use: the GPU box gives a 1-GPU job a 16-core share of a 256-core host;
    # more threads than that only oversubscribes (measured: 256 threads = 18 rays/s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, int(os.environ.get("RSN_CPU_THREADS", "16"))))


def cpu_baseline_level(args):
    """The oracle's single-level eval render on the host cores, bounded sample."""
    from oracle import cpu_ref

    torch.set_num_threads(host_threads())
    fs = cpu_ref.FieldSpec(num_layers=args.layers, width=args.width)
    P = cpu_ref.init_params(fs, seed=0)
    Rc = args.cpu_rays or 1024
    o, d, pa = cpu_ref.synthetic_rays(Rc, seed=0)
    nears, fars = torch.full((Rc, 1), 2.0), torch.full((Rc, 1), 6.0)
    with torch.no_grad():
        cpu_ref.render_level(P, fs, o, d, pa, nears, fars, args.samples)  # warm-up
        times = []
        t_end = time.time() + 12.0
        while len(times) < 3 or (time.time() < t_end and len(times) < 8):
            t0 = time.perf_counter()
            cpu_ref.render_level(P, fs, o, d, pa, nears, fars, args.samples)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": Rc / med, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{Rc} rays x {args.samples} samples, same field ({args.layers}x{args.width}) and level, "
                      f"median of {len(times)} runs, oracle/cpu_ref.render_level (eager PyTorch fp32)"}


def cpu_baseline_train(args, samples):
    """The oracle's training step on the host cores: get_outputs(training) + the 8-term loss + autograd backward +
    torch.optim.RAdam, bounded sample of the same workload (same field, same sample counts, fewer rays)."""
    from oracle import cpu_ref

    torch.set_num_threads(host_threads())
    fs = cpu_ref.FieldSpec(num_layers=args.layers, width=args.width)
    ms = cpu_ref.ModelSpec(*samples)
    P = {k: v.clone().requires_grad_(True) for k, v in cpu_ref.init_params(fs, seed=0, density_bias_shift=2.0).items()}
    opt = torch.optim.RAdam([p for p in P.values()], lr=1e-3, eps=1e-15)
    Rc = args.cpu_rays or 128
    o, d, pa = cpu_ref.synthetic_rays(Rc, seed=0)
    nears, fars = torch.full((Rc, 1), 2.0), torch.full((Rc, 1), 6.0)
    image = torch.rand(Rc, 3, generator=torch.Generator().manual_seed(1234))
    g = torch.Generator().manual_seed(5)
    times, M = [], 0
    t_end = time.time() + 25.0
    while len(times) < 3 or (time.time() < t_end and len(times) < 6):
        jit = {"coarse": torch.rand(Rc, samples[0] + 1, generator=g), "fine": torch.rand(Rc, samples[1] + 1, generator=g),
               "reflect_coarse": torch.rand(Rc, samples[2] + 1, generator=g),
               "reflect_fine": torch.rand(Rc, samples[3] + 1, generator=g)}
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = cpu_ref.get_outputs(P, fs, ms, o, d, pa, nears, fars, training=True, jitter=jit)
        loss = sum(cpu_ref.loss_dict(out, image).values())
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        M = int(out["mask"].sum())
    steady = sorted(times[1:])  # the first step pays allocator warm-up
    med = steady[len(steady) // 2]
    return {"value": Rc / med, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{Rc} rays x ({samples[0]} coarse + {samples[1]} fine + reflect {samples[2]} + {samples[3]} on "
                      f"M={M} rays), same field ({args.layers}x{args.width}): oracle/cpu_ref.get_outputs(training) + "
                      f"loss_dict + autograd backward + torch.optim.RAdam, median of {len(steady)} steps after 1 warm-up"}


class Watchdog:
    """Ends the process with rc 3 if the benchmark makes no progress for `limit` seconds (a rank lost in the gradient
    all-reduce, a hung kernel): a stall must never read as rc 0."""

    def __init__(self, rank, limit=240.0):
        self.t = time.time()
        self.rank, self.limit = rank, limit
        self.done = threading.Event()
        threading.Thread(target=self._run, daemon=True).start()

    def beat(self):
        self.t = time.time()

    def _run(self):
        while not self.done.wait(timeout=5.0):
            if time.time() - self.t > self.limit:
                sys.stderr.write(json.dumps({"error": "bench.py stalled: no step finished for %.0f s on rank %d"
                                                      % (self.limit, self.rank)}) + "\n")
                sys.stderr.flush()
                os._exit(3)


def timed_region(dist, share, dev, steps, fn):
    """barrier + synchronize | K steps | synchronize + barrier; returns the max over ranks of the wall time."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def newest_profile_traffic(kernel_name, rays, samples):
    """HBM bytes per launch of `kernel_name` from the newest profiles/*_summary.json taken on this workload size."""
    traffic = None
    try:
        for fn in sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_summary.json")):
            with open(os.path.join(REPO, "profiles", fn)) as fh:
                js = json.load(fh)
            if js.get("kernel", "").replace(" ", "") == kernel_name.replace(" ", "") and js.get("rays", 4096) == rays \
                    and js.get("samples", 128) == samples \
                    and js.get("hbm_traffic_bytes_per_launch") is not None:
                traffic = js["hbm_traffic_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    return traffic


TRAFFIC_NOTE = ("HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction), separate rocprofv3 --pmc passes of "
                "tools/profile_round.sh; newest profiles/*_summary.json taken on this kernel and workload size (null if none)")


def run_train(pkg, args, dev, rank, world, dist, share, samples, steps, warmup, dog, time_kernels=True):
    """`steps` timed optimisation steps on this rank's rays; returns the train-step record (rank-0 relevant)."""
    from reflect_sampling_nerf_amd import ops
    from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce, train_step
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    R = args.rays
    torch.manual_seed(0)  # identical random-init weights on every rank (data-parallel replicas)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=samples[0], num_importance_samples=samples[1],
                                            num_reflect_coarse_samples=samples[2],
                                            num_reflect_importance_samples=samples[3],
                                            base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0  # so that the reflect branch is exercised
    model.to(dev).train()
    if getattr(args, "reflect_capacity", ""):
        model.reflect_capacity = args.reflect_capacity
    if getattr(args, "wgrad_groups", 1) == 2:
        model.weight_grad_groups = 2
    model.field.set_mma_mode(args.mma if args.mma in ("f32", "bf16x6", "bf16") else "f32")
    o, d, pa = synthetic_rays(R, seed=rank)  # each rank renders its own rays
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(R, 1).to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    params = model.get_param_groups()["fields"]
    optimizer = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)  # config.py:50-53
    reducer = FlatGradAllReduce(params) if dist is not None else None
    if reducer is not None and world == 1:
        reducer.run_single_rank = True  # RSN_BENCH_FORCE_COLLECTIVE rehearsal
    batch = {"image": torch.rand(R, 3, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)}
    state = {"it": 100, "M": 0, "loss": None}  # past the 50-step loss warm-up: all eight loss terms are live
    m_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # reflected rays, accumulated on the device: no host read per step

    def step(_i=None):
        state["loss"] = train_step(model, rb, batch, optimizer, reducer, state["it"], ray_chunk=args.ray_chunk or None)
        state["it"] += 1
        nm = getattr(model, "_step_n_masked_dev", None)  # summed over the ray chunks by train_step
        if nm is not None:
            m_dev.add_(nm)
        dog.beat()

    for _ in range(warmup):
        step()
    m_dev.zero_()
    timer = ops.KernelTimer() if time_kernels else None
    torch.cuda.reset_peak_memory_stats()
    ops.TIMER = timer
    try:
        elapsed = timed_region(dist, share, dev, steps, step)
    finally:
        ops.TIMER = None
    state["M"] = int(m_dev.item())
    macs = algorithmic_macs(args.layers, args.width)
    rec = {"value": world * R * steps / elapsed, "unit": "rays/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "warmup": warmup, "n_gpus": world, "loss": float(state["loss"]),
           "reflect_ray_fraction": state["M"] / float(steps * R), "host_syncs_in_reducer":
               (reducer.host_syncs if reducer is not None else 0),
           "peak_device_memory_gb": torch.cuda.max_memory_allocated() / 2**30, "ray_chunk": args.ray_chunk or None,
           "reflect_capacity": getattr(args, "reflect_capacity", "") or None,
           "weight_grad_groups": getattr(args, "wgrad_groups", 1),
           "reflect_overflows": getattr(model, "reflect_overflows", 0)}
    if timer is not None:
        tot = timer.totals()
        pts = lambda k: tot.get(k, {"work": {}})["work"].get("points", 0)  # noqa: E731
        ms = lambda k: tot.get(k, {"ms": 0.0})["ms"]  # noqa: E731
        calls = lambda k: tot.get(k, {"calls": 0})["calls"]  # noqa: E731
        n_fwd = pts("field_forward_train_normals") + pts("field_forward_train")
        n_bwd = pts("field_backward") + pts("field_backward_input")
        flop = {
            "forward": 2.0 * (macs["forward"] * n_fwd + macs["normals"] * pts("field_forward_train_normals")),
            "backward": 2.0 * (macs["backward"] * pts("field_backward") + macs["backward_input"] * pts("field_backward_input")),
            "wgrad": 2.0 * macs["wgrad"] * n_bwd,
        }
        kern = {}
        ring = args.mma in ("bf16", "bf16x6") and args.width == 256  # plain / split bf16 training at width 256: the LDS-ring kernels
        rk = "x6" if args.mma == "bf16x6" else "bf16"
        mi = {"bf16x6": 1, "bf16": 3}.get(args.mma, 0)
        # plain-bf16 mode: the saved rows ARE the traffic (256 FLOP per byte of row: as much HBM- as MFMA-bound).  Algorithmic HBM
        # bytes per field point: forward writes L act rows + bottleneck (W bf16 each), mid hidden, encoded / SH inputs, raw heads,
        # ReLU bits and the 64 B of per-sample outputs (+ 12 B of normals on the primary levels); the backward sweep writes L + 1
        # wide gradient rows, d a_mid and the narrow head gradients and reads bits / heads / outputs (+ the encoded inputs where the
        # gradient reaches them); the weight gradients read every row once (the encoded inputs twice: layer 0 and the skip layer).
        Lw, Ww = args.layers, args.width
        row_b = 2 if args.mma == "bf16" else 4
        enc_b, sh_b = ((256, 128) if args.mma == "bf16" else (512, 256)) if ring else (416, 160)
        skipw = Lw > 5
        fwd_w = row_b * (Lw * Ww + Ww + MID_W) + enc_b + sh_b + 32 + 32 * (Lw + 1) + 64
        bwd_w = row_b * (Lw * Ww + Ww + MID_W) + 64 + 16
        bwd_r = 32 * (Lw + 1) + 32 + 64
        wg_r = row_b * (2 * Lw * Ww + 2 * Ww + 2 * MID_W) + enc_b * (2 if skipw else 1) + sh_b + 64 + 16
        hbm = {"forward": fwd_w * n_fwd + 12.0 * pts("field_forward_train_normals"),
               "backward": (bwd_w + bwd_r) * n_bwd + enc_b * pts("field_backward_input"),
               "wgrad": wg_r * n_bwd}
        for key, name, names in (
                ("forward", "rsn_field_%s_train_kernel<normals>" % rk if ring else "rsn_field_kernel<%d,true,%d>" % (args.width // 32, mi),
                 ("field_forward_train_normals", "field_forward_train")),
                ("backward", "rsn_field_%s_bwd_kernel<input>" % rk if ring else "rsn_field_bwd_kernel<%d,%d>" % (args.width // 32, mi),
                 ("field_backward", "field_backward_input")),
                ("wgrad", "rsn_wgrad_kernel", ("weight_grad",))):
            t_ms = sum(ms(n) for n in names)
            n_calls = sum(calls(n) for n in names)
            tf = flop[key] / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
            kern[key] = {"kernel": name, "launches_per_step": n_calls / steps, "avg_launch_ms": t_ms / max(n_calls, 1),
                         "ms_per_step": t_ms / steps, "algorithmic_flop_per_step": flop[key] / steps,
                         "achieved_tflops": tf, "frac_of_fp32_mfma_peak": tf / FP32_MFMA_PEAK_TFLOPS}
            if args.mma in ("bf16", "bf16x6"):  # these modes' two rooflines: the dense bf16 MFMA peak and the HBM stream of the saved rows
                tbps = hbm[key] / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
                issued = 6.0 if args.mma == "bf16x6" else 1.0  # split-bf16: six bf16 MFMA products per fp32 product
                kern[key].update({"frac_of_bf16_mfma_peak": issued * tf / BF16_MFMA_PEAK_TFLOPS, "bf16_products_per_flop": issued,
                                  "algorithmic_hbm_bytes_per_step": hbm[key] / steps, "achieved_hbm_tbps": tbps,
                                  "frac_of_hbm_peak": tbps / HBM_PEAK_TBPS})
        total_flop = sum(flop.values()) / steps
        # the same, split by launch kind (primary levels carry the analytic-normal sweep / no input gradient)
        per_kind = {}
        for n, fl in (("field_forward_train_normals", 2.0 * (macs["forward"] + macs["normals"])),
                      ("field_forward_train", 2.0 * macs["forward"]), ("field_backward", 2.0 * macs["backward"]),
                      ("field_backward_input", 2.0 * macs["backward_input"])):
            if calls(n):
                per_kind[n] = {"launches_per_step": calls(n) / steps, "avg_launch_ms": ms(n) / calls(n),
                               "achieved_tflops": fl * pts(n) / (ms(n) * 1e-3) / 1e12}
        rec["launch_kinds"] = per_kind
        rec["kernels"] = kern
        rec["algorithmic_flop_per_step"] = total_flop
        rec["flop_formula"] = (
            "per field point (MAC x 2): forward %d, analytic normals %d (primary levels), backward dX %d (%d where the "
            "gradient reaches pixel_area: reflect levels, get_inf_color), weight gradients %d; points per step = "
            "R*(Sc+Sf) primary + M*(Src+Srf+1) reflected with the MEASURED mean M = %.1f of R = %d" %
            (macs["forward"], macs["normals"], macs["backward"], macs["backward_input"], macs["wgrad"],
             state["M"] / float(steps), R))
        rec["end_to_end_tflops"] = total_flop / (elapsed / steps) / 1e12
        rec["end_to_end_frac_of_fp32_mfma_peak"] = rec["end_to_end_tflops"] / FP32_MFMA_PEAK_TFLOPS
        rec["other_ms_per_step"] = elapsed / steps * 1e3 - sum(k["ms_per_step"] for k in kern.values())
    return rec


def run_level(pkg, args, dev, steps, warmup, dog):
    """BASELINE configs[1] (secondary at N = 1): fused forward + composite of one sampling level, eval mode."""
    from reflect_sampling_nerf_amd import ops
    from reflect_sampling_nerf_amd._abi import RSN_SPACING_UNIFORM
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    R, S = args.rays, args.samples
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S,
                                            base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    model.to(dev).eval()
    fld = model.field
    fld.set_mma_mode(args.mma)
    o, d, pa = synthetic_rays(R, seed=0)
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears, fars = torch.full((R,), 2.0, device=dev), torch.full((R,), 6.0, device=dev)
    fld.packed_weights()
    flags = ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB

    def step(_i=None):
        sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
        lv = fld.evaluate_frustums(o, d, pa, eb, full=True)
        ops.composite(R, None, S, 1, flags, lv["sigma"], eb, lv["color"])
        dog.beat()

    for _ in range(warmup):
        step()
    timer = ops.KernelTimer()
    ops.TIMER = timer
    try:
        elapsed = timed_region(None, False, dev, steps, step)
    finally:
        ops.TIMER = None
    t = timer.totals()["field_forward_eval"]
    kms = t["ms"] / t["calls"]
    macs = algorithmic_macs(args.layers, args.width)
    flop = 2.0 * macs["forward"] * R * S
    achieved = flop / (kms * 1e-3) / 1e12
    mode_idx = {"f32": 0, "bf16x6": 1, "bf16x3": 2, "bf16": 3}[args.mma]
    kname = ("rsn_field_bf16_ring16_kernel" if args.width == 256 else "rsn_field_bf16_kernel<%d>" % (args.width // 32)) \
        if args.mma == "bf16" else \
        "rsn_field_kernel<%d, false, %d>" % (args.width // 32, mode_idx)
    # f32: algorithmic FLOP against the fp32-MFMA peak.  Split modes issue 6 (3) bf16 MFMA FLOP per algorithmic FLOP:
    # priced as issued bf16 FLOP against the bf16 dense peak.
    mult = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "bf16": 1}[args.mma]
    peak = FP32_MFMA_PEAK_TFLOPS if args.mma == "f32" else BF16_MFMA_PEAK_TFLOPS
    return {
        "workload": "%s: %d rays x %d samples, %d-layer %d-wide MLP, %s, forward + composite of one sampling level (eval): TWO launches, "
                    "the field kernel writes the per-sample outputs the reference's dict carries and rsn_composite_kernel (8 us) reads them"
                    % ("BASELINE configs[1]" if (R, S, args.mma) == (4096, 128, "f32") else
                       "BASELINE configs[3]" if (R, S, args.mma) == (16384, 192, "bf16") else "custom size",
                       R, S, args.layers, args.width,
                       {"f32": "fp32", "bf16x6": "fp32 emulated on bf16 MFMA (6 products)", "bf16x3": "bf16x3 split",
                        "bf16": "bf16 MFMA hidden GEMMs"}[args.mma]),
        "value": R * steps / elapsed, "unit": "rays/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
        "roofline": {"kernel": kname, "bound": "mfma", "achieved": achieved * mult, "algorithmic_tflops": achieved,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved * mult / peak,
                     "traffic": newest_profile_traffic(kname, R, S), "traffic_note": TRAFFIC_NOTE, "kernel_ms": kms,
                     "algorithmic_flop_per_launch": flop},
    }


def run_get_outputs(pkg, args, dev, steps, warmup, dog):
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    R, S = args.rays, args.samples
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S,
                                            base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).eval()
    model.field.set_mma_mode(args.mma)
    o, d, pa = synthetic_rays(R, seed=0)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(R, 1).to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    state = {}

    def step(_i=None):
        state["out"] = model(rb)
        dog.beat()

    for _ in range(warmup):
        step()
    elapsed = timed_region(None, False, dev, steps, step)
    return {"workload": "full eval get_outputs: %d rays x (%d coarse + %d fine + reflect 64 + 64), %dx%d field, M/R=%.2f"
                        % (R, S, S, args.layers, args.width, float(state["out"]["mask"].float().mean())),
            "value": R * steps / elapsed, "unit": "rays/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps}


def run_eval_image(pkg, args, dev, dog, side=800, chunks=(1024, 4096, 16384), repeats=2):
    """The eval-image path: a side x side image of camera rays through
    Model.get_outputs_for_camera_ray_bundle, `eval_num_rays_per_chunk` rays per forward -- the reference's 1024
    (config.py:41) and two larger chunk sizes -- full eval get_outputs per chunk (128 + 128 + reflect 64 + 64).  Every
    chunk is enqueued before anything is read back (no device-to-host read per chunk); rays/s = image rays / wall time
    of the whole image, best of `repeats`."""
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    torch.manual_seed(0)
    n = side * side
    o, d, pa = synthetic_rays(n, seed=3)
    rb = pkg.RayBundle(origins=o.reshape(side, side, 3).to(dev), directions=d.reshape(side, side, 3).to(dev),
                       pixel_area=pa.reshape(side, side, 1).to(dev))
    out = {"workload": "eval image: %d x %d camera rays through get_outputs_for_camera_ray_bundle (full eval get_outputs per "
                       "chunk: 128 + 128 + reflect 64 + 64 samples, %dx%d field, %s), no host read per chunk"
                       % (side, side, args.layers, args.width, args.mma), "unit": "rays/s", "chunks": {}}
    for chunk in chunks:
        cfg = pkg.ReflectSamplingNeRFModelConfig(base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width,
                                                eval_num_rays_per_chunk=chunk)
        model = cfg.setup(scene_box=None, num_train_data=1)
        with torch.no_grad():
            model.field.field_output_density.net.bias += 2.0
        model.to(dev).eval()
        model.field.set_mma_mode(args.mma)
        if os.environ.get("RSN_EVAL_SIDE_STREAMS"):  # A/B of the chunk overlap (default: three side streams)
            model.eval_side_streams = int(os.environ["RSN_EVAL_SIDE_STREAMS"])
        best = None
        for rep in range(repeats + 1):  # the first pass warms the allocator up and is not counted
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img = model.get_outputs_for_camera_ray_bundle(rb)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep >= 1:
                best = dt if best is None else min(best, dt)
            dog.beat()
        out["chunks"][str(chunk)] = {"rays_per_s": n / best, "ms_per_image": best * 1e3,
                                     "reflect_ray_fraction": float(img["mask"].float().mean())}
        del model, img
    out["value"] = out["chunks"][str(chunks[0])]["rays_per_s"]
    out["ratio_first_to_4096"] = out["chunks"][str(chunks[0])]["rays_per_s"] / out["chunks"].get("4096", out["chunks"][str(chunks[0])])["rays_per_s"]
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, popen=subprocess.Popen, grace=20.0, poll=0.2):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (the driver's command shape): this process
    touches NO GPU; it starts N fresh children of this same script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment: what torch.distributed.run would set; the reference's layout, one
    process per GPU under DDP, pipeline.py:72-77), relays rank 0's stdout (the one JSON line) and returns the WORST
    child exit code -- the watchdog's 3 and the exception path's 4 come through unchanged, a child ended by signal k
    reads as 128 + k.  Once one child has failed the others get `grace` seconds, then they are terminated by PID."""
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        procs.append(popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                           stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    relay = None
    if getattr(procs[0], "stdout", None) is not None:
        def _relay():
            for raw in procs[0].stdout:  # the JSON line goes to stdout, library chatter ("[Gloo] ...") to stderr
                text = raw.decode(errors="replace") if isinstance(raw, bytes) else raw
                dst = sys.stdout if text.lstrip().startswith("{") else sys.stderr
                dst.write(text)
                dst.flush()
        relay = threading.Thread(target=_relay, daemon=True)
        relay.start()
    codes = [None] * n
    failed_at = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                rc = p.poll()
                if rc is not None:
                    codes[i] = 128 - rc if rc < 0 else rc
                    if codes[i] != 0 and failed_at is None:
                        failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > grace:
            for i, p in enumerate(procs):  # the survivors sit in a collective that will never complete
                if codes[i] is None:
                    p.terminate()
            t_kill = time.time() + 5.0
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(poll)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                    p.wait()
                if codes[i] is None:
                    codes[i] = 5  # ended by the launcher after another rank failed
            break
        time.sleep(poll)
    if relay is not None:
        relay.join(timeout=10.0)
    worst = max(codes)
    if worst != 0:
        sys.stderr.write(json.dumps({"error": "bench.py --gpus %d: child exit codes %s" % (n, codes)}) + "\n")
    return worst


def run_selftest(rank, world):
    """--workload selftest: the launch plumbing without a GPU (CPU tests): process group over gloo, one all-reduce,
    rank 0 prints one line.  RSN_BENCH_SELFTEST_FAIL_RANK=r makes rank r die (SIGKILL) before the collective."""
    if os.environ.get("RSN_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        os.kill(os.getpid(), signal.SIGKILL)
    dog = Watchdog(rank, limit=float(os.environ.get("RSN_BENCH_WATCHDOG_S", "240")))
    total = float(rank + 1)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([total], dtype=torch.float64)
        dist.all_reduce(t)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    dog.done.set()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": total}), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's command shape: no launcher.  Start the ranks as fresh children BEFORE anything here touches the
        # GPU (never re-exec a process that initialised HIP) and exit with the worst child code.
        if args.workload != "selftest" and os.environ.get("RSN_BENCH_SHARE_GPU") != "1":
            have = visible_gpu_count()  # sysfs / environment only: the launcher never initialises HIP
            if have is not None and have < args.gpus:
                sys.stderr.write(json.dumps({"error": "bench.py --gpus %d: %d GPU(s) visible (RSN_BENCH_SHARE_GPU=1 "
                                                      "rehearses N ranks on one GPU over gloo)" % (args.gpus, have)}) + "\n")
                sys.exit(2)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload == "selftest":
        run_selftest(rank, world)
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    # rehearsal on a 1-GPU box: RSN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the real run is one rank per GPU over RCCL ("nccl" backend on ROCm).
    share = os.environ.get("RSN_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = None
    # RSN_BENCH_FORCE_COLLECTIVE=1: the N > 1 code path (process group on RCCL, gradient all-reduce on the side stream,
    # barriers, MAX over ranks) with ONE rank -- the rehearsal a 1-GPU box allows on the real backend
    force = os.environ.get("RSN_BENCH_FORCE_COLLECTIVE") == "1" and world == 1
    if force:
        os.environ.setdefault("MASTER_PORT", "29542")
    if world > 1 or force:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    dog = Watchdog(rank, limit=float(os.environ.get("RSN_BENCH_WATCHDOG_S", "240")))

    import reflect_sampling_nerf_amd as pkg

    pkg.load_library()
    R, S = args.rays, args.samples
    samples = (args.coarse or S, S, 64, 64)
    line = None
    try:
        if args.workload == "train":
            rec = run_train(pkg, args, dev, rank, world, dist, share, samples, args.steps, args.warmup, dog)
            if rank == 0:
                fwd = rec["kernels"]["forward"]
                line = {
                    "metric": "rays/sec (train step) at 4096 rays x 128 samples, 1/2/4/8 MI355X; PSNR vs ref",
                    "value": rec["value"], "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32" if args.mma == "f32" else "f32 (emulated: 3-way bf16 split, 6 bf16 MFMA products, f32 accumulate)",
                    "data": "synthetic",
                    "config": {
                        "workload": "training step per rank: %d rays x (%d coarse + %d fine + reflect %d + %d on the M "
                                    "reflected rays), %d-layer %d-wide field: forward (stratified jitter, analytic normals) "
                                    "+ 8-term loss + backward + %s + fused RAdam; M/R = %.2f"
                                    % (R, samples[0], samples[1], samples[2], samples[3], args.layers, args.width,
                                       ("ONE flat 618513-float gradient all-reduce (%s, %d ranks)" % (backend, world))
                                       if dist is not None else "no collective (N=1)", rec["reflect_ray_fraction"]),
                        "rays_per_gpu": R, "samples_per_ray": S, "global_rays_per_step": world * R,
                        **collective_config(dist, backend, world),
                        "weights": "random-init (nn.Linear default), seed 0, density bias +2",
                    },
                    "roofline": {
                        "kernel": fwd["kernel"] + " (training forward: saved activations + in-kernel analytic normals)",
                        "bound": "mfma", "achieved": fwd["achieved_tflops"], "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": fwd["frac_of_fp32_mfma_peak"],
                        "traffic": newest_profile_traffic(fwd["kernel"], R, S), "traffic_note": TRAFFIC_NOTE,
                        "avg_launch_ms": fwd["avg_launch_ms"], "launches_per_step": fwd["launches_per_step"],
                        "algorithmic_flop_per_step": fwd["algorithmic_flop_per_step"],
                        "share_of_step_time": fwd["ms_per_step"] / rec["ms_per_step"],
                    },
                    "train_step": {k: rec[k] for k in ("algorithmic_flop_per_step", "flop_formula", "kernels",
                                                       "end_to_end_tflops", "end_to_end_frac_of_fp32_mfma_peak",
                                                       "other_ms_per_step", "loss", "reflect_ray_fraction",
                                                       "host_syncs_in_reducer", "peak_device_memory_gb", "ray_chunk", "launch_kinds")},
                }
            if world == 1 and not args.no_secondary and args.mma == "f32":
                # secondary legs (untimed for the headline): configs[1] eval level and the configs[2] training shape
                lv = run_level(pkg, args, dev, args.steps, args.warmup, dog)
                c2 = run_train(pkg, args, dev, rank, world, None, share, (64, 128, 64, 64), min(args.steps, 10),
                               min(max(args.warmup, 1), 3), dog, time_kernels=False)
                x6_args = argparse.Namespace(**{**vars(args), "mma": "bf16x6"})
                x6 = run_train(pkg, x6_args, dev, rank, world, None, share, samples, min(args.steps, 10),
                               min(max(args.warmup, 1), 3), dog, time_kernels=True)
                bf_args = argparse.Namespace(**{**vars(args), "mma": "bf16"})
                bf = run_train(pkg, bf_args, dev, rank, world, None, share, samples, min(args.steps, 10),
                               min(max(args.warmup, 1), 3), dog, time_kernels=True)
                g2_args = argparse.Namespace(**{**vars(args), "wgrad_groups": 2})
                g2 = run_train(pkg, g2_args, dev, rank, world, None, share, samples, min(args.steps, 10),
                               min(max(args.warmup, 1), 3), dog, time_kernels=False)
                line["train_step_weight_grad_groups2"] = {
                    "workload": "the headline step with the reflect branch's weight gradients reduced before the primary levels' backward "
                                "sweeps and its buffers released (opt-in model.weight_grad_groups = 2: six more reduction launches, a "
                                "third less memory; same sample draws, gradients equal to the order of the atomics)",
                    **{k: g2[k] for k in ("value", "unit", "ms_per_step", "steps", "peak_device_memory_gb")}}
                line["eval_level"] = lv
                line["eval_image"] = run_eval_image(pkg, args, dev, dog)
                line["train_step_bf16_sweeps"] = {
                    "workload": "the headline step in REDUCED precision (opt-in; the reference itself trains under fp16 "
                                "autocast, config.py:33): forward (with the analytic-normal sweep) and backward sweeps on plain bf16 "
                                "MFMA operands with fp32 accumulation, weights shared per workgroup through an LDS ring "
                                "(rsn_field_bf16_train.hip); saved activations / layer gradients are bf16 rows, read as such by the "
                                "weight-gradient kernel (fp32 accumulation)",
                    **{k: bf[k] for k in ("value", "unit", "ms_per_step", "steps", "peak_device_memory_gb")},
                    "roofline": {"note": "two rooflines per kernel: algorithmic FLOP / time against the 2.5 PF dense bf16 MFMA peak, and "
                                         "algorithmic HBM bytes of the saved rows / time against 8 TB/s (HIP events on the launch stream)",
                                 **{k: {q: v[q] for q in ("kernel", "ms_per_step", "achieved_tflops", "frac_of_bf16_mfma_peak",
                                                          "algorithmic_hbm_bytes_per_step", "achieved_hbm_tbps", "frac_of_hbm_peak")}
                                    for k, v in bf["kernels"].items()}}}
                line["train_step_bf16x6_sweeps"] = {
                    "workload": "the headline step with the forward / backward sweeps on split-bf16 MFMA (3-way split, 6 "
                                "products, fp32 accumulate: fp32-equivalent, same parity bounds as the exact path; opt-in), weights "
                                "shared per workgroup through an LDS ring as three bf16 pieces (rsn_field_x6_train.hip); saved rows / "
                                "layer gradients are fp32; weight-gradient operands split the same way in-kernel",
                    **{k: x6[k] for k in ("value", "unit", "ms_per_step", "steps")},
                    "roofline": {"note": "per kernel: ISSUED bf16 MFMA work (6 x the algorithmic FLOP) / time against the 2.5 PF dense bf16 "
                                         "peak, and algorithmic HBM bytes of the fp32 saved rows / time against 8 TB/s",
                                 **{k: {q: v[q] for q in ("kernel", "ms_per_step", "achieved_tflops", "frac_of_bf16_mfma_peak",
                                                          "algorithmic_hbm_bytes_per_step", "achieved_hbm_tbps", "frac_of_hbm_peak")}
                                    for k, v in x6["kernels"].items()}}}
                line["train_step_configs2"] = {
                    "workload": "BASELINE configs[2]: %d rays x (64 coarse + 128 fine) + reflect 64 + 64, forward + backward "
                                "(+ loss + RAdam)" % R,
                    **{k: c2[k] for k in ("value", "unit", "ms_per_step", "steps", "reflect_ray_fraction")}}
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline_train(args, samples)
                dog.beat()
                if "eval_level" in line:
                    line["eval_level"]["cpu_baseline"] = cpu_baseline_level(args)
        else:
            assert world == 1, "--workload level / get_outputs are single-GPU measurements"
            if args.workload == "eval_image":
                rec = run_eval_image(pkg, args, dev, dog)
                rec["ms_per_step"] = rec["chunks"]["1024"]["ms_per_image"]
            else:
                rec = run_level(pkg, args, dev, args.steps, args.warmup, dog) if args.workload == "level" else \
                    run_get_outputs(pkg, args, dev, args.steps, args.warmup, dog)
            line = {
                "metric": "rays/sec (%s) at %d rays x %d samples, 1 MI355X" %
                          ("eval forward + composite of one level" if args.workload == "level" else
                           "eval image, 800 x 800, chunks of 1024 rays" if args.workload == "eval_image" else "eval get_outputs", R, S),
                "value": rec["value"], "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"f32": "f32", "bf16x6": "f32 (emulated: 3-way bf16 split, 6 bf16 MFMA products, f32 accumulate)",
                          "bf16x3": "bf16x3 (2-way bf16 split, f32 accumulate; reduced precision)",
                          "bf16": "bf16 (bf16 MFMA operands, f32 accumulate; BASELINE configs[3])"}[args.mma],
                "data": "synthetic",
                "config": {"workload": rec["workload"], "rays_per_gpu": R, "samples_per_ray": S, "parallelism": "dp1",
                           "weights": "random-init (nn.Linear default), seed 0"},
            }
            if "roofline" in rec:
                line["roofline"] = rec["roofline"]
            if "chunks" in rec:
                line["eval_image"] = rec
            if args.workload == "level" and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline_level(args)
    except BaseException:
        sys.stderr.write(json.dumps({"error": "bench.py failed on rank %d" % rank, "traceback": traceback.format_exc()}) + "\n")
        sys.stderr.flush()
        if world > 1:  # the other ranks may sit in a collective: do not wait for them in destroy_process_group
            os._exit(4)
        sys.exit(4)
    dog.done.set()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
