#!/usr/bin/env python3
"""bench.py -- rays/s of the reflect-sampling-nerf hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1], the configuration the metric is quoted on):
    4096 rays x 128 samples, 8-layer 256-wide trunk, fp32, fused forward + composite of one
    sampling level: uniform sampler -> fused field kernel (IPE, trunk, heads, SH, mid MLP, colour)
    -> per-ray compositing (weights, RGB on white, accumulation, median depth).
    Synthetic camera-shell rays, random-init weights, inputs resident in HBM.
With --gpus N every rank renders its own 4096-ray batch (weak scaling; rays are independent, the forward
path has no data-path collective -- the gradient all-reduce belongs to the training step).

One JSON line is printed by rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel (rsn_field_kernel), MFMA-bound: algorithmic FLOP (1,230,592 per sample,
                  SURVEY §8(d)) / its average duration, measured live with HIP events on its stream;
  cpu_baseline -- the CPU oracle (oracle/cpu_ref.py, a port of the reference's PyTorch op sequence) timed on the
                  host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = 1_230_592  # 2 x 615,296 MAC at W=256, L=8
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--workload", default="level", choices=["level", "get_outputs", "train"],
                    help="level = BASELINE configs[1] (default); get_outputs = full eval get_outputs "
                         "(coarse+fine+reflect); train = BASELINE configs[2]: full training step (forward, loss, "
                         "backward, gradient all-reduce, RAdam) with 64 coarse + 128 fine + reflect 64+64 samples")
    ap.add_argument("--mma", default="f32", choices=["f32", "bf16x6", "bf16x3", "bf16"],
                    help="matrix-core arithmetic of the eval field kernel: f32 = exact fp32 MFMA (default); bf16x6 = "
                         "fp32 emulation by 3-way bf16 splits (fp32-equivalent results); bf16x3 = reduced precision")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays of the bounded CPU-baseline sample (~10 s of CPU work)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle's single-level render on the host cores, bounded sample (cpu_rays x samples)."""
    from oracle import cpu_ref

    # host cores this process may use: the GPU box gives a 1-GPU job a 16-core share of a 256-core host;
    # more threads than that only oversubscribes (measured: 256 threads = 18 rays/s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, int(os.environ.get("RSN_CPU_THREADS", "16")))))
    fs = cpu_ref.FieldSpec(num_layers=args.layers, width=args.width)
    P = cpu_ref.init_params(fs, seed=0)
    Rc = args.cpu_rays
    o, d, pa = cpu_ref.synthetic_rays(Rc, seed=0)
    nears, fars = torch.full((Rc, 1), 2.0), torch.full((Rc, 1), 6.0)
    with torch.no_grad():
        cpu_ref.render_level(P, fs, o, d, pa, nears, fars, args.samples)  # warm-up
        times = []
        t_end = time.time() + 20.0
        while len(times) < 3 or (time.time() < t_end and len(times) < 10):
            t0 = time.perf_counter()
            cpu_ref.render_level(P, fs, o, d, pa, nears, fars, args.samples)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": Rc / med,
        "unit": "rays/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{Rc} rays x {args.samples} samples, same field ({args.layers}x{args.width}) and level, "
                  f"median of {len(times)} runs, oracle/cpu_ref.render_level (eager PyTorch fp32)",
    }


def train_leg(pkg, args, dev, rank, world, dist, share):
    """Secondary measurement of the default run: BASELINE configs[2]/[4], one full training step per rank (forward,
    twelve loss terms, backward, ONE flat gradient all-reduce over RCCL when N > 1, fused RAdam), timed with the same
    barrier + synchronize + max-over-ranks bracket.  Reported as `train_step` next to the headline; never `value`."""
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays
    from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce, train_step

    R = args.rays
    torch.manual_seed(0)
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=64, num_importance_samples=128,
                                            base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0  # so that the reflect branch is exercised
    model.to(dev).train()
    o, d, pa = synthetic_rays(R, seed=rank)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(R, 1).to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    params = model.get_param_groups()["fields"]
    optimizer = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)
    reducer = FlatGradAllReduce(params) if world > 1 else None
    batch = {"image": torch.rand(R, 3, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)}
    steps, warmup, it = min(args.steps, 20), min(max(args.warmup, 1), 3), 100  # past the 50-step loss warm-up
    for _ in range(warmup):
        train_step(model, rb, batch, optimizer, reducer, it)
        it += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(model, rb, batch, optimizer, reducer, it)
        it += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    with torch.no_grad():  # rays that take the reflect branch after the timed steps (training-mode forward, untimed)
        mask_frac = float(model(rb)["mask"].float().mean())
    return {"value": world * R * steps / elapsed, "unit": "rays/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "warmup": warmup, "n_gpus": world, "dtype": "f32", "loss": float(loss), "reflect_ray_fraction": mask_frac,
            "workload": "BASELINE configs[2]: %d rays x (64 coarse + 128 fine) + reflect (64 + 64) per rank, forward + "
                        "12-term loss + backward + %s + fused RAdam" %
                        (R, "one flat 618513-float gradient all-reduce (RCCL)" if world > 1 else "no collective (N=1)")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    # rehearsal on a 1-GPU box: RSN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the real run is one rank per GPU over RCCL ("nccl" backend on ROCm).
    share = os.environ.get("RSN_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays
    from reflect_sampling_nerf_amd import ops
    from reflect_sampling_nerf_amd._abi import RSN_SPACING_UNIFORM

    pkg.load_library()
    torch.manual_seed(0)  # identical random-init weights on every rank (data parallel replicas)
    R, S = args.rays, args.samples
    if args.workload == "train":
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=64, num_importance_samples=128,
                                                base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    else:
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S, num_importance_samples=S,
                                                base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    if args.workload in ("get_outputs", "train"):
        with torch.no_grad():
            model.field.field_output_density.net.bias += 2.0  # so that the reflect branch is exercised
    model.to(dev).eval()
    if args.workload == "train":
        model.train()
    fld = model.field
    fld.set_mma_mode(args.mma)
    o, d, pa = synthetic_rays(R, seed=rank)  # each rank renders its own rays
    o, d, pa = o.to(dev), d.to(dev), pa.reshape(R).to(dev)
    nears = torch.full((R,), 2.0, device=dev)
    fars = torch.full((R,), 6.0, device=dev)
    rb = pkg.RayBundle(origins=o, directions=d, pixel_area=pa[:, None], nears=nears[:, None], fars=fars[:, None])
    fld.packed_weights()
    flags = ops.RSN_COMP_EVAL | ops.RSN_COMP_CLIP_RGB
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    state = {}

    if args.workload == "train":
        from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce, train_step

        params = model.get_param_groups()["fields"]
        optimizer = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15, lr_final=1e-4, max_steps=50000)  # config.py:50-53
        reducer = FlatGradAllReduce(params) if world > 1 else None
        g = torch.Generator().manual_seed(1234 + rank)
        batch = {"image": torch.rand(R, 3, generator=g).to(dev)}
        state["it"] = 100  # past the 50-step loss warm-up: all twelve loss terms are live

    def step(i=None):
        if args.workload == "train":
            state["loss"] = train_step(model, rb, batch, optimizer, reducer, state["it"])
            state["it"] += 1
            return
        if args.workload == "get_outputs":
            state["out"] = model(rb)
            return
        sb, eb = ops.sample_spaced(R, None, S, RSN_SPACING_UNIFORM, 1.0, nears, fars, None)
        if i is not None:
            ev[i][0].record()
        lv = fld.evaluate_frustums(o, d, pa, eb, full=True)
        if i is not None:
            ev[i][1].record()
        state["out"] = ops.composite(R, None, S, 1, flags, lv["sigma"], eb, lv["color"])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.workload == "level" and args.mma == "f32" and world == 1:
        # informational: the same workload with the fp32-emulating split-bf16 matrix-core mode (results agree with
        # the exact-fp32 path to ~2e-7, tests/test_gpu_parity.py); `value` above stays the exact-fp32 number
        alt = {}
        for mode in ("bf16x6",):
            fld.set_mma_mode(mode)
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            alt[mode] = {"value": R * args.steps / dt, "unit": "rays/s", "ms_per_step": dt / args.steps * 1e3,
                         "note": "fp32 emulation: 3-way bf16 split, 6 bf16 MFMA products, f32 accumulate"}
        fld.set_mma_mode("f32")
        state["alt"] = alt
    if args.workload != "level":
        with torch.no_grad():
            model.eval()
            state["mask_frac"] = float(model(rb)["mask"].float().mean())
    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * R * args.steps / elapsed
        line = {
            "metric": "rays/sec (train step) at 4096 rays x 128 samples, 1/2/4/8 MI355X; PSNR vs ref",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (emulated: 3-way bf16 split, 6 bf16 MFMA products, f32 accumulate)",
                      "bf16x3": "bf16x3 (2-way bf16 split, f32 accumulate; reduced precision)",
                      "bf16": "bf16 (bf16 MFMA operands, f32 accumulate; BASELINE configs[3])"}[args.mma],
            "data": "synthetic",
            "config": {
                "workload": ("%s: %d rays x %d samples, %d-layer %d-wide MLP, %s, fused forward + composite of one "
                             "sampling level (eval)" %
                             ("BASELINE configs[1]" if (R, S, args.mma) == (4096, 128, "f32") else
                              "BASELINE configs[3]" if (R, S, args.mma) == (16384, 192, "bf16") else "custom size",
                              R, S, args.layers, args.width,
                              {"f32": "fp32", "bf16x6": "fp32 emulated on bf16 MFMA (6 products)",
                               "bf16x3": "bf16x3 split", "bf16": "bf16 MFMA hidden GEMMs"}[args.mma]))
                if args.workload == "level" else
                ("full %s: %d rays x (%d coarse + %d fine + reflect %d + %d), %dx%d field, M/R=%.2f" %
                 ("training step (forward+loss+backward+grad all-reduce+RAdam, BASELINE configs[2])"
                  if args.workload == "train" else "eval get_outputs",
                  R, cfg.num_coarse_samples, cfg.num_importance_samples, cfg.num_reflect_coarse_samples,
                  cfg.num_reflect_importance_samples, args.layers, args.width,
                  float(state.get("mask_frac", float("nan"))))),
                "rays_per_gpu": R,
                "samples_per_ray": S,
                "parallelism": ("dp%d (independent ray batches, no data-path collective)" % world) if args.workload != "train"
                else ("dp%d (one flat %d-float gradient all-reduce per step over RCCL)" % (world, 618513)),
                "weights": "random-init (nn.Linear default), seed 0",
            },
        }
        if args.workload == "level":
            kms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            flop = FLOP_PER_SAMPLE * R * S if (args.layers, args.width) == (8, 256) else None
            if flop is not None:
                achieved = flop / (kms * 1e-3) / 1e12
                traffic = None  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
                kname = "rsn_field_kernel<8, false, %d>" % {"f32": 0, "bf16x6": 1, "bf16x3": 2, "bf16": 3}[args.mma]
                try:  # newest summary (by name) that was taken on this kernel instantiation and this workload size
                    for fn in sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_summary.json")):
                        with open(os.path.join(REPO, "profiles", fn)) as fh:
                            js = json.load(fh)
                        if js.get("kernel") == kname and js.get("rays", 4096) == R and js.get("samples", 128) == S \
                                and js.get("hbm_traffic_bytes_per_launch") is not None:
                            traffic = js["hbm_traffic_bytes_per_launch"]
                except (OSError, ValueError):
                    pass
                # f32: algorithmic FLOP against the fp32-MFMA peak.  Split modes issue 6 (3) bf16 MFMA FLOP per
                # algorithmic FLOP: priced as issued bf16 FLOP against the bf16 dense peak.
                mult = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "bf16": 1}[args.mma]
                peak = FP32_MFMA_PEAK_TFLOPS if args.mma == "f32" else BF16_MFMA_PEAK_TFLOPS
                line["roofline"] = {
                    "kernel": "rsn_field_kernel<8,false,%d>" % {"f32": 0, "bf16x6": 1, "bf16x3": 2, "bf16": 3}[args.mma],
                    "bound": "mfma",
                    "achieved": achieved * mult,
                    "algorithmic_tflops": achieved,
                    "peak": peak,
                    "unit": "TFLOP/s",
                    "frac": achieved * mult / peak,
                    "traffic": traffic,
                    "traffic_note": "HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction), separate --pmc "
                                    "passes of tools/profile_round.sh; newest profiles/*_summary.json taken on this "
                                    "kernel instantiation and workload size (null if none)",
                    "kernel_ms": kms,
                    "algorithmic_flop_per_launch": flop,
                }
        if "alt" in state:
            line["alt_mma_modes"] = state["alt"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
    if args.workload == "level" and args.mma == "f32" and not args.no_train_leg:
        # Secondary leg, every rank.  It must never cost the headline line: a watchdog prints the line without it
        # and ends the process if the leg does not come back (a rank lost inside the gradient all-reduce).
        import threading

        finished = threading.Event()

        def watchdog():
            if finished.wait(timeout=240.0):
                return
            if rank == 0:
                line["train_step"] = {"error": "training leg did not finish within 240 s; headline unaffected"}
                print(json.dumps(line), flush=True)
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            res = train_leg(pkg, args, dev, rank, world, dist, share)
        except Exception as exc:
            res = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if world > 1:  # the other ranks may be waiting in a collective: leave it to their watchdogs
                finished.set()
                if rank == 0:
                    line["train_step"] = res
                    print(json.dumps(line), flush=True)
                os._exit(0)
        finished.set()
        if rank == 0:
            line["train_step"] = res
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
