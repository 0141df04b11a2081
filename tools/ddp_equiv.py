#!/usr/bin/env python3
"""DDP equivalence of the data-parallel training step (reference: pipeline.py:72-77, DistributedDataParallel with
find_unused_parameters=True): two ranks each run ONE real model training step (training-mode get_outputs, the eight
loss terms, backward) on disjoint 512-ray shards and average their gradients with parallel.FlatGradAllReduce; rank 0
then recomputes both shards' gradients single-process and checks that every post-reduce gradient equals the mean of
the two per-shard gradients (<= 1e-5 of the tensor's largest entry), `grad is None` pattern included.

Launch BEFORE anything else touches the GPU (one-GPU box: both ranks share cuda:0, gloo carries the collective):
    RSN_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29517 tools/ddp_equiv.py [--json gpurun_out/ddp_equiv.json]
On a multi-GPU node drop RSN_BENCH_SHARE_GPU: one rank per GPU over RCCL ("nccl").
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--samples", type=int, nargs=4, default=[32, 32, 16, 16])
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    share = os.environ.get("RSN_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if share:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce, apply_loss_warmup
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    R, S = args.rays, args.samples
    torch.manual_seed(0)  # identical replicas
    cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=S[0], num_importance_samples=S[1],
                                            num_reflect_coarse_samples=S[2], num_reflect_importance_samples=S[3],
                                            base_mlp_num_layers=args.layers, base_mlp_layer_width=args.width)
    model = cfg.setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    apply_loss_warmup(model, 100)
    params = model.get_param_groups()["fields"]
    names = [n for n, _ in model.field.named_parameters()]

    def shard_gradients(shard):
        """forward + loss + backward on shard `shard` (its own rays, target pixels and jitter stream)."""
        o, d, pa = synthetic_rays(R, seed=shard)
        rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.to(dev),
                           nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
        image = torch.rand(R, 3, generator=torch.Generator().manual_seed(777 + shard)).to(dev)
        for p in params:
            p.grad = None
        torch.manual_seed(1000 + shard)  # the samplers' stratified jitter (torch.rand on the device) is reproducible
        out = model(rb)
        loss = sum(model.get_loss_dict(out, {"image": image}).values())
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), int(out["mask"].sum())

    loss, M = shard_gradients(rank)
    reducer = FlatGradAllReduce(params)
    reducer()
    torch.cuda.synchronize()
    reduced = [None if p.grad is None else p.grad.detach().clone() for p in params]
    dist.barrier()
    result = None
    if rank == 0:
        per_shard = []
        for shard in range(world):
            shard_gradients(shard)
            per_shard.append([None if p.grad is None else p.grad.detach().clone() for p in params])
        rows, worst, ok = [], 0.0, True
        for i, name in enumerate(names):
            gs = [ps[i] for ps in per_shard]
            if all(g is None for g in gs):
                same = reduced[i] is None
                rows.append({"param": name, "pattern": "None on every rank", "ok": same})
                ok &= same
                continue
            mean = sum(torch.zeros_like(params[i]) if g is None else g for g in gs) / world
            if reduced[i] is None:
                rows.append({"param": name, "pattern": "missing after reduce", "ok": False})
                ok = False
                continue
            scale = float(mean.abs().max())
            err = float((reduced[i] - mean).abs().max()) / (scale + 1e-300)
            worst = max(worst, err)
            good = err <= 1e-5
            ok &= good
            rows.append({"param": name, "max_abs_err_over_tensor_max": err, "tensor_max": scale, "ok": good})
        result = {"tool": "tools/ddp_equiv.py", "world": world, "backend": dist.get_backend(),
                  "shared_gpu": share, "rays_per_rank": R, "samples": S, "field": f"{args.layers}x{args.width}",
                  "reflected_rays_rank0": M, "loss_rank0": loss,
                  "bound": "post-reduce gradient == mean of per-shard gradients, <= 1e-5 of the tensor max; None pattern equal",
                  "worst_rel_err": worst, "reducer_host_syncs": reducer.host_syncs,
                  "live_parameters": len(reducer.params), "all_ok": bool(ok), "per_parameter": rows}
        print(json.dumps({k: v for k, v in result.items() if k != "per_parameter"}))
        if args.json:
            os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
            with open(args.json, "w") as f:
                json.dump(result, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not result["all_ok"]:
        sys.exit(1)


if __name__ == "__main__":
    main()
