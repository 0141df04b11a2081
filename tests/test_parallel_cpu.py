"""world_size-2 gloo test (CPU) of the data-parallel gradient average: equals DDP's mean over ranks, tolerates
parameters without gradient on one rank, leaves globally-unused parameters at grad=None."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reflect_sampling_nerf_amd.parallel import FlatGradAllReduce, apply_loss_warmup


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)),
              torch.nn.Parameter(torch.randn(2, 2)), torch.nn.Parameter(torch.randn(4))]
    g = torch.Generator().manual_seed(100 + rank)
    params[0].grad = torch.randn(5, 3, generator=g)
    params[1].grad = torch.randn(7, generator=g)
    if rank == 0:  # parameter 2 only receives a gradient on rank 0 (reflect branch skipped on rank 1)
        params[2].grad = torch.randn(2, 2, generator=g)
    # parameter 3 is unused everywhere (field_output_low)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    reducer = FlatGradAllReduce(params)
    reducer()
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    ok = True
    for i, p in enumerate(params):
        gs = [gg[i] for gg in gathered]
        if all(x is None for x in gs):
            ok &= p.grad is None
        else:
            mean = sum(torch.zeros_like(p) if x is None else x for x in gs) / world
            ok &= p.grad is not None and torch.allclose(p.grad, mean, atol=1e-7)
    # the statically unused parameter was dropped from the flat buffer by the one-time decision
    ok &= len(reducer.params) == 3 and reducer.total == 15 + 7 + 4
    # steady state (every live parameter has a gradient on every rank, the reflect-sampling model's case): further
    # steps average correctly and issue NO device->host read -- the one-time decision stays the only one on rank 0
    syncs_before = reducer.host_syncs
    for step in range(3):
        for i in range(3):
            params[i].grad = torch.full_like(params[i], float(rank + 1 + step))
        reducer()
        for i in range(3):
            ok &= torch.allclose(params[i].grad, torch.full_like(params[i], 1.5 + step))
        ok &= params[3].grad is None
    ok &= reducer.host_syncs == syncs_before
    # a parameter dropped as "never used" receives a gradient later ON ONE RANK ONLY (rank 1).  No rank-local condition may
    # select the collective (the other rank would sit in a different all-reduce: a hang under RCCL): rank 1 raises the revive
    # flag inside the flat buffer both ranks reduce anyway and drops that gradient for this one step (replicas stay identical) ...
    for i in range(3):
        params[i].grad = torch.full_like(params[i], float(rank))
    if rank == 1:
        params[3].grad = torch.full_like(params[3], 8.0)
    reducer()
    ok &= len(reducer.params) == 3 and params[3].grad is None
    ok &= torch.allclose(params[0].grad, torch.full_like(params[0], 0.5))
    ok &= reducer.host_syncs == syncs_before
    # ... and one step later BOTH ranks see the reduced flag, take the decision again together (one more flag all-reduce: the one
    # extra host read) and average the revived parameter from then on -- also on a rank that has no gradient for it
    for i in range(3):
        params[i].grad = torch.full_like(params[i], float(rank))
    if rank == 1:
        params[3].grad = torch.full_like(params[3], 8.0)
    reducer()
    ok &= len(reducer.params) == 4 and params[3].grad is not None and torch.allclose(params[3].grad, torch.full_like(params[3], 4.0))
    ok &= torch.allclose(params[0].grad, torch.full_like(params[0], 0.5))
    ok &= reducer.host_syncs <= syncs_before + 2  # the decision (+ rank 0's read of the used-flags: it lacks a gradient rank 1 has)
    # a live parameter without a gradient on every rank this step keeps None; it stays live (no re-decision ping-pong)
    for i in range(3):
        params[i].grad = torch.full_like(params[i], 1.0)
    params[3].grad = None
    n_sync = reducer.host_syncs
    reducer()
    ok &= len(reducer.params) == 4 and params[3].grad is None
    ok &= reducer.host_syncs == n_sync + 1  # the used-flags read of a rank that lacks a live gradient, not a new decision
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_loss_warmup_schedule():
    class M:
        class config:
            loss_coefficients = {"predicted_normal_loss_coarse": 3e-5, "predicted_normal_loss_fine": 3e-4,
                                 "orientation_loss_coarse": 1e-2, "orientation_loss_fine": 1e-1, "loss_mid_fine": 1.0}

    apply_loss_warmup(M, 0)
    assert M.config.loss_coefficients["orientation_loss_fine"] == 0.0 and M.config.loss_coefficients["loss_mid_fine"] == 1.0
    apply_loss_warmup(M, 50)
    assert M.config.loss_coefficients["predicted_normal_loss_fine"] == 3e-4
    assert M.config.loss_coefficients["orientation_loss_coarse"] == 1e-2
