"""Data-parallel training glue: what the reference gets from torch DDP (reflect_sampling_nerf_pipeline.py:72-77).

One process per GPU, each rendering its own ray batch; the only exchange is the gradient average.  The Field
has 618,513 fp32 parameters (2.47 MB), so the collective is latency bound: ONE flat buffer, ONE all-reduce
(RCCL over xGMI when the backend is "nccl"), then a scale by 1/world -- exactly DDP's averaging, including its
`find_unused_parameters=True` behaviour: a parameter that received no gradient on this rank contributes zeros, and
a parameter unused on EVERY rank keeps `grad is None`.

Steady state has no host work beyond Python bookkeeping: no host->device copy, no device->host read, no
synchronisation.  The all-reduce runs on the reducer's own stream (pack on the compute stream -> event ->
all-reduce + 1/world scale on the side stream -> event -> the compute stream waits before it unpacks), so anything
the caller enqueues that does not touch the gradients overlaps with the collective.

  * Which parameters never receive a gradient on any rank (field_output_low: built by field.py:67, never evaluated)
    is decided on the first call by one extra flag all-reduce; they are dropped from the flat buffer.  Should a dropped
    parameter receive a gradient on SOME rank later, that rank raises a "revive" flag in the tail of the flat buffer --
    the buffer every rank reduces every step anyway -- and drops the gradient for this one step; the reduced flag is
    copied to pinned host memory asynchronously and looked at one step late (its event has long fired: no stall), where
    EVERY rank sees the same value and takes the decision again in the same call, the new live set united with the old
    one.  No rank-local condition ever selects which collective is issued (a mismatch would hang RCCL).
  * The per-step "was used" flags of the live parameters travel in the tail of the same flat buffer.  They are kept on
    the device (one cached tensor per distinct None-pattern) and are only read back on a step where THIS rank lacks a
    gradient some other rank may have produced -- which the reflect-sampling model never does (train_graph.py hands
    every live parameter a gradient even when no ray of the rank is reflected, like the reference: model.py:240-241).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Averages `.grad` of `params` across the process group with a single all-reduce of one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, side_stream: bool = True):
        self.all_params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.params: Optional[List[torch.nn.Parameter]] = None  # live parameters, fixed by the first call
        self.group = process_group
        self.use_side_stream = side_stream
        self._flat: Optional[torch.Tensor] = None
        self._flags: Dict[Tuple[bool, ...], torch.Tensor] = {}
        self._stream = None
        self.host_syncs = 0  # device->host reads issued by the reducer (tests assert it stays at the one-time 1)
        self.run_single_rank = False  # tests: issue the collective in a one-rank group too (RCCL + side stream on one GPU)

    # ------------------------------------------------------------------------------------------------ one-time set-up
    def _decide_live(self) -> None:
        """Parameters without a gradient on EVERY rank are statically unused -> never reduced.  Taken on the first call and
        again -- by all ranks in the same call -- one step after any rank has flagged a gradient on a dropped parameter;
        a parameter once live stays live (a step without its gradient contributes zeros)."""
        p0 = self.all_params[0]
        was_live = set() if self.params is None else {id(p) for p in self.params}
        have = torch.tensor([1.0 if (p.grad is not None or id(p) in was_live) else 0.0 for p in self.all_params],
                            dtype=torch.float32)
        gloo = dist.get_backend(self.group) == "gloo"
        t = have if gloo else have.to(p0.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        used = t.cpu()
        self.host_syncs += 1
        self.params = [p for p, u in zip(self.all_params, used.tolist()) if u > 0.0]
        self._dropped = [p for p, u in zip(self.all_params, used.tolist()) if u <= 0.0]
        self.sizes = [p.numel() for p in self.params]
        self.total = sum(self.sizes)
        n = self.total + len(self.params) + 1  # gradients + one "was used" flag per live parameter + the revive flag
        self._flat = torch.zeros(n, device=p0.device, dtype=p0.dtype)
        self._views = [v.view_as(p) for v, p in zip(self._flat[: self.total].split(self.sizes), self.params)]
        self._flags = {}
        self._revive_host = torch.zeros(1, dtype=p0.dtype, pin_memory=p0.is_cuda)  # last step's reduced revive flag
        self._revive_ev = torch.cuda.Event() if p0.is_cuda else None
        self._revive_pending = False
        if self.use_side_stream and p0.is_cuda and self._stream is None:
            self._stream = torch.cuda.Stream(device=p0.device)
            self._packed_ev = torch.cuda.Event()
            self._reduced_ev = torch.cuda.Event()

    def _flag_tensor(self, have: Tuple[bool, ...]) -> torch.Tensor:
        t = self._flags.get(have)
        if t is None:  # one host->device copy per distinct pattern, ever
            t = torch.tensor([1.0 if h else 0.0 for h in have], dtype=self._flat.dtype).to(self._flat.device)
            self._flags[have] = t
        return t

    # ------------------------------------------------------------------------------------------------ every step
    @torch.no_grad()
    def __call__(self) -> None:
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1 and not self.run_single_rank:
            return
        if self.params is None:
            self._decide_live()
        elif self._revive_pending:
            # last step's REDUCED revive flag (the same number on every rank): copied to pinned memory behind that step's
            # all-reduce, its event fired long ago.  Non-zero: some rank saw a gradient on a dropped parameter -> every rank
            # takes the decision again, here, together.
            if self._revive_ev is not None:
                self._revive_ev.synchronize()
            self._revive_pending = False
            if float(self._revive_host[0]) > 0.0:
                self._decide_live()
        revive = False
        for p in self._dropped:
            if p.grad is not None:  # host-only check.  Only this rank may have it: not applied this step (replicas stay
                p.grad = None       # identical); flagged, and reduced from the next step on
                revive = True
        flat = self._flat
        have = tuple(p.grad is not None for p in self.params)
        complete = all(have)
        # pack (compute stream): one multi-tensor copy for the gradients that exist, zeros elsewhere, flags device-side
        if not complete:
            flat[: self.total].zero_()
        src = [p.grad for p, h in zip(self.params, have) if h]
        dst = self._views if complete else [v for v, h in zip(self._views, have) if h]
        if src:
            torch._foreach_copy_(dst, src)
        flat[self.total:].copy_(self._flag_tensor(have + (revive,)))
        gloo_cuda = dist.get_backend(self.group) == "gloo" and flat.is_cuda
        if gloo_cuda:  # CPU rehearsal backend (two ranks sharing one GPU): stage through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
            flat[: self.total].mul_(1.0 / world)
        elif self._stream is not None:
            cur = torch.cuda.current_stream(flat.device)
            self._packed_ev.record(cur)
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(self._packed_ev)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)  # RCCL; its work is ordered on this stream
                flat[: self.total].mul_(1.0 / world)
                self._reduced_ev.record(self._stream)
            cur.wait_event(self._reduced_ev)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat[: self.total].mul_(1.0 / world)
        # the reduced revive flag goes to pinned host memory asynchronously; the NEXT call looks at it
        if flat.is_cuda:
            self._revive_host.copy_(flat[-1:], non_blocking=True)
            self._revive_ev.record(torch.cuda.current_stream(flat.device))
        else:
            self._revive_host.copy_(flat[-1:])
        self._revive_pending = True
        # unpack: gradients that exist locally are overwritten in one multi-tensor copy
        if src:
            torch._foreach_copy_(src, dst)
        if not complete:
            # this rank lacks a gradient: the summed flags tell "zeros from me, data from others" apart from "unused
            # on every rank this step -> stays None" (DDP, find_unused_parameters=True).  The only host read.
            used = flat[self.total:-1].cpu()
            self.host_syncs += 1
            for i, (p, h) in enumerate(zip(self.params, have)):
                if not h and float(used[i]) > 0.0:
                    p.grad = self._views[i].clone()


def apply_loss_warmup(model, step: int) -> None:
    """reflect_sampling_nerf_pipeline.py:79-91: the four normal/orientation coefficients are 0 for step < 50."""
    c = model.config.loss_coefficients
    if step < 50:
        for k in ("predicted_normal_loss_coarse", "predicted_normal_loss_fine", "orientation_loss_coarse",
                  "orientation_loss_fine"):
            c[k] = 0.0
    else:
        c["predicted_normal_loss_coarse"] = 3e-5
        c["predicted_normal_loss_fine"] = 3e-4
        c["orientation_loss_coarse"] = 1e-2
        c["orientation_loss_fine"] = 1e-1


_MEAN_TERMS = ("loss_mid_coarse", "loss_mid_fine", "loss_reflect_mid_coarse", "loss_reflect_mid_fine")


def train_step(model, ray_bundle, batch, optimizer, reducer: Optional[FlatGradAllReduce], step: int,
               ray_chunk: Optional[int] = None) -> torch.Tensor:
    """One optimisation step: warm-up -> forward -> get_loss_dict -> backward -> gradient average -> optimiser.

    ray_chunk: bound the step's activation memory.  The training forward keeps ~10.4 KB per sample for the backward
    pass and the backward sweep writes ~9.8 KB per sample of layer gradients for the weight-gradient kernels (DESIGN
    4.3; half of that in the bf16 mode).  The reflect branch's buffers are sized for ALL R rays whatever the number M of
    reflected ones is (its launches take M from device memory, the host never reads it): 30.2 GB peak at 4096 rays x
    (128 + 128 + reflect 64 + 64) in fp32, 15.9 GB in the bf16 mode, proportionally more for larger batches.  With ray_chunk = n
    the batch is walked in chunks of n rays -- forward, loss, backward per chunk, the parameter gradients ACCUMULATING
    (autograd adds into .grad; the weight-gradient kernels accumulate anyway) -- so the live activations are those of one
    chunk whatever the batch size.  The result is the whole-batch step exactly: the four MSE terms are means over the
    batch (a chunk of m of R rays enters with weight m/R), the normal / orientation terms are sums (model.py:395-407);
    the gradient all-reduce and the optimiser run once, after the last chunk."""
    apply_loss_warmup(model, step)
    optimizer.zero_grad(set_to_none=True)
    R = ray_bundle.origins.shape[0]
    n_masked = None  # reflected rays of the step, summed over the chunks ON THE DEVICE (model._step_n_masked_dev)
    if not ray_chunk or ray_chunk >= R:
        outputs = model(ray_bundle)
        loss_dict = model.get_loss_dict(outputs, batch)
        loss = sum(loss_dict.values())
        loss.backward()
        total = loss.detach()
        n_masked = getattr(model, "_last_n_masked_dev", None)
    else:
        total = None
        for lo in range(0, R, ray_chunk):
            hi = min(lo + ray_chunk, R)
            outputs = model(ray_bundle[lo:hi])
            loss_dict = model.get_loss_dict(outputs, {"image": batch["image"][lo:hi]})
            w = (hi - lo) / float(R)
            loss = sum(v * w if k in _MEAN_TERMS else v for k, v in loss_dict.items())
            loss.backward()
            total = loss.detach() if total is None else total + loss.detach()
            nm = getattr(model, "_last_n_masked_dev", None)
            if nm is not None:
                n_masked = nm.clone() if n_masked is None else n_masked + nm
            del outputs, loss_dict, loss  # this chunk's activations are released before the next chunk's forward
    model._step_n_masked_dev = n_masked
    if reducer is not None:
        reducer()
    optimizer.step()
    return total
