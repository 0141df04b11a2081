"""Data-parallel training glue: what the reference gets from torch DDP (reflect_sampling_nerf_pipeline.py:72-77).

One process per GPU, each rendering its own ray batch; the only exchange is the gradient average.  The Field
has 618,513 fp32 parameters (2.47 MB), so the collective is latency bound: ONE flat buffer, ONE all-reduce
(RCCL over xGMI when the backend is "nccl"), then a scale by 1/world -- exactly DDP's averaging, including its
`find_unused_parameters=True` behaviour: a parameter that received no gradient on this rank (field_output_low
always; the reflect-only path when no ray of this rank is reflected) contributes zeros, and a parameter unused on
EVERY rank keeps `grad is None`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Averages `.grad` of `params` across the process group with a single all-reduce of one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.sizes = [p.numel() for p in self.params]
        self.total = sum(self.sizes)
        self._flat: Optional[torch.Tensor] = None

    def _buffer(self, device, dtype):
        n = self.total + len(self.params)  # gradients + one "was used" flag per parameter
        if self._flat is None or self._flat.device != device or self._flat.dtype != dtype:
            self._flat = torch.zeros(n, device=device, dtype=dtype)
            self._views = [v.view_as(p) for v, p in zip(self._flat[: self.total].split(self.sizes), self.params)]
        return self._flat

    @torch.no_grad()
    def __call__(self) -> None:
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        p0 = self.params[0]
        flat = self._buffer(p0.device, p0.dtype)
        have = [p.grad is not None for p in self.params]
        # pack: one multi-tensor copy for the gradients that exist, zeros elsewhere, the flags in one transfer
        if not all(have):
            flat[: self.total].zero_()
        src = [p.grad for p, h in zip(self.params, have) if h]
        if src:
            torch._foreach_copy_([v for v, h in zip(self._views, have) if h], src)
        flat[self.total:].copy_(torch.tensor([1.0 if h else 0.0 for h in have], dtype=flat.dtype), non_blocking=True)
        if dist.get_backend(self.group) == "gloo" and flat.is_cuda:  # CPU rehearsal backend: stage through host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat[: self.total].mul_(1.0 / world)
        # unpack: gradients that exist locally are overwritten in one multi-tensor copy; only when this rank lacks
        # some does it need the summed flags (one host read) to tell "zeros" from "unused everywhere -> stays None"
        if src:
            torch._foreach_copy_(src, [v for v, h in zip(self._views, have) if h])
        if not all(have):
            used = flat[self.total:].cpu()
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


def train_step(model, ray_bundle, batch, optimizer, reducer: Optional[FlatGradAllReduce], step: int) -> torch.Tensor:
    """One optimisation step: warm-up -> forward -> get_loss_dict -> backward -> gradient average -> optimiser."""
    apply_loss_warmup(model, step)
    optimizer.zero_grad(set_to_none=True)
    outputs = model(ray_bundle)
    loss_dict = model.get_loss_dict(outputs, batch)
    loss = sum(loss_dict.values())
    loss.backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return loss.detach()
