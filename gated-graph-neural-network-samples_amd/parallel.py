"""Data parallelism over graph batches: one process per GPU, torch.distributed (backend "nccl" == RCCL
over xGMI on ROCm; "gloo" for the CPU tests of the reduction math).

The reference has no distributed code.  A batch is a disjoint union of molecule
graphs (chem_tensorflow_sparse.py:278-350), so propagation needs no communication; only the weights
are shared.  Per optimisation step there is exactly ONE sum all-reduce over ONE flat fp32 buffer
holding every gradient (591,802 floats = 2.37 MB for the default model) plus the per-task mask counts
-- at this size the collective is latency-bound on xGMI, so a single call beats any bucketing.

Loss normalisation (chem_tensorflow.py:161-169): loss = sum_g 0.5*diff_g^2 / (sum_g mask_g + 1e-7) over
the WHOLE batch.  With the batch sharded, rank r back-propagates  num_r / (sum_r den_r + 1e-7); the
SUM all-reduce of those gradients is exactly the gradient of the unsharded loss, for any shard sizes
(averaging per-rank losses would be wrong for unequal graph counts).  Per-variable clip and Adam then
run redundantly on every rank (chem_tensorflow.py:183-191).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .utils import SMALL_NUMBER


class DataParallelContext:
    def __init__(self, rank: int, world_size: int, device: torch.device, group=None, force_collectives: bool = False):
        self.rank, self.world_size, self.device, self.group = rank, world_size, device, group
        # force_collectives (GGNN_FORCE_COLLECTIVES=1): issue the collectives even at world_size 1 -- a one-rank RCCL
        # smoke test of the exact calls an N-GPU run makes (needs an initialised process group)
        self.force_collectives = force_collectives
        self._flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []
        self._layout = None

    @property
    def active(self) -> bool:
        """True when the training step has to reduce across ranks."""
        return self.world_size > 1 or self.force_collectives

    @classmethod
    def from_env(cls, backend: Optional[str] = None) -> "DataParallelContext":
        """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        # GGNN_LOCAL_DEVICE / GGNN_DIST_BACKEND: test hooks (several ranks sharing one GPU over gloo on a 1-GPU box)
        local = int(os.environ.get("GGNN_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        backend = backend or os.environ.get("GGNN_DIST_BACKEND")
        use_cuda = torch.cuda.is_available()
        device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
        if use_cuda:
            torch.cuda.set_device(device)
        force = os.environ.get("GGNN_FORCE_COLLECTIVES", "0") != "0"
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = backend or ("nccl" if use_cuda else "gloo")
            kwargs = {"device_id": device} if (use_cuda and backend == "nccl") else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
        return cls(rank, world, device, force_collectives=force)

    # ---- collectives ------------------------------------------------------------------------------
    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def barrier(self) -> None:
        if self.active:
            dist.barrier(group=self.group)

    def broadcast_(self, tensors: List[torch.Tensor], src: int = 0) -> None:
        """Make every rank start from rank-0's weights (one flat broadcast)."""
        if not self.active:
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=self.group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n

    # ---- the training-step hooks used by train.train_step ---------------------------------------------
    def global_loss(self, model) -> torch.Tensor:
        """This rank's share of the global loss: sum_tasks num_task(rank) / (sum_ranks den_task + 1e-7)."""
        tasks = model.params['task_ids']
        den = torch.stack([model.ops['loss_denominator_task%i' % t].detach() for t in tasks]).to(torch.float32)
        self.all_reduce_sum_(den)
        total = None
        for i, t in enumerate(tasks):
            ratio = 1.0 / (model.params['task_sample_ratios'].get(t) or 1.0)   # chem_tensorflow.py:168
            term = model.ops['loss_numerator_task%i' % t] / (den[i] + SMALL_NUMBER) * ratio
            total = term if total is None else total + term
        return total

    def reduce_gradients(self, variables: List[torch.Tensor], grads: List[Optional[torch.Tensor]]) -> None:
        """ONE sum all-reduce of one flat buffer with every gradient.  The gradients are packed into the buffer with a
        single multi-tensor copy (torch._foreach_copy_ onto persistent views of the buffer) and `grads[i]` is then
        REPLACED by the view of the reduced buffer -- no copy back, no per-variable launches on a latency-bound step.
        A variable with no gradient on this rank (e.g. an empty padding batch) contributes zeros.  The views are valid
        until the next call."""
        layout = tuple((v.data_ptr(), tuple(v.shape)) for v in variables)
        if self._flat is None or self._layout != layout or self._flat.device != variables[0].device:
            n = sum(v.numel() for v in variables)
            self._flat = torch.empty(n, dtype=torch.float32, device=variables[0].device)
            self._views, off = [], 0
            for v in variables:
                self._views.append(self._flat[off:off + v.numel()].view(v.shape))
                off += v.numel()
            self._layout = layout
        live_dst = [w for w, g in zip(self._views, grads) if g is not None]
        live_src = [g for g in grads if g is not None]
        if len(live_src) < len(grads):
            self._flat.zero_()
        if live_src:
            torch._foreach_copy_(live_dst, live_src)
        self.all_reduce_sum_(self._flat)
        for i, w in enumerate(self._views):
            grads[i] = w
