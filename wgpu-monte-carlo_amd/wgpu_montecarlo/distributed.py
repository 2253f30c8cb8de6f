"""Multi-GPU plumbing: which ranks a call is sharded over, and the single collective that joins them.

The reference is single-device; this is new design. One process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" for CPU rehearsal). Each rank runs its
shard of the SAME logical sample grid (libmcx: mcx_shard_integrate / mcx_shard_chains), producing K
partial sums in f64; one sum all-reduce of K (MCMC: K+1) doubles combines them. 32..520 bytes: the
collective is latency-bound, so there is exactly one per call and no other data-path communication.
"""
from __future__ import annotations

import os
import sys
from typing import Optional

import numpy as np


class Group:
    """The ranks one call is sharded over."""

    def __init__(self, process_group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)


def default_group() -> Optional[Group]:
    """The world group if the caller initialised torch.distributed with more than one rank, else None.
    Never imports torch by itself: a process that has not imported torch cannot have a process group."""
    torch = sys.modules.get("torch")
    if torch is None or os.environ.get("MCX_DISTRIBUTED", "1") == "0":
        return None
    dist = getattr(torch, "distributed", None)
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() < 2:
        return None
    return Group()


def all_reduce_host(group: Optional[Group], sums: np.ndarray) -> np.ndarray:
    """Sum `sums` (float64 host array) over the group: the gloo / host-staged form of the collective."""
    if group is None or group.world < 2:
        return sums
    import torch

    t = torch.from_numpy(np.ascontiguousarray(sums, dtype=np.float64).copy())
    if group.backend == "nccl":
        t = t.cuda()
    group.dist.all_reduce(t, op=group.dist.ReduceOp.SUM, group=group.group)
    return t.cpu().numpy()


def all_reduce_device(group: Group, buf, async_op: bool = False):
    """Sum a float64 CUDA tensor in place over the group (RCCL over xGMI).

    The collective is ordered after the work already enqueued on torch's current stream. async_op=False:
    the current stream then waits for it (later kernels see the reduced values). async_op=True: it runs on
    the communicator's stream without blocking the current one; returns the work handle to wait on."""
    return group.dist.all_reduce(buf, op=group.dist.ReduceOp.SUM, group=group.group, async_op=async_op)
