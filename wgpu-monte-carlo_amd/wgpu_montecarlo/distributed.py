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


def world_group() -> Optional[Group]:
    """The world group of an initialised torch.distributed job with more than one rank, else None.
    Never imports torch by itself: a process that has not imported torch cannot have a process group."""
    torch = sys.modules.get("torch")
    if torch is None:
        return None
    dist = getattr(torch, "distributed", None)
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() < 2:
        return None
    return Group()


def default_group() -> Optional[Group]:
    """What an integrator built without `process_group` shards over: nothing, unless MCX_DISTRIBUTED=1 is set.

    Sharding is opt-in (MonteCarloIntegrator(process_group="world" | <torch group>)): every sharded call is a
    collective, so an integrate() issued by one rank only of a data-parallel training job -- a drop-in user of the
    reference's single-device API -- would otherwise hang in the all-reduce."""
    if os.environ.get("MCX_DISTRIBUTED", "0") != "1":
        return None
    return world_group()


def resolve_group(process_group) -> Optional[Group]:
    """None -> default_group(); "world" -> the world group (None when the job has a single rank); else a torch group."""
    if process_group is None:
        return default_group()
    if isinstance(process_group, str):
        if process_group != "world":
            raise ValueError("process_group must be None, 'world' or a torch.distributed process group")
        return world_group()
    return Group(process_group)


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
