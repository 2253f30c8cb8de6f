"""N > 1 path rehearsed on CPU: world_size-2 and -3 `gloo` process groups.

Each rank takes its shard of the logical sample grid from the C ABI (mcx_shard_integrate /
mcx_shard_chains -- the same calls the GPU path makes), computes that shard's partial sums with the CPU
oracle standing in for the kernel, and joins the ranks with the product's one collective
(wgpu_montecarlo.distributed.all_reduce_host). The result must equal the single-process sums: sharding
changes neither the samples drawn nor N_eff.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, dist_code, out_dir):
    for p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from wgpu_montecarlo import distributed
    from wgpu_montecarlo import runtime as rt

    assert distributed.default_group() is None           # sharding is opt-in: nothing is picked up implicitly
    group = distributed.resolve_group("world")
    assert group is not None and (group.rank, group.world, group.backend) == (rank, world, "gloo")
    d = rt.dispatch_config(n_samples)
    shard = rt.shard_integrate(d, dist_code, group.rank, group.world)
    xs = oracle.samples(dist_code, 0.5, 1.5, n_samples=n_samples, seed=11, guard=1,
                        idx0=shard.idx_begin, nidx=shard.idx_count).astype(np.float64)
    per_unit = 2 if dist_code == rt.DIST_NORMAL else 1          # a normal unit is a Box-Muller pair
    lo, hi = shard.unit_begin * per_unit, min(shard.unit_end * per_unit, d.loops_per_thread)
    mine = xs[:, lo:hi]
    partial = np.array([mine.sum(), (mine**2).sum(), float(mine.size)])
    total = distributed.all_reduce_host(group, partial)
    # chains: contiguous, disjoint, covering
    spans = [rt.shard_chains(4096, r, world) for r in range(world)]
    assert sum(n for _, n in spans) == 4096
    np.save(Path(out_dir) / f"rank{rank}.npy", total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_code,n_samples", [(2, 1, 700_000), (2, 0, 300_000), (3, 1, 70_000), (2, 1, 65_536)])
def test_sharded_sums_equal_single_process(tmp_path, world, dist_code, n_samples):
    import torch.multiprocessing as mp

    import oracle

    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_samples, dist_code, str(tmp_path)), nprocs=world, join=True)
    xs = oracle.samples(dist_code, 0.5, 1.5, n_samples=n_samples, seed=11, guard=1).astype(np.float64)
    want = np.array([xs.sum(), (xs**2).sum(), float(xs.size)])
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got[2] == want[2]                                  # every sample of the grid counted exactly once
        assert np.allclose(got[:2], want[:2], rtol=1e-11), (got, want)
