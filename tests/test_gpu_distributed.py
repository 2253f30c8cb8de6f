"""The N > 1 code path on real kernels: two `gloo` ranks that share GPU 0 (the 8-GPU RCCL runs are the
driver's; here every piece except the RCCL transport itself is exercised), and the device-resident
launch path on torch streams."""
import math
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bimodal(x):
    return 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2))


def _calls(mc, D):
    f1 = lambda x: x
    f2 = lambda x: x**2
    out = {}
    out["normal"] = mc.integrate([f1, f2], D.normal(0.0, 1.0), n_samples=3_000_001, seed=5).values
    out["beta"] = mc.integrate([f1, f2], D.beta(2.0, 5.0), n_samples=2_000_000, seed=6).values
    out["tiny"] = mc.integrate([f1, f2], D.normal(0.0, 1.0), n_samples=1000, seed=7).values      # L = 1: idx sharding
    xs = np.linspace(0, 10, 512)
    out["is"] = mc.integrate_importance_sampling([f1, f2], D.from_pdf_table(xs, np.exp(-xs)), D.normal(2.0, 3.0),
                                                 n_samples=2_000_000, seed=8).values
    res = mc.integrate_mcmc([f1, f2], D.from_pdf(_bimodal, support=(-10, 10)), D.normal(0.0, 2.0), n_steps=600,
                            n_chains=2048, n_burnin=50, seed=9)
    out["mcmc"] = np.append(res.values, res.meta["accept_rate"])
    return out


def _worker(rank, world, port, out_dir):
    for p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    assert MonteCarloIntegrator(device=0)._rank_world() == (0, 1)      # sharding is opt-in
    mc = MonteCarloIntegrator(device=0, process_group="world")         # every call is sharded over the world group
    assert mc._rank_world() == (rank, world)
    res = _calls(mc, Distribution)
    np.savez(Path(out_dir) / f"rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_gpu_results(tmp_path, integrator):
    import torch.multiprocessing as mp

    from wgpu_montecarlo import Distribution

    want = _calls(integrator, Distribution)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key, ref in want.items():
            assert np.allclose(got[key], ref, rtol=1e-9, atol=1e-9), (r, key, got[key], ref)


def test_device_resident_launch_follows_torch_streams(integrator):
    """launch() enqueues on torch's CURRENT stream (the null stream by default) and leaves K sums on the GPU."""
    import torch

    from wgpu_montecarlo import Distribution

    fns = [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4]
    prepared = integrator.prepare_integrate(fns, Distribution.normal(0.0, 1.0))
    want = integrator.integrate(fns, Distribution.normal(0.0, 1.0), n_samples=50_000_000, seed=3)
    dev = torch.device("cuda", 0)
    out = torch.full((3, 4), float("nan"), dtype=torch.float64, device=dev)
    n_eff = prepared.launch(50_000_000, 3, out[0])                      # default (null) stream
    copy0 = out[0] + 0.0                                                # consumer on the same stream: ordered
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        prepared.launch(50_000_000, 3, out[1])
        copy1 = out[1] + 0.0
    n_eff2, work = prepared.launch(50_000_000, 3, out[2], async_op=True)
    assert work is None and n_eff2 == n_eff == want.meta["n_eff"]
    torch.cuda.synchronize()
    for got in (copy0, copy1, out[2]):
        assert np.array_equal(got.cpu().numpy() / float(n_eff), want.values)
    assert np.array_equal(prepared.run(50_000_000, 3).values, want.values)


def test_launches_on_two_streams_do_not_share_scratch(integrator):
    """Two torch streams launch on the same engine in alternation: each stream has its own per-workgroup partials
    buffer (the fold kernel reads what the main kernel of the same launch wrote), so the results are the blocking ones."""
    import torch

    from wgpu_montecarlo import Distribution

    fns = [lambda x: x, lambda x: x**2]
    prepared = integrator.prepare_integrate(fns, Distribution.normal(0.0, 1.0))
    n = 30_000_000
    want = [prepared.run(n, seed).values for seed in range(8)]
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    out = torch.zeros(16, 2, dtype=torch.float64, device=dev)
    n_eff = None
    for rep in range(2):
        for seed in range(8):
            with torch.cuda.stream(streams[seed % 2]):
                n_eff = prepared.launch(n, seed, out[rep * 8 + seed])
    torch.cuda.synchronize()
    got = out.cpu().numpy() / float(n_eff)
    for rep in range(2):
        for seed in range(8):
            assert np.array_equal(got[rep * 8 + seed], want[seed]), (rep, seed)


_RCCL_SCRIPT = r"""
import os, sys
sys.path[:0] = [%(pkg)r, %(root)r]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator, distributed, runtime
mc = MonteCarloIntegrator(device=0)
assert "already loaded" in runtime.hip_runtime(), runtime.hip_runtime()      # torch's HIP runtime instance
f1 = lambda x: x
f2 = lambda x: x**2
prepared = mc.prepare_integrate([f1, f2], Distribution.normal(0.0, 1.0))
want = mc.integrate([f1, f2], Distribution.normal(0.0, 1.0), n_samples=200_000_000, seed=11)
group = distributed.Group()
assert (group.backend, group.world) == ("nccl", 1)
out = torch.zeros(8, 2, dtype=torch.float64, device="cuda")
works = []
for i in range(8):                                   # kernel i -> RCCL all-reduce i (async) -> kernel i+1 overlaps it
    n_eff = prepared.launch(200_000_000, 11, out[i])
    works.append(distributed.all_reduce_device(group, out[i], async_op=True))
for w in works:
    w.wait()
torch.cuda.synchronize()
got = out.cpu().numpy() / float(n_eff)
assert all(np.array_equal(row, want.values) for row in got), (got, want.values)
host = distributed.all_reduce_host(group, np.array([1.5, 2.5]))
assert np.array_equal(host, [1.5, 2.5])
dist.destroy_process_group()
print("RCCL-OK")
"""


def test_rccl_all_reduce_is_ordered_after_our_kernels(tmp_path):
    """A real RCCL communicator (1 rank: the box has one GPU) in the same process as libmcx: the collective is
    enqueued behind our kernels on torch's stream and both use ONE HIP runtime instance."""
    import subprocess

    script = tmp_path / "rccl_one_rank.py"          # a file: lambdas need recoverable source
    script.write_text(_RCCL_SCRIPT % dict(pkg=str(ROOT / "wgpu-monte-carlo_amd"), root=str(ROOT), port=str(_free_port())))
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "RCCL-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_one_thread_drives_several_engines(integrator):
    """mcx_integrate_multi / mcx_mcmc_multi: the C-level multi-device path (no torch, no RCCL). Three engines on the
    one GPU of the test box stand in for three devices: same sums as the single-engine call."""
    import numpy as np
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    fns = [lambda x: x, lambda x: x**2, lambda x: x**3]
    src = functions_to_hip(fns)
    beta = Distribution.beta(2.0, 5.0)
    engines = [rt.Engine(0) for _ in range(3)]
    try:
        # K1, custom distribution: every engine owns its copy of the CDF table
        desc = rt.make_desc(rt.KIND_INTEGRATE, 3, rt.DIST_CUSTOM)
        shards = [(e, e.module(src, desc), dict(cdf=e.cached_table(rt.TABLE_CDF, beta._cdf_table, beta._x_table))) for e in engines]
        sums, n_eff = rt.integrate_multi(shards, 3_000_001, 11, 0.0, 0.0)
        e0, m0, t0 = shards[0]
        whole, n_eff1 = e0.integrate(m0, 3_000_001, 11, 0.0, 0.0, cdf=t0["cdf"])
        assert n_eff == n_eff1
        assert np.allclose(sums, whole, rtol=1e-9, atol=1e-9 * n_eff)
        # K3, normal proposal (log q from the deviate): chain shards
        target = Distribution.normal(0.3, 1.0)
        tx, tl = target.get_log_pdf_table()
        desc3 = rt.make_desc(rt.KIND_MCMC, 3, rt.DIST_NORMAL, q_sampler=True, cell_tables=True)
        shards3 = [(e, e.module(src, desc3), dict(target_logpdf=e.cached_table(rt.TABLE_LOGPDF, tx, tl))) for e in engines]
        sums3, n_eff3 = rt.mcmc_multi(shards3, 300, 2000, 40, 5, 0.0, 2.0)
        e0, m0, t0 = shards3[0]
        whole3, _ = e0.mcmc(m0, 300, 2000, 40, 5, 0.0, 2.0, t0["target_logpdf"], None)
        assert sums3.shape == (4,) and np.allclose(sums3, whole3, rtol=1e-9, atol=1e-9 * n_eff3)
        # contract errors
        with pytest.raises(ValueError, match="own engine"):
            rt.integrate_multi([shards[0], shards[0]], 1000, 1, 0.0, 0.0)
    finally:
        for e in engines:
            e.close()


def test_one_process_drives_several_devices_through_the_public_api(integrator):
    """MonteCarloIntegrator(devices=[...]): one host thread, one engine per listed device, libmcx joins the shards
    (RCCL communicator for distinct devices, host-side sum otherwise). The test box has one GPU, so the three
    "devices" are three engines on GPU 0 (host-sum path; the RCCL path is exercised at runtime level in
    test_gpu_runtime_limits.py and from plain C): every public entry point must reproduce the single-device values."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    multi = MonteCarloIntegrator(devices=[0, 0, 0])
    want = _calls(integrator, Distribution)
    got = _calls(multi, Distribution)
    for key, ref in want.items():
        assert np.allclose(got[key], ref, rtol=1e-9, atol=1e-9), (key, got[key], ref)
    res = multi.integrate([lambda x: x, lambda x: x**2], Distribution.normal(0.0, 1.0), n_samples=3_000_001, seed=5)
    assert res.meta["devices"] == [0, 0, 0] and res.meta["collective"] == "host-sum"
    assert res.meta["n_eff"] == integrator.integrate([lambda x: x], Distribution.normal(0.0, 1.0), n_samples=3_000_001).meta["n_eff"]
    single = MonteCarloIntegrator(devices=[0])
    assert single.integrate([lambda x: x], Distribution.normal(0.0, 1.0), n_samples=1000).meta["collective"] is None
    with pytest.raises(ValueError, match="alternatives"):
        MonteCarloIntegrator(devices=[0], process_group="world")
    import torch

    prepared = multi.prepare_integrate([lambda x: x], Distribution.normal(0.0, 1.0))
    with pytest.raises(RuntimeError, match="devices"):
        prepared.launch(1000, 1, torch.zeros(1, dtype=torch.float64, device="cuda"))
    assert np.allclose(prepared.run(3_000_001, 5).values, want["normal"][:1], rtol=1e-9, atol=1e-9)


def test_the_bench_line_validates_itself_at_two_ranks():
    """`python bench.py --gpus 2` through its own launcher, two gloo ranks sharing this box's GPU (RCCL refuses two ranks
    on one device; every other piece of the N > 1 line is the real one): the one JSON line carries the headline, the
    other BASELINE configs as legs (C2 / C3 weak, C4 / C5 strong) and `self_check` -- an all-reduce of ones = 2, and per
    config the all-reduced shard sums equal to the whole grid run on rank 0 (C4: accepted-step counts exactly)."""
    import json
    import subprocess

    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "3",
                          "--warmup", "1", "--scale", "0.02", "--legs", "c3,c4,c5", "--no-cpu-baseline", "--no-cold", "--prewarm-ms", "0"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["name"] == "c2" and line["scaling"] == "weak"
    assert set(line["configs"]) == {"c3", "c4", "c5"} and not any("error" in leg for leg in line["configs"].values()), line["configs"]
    assert [line["configs"][c]["scaling"] for c in ("c3", "c4", "c5")] == ["weak", "strong", "strong"]
    check = line["self_check"]
    assert check["rccl_sum_of_ones"] == 2.0 and check["rccl_sum_of_ones_ok"]
    for cfg in ("c2", "c3", "c4", "c5"):
        c = check["sharded_equals_single"][cfg]
        assert c["ok"] and c["max_mean_diff_over_scale"] <= 1e-7, (cfg, c)
    assert check["sharded_equals_single"]["c4"]["accepted_steps_equal"] is True
    assert len(line["roofline"]["per_rank_kernel_ms"]) == 2
    for leg in [line] + list(line["configs"].values()):
        assert leg["value"] > 0 and leg["philox"]["value"] > 0 and set(leg["stream_valid"]) == {"pcg_ref", "philox"}
