"""Runtime corners of libmcx on the GPU: the LDS budget next to a kernel's static scratch, chunked launches for very
long calls, the single-process RCCL communicator, module eviction while launches are in flight."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_lds_budget_accounts_for_static_scratch(integrator):
    """MCMC with std_error=True and K = 16 declares 16 waves x 49 rows x 8 B = 6.1 KiB of static reduction scratch
    (block 1024): a ~155 KiB target table used to pass the host check (<= 156 KiB) and then fail at
    hipModuleLaunchKernel. Now the budget is 160 KiB minus the module's static LDS: the API falls back to
    tables_lds = 0 when the tables do not fit, and a C caller gets MCX_E_INVALID naming the budget, not a HIP error."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt

    desc = rt.make_desc(rt.KIND_MCMC, 16, rt.DIST_NORMAL, second_moments=True, cell_tables=True, q_sampler=True)
    budget = rt.lds_table_budget(desc)
    assert 150 * 1024 <= budget <= 160 * 1024 - 16 * 49 * 8
    assert rt.lds_table_budget(rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_CUSTOM)) > budget    # less scratch, more room

    mc = MonteCarloIntegrator(std_error=True)
    mc._engine = rt.Engine(0)                                     # own engine: its table cache is dropped at the end
    try:
        fns = [lambda x, k=k: x**k for k in range(1, 17)]
        pdf = lambda x: np.exp(-0.5 * x * x)
        proposal = Distribution.normal(0.0, 1.5)                  # log q from the sampler's deviate: only the target table is staged
        fits = Distribution.from_pdf(pdf, support=(-8.0, 8.0), table_size=budget // 8 - 64)
        # 1 048 576 chains: the launch uses 1024-thread workgroups (mcx_mcmc_block_hint), whose 16 waves need the 6.1 KiB
        chains = 1_048_576
        res = mc.integrate_mcmc(fns, fits, proposal, n_steps=12, n_chains=chains, n_burnin=3, seed=3)
        assert np.all(np.isfinite(res.values))
        assert budget - 1024 < res.meta["lds_bytes"] <= budget    # staged, right up to the budget
        # a table inside the old 156 KiB limit but beyond what the static scratch leaves: must fall back, not fail
        big = Distribution.from_pdf(pdf, support=(-8.0, 8.0), table_size=(155 * 1024) // 8)
        res_big = mc.integrate_mcmc(fns, big, proposal, n_steps=12, n_chains=chains, n_burnin=3, seed=3)
        assert np.all(np.isfinite(res_big.values)) and res_big.meta["lds_bytes"] == 0
        assert np.allclose(res_big.values[:2], res.values[:2], atol=0.05)        # same chains, finer table of the same density
        # the C-level contract: forcing the staged build for that table is refused with a clear message
        tx, tl = big.get_log_pdf_table()
        tb = mc._engine.cached_table(rt.TABLE_LOGPDF, tx, tl)
        assert budget < tb.lds_bytes <= 156 * 1024
        from wgpu_montecarlo.api import functions_to_hip

        forced = rt.make_desc(rt.KIND_MCMC, 16, rt.DIST_NORMAL, second_moments=True, cell_tables=tb.has_cells, q_sampler=True)
        mod = mc._engine.module(functions_to_hip(fns), forced)
        with pytest.raises(ValueError, match="do not fit in LDS"):
            mc._engine.mcmc(mod, 12, chains, 3, 3, 0.0, 1.5, tb, None)
        # a small call picks 256-thread workgroups: 4 waves of scratch instead of 16, and the same table is staged
        small = mc.integrate_mcmc(fns, big, proposal, n_steps=60, n_chains=1024, n_burnin=10, seed=3)
        assert small.meta["block"] == 256 and small.meta["lds_bytes"] == tb.lds_bytes
    finally:
        mc._engine.close()


def test_static_lds_estimate_covers_the_code_objects(integrator):
    """mcx_lds_table_budget reserves an upper bound of the static LDS; the launch check uses the code object's real
    figure. The bound must never be below the real one (else a call the API staged would fail at launch)."""
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    eng = integrator._engine
    for kind, k, kw in [(rt.KIND_INTEGRATE, 4, dict(dist_type=rt.DIST_CUSTOM)), (rt.KIND_INTEGRATE, 32, dict(dist_type=rt.DIST_CUSTOM)),
                        (rt.KIND_INTEGRATE, 32, dict(dist_type=rt.DIST_CUSTOM, second_moments=True)),
                        (rt.KIND_INTEGRATE, 64, dict(dist_type=rt.DIST_CUSTOM)),
                        (rt.KIND_MCMC, 2, dict(dist_type=rt.DIST_NORMAL)), (rt.KIND_MCMC, 16, dict(dist_type=rt.DIST_NORMAL, second_moments=True)),
                        (rt.KIND_MCMC, 16, dict(dist_type=rt.DIST_NORMAL, second_moments=True, walk=rt.WALK_ADAPTIVE))]:
        desc = rt.make_desc(kind, k, kw.pop("dist_type"), **kw)
        src = functions_to_hip([lambda x, j=j: x**j for j in range(1, k + 1)])
        mod = eng.module(src, desc)
        reserved = 160 * 1024 - rt.lds_table_budget(desc)
        assert 0 < mod.static_lds <= reserved, (kind, k, kw, mod.static_lds, reserved)


def test_long_calls_are_split_into_several_launches(integrator):
    """Calls above the per-launch work bound become several launches on one stream writing disjoint partial records,
    folded once: same samples, same sums (up to summation order), launch count reported."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt

    fns = [lambda x: x, lambda x: x**2, lambda x: x**3]
    n = 50_000_011
    whole = integrator.integrate(fns, Distribution.normal(0.5, 1.5), n_samples=n, seed=9)
    whole_beta = integrator.integrate(fns, Distribution.beta(2.0, 5.0), n_samples=n, seed=9)
    target = Distribution.from_pdf(lambda x: np.exp(-0.5 * x * x), support=(-8.0, 8.0))
    whole_mc = integrator.integrate_mcmc(fns[:2], target, Distribution.normal(0.0, 1.5), n_steps=200, n_chains=5000, n_burnin=20, seed=4)
    assert whole.meta["n_eff"] > n and integrator._engine.last_launch()["launches"] == 1
    try:
        rt.set_max_launch_units(7_000_000)                         # -> 8 launches of the 5e7-sample calls
        split = integrator.integrate(fns, Distribution.normal(0.5, 1.5), n_samples=n, seed=9)
        launches = integrator._engine.last_launch()["launches"]
        split_beta = integrator.integrate(fns, Distribution.beta(2.0, 5.0), n_samples=n, seed=9)
        launches_beta = integrator._engine.last_launch()["launches"]
        rt.set_max_launch_units(200_000)                           # 5120 padded chains x 220 steps = 1.1e6 chain-steps
        split_mc = integrator.integrate_mcmc(fns[:2], target, Distribution.normal(0.0, 1.5), n_steps=200, n_chains=5000,
                                             n_burnin=20, seed=4)
        launches_mc = integrator._engine.last_launch()["launches"]
    finally:
        rt.set_max_launch_units(0)
    assert launches == 8 and launches_beta == 8 and launches_mc >= 5
    assert split.meta["n_eff"] == whole.meta["n_eff"]
    assert np.allclose(split.values, whole.values, rtol=1e-10, atol=1e-10)
    assert np.allclose(split_beta.values, whole_beta.values, rtol=1e-10, atol=1e-12)
    assert np.allclose(split_mc.values, whole_mc.values, rtol=1e-10, atol=1e-12)       # chains are independent of the cut
    assert split_mc.meta["accept_rate"] == pytest.approx(whole_mc.meta["accept_rate"], abs=1e-12)
    # ... and the split sums agree with the oracle like any other launch
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2), (oracle.FN_POW, 3)], oracle.NORMAL, 0.5, 1.5,
                           n_samples=n, seed=9, guard=1)
    want = ref["sums"] / ref["n_eff"]
    assert np.all(np.abs(split.values - want) < 2e-5 + 2e-5 * np.abs(want))


def test_rccl_communicator_single_process(integrator):
    """mcx_comm_*: ncclCommInitAll over this process's engines (one GPU on the test box, RCCL refuses duplicate
    devices) + a grouped ncclAllReduce of the K doubles on the engine's stream instead of the host-side sum. With one
    rank the collective is the identity: the sums must be bit-identical to the plain call, for K1 and K3."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    fns = [lambda x: x, lambda x: x**2, lambda x: x**3]
    src = functions_to_hip(fns)
    eng = rt.Engine(0)
    comm = None
    try:
        comm = rt.Comm([eng])
        assert comm.size == 1 and "rccl" in rt.rccl_library().lower()
        desc = rt.make_desc(rt.KIND_INTEGRATE, 3, rt.DIST_NORMAL)
        mod = eng.module(src, desc)
        sums, n_eff = comm.integrate([(eng, mod, {})], 30_000_001, 11, 0.25, 2.0)
        whole, n_eff1 = eng.integrate(mod, 30_000_001, 11, 0.25, 2.0)
        assert n_eff == n_eff1 and np.array_equal(sums, whole)
        for _ in range(3):                                          # the communicator is reusable
            again, _ = comm.integrate([(eng, mod, {})], 30_000_001, 11, 0.25, 2.0)
            assert np.array_equal(again, whole)
        target = Distribution.normal(0.3, 1.0)
        tx, tl = target.get_log_pdf_table()
        desc3 = rt.make_desc(rt.KIND_MCMC, 3, rt.DIST_NORMAL, q_sampler=True, cell_tables=True)
        mod3 = eng.module(src, desc3)
        tb = dict(target_logpdf=eng.cached_table(rt.TABLE_LOGPDF, tx, tl))
        sums3, _ = comm.mcmc([(eng, mod3, tb)], 300, 2000, 40, 5, 0.0, 2.0)
        whole3, _ = eng.mcmc(mod3, 300, 2000, 40, 5, 0.0, 2.0, tb["target_logpdf"], None)
        assert np.array_equal(sums3, whole3)
        other = rt.Engine(0)
        try:
            with pytest.raises(ValueError, match="distinct device"):
                rt.Comm([eng, other])
        finally:
            other.close()
    finally:
        if comm is not None:
            comm.close()
        eng.close()


def test_module_eviction_waits_for_launches_in_flight():
    """Engine.module() keeps the MAX_MODULES most recently used modules; an evicted module is unloaded only after its
    last launch has finished (mcx_module_release waits on the module's last-use event). Here: a launch on a torch
    stream is still running when its module is evicted and released."""
    import torch

    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt

    eng = rt.Engine(0)
    old_max = rt.Engine.MAX_MODULES
    rt.Engine.MAX_MODULES = 2
    try:
        mc = MonteCarloIntegrator()
        mc._engine = eng
        out = torch.zeros(3, 2, dtype=torch.float64, device="cuda")
        want = []
        stream = torch.cuda.Stream()
        for i, c in enumerate((1.0, 2.0, 3.0)):
            fns = [lambda x, c=c: c * x * x, lambda x, c=c: x + c]
            prepared = mc.prepare_integrate(fns, Distribution.normal(0.0, 1.0))
            with torch.cuda.stream(stream):
                n_eff = prepared.launch(400_000_000, 5, out[i])       # ~0.2 ms each, queued back to back
            del prepared                                              # the cache holds the only other reference
            want.append([c, c])
        assert len(eng._modules) == 2                                 # the first module was evicted while queued / running
        import gc
        gc.collect()
        torch.cuda.synchronize()
        got = out.cpu().numpy() / float(n_eff)
        assert np.allclose(got, want, atol=2e-3), got
    finally:
        rt.Engine.MAX_MODULES = old_max
        eng.close()


@pytest.mark.parametrize("mode", ["plain", "torch_first", "torch_after_cuda", "torch_after_noclose"])
def test_process_exits_cleanly_whatever_the_import_order(mode):
    """RCCL is bound at run time; PyTorch ships its own copy. Creating a communicator and importing torch afterwards
    used to end in `double free or corruption` at interpreter exit, and a communicator left open segfaulted during
    teardown (tools/exit_order_probe.py). Each mode runs in its own process and must exit with code 0."""
    import subprocess
    import sys
    from pathlib import Path

    probe = Path(__file__).resolve().parent.parent / "tools" / "exit_order_probe.py"
    res = subprocess.run([sys.executable, str(probe), mode], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and f"done {mode}" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-1500:])


def test_launch_geometry_follows_the_work_per_workgroup(integrator):
    """Workgroups per launch: 4096 (16 per CU) when every workgroup samples enough to pay for staging its tables into LDS
    (6 samples per staged byte), never fewer than 1 Mi physical threads; table-less kernels always take 4096 workgroups of
    256 once the call is large enough to be cut that far (profiles/r02b_launch_geometry_vs_call_size.txt)."""
    from wgpu_montecarlo import Distribution

    beta = Distribution.beta(2.0, 5.0)
    f4 = [lambda x, p=p: x**p for p in range(1, 5)]
    geometry = lambda res: (res.meta["n_blocks"], res.meta["block"])
    assert geometry(integrator.integrate(f4, beta, n_samples=10**7)) == (1024, 1024)            # 72 KiB staged per workgroup
    assert geometry(integrator.integrate(f4, beta, n_samples=3 * 10**9)) == (4096, 1024)
    mid = geometry(integrator.integrate(f4, beta, n_samples=10**9))
    assert mid[1] == 1024 and 1024 < mid[0] < 4096
    assert geometry(integrator.integrate(f4, Distribution.normal(0.0, 1.0), n_samples=10**9)) == (4096, 256)
    integrator._engine.set_target_threads(1 << 19)                # the override still rules
    try:
        assert geometry(integrator.integrate(f4, beta, n_samples=3 * 10**9)) == (512, 1024)
    finally:
        integrator._engine.set_target_threads(0)


@pytest.mark.parametrize("segments", [2, 5, 8])
def test_mcmc_time_segments_run_the_same_chains(integrator, segments):
    """mcx_engine_set_mcmc_segments with an explicit count: two halves of the chains on two streams, each cut into step segments that
    resume every chain from its saved {x, w}. The draws are functions of (seed, chain, step), so the chains are the SAME
    chains: accepted-step count identical, sums equal up to the regrouping of the f32 accumulation blocks -- for odd and
    even burn-in lengths, step counts that leave a single trailing step, chain counts that do not halve into whole
    workgroups, and a launch on a caller's stream."""
    import torch
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt

    target = Distribution.from_pdf(lambda x: 0.5 * (np.exp(-0.5 * (x - 2) ** 2) + np.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    f2 = [lambda x: x, lambda x: x * x]
    eng = integrator._engine
    cases = [(2000, 4096, 300), (1999, 4096 + 256, 301), (64, 768, 7), (501, 65_536, 0), (37, 512, 100)]
    plain = [integrator.integrate_mcmc(f2, target, proposal, n_steps=s, n_chains=c, n_burnin=b, seed=11) for s, c, b in cases]
    assert all(eng.last_launch()["launches"] == 1 for _ in (0,))
    eng.set_mcmc_segments(segments)
    try:
        cut = [integrator.integrate_mcmc(f2, target, proposal, n_steps=s, n_chains=c, n_burnin=b, seed=11) for s, c, b in cases]
        launches = eng.last_launch()["launches"]
        # the asynchronous form on a caller's stream
        prep = integrator.prepare_mcmc(f2, target, proposal)
        out = torch.zeros(prep.rows, dtype=torch.float64, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            n_eff = prep.launch(2000, 4096, 300, 11, out)
        side.synchronize()
        # precise math / random walks are not segmentable: one launch, whatever the setting
        other = integrator.integrate_mcmc(f2, target, proposal, n_steps=200, n_chains=1024, n_burnin=20, seed=3,
                                          proposal_kind="random_walk")
        assert eng.last_launch()["launches"] == 1 and np.all(np.isfinite(other.values))
    finally:
        eng.set_mcmc_segments(rt.SEGMENTS_AUTO)
    assert launches == 2 * min(segments, (37 + 100) // 4) or launches == 2 * segments
    for a, b, (s, c, bn) in zip(plain, cut, cases):
        assert a.meta["n_eff"] == b.meta["n_eff"]
        assert a.meta["accept_rate"] == b.meta["accept_rate"], (s, c, bn)          # the same accept decisions, all of them
        assert np.allclose(a.values, b.values, rtol=2e-6, atol=2e-6), (s, c, bn, a.values, b.values)
    assert n_eff == plain[0].meta["n_eff"]
    assert np.allclose(out.cpu().numpy()[:2] / n_eff, plain[0].values, rtol=2e-6, atol=2e-6)
    with pytest.raises(ValueError):
        eng.set_mcmc_segments(65)


def test_full_size_mcmc_calls_are_time_segmented_by_default(integrator):
    """The default (MCX_SEGMENTS_AUTO): a launch of >= 131 072 chains -- two waves per SIMD and more; C4's full size and
    its 2 / 4 / 8-GPU shards -- with >= 1.4e9 chain-steps of work runs as 8 segments x 2 chain halves, with >= 7e8 as 4,
    anything smaller or shorter as one launch (there the launches cost more than they cover); set_mcmc_segments(0) turns
    it off. Same chains either way: identical accepted-step counts (every accept decision), sums equal up to the
    regrouping of the f32 blocks."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt

    target = Distribution.from_pdf(lambda x: 0.5 * (np.exp(-0.5 * (x - 2) ** 2) + np.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    f2 = [lambda x: x, lambda x: x * x]
    eng = integrator._engine
    run = lambda chains, steps, burn: integrator.integrate_mcmc(f2, target, proposal, n_steps=steps, n_chains=chains, n_burnin=burn, seed=5)
    auto = run(1_048_576, 1301, 40)                                          # 1.41e9 chain-steps
    assert (auto.meta["segments"], auto.meta["launches"]) == (8, 16)
    assert auto.meta["n_blocks"] * auto.meta["block"] == 1_048_576          # the workgroups of one segment, both halves
    assert (run(131_072, 10_000, 1000).meta["segments"], run(131_072, 5_400, 0).meta["segments"]) == (8, 4)      # 1.44e9, 7.1e8
    for chains, steps in ((1_048_576, 400), (131_072, 3000), (65_536 + 256, 30_000)):          # short, short, too few chains
        assert run(chains, steps, 0).meta["launches"] == 1, (chains, steps)
    eng.set_mcmc_segments(0)
    try:
        one = run(1_048_576, 1301, 40)
    finally:
        eng.set_mcmc_segments(rt.SEGMENTS_AUTO)
    assert (one.meta["segments"], one.meta["launches"]) == (0, 1)
    assert one.meta["accept_rate"] == auto.meta["accept_rate"]
    assert np.allclose(one.values, auto.values, rtol=2e-6, atol=2e-6), (one.values, auto.values)


def test_segmented_mcmc_calls_in_flight_on_two_streams_keep_their_own_state(integrator):
    """the side stream, its fork / join events and the {x, w} buffer of a time-segmented call used to be one
    per ENGINE, so two segmented calls in flight on different streams of one engine resumed from each other's chain
    state. They are kept per caller stream now (like the per-workgroup partial sums): two such calls, different seeds,
    queued on two torch streams at once, each reproduce their own unsegmented result -- accept counts exactly."""
    import torch
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt

    target = Distribution.from_pdf(lambda x: 0.5 * (np.exp(-0.5 * (x - 2) ** 2) + np.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    f2 = [lambda x: x, lambda x: x * x]
    eng = integrator._engine
    prep = integrator.prepare_mcmc(f2, target, proposal)
    sizes = (3000, 131_072, 500)                                   # ~0.5 ms per call: long enough to overlap
    want = torch.zeros(2, prep.rows, dtype=torch.float64, device="cuda")
    eng.set_mcmc_segments(0)
    try:
        for i, seed in enumerate((21, 22)):
            n_eff = prep.launch(*sizes, seed, want[i])
        torch.cuda.synchronize()
        assert eng.last_launch()["launches"] == 1
        eng.set_mcmc_segments(6)
        got = torch.zeros(4, 2, prep.rows, dtype=torch.float64, device="cuda")
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(4):                                       # keep both streams busy with alternating seeds
            for i, seed in enumerate((21, 22)):
                with torch.cuda.stream(streams[i]):
                    prep.launch(*sizes, seed, got[rep, i])
        assert eng.last_launch()["launches"] == 12 and eng.last_launch()["segments"] == 6
        torch.cuda.synchronize()
    finally:
        eng.set_mcmc_segments(rt.SEGMENTS_AUTO)
    w, g = want.cpu().numpy(), got.cpu().numpy()
    for rep in range(4):
        assert np.array_equal(g[rep, :, 2], w[:, 2]), (rep, g[rep, :, 2], w[:, 2])       # accepted steps: the same chains
        assert np.allclose(g[rep, :, :2] / n_eff, w[:, :2] / n_eff, rtol=2e-6, atol=2e-6)
    assert not np.array_equal(w[0], w[1])


@pytest.mark.parametrize("segments", [2, 7])
def test_philox_mcmc_time_segments_run_the_same_chains(segments):
    """The Philox stream's MH loop resumes a chain from {x, w} at an even step as well (call index = step / 2): the same
    chains cut into segments -- identical accepted-step counts -- for odd / even burn-in and a trailing single step."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt

    target = Distribution.from_pdf(lambda x: 0.5 * (np.exp(-0.5 * (x - 2) ** 2) + np.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    f2 = [lambda x: x, lambda x: x * x]
    mc = MonteCarloIntegrator(rng="philox")
    eng = mc._engine
    cases = [(2000, 4096, 300), (1999, 4096 + 256, 301), (64, 768, 7), (37, 512, 100)]
    eng.set_mcmc_segments(0)
    try:
        plain = [mc.integrate_mcmc(f2, target, proposal, n_steps=s, n_chains=c, n_burnin=b, seed=11) for s, c, b in cases]
        assert eng.last_launch()["launches"] == 1
        eng.set_mcmc_segments(segments)
        cut = [mc.integrate_mcmc(f2, target, proposal, n_steps=s, n_chains=c, n_burnin=b, seed=11) for s, c, b in cases]
        assert eng.last_launch()["segments"] == segments
    finally:
        eng.set_mcmc_segments(rt.SEGMENTS_AUTO)
    for a, b, case in zip(plain, cut, cases):
        assert a.meta["rng"] == b.meta["rng"] == "philox"
        assert a.meta["accept_rate"] == b.meta["accept_rate"], case
        assert np.allclose(a.values, b.values, rtol=2e-6, atol=2e-6), (case, a.values, b.values)


@pytest.mark.parametrize("env", [{"MCX_POLL_US": "0"}, {"MCX_NO_ZERO_COPY": "1"}, {"MCX_NO_TIMING": "1"},
                                 {"MCX_POLL_US": "1"}])
def test_blocking_calls_give_the_same_bits_on_every_result_path(env, tmp_path):
    """A blocking call's K doubles reach the host in one of three ways: folded straight into pinned host memory and
    found by polling the per-row tickets (default), the same with the stream's completion signal (MCX_POLL_US=0, or a
    call that outlasts the polling window: MCX_POLL_US=1 us), or through the device buffer and a copy
    (MCX_NO_ZERO_COPY=1: what a runtime without mapped host memory gets). Same kernels, same fold order: the same bits."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = r"""
import json, sys
sys.path[:0] = [%r, %r]
import numpy as np
from wgpu_montecarlo import Distribution, MonteCarloIntegrator
mc = MonteCarloIntegrator()
f = [lambda x: x, lambda x: x * x, lambda x: x > 0.5]
out = {}
out["k1_small"] = mc.integrate(f, Distribution.normal(0.2, 1.3), n_samples=10_000, seed=3).values.tolist()
out["k1_large"] = mc.integrate(f, Distribution.normal(0.2, 1.3), n_samples=300_000_000, seed=3).values.tolist()
out["beta"] = mc.integrate(f, Distribution.beta(2.0, 5.0), n_samples=5_000_000, seed=4).values.tolist()
r = mc.integrate_mcmc(f[:2], Distribution.normal(0.5, 1.0), Distribution.normal(0.0, 2.0), n_steps=500, n_chains=8192, n_burnin=50, seed=5)
out["mcmc"] = r.values.tolist() + [r.meta["accept_rate"]]
out["kernel_ms"] = r.meta["kernel_ms"]
print(json.dumps(out))
""" % (str(root / "wgpu-monte-carlo_amd"), str(root))

    script = tmp_path / "result_paths.py"               # a file: lambdas are lowered from their source text
    script.write_text(code)

    def run(extra):
        import os

        res = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-1500:]
        return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])

    want, got = run({}), run(env)
    for key in ("k1_small", "k1_large", "beta", "mcmc"):
        assert got[key] == want[key], (env, key, got[key], want[key])
    assert (got["kernel_ms"] < 0) == ("MCX_NO_TIMING" in env) and want["kernel_ms"] > 0
