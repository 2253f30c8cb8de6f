"""Randomised parity sweep (fixed seeds): small odd geometries of the logical grid and of the physical launch,
every sampler, random shards and physical-thread targets, against the oracle on the same streams."""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
F = [lambda x: x, lambda x: x * x, lambda x: math.cos(x)]


def _expected(xs):
    xs = xs.astype(np.float64)
    return np.array([xs.sum(), (xs * xs).sum(), np.cos(xs).sum()])


@pytest.mark.parametrize("case", range(24))
def test_random_geometry_k1(case):
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import MonteCarloIntegrator, functions_to_hip

    rng = np.random.default_rng(1000 + case)
    target = int(rng.choice([1, 255, 256, 257, 700, 1000, 4096, 5000, 65536]))
    n = int(rng.choice([1, 2, 3, 5, 17, 255, 1001, 4097, 99_999, 300_001, 1_000_003]))
    if target * 1 > 65536 and n > 300_001:
        n = 300_001
    seed = int(rng.integers(0, 2**32))
    kind = int(rng.integers(0, 4))
    beta = Distribution.beta(float(rng.uniform(1.5, 4)), float(rng.uniform(1.5, 6)), table_size=int(rng.choice([1000, 1500, 2048, 3000])))
    dist, p1, p2, kw = [
        (Distribution.uniform(-2.0, 5.0), -2.0, 5.0, {}),
        (Distribution.normal(1.0, 0.5), 1.0, 0.5, {}),
        (Distribution.exponential(3.0), 3.0, 0.0, {}),
        (beta, 0.0, 0.0, dict(cdf_table=beta._cdf_table, x_table=beta._x_table)),
    ][kind]
    mc = MonteCarloIntegrator(target_threads=target)
    mc._engine.set_target_threads(int(rng.choice([64, 4096, 100_000, 1 << 20, 1 << 22])))   # engines are shared:
    request_default = lambda: mc._engine.set_target_threads(0)                               # restored below
    try:
        res = mc.integrate(F, dist, n_samples=n, seed=seed)
    except Exception:
        request_default()
        raise
    cfg = oracle.dispatch_config(n, target)
    assert res.meta["n_eff"] == cfg["total_threads"] * cfg["loops_per_thread"]
    xs = oracle.samples(kind, p1, p2, n_samples=n, seed=seed, target_threads=target, guard=1, **kw)
    want = _expected(xs) / xs.size
    assert np.allclose(res.values, want, rtol=3e-5, atol=3e-5), (case, target, n, kind, res.values, want)
    # the same call as shards of a random world size
    world = int(rng.choice([2, 3, 5, 8]))
    cdf = mc._cdf_table(dist)
    mod = mc._engine.module(functions_to_hip(F), rt.make_desc(rt.KIND_INTEGRATE, 3, kind))
    parts = [mc._engine.integrate(mod, n, seed, p1, p2, target, cdf=cdf, rank=r, world=world)[0] for r in range(world)]
    request_default()
    assert np.allclose(np.sum(parts, axis=0) / xs.size, want, rtol=3e-5, atol=3e-5), (case, world)


@pytest.mark.parametrize("case", range(8))
def test_random_geometry_mcmc(case):
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo.api import MonteCarloIntegrator

    rng = np.random.default_rng(5000 + case)
    n_chains = int(rng.choice([1, 100, 256, 257, 1000, 3000]))
    n_steps = int(rng.choice([1, 2, 7, 100, 513]))
    n_burnin = int(rng.choice([0, 1, 2, 50]))
    seed = int(rng.integers(0, 2**32))
    kind = int(rng.integers(0, 4))
    target = Distribution.normal(0.5, 1.0)
    lap = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-12, 12), table_size=1200)
    proposal, p1, p2 = [(Distribution.uniform(-6.0, 7.0), -6.0, 7.0), (Distribution.normal(0.0, 2.0), 0.0, 2.0),
                        (Distribution.exponential(0.7), 0.7, 0.0), (lap, 0.0, 0.0)][kind]
    mc = MonteCarloIntegrator()
    res = mc.integrate_mcmc(F[:2], target, proposal, n_steps=n_steps, n_chains=n_chains, n_burnin=n_burnin, seed=seed)
    tx, tl = target.get_log_pdf_table()
    px, pl = proposal.get_log_pdf_table()
    ref = oracle.mcmc([(oracle.FN_IDENTITY, 0), (oracle.FN_SQ, 0)], kind, p1, p2, tx, tl, px, pl, n_steps=n_steps,
                      n_chains=n_chains, n_burnin=n_burnin, seed=seed, guard=1,
                      cdf_table=proposal._cdf_table, x_table=proposal._x_table if proposal._cdf_table is not None else None)
    assert res.meta["n_eff"] == ref["n_eff"]
    assert np.allclose(res.values, ref["sums"][:2] / ref["n_eff"], rtol=5e-4, atol=5e-4), (case, res.values, ref["sums"][:2] / ref["n_eff"])
    total_steps = (ref["n_eff"] // n_steps) * (n_steps + n_burnin)
    assert abs(res.meta["accept_rate"] - ref["sums"][2] / total_steps) < 2e-3


def test_many_fused_functions(integrator):
    """K = 64 (the maximum) on U(0,1): E[x^k] = 1/(k+1); K = 33 with importance sampling; K = 20 in the MH kernel."""
    from wgpu_montecarlo import Distribution

    fns = [lambda x, k=k: x**k for k in range(1, 65)]
    r = integrator.integrate(fns, Distribution.uniform(0.0, 1.0), n_samples=20_000_000, seed=3)
    truth = 1.0 / (np.arange(1, 65) + 1.0)
    sigma = np.sqrt((1.0 / (2 * np.arange(1, 65) + 1.0) - truth**2) / r.meta["n_eff"])
    assert np.all(np.abs(r.values - truth) < 4 * sigma), np.abs(r.values - truth) / sigma
    xs = oracle.samples(oracle.UNIFORM, 0.0, 1.0, n_samples=65536 * 4, seed=8, guard=1).astype(np.float64)
    small = integrator.integrate(fns, Distribution.uniform(0.0, 1.0), n_samples=65536 * 4, seed=8)
    want = np.array([(xs**k).mean() for k in range(1, 65)])
    assert np.allclose(small.values, want, rtol=2e-5, atol=1e-7)
    with pytest.raises(ValueError, match="at most 64"):
        integrator.integrate(fns + [lambda x: x], Distribution.uniform(0.0, 1.0), n_samples=1000)
    r = integrator.integrate_importance_sampling(fns[:33], Distribution.uniform(0.0, 1.0), Distribution.uniform(-0.5, 1.5),
                                                 n_samples=5_000_000, seed=5)
    assert np.all(np.abs(r.values - truth[:33]) < 0.01)
    r = integrator.integrate_mcmc(fns[:20], Distribution.beta(2.0, 5.0), Distribution.uniform(0.0, 1.0), n_steps=4000,
                                  n_chains=2048, n_burnin=200, seed=7)

    def beta_moment(k):
        m = 1.0
        for j in range(k):
            m *= (2 + j) / (7 + j)
        return m

    assert np.all(np.abs(r.values - [beta_moment(k) for k in range(1, 21)]) < 0.01)


@pytest.mark.parametrize("case", range(12))
def test_random_geometry_importance_sampling(case):
    """K2 on random small geometries: every way the weight can be formed -- target analytic or from a table (strict
    grid -> cells, perturbed grid -> verified guess), proposal normal (1/q from the deviate), uniform (emitted
    closure) or custom (CDF sampling + PDF table) -- against the oracle's f * p / q on the same samples."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo.api import MonteCarloIntegrator

    rng = np.random.default_rng(9000 + case)
    target_threads = int(rng.choice([256, 700, 4096, 65536]))
    n = int(rng.choice([3, 255, 1001, 99_999, 300_001, 700_001]))
    seed = int(rng.integers(0, 2**32))
    s2pi = float(np.float32(np.sqrt(2 * np.pi)))
    # target
    t_kind = int(rng.integers(0, 3))
    if t_kind == 0:
        target, p_spec = Distribution.normal(0.2, 0.7), (oracle.PDF_NORMAL, 0.2, 0.7, s2pi)
    else:
        xt = np.linspace(-3.0, 3.5, int(rng.choice([300, 512, 1000])))
        if t_kind == 2:
            xt[1:-1] += rng.uniform(-0.2, 0.2, len(xt) - 2) * (xt[1] - xt[0])
        target = Distribution.from_pdf_table(xt, np.exp(-0.5 * ((xt - 0.2) / 0.7) ** 2))
        p_spec = (oracle.PDF_TABLE, target._x_table, target._pdf_table)
    # proposal
    q_kind = int(rng.integers(0, 3))
    kw = {}
    if q_kind == 0:
        proposal, code, p1, p2 = Distribution.normal(0.0, 1.5), oracle.NORMAL, 0.0, 1.5
        q_spec = (oracle.PDF_NORMAL, 0.0, 1.5, s2pi)
    elif q_kind == 1:
        proposal, code, p1, p2 = Distribution.uniform(-4.0, 4.5), oracle.UNIFORM, -4.0, 4.5
        q_spec = (oracle.PDF_UNIFORM, -4.0, 4.5, 8.5)
    else:
        proposal = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-9, 9), table_size=int(rng.choice([1200, 2048])))
        code, p1, p2 = oracle.CUSTOM, 0.0, 0.0
        q_spec = (oracle.PDF_TABLE, *proposal.get_or_compute_pdf_table())
        kw = dict(cdf_table=proposal._cdf_table, x_table=proposal._x_table)
    mc = MonteCarloIntegrator(target_threads=target_threads)
    res = mc.integrate_importance_sampling(F, target, proposal, n_samples=n, seed=seed)
    fns = [(oracle.FN_IDENTITY, 0), (oracle.FN_SQ, 0), (oracle.FN_COS, 0)]
    ref = oracle.integrate(fns, code, p1, p2, n_samples=n, seed=seed, target_threads=target_threads, guard=1,
                           p=p_spec, q=q_spec, **kw)
    assert res.meta["n_eff"] == ref["n_eff"]
    want = ref["sums"] / ref["n_eff"]
    scale = max(1.0, float(np.max(np.abs(want))))
    assert np.allclose(res.values, want, rtol=5e-5, atol=5e-5 * scale), (case, t_kind, q_kind, n, res.values, want)


@pytest.mark.parametrize("case", range(8))
def test_random_geometry_mcmc_extensions(case):
    """Random-walk proposals, the Philox stream and the batch-means rows on random small geometries."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo.api import MonteCarloIntegrator

    rng = np.random.default_rng(7000 + case)
    n_chains = int(rng.choice([1, 256, 257, 1000]))
    n_steps = int(rng.choice([1, 2, 7, 100, 513]))
    n_burnin = int(rng.choice([0, 1, 3, 50]))
    seed = int(rng.integers(0, 2**32))
    philox = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    x0 = float(rng.choice([0.0, 0.5, -1.0]))
    target = Distribution.normal(0.5, 1.0)
    step, code, p1, p2, walk = [(Distribution.normal(0.0, 1.2), oracle.NORMAL, 0.0, 1.2, 2),
                                (Distribution.normal(0.4, 1.2), oracle.NORMAL, 0.4, 1.2, 1),
                                (Distribution.uniform(-1.5, 2.0), oracle.UNIFORM, -1.5, 2.0, 1)][kind]
    mc = MonteCarloIntegrator(rng="philox" if philox else "pcg_ref", std_error=True)
    res = mc.integrate_mcmc(F[:2], target, step, n_steps=n_steps, n_chains=n_chains, n_burnin=n_burnin, seed=seed,
                            proposal_kind="random_walk", initial_state=x0)
    tx, tl = target.get_log_pdf_table()
    px, pl = step.get_log_pdf_table()
    ref = oracle.mcmc([(oracle.FN_IDENTITY, 0), (oracle.FN_SQ, 0)], code, p1, p2, tx, tl, px, pl, n_steps=n_steps,
                      n_chains=n_chains, n_burnin=n_burnin, seed=seed, guard=1, walk=walk, x0=x0, rng=int(philox))
    assert res.meta["n_eff"] == ref["n_eff"]
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.allclose(res.values, want, rtol=4e-3, atol=4e-3), (case, kind, philox, res.values, want)
    total_steps = (ref["n_eff"] // n_steps) * (n_steps + n_burnin)
    assert abs(res.meta["accept_rate"] - ref["sums"][2] / total_steps) < 3e-3
    assert np.all(np.isfinite(res.meta["std_error"]))


@pytest.mark.parametrize("case", range(16))
def test_random_geometry_moment_families(case):
    """x, x**2, .., x**K for K from 8 to 32 on random small geometries, both streams, every sampler: power sums by Newton
    pairs (below 12 rows on the reference stream) or by quads (the recurrence of the four samples' quartic), the CDF
    sampler through its bucket-direct records with the append queue or the exchange ring -- against the f64 power sums
    of the oracle's samples on the same stream, and as shards. Bound relative to sum |x|^k (sums of mixed sign cancel)."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import MonteCarloIntegrator, functions_to_hip

    rng = np.random.default_rng(4000 + case)
    k = int(rng.choice([8, 11, 12, 13, 16, 24, 32]))
    philox = bool(rng.integers(0, 2))
    target = int(rng.choice([256, 257, 700, 4096, 65536]))
    n = int(rng.choice([5, 255, 1001, 4097, 99_999, 300_001, 1_000_003, 2_500_001]))
    seed = int(rng.integers(0, 2**32))
    kind = int(rng.integers(0, 4)) if case % 2 else 3           # every other case: the CDF sampler
    beta = Distribution.beta(float(rng.uniform(1.5, 4)), float(rng.uniform(1.5, 6)), table_size=int(rng.choice([1000, 2048, 3000])))
    dist, p1, p2, kw = [
        (Distribution.uniform(-1.0, 1.25), -1.0, 1.25, {}),
        (Distribution.normal(0.1, 0.4), 0.1, 0.4, {}),
        (Distribution.exponential(4.0), 4.0, 0.0, {}),
        (beta, 0.0, 0.0, dict(cdf_table=beta._cdf_table, x_table=beta._x_table)),
    ][kind]
    fns = [lambda x, p=p: x**p for p in range(1, k + 1)]
    mc = MonteCarloIntegrator(target_threads=target, rng="philox" if philox else "pcg_ref")
    res = mc.integrate(fns, dist, n_samples=n, seed=seed)
    xs = oracle.samples(kind, p1, p2, n_samples=n, seed=seed, target_threads=target, guard=1, rng=int(philox), **kw)
    assert res.meta["n_eff"] == xs.size
    xs = xs.astype(np.float64).ravel()
    want = np.array([(xs**p).sum() for p in range(1, k + 1)])
    mag = np.array([(np.abs(xs) ** p).sum() for p in range(1, k + 1)])
    got = res.values * xs.size
    assert np.all(np.abs(got - want) <= 3e-5 * mag + 1e-30), (case, k, philox, kind, n, np.abs(got - want) / mag)
    world = int(rng.choice([2, 3, 8]))
    cdf = mc._cdf_table(dist)
    desc = rt.make_desc(rt.KIND_INTEGRATE, k, kind, rng=rt.RNG_PHILOX if philox else rt.RNG_PCG_REF, moment_family=True,
                        cdf_direct=cdf is not None and cdf.direct_bits > 0)
    mod = mc._engine.module(functions_to_hip(fns), desc)
    parts = [mc._engine.integrate(mod, n, seed, p1, p2, target, cdf=cdf, rank=r, world=world)[0] for r in range(world)]
    assert np.all(np.abs(np.sum(parts, axis=0) - want) <= 3e-5 * mag + 1e-30), (case, world)
