"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical counter streams.

Tolerances (stated per test): the integer side -- T, L, N_eff, padded chain counts -- is bit-exact. The
floating-point side is compared with the oracle's f64 sums of the same f32 samples; the remaining
difference comes from (a) v_log/v_sin/v_cos/v_sqrt vs glibc logf/sinf/cosf/sqrtf in the samplers
(<= a few ulp per sample), (b) x*x*.. vs powf, (c) p/q computed once vs f*p/q, (d) summation order.
Observed |diff| is ~1e-7..1e-6 on O(1) means; the asserted bound is 2e-5 (absolute + relative), about
100x below the Monte-Carlo 3-sigma at these sizes. math="precise" (ocml in the samplers) is held to 2e-6.
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

MOMENTS = [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4]
ORC_MOMENTS = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2), (oracle.FN_POW, 3), (oracle.FN_POW, 4)]
TOL = 2e-5


def close(got, want, tol=TOL):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    ok = np.all(err <= tol + tol * np.abs(want))
    return ok, f"got {got} want {want} |diff| {err}"


def test_engine_reports_gfx950(integrator):
    assert integrator._engine.device == 0


@pytest.mark.parametrize("n_samples,target", [(1_000_000, None), (1_000, None), (3_000_000, 32768),
                                             (1_114_112, None), (200_000, 1000)])
def test_normal_moments_match_oracle(integrator, n_samples, target):
    """K1, normal sampler: same samples as the oracle for several (T, L), incl. odd L and L = 1."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    integ = integrator if target is None else MonteCarloIntegrator(target_threads=target)
    res = integ.integrate(MOMENTS, Distribution.normal(0.5, 1.5), n_samples=n_samples, seed=1234)
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 0.5, 1.5, n_samples=n_samples, seed=1234,
                           target_threads=target, guard=1)
    assert res.meta["n_eff"] == ref["n_eff"]
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    # and the reference's own f32 accumulation stays within its rounding noise of our f64 result
    ok, msg = close(res.values, ref["ref"], tol=5e-4)
    assert ok, msg


def test_normal_precise_math_is_tighter():
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    integ = MonteCarloIntegrator(math="precise")
    res = integ.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=1_000_000, seed=42)
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 0.0, 1.0, n_samples=1_000_000, seed=42, guard=1)
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"], tol=2e-6)
    assert ok, msg


def test_strict_reference_uniform_matches_strict_oracle():
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    integ = MonteCarloIntegrator(strict_reference_uniform=True)
    res = integ.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=1_000_000, seed=42)
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 0.0, 1.0, n_samples=1_000_000, seed=42, guard=0)
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg


def test_uniform_and_exponential_match_oracle(integrator):
    from wgpu_montecarlo import Distribution

    fns = [lambda x: x, lambda x: x**2, lambda x: math.sin(x), lambda x: math.exp(-x)]
    for dist, code, p1, p2 in ((Distribution.uniform(-1.0, 3.0), oracle.UNIFORM, -1.0, 3.0),
                               (Distribution.exponential(2.0), oracle.EXPONENTIAL, 2.0, 0.0)):
        res = integrator.integrate(fns, dist, n_samples=2_000_000, seed=7)
        xs = oracle.samples(code, p1, p2, n_samples=2_000_000, seed=7, guard=1).astype(np.float64)
        want = [xs.mean(), (xs**2).mean(), np.sin(xs).mean(), np.exp(-xs).mean()]
        assert res.meta["n_eff"] == xs.size
        ok, msg = close(res.values, want)
        assert ok, msg


def test_custom_cdf_table_beta_matches_oracle(integrator):
    """K1 with the 2048-point CDF table of Beta(2,5) (LDS-staged, guide-table search)."""
    from wgpu_montecarlo import Distribution

    dist = Distribution.beta(2.0, 5.0)
    fns = [lambda x: x, lambda x: x**2, lambda x: x**3]
    res = integrator.integrate(fns, dist, n_samples=2_000_000, seed=99)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2), (oracle.FN_POW, 3)], oracle.CUSTOM,
                           n_samples=2_000_000, seed=99, cdf_table=dist._cdf_table, x_table=dist._x_table, guard=1)
    assert res.meta["n_eff"] == ref["n_eff"]
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"], tol=2e-6)
    assert ok, msg
    truth = [2 / 7, 2 * 3 / (7 * 8), 2 * 3 * 4 / (7 * 8 * 9)]
    assert np.all(np.abs(res.values - truth) < 2e-3)


def test_custom_cdf_table_not_monotone_uses_reference_search(integrator):
    """A user CDF table with a dip cannot use the guide table; the capped 12-step search is reproduced."""
    from wgpu_montecarlo import Distribution

    x = np.linspace(0.0, 1.0, 300, dtype=np.float32)
    pdf = np.ones_like(x)
    cdf = np.linspace(0.0, 1.0, 300, dtype=np.float32)
    cdf[100], cdf[101] = cdf[101], cdf[100]
    dist = Distribution.from_pdf_table(x, pdf, cdf)
    res = integrator.integrate([lambda x: x, lambda x: x**2], dist, n_samples=500_000, seed=5)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.CUSTOM, n_samples=500_000, seed=5,
                           cdf_table=cdf, x_table=x, guard=1)
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"], tol=2e-6)
    assert ok, msg


def test_importance_sampling_analytic(integrator):
    """K2, both PDFs analytic (emitted from the Distribution closures)."""
    from wgpu_montecarlo import Distribution

    target, proposal = Distribution.normal(0.0, 1.0), Distribution.normal(0.5, 1.5)
    res = integrator.integrate_importance_sampling(MOMENTS, target, proposal, n_samples=2_000_000, seed=11)
    s2pi = float(np.float32(np.sqrt(2 * np.pi)))
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 0.5, 1.5, n_samples=2_000_000, seed=11, guard=1,
                           p=(oracle.PDF_NORMAL, 0.0, 1.0, s2pi), q=(oracle.PDF_NORMAL, 0.5, 1.5, s2pi))
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    assert np.all(np.abs(res.values - [0, 1, 0, 3]) < [0.01, 0.02, 0.05, 0.2])


def test_q_from_the_sampler_deviate_equals_the_emitted_density(integrator):
    """desc.q_sampler: 1/q from the Box-Muller deviate against q evaluated by the emitted Distribution.normal closure,
    same stream: the weights agree to float rounding, sample by sample."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import _pdf_to_hip, functions_to_hip

    eng = integrator._engine
    target, proposal = Distribution.normal(0.3, 0.8), Distribution.normal(-0.4, 1.7)
    src = functions_to_hip(MOMENTS) + "\n\n" + _pdf_to_hip(target, "mcx_pdf_p")
    with_q = src + "\n\n" + _pdf_to_hip(proposal, "mcx_pdf_q")
    plain = eng.module(with_q, rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, weight=True))
    fused = eng.module(src, rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, weight=True, q_sampler=True))
    a, n_eff = eng.integrate(plain, 3_000_000, 9, -0.4, 1.7)
    b, _ = eng.integrate(fused, 3_000_000, 9, -0.4, 1.7)
    assert np.allclose(a, b, rtol=2e-6, atol=2e-6 * n_eff * 1e-3), (a, b)
    m = b / n_eff
    assert abs(m[0] - 0.3) < 0.01 and abs(m[1] - (0.64 + 0.09)) < 0.02
    with pytest.raises(ValueError, match="q_sampler"):
        rt.module_source("", rt.make_desc(rt.KIND_INTEGRATE, 1, rt.DIST_UNIFORM, weight=True, q_sampler=True))


def test_importance_sampling_target_table(integrator):
    """K2 with a 512-point target PDF table (BASELINE config 3 shape, reference tests/test_importance_sampling.py:335-346)."""
    from wgpu_montecarlo import Distribution

    x = np.linspace(0, 10, 512)
    target = Distribution.from_pdf_table(x, np.exp(-x))
    proposal = Distribution.normal(2.0, 3.0)
    res = integrator.integrate_importance_sampling(MOMENTS, target, proposal, n_samples=2_000_000, seed=3)
    s2pi = float(np.float32(np.sqrt(2 * np.pi)))
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 2.0, 3.0, n_samples=2_000_000, seed=3, guard=1,
                           p=(oracle.PDF_TABLE, target._x_table, target._pdf_table),
                           q=(oracle.PDF_NORMAL, 2.0, 3.0, s2pi))
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    # truth: integral of x^k * exp(-x) on [0,10] ~= 1, 2, 6, 24 minus tails
    assert abs(res.values[0] - 1.0) < 0.01 and abs(res.values[1] - 2.0) < 0.03


def test_importance_sampling_both_tables_nonuniform_grid(integrator):
    """Both PDFs from tables; the proposal's grid is non-uniform so the binary-search path runs."""
    from wgpu_montecarlo import Distribution

    xt = np.linspace(-4, 4, 700)
    target = Distribution.from_pdf_table(xt, np.exp(-0.5 * xt * xt) / np.sqrt(2 * np.pi))
    xq = np.sign(np.linspace(-1, 1, 901)) * np.abs(np.linspace(-1, 1, 901)) ** 1.5 * 6.0
    proposal = Distribution.from_pdf_table(xq, np.exp(-0.5 * (xq / 2) ** 2) / (2 * np.sqrt(2 * np.pi)))
    res = integrator.integrate_importance_sampling([lambda x: x, lambda x: x**2], target, proposal,
                                                   n_samples=1_000_000, seed=21)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.CUSTOM, n_samples=1_000_000,
                           seed=21, guard=1, cdf_table=proposal._cdf_table, x_table=proposal._x_table,
                           p=(oracle.PDF_TABLE, target._x_table, target._pdf_table),
                           q=(oracle.PDF_TABLE, proposal._x_table, proposal._pdf_table))
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    assert abs(res.values[1] - 1.0) < 0.02


@pytest.mark.parametrize("math_mode", ["default", "precise"])
def test_importance_sampling_nearly_uniform_grid(math_mode):
    """A target table whose keys are only *nearly* a linspace is not given slope-intercept cells
    (mcx_table_cells = 0), so the module is built without cell_tables and both lookups run the key/value form
    (guessed-and-verified for the target, also for the strict-grid proposal); math="precise" never uses cells."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt

    rng = np.random.default_rng(0)
    xt = np.linspace(-4, 4, 700)
    xt[1:-1] += rng.uniform(-0.2, 0.2, 698) * (xt[1] - xt[0])
    target = Distribution.from_pdf_table(xt, np.exp(-0.5 * xt * xt) / np.sqrt(2 * np.pi))
    xq = np.linspace(-9, 9, 1200)
    proposal = Distribution.from_pdf_table(xq, np.exp(-0.5 * (xq / 2) ** 2) / (2 * np.sqrt(2 * np.pi)))
    assert rt.table_cells(target._x_table, target._pdf_table) is None
    assert rt.table_cells(proposal._x_table, proposal._pdf_table) is not None
    mc = MonteCarloIntegrator(math=math_mode)
    res = mc.integrate_importance_sampling([lambda x: x, lambda x: x**2], target, proposal, n_samples=1_000_000, seed=5)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.CUSTOM, n_samples=1_000_000,
                           seed=5, guard=1, cdf_table=proposal._cdf_table, x_table=proposal._x_table,
                           p=(oracle.PDF_TABLE, target._x_table, target._pdf_table),
                           q=(oracle.PDF_TABLE, proposal._x_table, proposal._pdf_table))
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"], tol=2e-6 if math_mode == "precise" else TOL)
    assert ok, msg
    assert abs(res.values[1] - 1.0) < 0.02
    assert not mc._table(rt.TABLE_PDF, target._x_table, target._pdf_table).has_cells


def test_cell_tables_module_rejects_a_table_without_cells(integrator):
    """desc.cell_tables is a promise about the tables bound at launch; libmcx checks it (MCX_E_INVALID)."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    eng = integrator._engine
    xs = np.linspace(-3, 3, 300)
    strict = integrator._table(rt.TABLE_PDF, xs, np.exp(-0.5 * xs * xs))
    bumpy_x = xs.copy()
    bumpy_x[100] += 0.2 * (xs[1] - xs[0])
    bumpy = integrator._table(rt.TABLE_PDF, bumpy_x, np.exp(-0.5 * bumpy_x * bumpy_x))
    assert strict.has_cells and not bumpy.has_cells
    mod = eng.module(functions_to_hip(MOMENTS[:1]), rt.make_desc(rt.KIND_INTEGRATE, 1, rt.DIST_NORMAL, weight=True,
                                                                   p_table=True, q_table=True, cell_tables=True))
    wide = np.linspace(-8, 8, 500)
    q = integrator._table(rt.TABLE_PDF, wide, np.exp(-0.5 * wide * wide))
    sums, n_eff = eng.integrate(mod, 100_000, 1, 0.0, 1.0, target_pdf=strict, proposal_pdf=q)
    assert np.isfinite(sums[0])
    with pytest.raises(ValueError, match="no cell form"):
        eng.integrate(mod, 100_000, 1, 0.0, 1.0, target_pdf=bumpy, proposal_pdf=q)
    with pytest.raises(ValueError, match="precise_sampler"):
        rt.module_source("", rt.make_desc(rt.KIND_INTEGRATE, 1, rt.DIST_NORMAL, cell_tables=True, precise_sampler=True))


def _mcmc_oracle(target, proposal, code, p1, p2, **kw):
    tx, tl = target.get_log_pdf_table()
    px, pl = proposal.get_log_pdf_table()
    return oracle.mcmc([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], code, p1, p2, tx, tl, px, pl,
                       cdf_table=proposal._cdf_table, x_table=proposal._x_table if proposal._cdf_table is not None else None,
                       guard=1, **kw)


@pytest.mark.parametrize("n_chains,n_steps,n_burnin", [(256, 2000, 200), (1000, 501, 0), (1, 300, 7)])
def test_mcmc_normal_proposal_matches_oracle(integrator, n_chains, n_steps, n_burnin):
    """K3: bimodal custom target, normal proposal (BASELINE config 4 shape); padded chain count is bit-exact."""
    from wgpu_montecarlo import Distribution

    target = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)),
                                   support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    res = integrator.integrate_mcmc([lambda x: x, lambda x: x**2], target, proposal, n_steps=n_steps,
                                    n_chains=n_chains, n_burnin=n_burnin, seed=42)
    ref = _mcmc_oracle(target, proposal, oracle.NORMAL, 0.0, 2.0, n_steps=n_steps, n_chains=n_chains,
                       n_burnin=n_burnin, seed=42)
    assert res.meta["n_eff"] == ref["n_eff"] == ((n_chains + 255) // 256) * 256 * n_steps
    assert res.n_samples == n_chains * n_steps
    # an accept decision can flip where log(u) and log_alpha agree to ~1 ulp (v_log_f32 vs logf): allow 2e-4
    ok, msg = close(res.values, ref["sums"][:2] / ref["n_eff"], tol=2e-4)
    assert ok, msg
    total_steps = (ref["n_eff"] // n_steps) * (n_steps + n_burnin)
    assert abs(res.meta["accept_rate"] - ref["sums"][2] / total_steps) < 1e-4


@pytest.mark.parametrize("kind", ["uniform", "exponential", "custom"])
def test_mcmc_other_proposals_match_oracle(integrator, kind):
    from wgpu_montecarlo import Distribution

    if kind == "uniform":
        target, proposal, code, p1, p2 = Distribution.beta(2.0, 5.0), Distribution.uniform(0.0, 1.0), oracle.UNIFORM, 0.0, 1.0
    elif kind == "exponential":
        target, proposal, code, p1, p2 = Distribution.exponential(1.0), Distribution.exponential(0.5), oracle.EXPONENTIAL, 0.5, 0.0
    else:
        target = Distribution.normal(0.0, 1.0)
        proposal = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-12, 12))
        code, p1, p2 = oracle.CUSTOM, 0.0, 0.0
    res = integrator.integrate_mcmc([lambda x: x, lambda x: x**2], target, proposal, n_steps=1500, n_chains=512,
                                    n_burnin=100, seed=8)
    ref = _mcmc_oracle(target, proposal, code, p1, p2, n_steps=1500, n_chains=512, n_burnin=100, seed=8)
    ok, msg = close(res.values, ref["sums"][:2] / ref["n_eff"], tol=2e-4)
    assert ok, msg


def test_same_call_is_bit_reproducible(integrator):
    from wgpu_montecarlo import Distribution

    a = integrator.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=5_000_000, seed=77).values
    b = integrator.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=5_000_000, seed=77).values
    assert np.array_equal(a, b)
    c = integrator.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=5_000_000, seed=78).values
    assert not np.array_equal(a, c)


@pytest.mark.parametrize("dist_code", [0, 1, 3])
def test_shards_partition_the_sample_grid(integrator, dist_code):
    """Sum over (rank, world) shards == the single-GPU sums: the N-GPU job draws the same samples."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    eng = integrator._engine
    dist = {0: Distribution.uniform(0, 2), 1: Distribution.normal(0, 1), 3: Distribution.beta(2, 5)}[dist_code]
    cdf = integrator._cdf_table(dist)
    p1, p2 = {0: (0.0, 2.0), 1: (0.0, 1.0), 3: (0.0, 0.0)}[dist_code]
    mod = eng.module(functions_to_hip(MOMENTS), rt.make_desc(rt.KIND_INTEGRATE, 4, dist_code))
    for n in (3_000_001, 70_000):      # L = 46 / 2: the second case has fewer units than 8 ranks for the normal
        whole, n_eff = eng.integrate(mod, n, 5, p1, p2, cdf=cdf)
        for world in (2, 3, 8):
            parts = [eng.integrate(mod, n, 5, p1, p2, cdf=cdf, rank=r, world=world) for r in range(world)]
            assert all(pe == n_eff for _, pe in parts)
            total = np.sum([p for p, _ in parts], axis=0)
            # f32 flush blocks move with the shard boundaries: compare relative to the summed magnitude (~N_eff)
            assert np.allclose(total, whole, rtol=1e-8, atol=1e-9 * n_eff), (world, total, whole)


def test_mcmc_chain_shards_partition(integrator):
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    eng = integrator._engine
    target, proposal = Distribution.normal(0.0, 1.0), Distribution.normal(0.0, 1.5)
    tt = integrator._table(rt.TABLE_LOGPDF, *target.get_log_pdf_table())
    qt = integrator._table(rt.TABLE_LOGPDF, *proposal.get_log_pdf_table())
    mod = eng.module(functions_to_hip(MOMENTS[:2]), rt.make_desc(rt.KIND_MCMC, 2, 1))
    whole, n_eff = eng.mcmc(mod, 400, 2048, 50, 9, 0.0, 1.5, tt, qt)
    for world in (2, 3, 8):
        parts = [eng.mcmc(mod, 400, 2048, 50, 9, 0.0, 1.5, tt, qt, rank=r, world=world)[0] for r in range(world)]
        assert np.allclose(np.sum(parts, axis=0), whole, rtol=1e-8, atol=1e-9 * n_eff)


def test_full_size_c2_within_three_sigma(integrator):
    """BASELINE config 2 at full size: n = 1e9, N_eff bit-exact, every moment within 3 sigma of truth."""
    from wgpu_montecarlo import Distribution

    res = integrator.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=10**9, seed=42)
    assert res.meta["n_eff"] == 1_000_013_824
    sigma = np.sqrt(np.array([1.0, 2.0, 15.0, 96.0]) / res.meta["n_eff"])
    assert np.all(np.abs(res.values - [0, 1, 0, 3]) < 3 * sigma), res.values
    # linearity property, size independent: E[2x + 3x^2] == 2 E[x] + 3 E[x^2] on the same stream
    lin = integrator.integrate([lambda x: 2 * x + 3 * x**2], Distribution.normal(0.0, 1.0), n_samples=10**9, seed=42)
    assert abs(lin.values[0] - (2 * res.values[0] + 3 * res.values[1])) < 1e-6


def test_reference_api_errors(integrator):
    from wgpu_montecarlo import Distribution

    d = Distribution.normal(0, 1)
    with pytest.raises(ValueError):
        integrator.integrate([], d, n_samples=1000)
    with pytest.raises(TypeError):
        integrator.integrate([123], d, n_samples=1000)
    with pytest.raises(ValueError, match="n_steps must be positive"):
        integrator.integrate_mcmc([lambda x: x], d, d, n_steps=0)
    with pytest.raises(ValueError, match="n_chains must be positive"):
        integrator.integrate_mcmc([lambda x: x], d, d, n_chains=0)
    with pytest.raises(ValueError, match="n_burnin must be non-negative"):
        integrator.integrate_mcmc([lambda x: x], d, d, n_burnin=-1)
    with pytest.raises(RuntimeError):
        integrator.integrate(["fn f(x: f32) -> f32 { return undefined_thing(x); }"], d, n_samples=1000)


def test_wgsl_string_functions(integrator):
    """Raw WGSL function strings (reference tests/test_integrator.py:48-71)."""
    from wgpu_montecarlo import Distribution

    d = Distribution.normal(0.0, 1.0)
    res = integrator.integrate([lambda x: x, "fn f(x: f32) -> f32 { return x * x; }"], d, n_samples=1_000_000, seed=42)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_SQ, 0)], oracle.NORMAL, 0.0, 1.0, n_samples=1_000_000,
                           seed=42, guard=1)
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg


@pytest.mark.parametrize("n_samples", [1_000_000, 1_000, 3_211_264, 200_000])
def test_philox_stream_matches_its_oracle(n_samples):
    """rng="philox" (opt-in, not in the reference): same samples as the oracle's restatement for L % 4 = 0, 1, 3, 0
    and every distribution family; the Philox block function itself is pinned by the Random123 KATs."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    mc = MonteCarloIntegrator(rng="philox")
    res = mc.integrate(MOMENTS, Distribution.normal(0.5, 1.5), n_samples=n_samples, seed=77)
    ref = oracle.integrate(ORC_MOMENTS, oracle.NORMAL, 0.5, 1.5, n_samples=n_samples, seed=77, guard=1, rng=1)
    assert res.meta["n_eff"] == ref["n_eff"]
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    beta = Distribution.beta(2.0, 5.0)
    for dist, code, p1, p2, kw in ((Distribution.uniform(-1.0, 3.0), oracle.UNIFORM, -1.0, 3.0, {}),
                                   (Distribution.exponential(2.0), oracle.EXPONENTIAL, 2.0, 0.0, {}),
                                   (beta, oracle.CUSTOM, 0.0, 0.0, dict(cdf_table=beta._cdf_table, x_table=beta._x_table))):
        res = mc.integrate(MOMENTS[:2], dist, n_samples=n_samples, seed=5)
        ref = oracle.integrate(ORC_MOMENTS[:2], code, p1, p2, n_samples=n_samples, seed=5, guard=1, rng=1, **kw)
        ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
        assert ok, msg


def test_philox_shards_and_full_size():
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    mc = MonteCarloIntegrator(rng="philox")
    eng = mc._engine
    mod = eng.module(functions_to_hip(MOMENTS), rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, rng=rt.RNG_PHILOX))
    for n in (3_000_001, 70_000):
        whole, n_eff = eng.integrate(mod, n, 5, 0.0, 1.0)
        for world in (2, 3, 8):
            parts = [eng.integrate(mod, n, 5, 0.0, 1.0, rank=r, world=world)[0] for r in range(world)]
            assert np.allclose(np.sum(parts, axis=0), whole, rtol=1e-8, atol=1e-9 * n_eff)
    res = mc.integrate(MOMENTS, Distribution.normal(0.0, 1.0), n_samples=10**9, seed=42)
    sigma = np.sqrt(np.array([1.0, 2.0, 15.0, 96.0]) / res.meta["n_eff"])
    assert np.all(np.abs(res.values - [0, 1, 0, 3]) < 3.5 * sigma), res.values


@pytest.mark.parametrize("n_prop,n_tgt,in_lds", [(5800, 5000, True), (1001, 777, True), (9000, 7000, False)])
def test_large_tables_and_capped_search(integrator, n_prop, n_tgt, in_lds):
    """Tables up to 156 KiB are staged in LDS (one 1024-thread workgroup per CU); beyond that they are read from
    HBM/L2 (tables_lds = 0). With n > 4096 points the reference's 12-step CDF search is NOT an exact lower bound
    any more: the capped search itself must be reproduced."""
    from wgpu_montecarlo import Distribution

    proposal = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-12, 12), table_size=n_prop)
    xt = np.linspace(-6, 6, n_tgt)
    target = Distribution.from_pdf_table(xt, np.exp(-0.5 * xt * xt) / np.sqrt(2 * np.pi))
    res = integrator.integrate_importance_sampling([lambda x: x, lambda x: x**2], target, proposal, n_samples=1_000_000, seed=4)
    # staged: the proposal and target PDF tables in cell form ((n + 1) * 8 bytes each: two sentinel cells), and for the
    # sampling CDF table either its bucket-direct records (n <= 4096: G = 4 * pow2ceil(n) records of 8 bytes, plus
    # the 16 per-wave queues of 128 words; {cdf, x} and slopes stay in global memory) or, without them (n > 4096: the
    # capped search), {cdf, x} + slopes
    if n_prop <= 4096:
        cdf_bytes = 8 * min(4 * (1 << (n_prop - 1).bit_length()), 8192) + 16 * 128 * 4
    else:
        cdf_bytes = n_prop * 8 + (n_prop * 4 + 7) // 8 * 8
    bare = (n_prop + n_tgt + 2) * 8 + cdf_bytes
    # + the sentinel cells that cover the sampler's range left and right of the target table (desc.cell_noclamp: the proposal
    # on (-12, 12) reaches past the target's (-6, 6)) when they cost no occupancy -- not for the 156 KiB case
    from wgpu_montecarlo import runtime as rt

    t_tab = integrator._table(rt.TABLE_PDF, target._x_table, target._pdf_table)
    q_tab = integrator._table(rt.TABLE_PDF, *proposal.get_or_compute_pdf_table())
    cdf = integrator._cdf_table(proposal)
    if n_prop <= 4096:
        pads = 8 * sum(sum(rt.cell_pads(t, rt.DIST_CUSTOM, 0.0, 0.0, cdf)) for t in (t_tab, q_tab))
        assert pads > 8 * 4 and cdf.reach_known
    else:
        # no guide / bucket-direct form beyond 4096 points: the capped search can stop short of the cell that holds u, so
        # the draws are not provably inside the x column -- the lookups keep their index clamp
        assert not cdf.reach_known and all(rt.cell_pads(t, rt.DIST_CUSTOM, 0.0, 0.0, cdf) is None for t in (t_tab, q_tab))
        pads = 0
    assert res.meta["lds_bytes"] == ((bare + pads if bare + pads <= 80 * 1024 else bare) if in_lds else 0)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.CUSTOM, n_samples=1_000_000, seed=4, guard=1,
                           cdf_table=proposal._cdf_table, x_table=proposal._x_table,
                           p=(oracle.PDF_TABLE, target._x_table, target._pdf_table),
                           q=(oracle.PDF_TABLE, *proposal.get_or_compute_pdf_table()))
    ok, msg = close(res.values, ref["sums"] / ref["n_eff"])
    assert ok, msg
    plain = integrator.integrate([lambda x: x, lambda x: x**2], proposal, n_samples=1_000_000, seed=4)
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.CUSTOM, n_samples=1_000_000, seed=4, guard=1,
                           cdf_table=proposal._cdf_table, x_table=proposal._x_table)
    ok, msg = close(plain.values, ref["sums"] / ref["n_eff"], tol=2e-6)
    assert ok, msg


@pytest.mark.parametrize("kind", ["normal", "uniform", "custom"])
def test_philox_mcmc_matches_its_oracle(kind):
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    mc = MonteCarloIntegrator(rng="philox")
    target = Distribution.normal(0.3, 1.0)
    if kind == "normal":
        proposal, code, p1, p2 = Distribution.normal(0.0, 2.0), oracle.NORMAL, 0.0, 2.0
    elif kind == "uniform":
        proposal, code, p1, p2 = Distribution.uniform(-6.0, 7.0), oracle.UNIFORM, -6.0, 7.0
    else:
        proposal = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-12, 12))
        code, p1, p2 = oracle.CUSTOM, 0.0, 0.0
    res = mc.integrate_mcmc([lambda x: x, lambda x: x**2], target, proposal, n_steps=1200, n_chains=700, n_burnin=33, seed=21)
    ref = _mcmc_oracle(target, proposal, code, p1, p2, n_steps=1200, n_chains=700, n_burnin=33, seed=21, rng=1)
    assert res.meta["n_eff"] == ref["n_eff"] == 768 * 1200
    ok, msg = close(res.values, ref["sums"][:2] / ref["n_eff"], tol=2e-4)
    assert ok, msg
    assert abs(res.values[0] - 0.3) < 0.02 and abs(res.values[1] - 1.09) < 0.03


@pytest.mark.parametrize("case", ["beta_k32", "normal_k12", "is_table_k9", "philox_uniform_k16"])
def test_moment_family_pairs_match_per_sample_evaluation(integrator, case):
    """desc.moment_family: x, x**2, .., x**K accumulated two samples at a time through Newton's identity
    s_k = (a + b) s_{k-1} - a b s_{k-2} (below 12 rows on the reference stream), four at a time through the recurrence of
    their quartic above, against the per-sample multiply chain on the same stream, and against the oracle. Sums of powers of samples of mixed sign cancel in both forms; the bound is relative to sum |x|^k."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import _moment_family, functions_to_hip

    k = int(case.rsplit("k", 1)[1])
    fns = [lambda x, p=p: x**p for p in range(1, k + 1)]
    assert _moment_family(fns)
    mc = MonteCarloIntegrator(rng="philox" if case.startswith("philox") else "pcg_ref")
    eng = mc._engine
    n, seed = 2_000_001, 13
    src = functions_to_hip(fns)
    if case == "beta_k32":
        dist = Distribution.beta(2.0, 5.0)
        common = dict(kind=rt.KIND_INTEGRATE, k=k, dist_type=rt.DIST_CUSTOM)
        call = lambda mod: eng.integrate(mod, n, seed, 0.0, 0.0, cdf=mc._cdf_table(dist))
        res = mc.integrate(fns, dist, n_samples=n, seed=seed)
        xs = oracle.samples(oracle.CUSTOM, n_samples=n, seed=seed, guard=1, cdf_table=dist._cdf_table, x_table=dist._x_table)
    elif case == "normal_k12":
        dist = Distribution.normal(0.2, 0.9)
        common = dict(kind=rt.KIND_INTEGRATE, k=k, dist_type=rt.DIST_NORMAL)
        call = lambda mod: eng.integrate(mod, n, seed, 0.2, 0.9)
        res = mc.integrate(fns, dist, n_samples=n, seed=seed)
        xs = oracle.samples(oracle.NORMAL, 0.2, 0.9, n_samples=n, seed=seed, guard=1)
    elif case == "philox_uniform_k16":
        dist = Distribution.uniform(-1.0, 1.5)
        common = dict(kind=rt.KIND_INTEGRATE, k=k, dist_type=rt.DIST_UNIFORM, rng=rt.RNG_PHILOX)
        call = lambda mod: eng.integrate(mod, n, seed, -1.0, 1.5)
        res = mc.integrate(fns, dist, n_samples=n, seed=seed)
        xs = None
    else:
        xt = np.linspace(-3.0, 3.0, 400)
        target, proposal = Distribution.from_pdf_table(xt, np.exp(-0.5 * xt * xt)), Distribution.normal(0.0, 1.3)
        tb = mc._table(rt.TABLE_PDF, target._x_table, target._pdf_table)
        common = dict(kind=rt.KIND_INTEGRATE, k=k, dist_type=rt.DIST_NORMAL, weight=True, p_table=True, q_sampler=True,
                      cell_tables=True)
        call = lambda mod: eng.integrate(mod, n, seed, 0.0, 1.3, target_pdf=tb)
        res = mc.integrate_importance_sampling(fns, target, proposal, n_samples=n, seed=seed)
        xs = None
    plain, n_eff = call(eng.module(src, rt.make_desc(**common)))
    paired, _ = call(eng.module(src, rt.make_desc(moment_family=True, **common)))
    scale = np.maximum(np.abs(plain), 1e-3 * n_eff)          # sums of x^k with mixed signs cancel
    assert np.all(np.abs(paired - plain) <= 3e-6 * scale + 1e-4), (paired, plain)
    assert np.allclose(res.values * n_eff, paired, rtol=1e-12, atol=1e-9 * n_eff)     # the API took the paired path
    if xs is not None:
        xs = xs.astype(np.float64).ravel()
        want = np.array([(xs**p).sum() for p in range(1, k + 1)])
        mag = np.array([(np.abs(xs) ** p).sum() for p in range(1, k + 1)])
        assert np.all(np.abs(paired - want) <= 2e-5 * mag), np.abs(paired - want) / mag
