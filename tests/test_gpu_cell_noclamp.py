"""desc.cell_noclamp on the GPU: the cell lookup without its index clamp, the tables padded with sentinel cells over the
sampler's whole range instead. Same cell for every in-table x, the reference's outside value (0 / -100,
src/distribution.rs:190-195, 384-389) for every other x the sampler can produce -- so the sums must equal those of the
clamped lookup (MCX_NO_NOCLAMP=1) on the same stream."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bump(x):
    return math.exp(-0.5 * (x - 1.0) ** 2) + 0.3 * math.exp(-2.0 * (x + 1.5) ** 2)


def _calls(mc):
    from wgpu_montecarlo import Distribution

    xs = np.linspace(-4.0, 5.0, 700)
    table = Distribution.from_pdf_table(xs, np.array([_bump(float(x)) for x in xs]))
    custom = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-9.0, 9.0))
    f2 = [lambda x: x, lambda x: x * x]
    out = {}
    # importance sampling: p from a table narrower than the proposal's range, on both sides
    for name, proposal in (("normal", Distribution.normal(0.5, 2.5)), ("uniform", Distribution.uniform(-6.0, 7.5)),
                           ("exponential", Distribution.exponential(0.4)), ("custom", custom)):
        out["is_" + name] = mc.integrate_importance_sampling(f2, table, proposal, n_samples=3_000_001, seed=11)
    # Metropolis-Hastings, independent proposals: log p (and log q for the non-normal proposals) from tables
    target = Distribution.from_pdf(_bump, support=(-4.0, 5.0))
    for name, proposal in (("normal", Distribution.normal(0.0, 3.0)), ("uniform", Distribution.uniform(-5.0, 6.0)),
                           ("custom", custom)):
        out["mh_" + name] = mc.integrate_mcmc(f2, target, proposal, n_steps=400, n_chains=4096, n_burnin=50, seed=5)
    return out, table, target


@pytest.mark.parametrize("rng", ["pcg_ref", "philox"])
def test_noclamp_sums_equal_the_clamped_lookup(monkeypatch, rng):
    from wgpu_montecarlo import MonteCarloIntegrator

    free, table, target = _calls(MonteCarloIntegrator(rng=rng))
    monkeypatch.setenv("MCX_NO_NOCLAMP", "1")
    clamped, _, _ = _calls(MonteCarloIntegrator(rng=rng))
    for key in free:
        a, b = free[key], clamped[key]
        assert a.meta["n_eff"] == b.meta["n_eff"]
        assert a.meta["lds_bytes"] > b.meta["lds_bytes"], key              # the padded sentinels were staged
        # same cells; the index FMA rounds differently where the padded base shifts its constant, so an x within 1e-3 cell of
        # a node may take the neighbouring cell's line there (continuous at the node): agreement far below f32 resolution of the sums
        assert np.allclose(a.values, b.values, rtol=1e-6, atol=1e-9), (key, a.values, b.values)
        if key.startswith("mh_"):
            assert abs(a.meta["accept_rate"] - b.meta["accept_rate"]) < 1e-6


def test_too_wide_a_range_keeps_the_clamp(integrator):
    """A proposal whose range would need more than 4096 extra cells on a side: the API builds the clamped module (same
    LDS bytes as the bare table); a C caller that forces the flag gets MCX_E_INVALID, not an out-of-table read."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    xs = np.linspace(0.0, 1.0, 2048)
    table = Distribution.from_pdf_table(xs, 6.0 * xs * (1.0 - xs))
    wide = Distribution.normal(0.5, 40.0)
    f2 = [lambda x: x, lambda x: x * x]
    res = integrator.integrate_importance_sampling(f2, table, wide, n_samples=2_000_000, seed=3)
    tb = integrator._table(rt.TABLE_PDF, table._x_table, table._pdf_table)
    assert res.meta["lds_bytes"] == tb.lds_bytes
    assert rt.cell_pads(tb, rt.DIST_NORMAL, 0.5, 40.0) is None
    assert rt.cell_pads(tb, rt.DIST_NORMAL, 0.5, 0.1) is not None
    assert rt.cell_pads(tb, rt.DIST_NORMAL, 0.5, 0.1, guard=False) is None          # u1 = 0 -> an infinite deviate
    assert rt.cell_pads(tb, rt.DIST_EXPONENTIAL, -1.0, 0.0) is None
    assert abs(res.values[0] - 0.5) < 0.02

    eng = integrator._engine
    forced = rt.make_desc(rt.KIND_INTEGRATE, 2, rt.DIST_NORMAL, weight=True, p_table=True, q_sampler=True, cell_tables=True,
                          cell_noclamp=True)
    mod = eng.module(functions_to_hip(f2), forced)
    with pytest.raises(ValueError, match="cannot be padded over the sampler's range"):
        eng.integrate(mod, 1_000_000, 3, 0.5, 40.0, target_pdf=tb)
    sums, n_eff = eng.integrate(mod, 1_000_000, 3, 0.5, 0.1, target_pdf=tb)           # the same module, a range that fits
    assert np.all(np.isfinite(sums)) and n_eff >= 1_000_000
    with pytest.raises(ValueError, match="cell_noclamp needs cell_tables"):
        eng.module(functions_to_hip(f2), rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, cell_tables=True, cell_noclamp=True,
                                                      walk=rt.WALK_RANDOM_SYMMETRIC))


def test_cell_address_from_the_mantissa(monkeypatch):
    """desc.cell_addr16 (MCMC modules within 64 KiB of LDS): the cell's byte address is read out of the index FMA's
    mantissa instead of converted. Same chains as with the convert (MCX_NO_ADDR16=1) up to the neighbour-cell rounding at
    nodes; a launch that needs more than 64 KiB of LDS is refused."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    f2 = [lambda x: x, lambda x: x * x]
    target = Distribution.from_pdf(_bump, support=(-4.0, 5.0))

    def run(mc):
        return [mc.integrate_mcmc(f2, target, q, n_steps=400, n_chains=4096, n_burnin=50, seed=5)
                for q in (Distribution.normal(0.0, 3.0), Distribution.uniform(-5.0, 6.0), Distribution.normal(0.0, 40.0))]

    fast = run(MonteCarloIntegrator())
    monkeypatch.setenv("MCX_NO_ADDR16", "1")
    plain = run(MonteCarloIntegrator())
    for a, b in zip(fast, plain):
        assert a.meta["n_eff"] == b.meta["n_eff"]
        assert np.allclose(a.values, b.values, rtol=1e-6, atol=1e-9), (a.values, b.values)
        assert abs(a.meta["accept_rate"] - b.meta["accept_rate"]) < 1e-6
    monkeypatch.delenv("MCX_NO_ADDR16")

    mc = MonteCarloIntegrator()
    big = Distribution.from_pdf(_bump, support=(-4.0, 5.0), table_size=9000)           # 72 KiB of cells
    tx, tl = big.get_log_pdf_table()
    tb = mc._engine.cached_table(rt.TABLE_LOGPDF, tx, tl)
    forced = rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, cell_tables=True, q_sampler=True, cell_addr16=True)
    mod = mc._engine.module(functions_to_hip(f2), forced)
    with pytest.raises(ValueError, match="does not fit 16 bits"):
        mc._engine.mcmc(mod, 50, 4096, 10, 3, 0.0, 3.0, tb, None)
    res = mc.integrate_mcmc(f2, big, Distribution.normal(0.0, 3.0), n_steps=200, n_chains=4096, n_burnin=20, seed=5)   # the API does not set it
    assert np.all(np.isfinite(res.values))
