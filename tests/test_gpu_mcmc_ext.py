"""GPU tests for the K3 extensions (random-walk MH, batch-means diagnostics) against the oracle's restatement of
libmcx's definition. Parity unpinned with respect to the reference (it has neither, src/shader_gen.rs:514);
the chains use the reference's counter stream, so the oracle and the kernel walk the same chains.

Tolerance: an accept decision flips where log(u) and log_alpha agree to ~1 ulp (v_log_f32 vs logf). With
independent proposals the chains re-join at the next acceptance (2e-4 as in test_gpu_parity); a random-walk chain
that flipped stays apart for the rest of its run, so means over ~1e6 steps are held to 3e-3 instead.
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

F = [lambda x: x, lambda x: x**2]
ORC_F = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)]


def _oracle(target, step, code, p1, p2, **kw):
    tx, tl = target.get_log_pdf_table()
    px, pl = step.get_log_pdf_table()
    return oracle.mcmc(ORC_F, code, p1, p2, tx, tl, px, pl, cdf_table=step._cdf_table,
                       x_table=step._x_table if step._cdf_table is not None else None, guard=1, **kw)


def _bimodal():
    from wgpu_montecarlo import Distribution

    return Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)),
                                 support=(-10, 10))


@pytest.mark.parametrize("rng", ["pcg_ref", "philox"])
def test_random_walk_symmetric_matches_oracle(rng):
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    mc = MonteCarloIntegrator(rng=rng)
    target, step = _bimodal(), Distribution.normal(0.0, 2.5)
    res = mc.integrate_mcmc(F, target, step, n_steps=1500, n_chains=700, n_burnin=200, seed=5,
                            proposal_kind="random_walk")
    ref = _oracle(target, step, oracle.NORMAL, 0.0, 2.5, n_steps=1500, n_chains=700, n_burnin=200, seed=5,
                  walk=2, rng=int(rng == "philox"))
    assert res.meta["n_eff"] == ref["n_eff"] == 768 * 1500
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.all(np.abs(res.values - want) < 3e-3), (res.values, want)
    assert abs(res.meta["accept_rate"] - ref["sums"][2] / (768 * 1700)) < 1e-3
    assert abs(res.values[0]) < 0.05 and abs(res.values[1] - 5.0) < 0.1
    assert res.meta["proposal_kind"] == "random_walk"


@pytest.mark.parametrize("kind", ["normal_drift", "uniform_skew", "custom"])
def test_random_walk_general_matches_oracle(integrator, kind):
    """Asymmetric increments: the Hastings correction log q(-d) - log q(d) keeps the chain on target."""
    from wgpu_montecarlo import Distribution

    target = Distribution.normal(0.0, 1.0)
    if kind == "normal_drift":
        step, code, p1, p2 = Distribution.normal(0.7, 1.0), oracle.NORMAL, 0.7, 1.0
    elif kind == "uniform_skew":
        step, code, p1, p2 = Distribution.uniform(-1.0, 1.5), oracle.UNIFORM, -1.0, 1.5
    else:
        step = Distribution.from_pdf(lambda x: math.exp(-abs(x - 0.2)) / 2, support=(-8, 8))
        code, p1, p2 = oracle.CUSTOM, 0.0, 0.0
    res = integrator.integrate_mcmc(F, target, step, n_steps=2000, n_chains=512, n_burnin=300, seed=11,
                                    proposal_kind="random_walk")
    ref = _oracle(target, step, code, p1, p2, n_steps=2000, n_chains=512, n_burnin=300, seed=11, walk=1)
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.all(np.abs(res.values - want) < 3e-3), (res.values, want)
    assert abs(res.values[0]) < 0.03 and abs(res.values[1] - 1.0) < 0.05, res.values


def test_initial_state_and_one_step(integrator):
    """n_steps = 1, no burn-in: the state after one step from x0 + d_0, same as the oracle's trace."""
    from wgpu_montecarlo import Distribution

    target, step = Distribution.normal(3.0, 1.0), Distribution.normal(0.0, 0.25)
    res = integrator.integrate_mcmc(F, target, step, n_steps=1, n_chains=512, n_burnin=0, seed=2,
                                    proposal_kind="random_walk", initial_state=3.0)
    ref = _oracle(target, step, oracle.NORMAL, 0.0, 0.25, n_steps=1, n_chains=512, n_burnin=0, seed=2, walk=2, x0=3.0)
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.all(np.abs(res.values - want) < 2e-4), (res.values, want)
    assert abs(res.values[0] - 3.0) < 0.05


def test_start_outside_the_support(integrator):
    """Chains started outside the target table wait for a proposal that lands inside (proposals with
    log p <= -100 are rejected); same waiting chains as the oracle."""
    from wgpu_montecarlo import Distribution

    target, step = _bimodal(), Distribution.normal(0.0, 2.5)
    kw = dict(n_steps=3000, n_chains=512, n_burnin=500, seed=4)
    res = integrator.integrate_mcmc(F, target, step, proposal_kind="random_walk", initial_state=10.5, **kw)
    ref = _oracle(target, step, oracle.NORMAL, 0.0, 2.5, walk=2, x0=10.5, **kw)
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.all(np.abs(res.values - want) < 3e-3 * (1 + np.abs(want))), (res.values, want)
    # the few chains that start several sigma outside wait long (x^2 ~ 300 while they do); the bulk is on target
    assert abs(res.values[0]) < 0.3 and abs(res.values[1] - 5.0) < 2.0, res.values


@pytest.mark.parametrize("proposal_kind", ["independent", "random_walk"])
def test_batch_means_rows_match_oracle(proposal_kind):
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    target = _bimodal()
    proposal = Distribution.normal(0.0, 2.0)
    walk = 2 if proposal_kind == "random_walk" else 0
    tol = 3e-3 if walk else 2e-4
    kw = dict(n_steps=800, n_chains=1024, n_burnin=100, seed=17)
    plain = MonteCarloIntegrator().integrate_mcmc(F, target, proposal, proposal_kind=proposal_kind, **kw)
    diag = MonteCarloIntegrator(std_error=True).integrate_mcmc(F, target, proposal, proposal_kind=proposal_kind, **kw)
    assert np.array_equal(plain.values, diag.values)           # the extra rows do not disturb the chains
    assert plain.meta["accept_rate"] == diag.meta["accept_rate"]
    ref = _oracle(target, proposal, oracle.NORMAL, 0.0, 2.0, walk=walk, **kw)
    n_eff, T = ref["n_eff"], 1024
    mean = ref["sums"][:2] / n_eff
    var_f = ref["sumsq"] / n_eff - mean**2
    var_between = ref["chain_mean_sq"] / T - mean**2
    want_se = np.sqrt(var_between / (T - 1))
    want_tau = 800 * var_between / var_f
    assert np.allclose(diag.meta["std_error"], want_se, rtol=20 * tol), (diag.meta["std_error"], want_se)
    assert np.allclose(diag.meta["tau_int"], want_tau, rtol=40 * tol), (diag.meta["tau_int"], want_tau)
    assert np.allclose(diag.meta["ess"], n_eff / want_tau, rtol=40 * tol)
    # the standard error is honest: truth (0, 5) lies within 4 of them
    assert np.all(np.abs(diag.values - [0.0, 5.0]) < 4 * diag.meta["std_error"] + 1e-3), (diag.values, diag.meta["std_error"])
    assert np.all(diag.meta["ess"] <= 1.2 * n_eff) and np.all(diag.meta["ess"] > 0.01 * n_eff)


def test_diagnostics_shard_and_sum(integrator):
    """All 3k + 1 rows are plain sums over chains: rank shards add up to the whole."""
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    eng = integrator._engine
    target, step = Distribution.normal(0.0, 1.0), Distribution.normal(0.0, 1.5)
    tt = integrator._table(rt.TABLE_LOGPDF, *target.get_log_pdf_table())
    qt = integrator._table(rt.TABLE_LOGPDF, *step.get_log_pdf_table())
    for walk in (0, 1, 2):
        desc = rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, second_moments=True, walk=walk)
        assert rt.result_rows(desc) == 7
        mod = eng.module(functions_to_hip(F), desc)
        whole, n_eff = eng.mcmc(mod, 300, 2048, 40, 9, 0.0, 1.5, tt, qt, x0=0.5)
        assert whole.shape == (7,) and np.all(whole[2:4] > 0) and np.all(whole[5:7] >= 0)
        for world in (2, 3):
            parts = [eng.mcmc(mod, 300, 2048, 40, 9, 0.0, 1.5, tt, qt, rank=r, world=world, x0=0.5)[0] for r in range(world)]
            assert np.allclose(np.sum(parts, axis=0), whole, rtol=1e-9, atol=1e-9 * n_eff)


def test_unknown_proposal_kind(integrator):
    from wgpu_montecarlo import Distribution

    d = Distribution.normal(0, 1)
    with pytest.raises(ValueError, match="proposal_kind"):
        integrator.integrate_mcmc(F, d, d, proposal_kind="langevin")


@pytest.mark.parametrize("case", ["normal_small", "normal_large", "uniform_philox"])
def test_adaptive_random_walk_matches_oracle(case):
    """proposal_kind="adaptive_random_walk": the per-chain step scale tuned during burn-in, same as the oracle's
    restatement; acceptance of the sampling phase near the target; the mean final scale reported."""
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    target = Distribution.normal(0.5, 1.0)
    step, code, p1, p2, rng = {"normal_small": (Distribution.normal(0.0, 0.05), oracle.NORMAL, 0.0, 0.05, "pcg_ref"),
                               "normal_large": (Distribution.normal(0.0, 25.0), oracle.NORMAL, 0.0, 25.0, "pcg_ref"),
                               "uniform_philox": (Distribution.uniform(-0.3, 0.3), oracle.UNIFORM, -0.3, 0.3, "philox")}[case]
    kw = dict(n_steps=1200, n_chains=512, n_burnin=1200, seed=9)
    mc = MonteCarloIntegrator(rng=rng, std_error=True)
    res = mc.integrate_mcmc(F, target, step, proposal_kind="adaptive_random_walk", initial_state=0.5, target_accept=0.4, **kw)
    ref = _oracle(target, step, code, p1, p2, walk=3, x0=0.5, target_accept=0.4, rng=int(rng == "philox"), **kw)
    want = ref["sums"][:2] / ref["n_eff"]
    assert np.all(np.abs(res.values - want) < 6e-3), (res.values, want)
    assert abs(res.meta["step_scale"] - ref["scale_sum"] / 512) < 0.02 * ref["scale_sum"] / 512
    assert abs(res.values[0] - 0.5) < 0.05 and abs(res.values[1] - 1.25) < 0.1, res.values
    # acceptance over burn-in + sampling is pulled towards the target from either side
    assert 0.3 < res.meta["accept_rate"] < 0.55, res.meta["accept_rate"]
    assert np.all(np.isfinite(res.meta["std_error"]))


def test_adaptive_random_walk_validation(integrator):
    from wgpu_montecarlo import Distribution

    d = Distribution.normal(0, 1)
    with pytest.raises(ValueError, match="symmetric"):
        integrator.integrate_mcmc(F, d, Distribution.normal(0.3, 1.0), proposal_kind="adaptive_random_walk")
    with pytest.raises(ValueError, match="target_accept"):
        integrator.integrate_mcmc(F, d, d, proposal_kind="adaptive_random_walk", target_accept=1.5)
