"""CPU tests for the two K3 extensions that sit either side of the reference's sampler:
random-walk Metropolis-Hastings and the batch-means diagnostics. The reference has neither
("For now, we use independent proposal", src/shader_gen.rs:514), so these are *parity unpinned*: the oracle
restates libmcx's own definition (include/mcx.h: mcx_module_desc.walk / .second_moments) and is checked here
against closed-form truths; the GPU tests then hold the HIP kernel to the oracle.
"""
import math

import numpy as np
import pytest

import oracle
from wgpu_montecarlo import Distribution
from wgpu_montecarlo import runtime as rt

FNS = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)]


def _tables(target, proposal):
    tx, tl = target.get_log_pdf_table()
    px, pl = proposal.get_log_pdf_table()
    return tx, tl, px, pl


def test_random_walk_symmetric_recovers_bimodal_moments():
    target = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)),
                                   support=(-10, 10))
    step = Distribution.normal(0.0, 2.5)
    r = oracle.mcmc(FNS, oracle.NORMAL, 0.0, 2.5, *_tables(target, step), n_steps=2000, n_chains=512, n_burnin=300,
                    seed=5, guard=1, walk=2)
    mean = r["sums"][:2] / r["n_eff"]
    assert abs(mean[0]) < 0.05 and abs(mean[1] - 5.0) < 0.1, mean
    rate = r["sums"][2] / (512 * 2300)
    assert 0.3 < rate < 0.8


def test_hastings_correction_for_drifting_increments():
    """Increments N(0.7, 1) are not symmetric: walk = 1 stays unbiased on N(0,1), walk = 2 (no correction) does not."""
    target = Distribution.normal(0.0, 1.0)
    step = Distribution.normal(0.7, 1.0)
    args = dict(n_steps=3000, n_chains=512, n_burnin=300, seed=11, guard=1)
    good = oracle.mcmc(FNS, oracle.NORMAL, 0.7, 1.0, *_tables(target, step), walk=1, **args)
    bad = oracle.mcmc(FNS, oracle.NORMAL, 0.7, 1.0, *_tables(target, step), walk=2, **args)
    g, b = good["sums"][:2] / good["n_eff"], bad["sums"][:2] / bad["n_eff"]
    assert abs(g[0]) < 0.02 and abs(g[1] - 1.0) < 0.04, g
    assert b[0] > 0.15, b                      # the drift shows without the correction


def test_symmetric_and_general_walk_agree_for_symmetric_increments():
    target = Distribution.normal(1.0, 0.5)
    step = Distribution.uniform(-1.0, 1.0)
    args = dict(n_steps=1500, n_chains=256, n_burnin=100, seed=3, guard=1, x0=1.0)
    a = oracle.mcmc(FNS, oracle.UNIFORM, -1.0, 1.0, *_tables(target, step), walk=1, **args)
    b = oracle.mcmc(FNS, oracle.UNIFORM, -1.0, 1.0, *_tables(target, step), walk=2, **args)
    # log q(d) = log q(-d) = log(1/2) exactly inside the support: identical chains
    assert np.array_equal(a["sums"], b["sums"])
    m = a["sums"][:2] / a["n_eff"]
    assert abs(m[0] - 1.0) < 0.01 and abs(m[1] - 1.25) < 0.03


def test_initial_state_moves_the_start():
    target = Distribution.normal(0.0, 1.0)
    step = Distribution.normal(0.0, 0.1)
    base = dict(n_steps=1, n_chains=256, n_burnin=0, seed=2, guard=1, walk=2, trace_chains=256)
    a = oracle.mcmc(FNS, oracle.NORMAL, 0.0, 0.1, *_tables(target, step), x0=0.0, **base)
    b = oracle.mcmc(FNS, oracle.NORMAL, 0.0, 0.1, *_tables(target, step), x0=3.0, **base)
    assert abs(np.mean(a["trace"])) < 0.1
    assert abs(np.mean(b["trace"]) - 3.0) < 0.25       # one step from 3 + d_0 towards the mode


def test_chains_started_outside_the_support_walk_back_in():
    """log p = -100 outside the table is 'density 0' for the random walk: proposals there are rejected, so a chain
    that starts outside waits for a move that lands inside instead of diffusing over a flat landscape."""
    target = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)),
                                   support=(-10, 10))
    step = Distribution.normal(0.0, 2.5)
    r = oracle.mcmc(FNS, oracle.NORMAL, 0.0, 2.5, *_tables(target, step), n_steps=3000, n_chains=256, n_burnin=500,
                    seed=4, guard=1, walk=2, x0=12.0, trace_chains=256)
    tr = r["trace"].astype(np.float64)
    assert np.all(np.abs(tr[:, -1]) <= 10.0)                    # every chain ended inside
    assert np.all(np.abs(tr) <= 12.0 + 4 * 2.5)                 # and none ever moved further out than its start
    # a chain that started k sigma beyond the edge waits ~1/P(d < -k sigma) steps: the typical chain is in early
    late = np.mean(tr[:, -1000:] ** 2, axis=1)
    assert abs(np.median(late) - 5.0) < 0.6, np.median(late)


def test_batch_means_rows_of_the_oracle():
    """sumsq / chain_mean_sq against the same quantities recomputed from the traced chains."""
    target = Distribution.normal(0.0, 1.0)
    proposal = Distribution.normal(0.0, 2.0)
    r = oracle.mcmc(FNS, oracle.NORMAL, 0.0, 2.0, *_tables(target, proposal), n_steps=400, n_chains=256, n_burnin=50,
                    seed=9, guard=1, trace_chains=256)
    tr = r["trace"].astype(np.float64)
    f32 = r["trace"]
    assert np.allclose(r["sumsq"][0], np.sum((f32 * f32).astype(np.float64)), rtol=1e-12)
    assert np.allclose(r["chain_mean_sq"][0], np.sum(np.mean(tr, axis=1) ** 2), rtol=1e-12)
    # independent sampler with a wide proposal: ESS is a sizeable fraction of the draws, never more than all of them
    n_eff, T = r["n_eff"], 256
    mean = r["sums"][0] / n_eff
    var_f = r["sumsq"][0] / n_eff - mean**2
    var_between = r["chain_mean_sq"][0] / T - mean**2
    tau = 400 * var_between / var_f
    assert 0.8 < tau < 6.0


def test_result_rows_and_walk_validation():
    d = rt.make_desc(rt.KIND_MCMC, 3, rt.DIST_NORMAL)
    assert rt.result_rows(d) == 4
    d = rt.make_desc(rt.KIND_MCMC, 3, rt.DIST_NORMAL, second_moments=True)
    assert rt.result_rows(d) == 10
    d = rt.make_desc(rt.KIND_INTEGRATE, 5, rt.DIST_NORMAL, second_moments=True)
    assert rt.result_rows(d) == 10
    with pytest.raises(ValueError, match="k <= 16"):
        rt.result_rows(rt.make_desc(rt.KIND_MCMC, 17, rt.DIST_NORMAL, second_moments=True))
    with pytest.raises(ValueError, match="walk"):
        rt.result_rows(rt.make_desc(rt.KIND_INTEGRATE, 2, rt.DIST_NORMAL, walk=1))
    with pytest.raises(ValueError, match="walk"):
        rt.result_rows(rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, walk=4))
    assert rt.result_rows(rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, walk=3)) == 4          # + the step-scale row
    assert rt.result_rows(rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, walk=3, second_moments=True)) == 8


@pytest.mark.parametrize("walk", [1, 2])
def test_walk_modules_compile_for_gfx950(walk):
    """hiprtc needs no GPU: the random-walk + diagnostics specialisations build into the code-object cache."""
    from wgpu_montecarlo.api import functions_to_hip

    src = functions_to_hip([lambda x: x, lambda x: x**2])
    desc = rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, second_moments=True, walk=walk)
    text = rt.module_source(src, desc)
    assert f"#define MCX_WALK {walk}" in text and "#define MCX_NF 2" in text and "#define MCX_K 4" in text
    rt.precompile(src, desc)


@pytest.mark.parametrize("s0", [0.05, 1.0, 30.0])
def test_adaptive_walk_tunes_the_step_scale(s0):
    """walk = 3: whatever the increments' own scale, the per-chain scale adapts during burn-in until the acceptance
    rate of the sampling phase is near the target, and the moments of N(0,1) come out right."""
    target = Distribution.normal(0.0, 1.0)
    step = Distribution.normal(0.0, s0)
    args = dict(n_steps=1500, n_chains=256, n_burnin=1500, seed=21, guard=1, walk=3, target_accept=0.44, trace_chains=256)
    r = oracle.mcmc(FNS, oracle.NORMAL, 0.0, s0, *_tables(target, step), **args)
    mean_scale = r["scale_sum"] / 256
    # optimal proposal std for a unit normal target at 44 % acceptance is about 2.4: scale * s0 should land near it
    assert 1.2 < mean_scale * s0 < 4.5, mean_scale * s0
    tr = r["trace"]
    moved = (tr[:, 1:] != tr[:, :-1]).mean()
    assert abs(moved - 0.44) < 0.08, moved
    m = r["sums"][:2] / r["n_eff"]
    assert abs(m[0]) < 0.05 and abs(m[1] - 1.0) < 0.08, m
