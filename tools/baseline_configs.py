"""The BASELINE.json configurations C1..C5 as concrete synthetic workloads, with their truths and
Monte-Carlo error bands. Shared by bench.py (--config), tools/run_configs.py and the full-size GPU tests, so that all
three run literally the same calls.

Scenario sources in the reference's own tests (file:line into NightingaleCen/wgpu-monte-carlo):
  C2  tests/test_integrator.py:112          K = 4 moments on N(0,1)
  C3  tests/test_importance_sampling.py:335-346   from_pdf_table(linspace(0,10,512), exp(-x)) target
  C4  tests/test_mcmc.py:351-372            bimodal custom-PDF target, Normal(0,2) proposal
  C5  tests/test_distributions.py:78-110    Beta(2,5), here with K = 32 fused moments
"""
from __future__ import annotations

import math

import numpy as np

# VALU cost per unit (sample or MH step) in lane-op equivalents with the survey's weights (plain op 1 -- an
# FMA is one op --, integer multiply 4, transcendental 2), counted on the ALGORITHM THE SHIPPED KERNEL RUNS: the hot loop of
# the cached code object, disassembled and classified by tools/issue_model.py -> profiles/rNN_issue_model.json
# (`survey_weighted_ops_per_unit`, keyed by the module's cache key). The constants below are that file's figures for the
# round-3 kernels, used when the file is absent; ops_per_unit() prefers the file and says which it used.
#   C2  per Box-Muller pair: state add, 2 hashes (8 + 6 instructions, one v_mul_lo each), cvt, guard max, log, fma, sqrt,
#       alignbit, cos, sin, 2 mul = 24 instructions -> 18 plain + 2 x 4 + 4 x 2 = 34; per sample x, x^2, x^3, x^4 accumulate in
#       5 (add, fma, mul, fma, fma) -> 17 + 5                                                                  = 22.0
#   C3  C2's sampler + affine fma + cell lookup of p (fma, cvt, and, ds_read, fma) + 1/q from the deviate (mul, mul, exp,
#       mul, mul) + weighted accumulation (3 mul + 4 fma)                                                      = 35.0
#   C4  per MH step: half a pair's proposal (hash 8 + 6, Box-Muller 10, affine 2) / 2, accept hash 8, cell lookup 5,
#       w = fma(z/2, z, lp), log alpha, accept test (cvt, log, fma, cmp), 2 state selects, x^2 + 2 accumulates  = 44.5
#   C5  per sample: hash 8, bucket-direct read + flag compare + line fma 5, ring exchange 4, 32 powers by quads at 1.25 per
#       power = 40, quad set-up 9 / 4, resolve of flagged draws 0.17 / 64 per sample                           = 63.1
OPS_PER_UNIT = {"c1": 20.0, "c2": 22.0, "c3": 35.0, "c4": 44.5, "c5": 63.1}
NOMINAL_CLOCK_GHZ = 2.4


def ops_per_unit(name: str, module_key=None):
    """(lane-op equivalents per unit, where the figure comes from). Prefers profiles/rNN_issue_model.json (latest round): the
    entry of this very code object when module_key is given and known there, else the config's reference-stream entry."""
    import json
    from pathlib import Path

    hits = sorted((Path(__file__).resolve().parent.parent / "profiles").glob("r[0-9][0-9]_issue_model.json"))
    if hits:
        try:
            modules = json.loads(hits[-1].read_text()).get("modules", {})
        except ValueError:
            modules = {}
        if module_key in modules:
            return float(modules[module_key]["survey_weighted_ops_per_unit"]), f"profiles/{hits[-1].name}: this code object ({module_key})"
        if module_key is None:
            for entry in modules.values():
                if entry.get("config") == name and entry.get("rng") == "pcg_ref":
                    return float(entry["survey_weighted_ops_per_unit"]), f"profiles/{hits[-1].name}: config {name}, reference stream"
    return OPS_PER_UNIT[name], "tools/baseline_configs.py OPS_PER_UNIT (hand count of the round-3 kernel; this code object is not in profiles/*_issue_model.json)"


VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9      # 7.86e13: CUs x SIMDs x lanes x clock (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0


def moment_functions(k: int = 4):
    """[x, x**2, ..., x**k] as separate lambdas (each on its own line: source recovery on Python 3.10)."""
    if k == 4:
        f1 = lambda x: x
        f2 = lambda x: x**2
        f3 = lambda x: x**3
        f4 = lambda x: x**4
        return [f1, f2, f3, f4]
    if k == 2:
        f1 = lambda x: x
        f2 = lambda x: x**2
        return [f1, f2]
    return [lambda x, k=j: x**k for j in range(1, k + 1)]      # bound defaults -> constants -> shared multiply chain


def bimodal(x):
    return 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2))


def beta25_moment(k: int) -> float:
    m = 1.0
    for j in range(k):
        m *= (2 + j) / (7 + j)
    return m


def table_moments(xs, ps, qpdf, kmax):
    """mu_k = integral x^k p~(x) dx for the piecewise-linear interpolant p~ of the (unnormalised) table, and the
    importance-sampling variance integral (x^k p~/q)^2 q dx - mu_k^2, by 4-point Gauss-Legendre on 8 sub-cells of
    every table cell."""
    gx, gw = np.polynomial.legendre.leggauss(4)
    sub = 8
    edges = np.concatenate([np.linspace(xs[i], xs[i + 1], sub + 1)[:-1] for i in range(len(xs) - 1)] + [[xs[-1]]])
    a, b = edges[:-1], edges[1:]
    mid, half = (a + b) / 2, (b - a) / 2
    pts = mid[:, None] + half[:, None] * gx[None, :]
    w = half[:, None] * gw[None, :]
    dens = np.interp(pts, xs, ps)
    q = qpdf(pts)
    mus, variances = [], []
    for k in range(1, kmax + 1):
        mu = float((pts**k * dens * w).sum())
        second = float(((pts**k * dens) ** 2 / q * w).sum())
        mus.append(mu)
        variances.append(second - mu * mu)
    return np.array(mus), np.array(variances)


class Workload:
    """One BASELINE config: how to prepare it on an integrator, its per-step size and its truth.

    prepare(integrator) -> prepared object; launch(prepared, scale, seed, out, **kw) enqueues one step whose size is
    `scale` x the config's nominal size; units(n_eff) = what one step processed in the metric's unit;
    check(sums / n_eff, n_eff) -> (abs_err, three_sigma) per function."""

    def __init__(self, name, title, unit, k, rows, nominal, prepare, launch, units, band, blocking, n_steps=0, n_burnin=0):
        self.name, self.title, self.unit, self.k, self.rows, self.nominal = name, title, unit, k, rows, nominal
        self.prepare, self.launch, self.units, self.band, self.blocking = prepare, launch, units, band, blocking
        self.n_steps, self.n_burnin = n_steps, n_burnin          # MCMC configs: steps per chain


def get(name: str, Distribution) -> Workload:
    """Build workload `name` ("c1".."c5") against the given Distribution class (the product's)."""
    name = name.lower()
    if name in ("c1", "c2"):
        k = 2 if name == "c1" else 4
        nominal = 1_000_000 if name == "c1" else 1_000_000_000
        truth = np.array([0.0, 1.0, 0.0, 3.0])[:k]
        var = np.array([1.0, 2.0, 15.0, 96.0])[:k]
        dist = Distribution.normal(0.0, 1.0)
        fns = moment_functions(k)
        return Workload(
            name, f"integrate([x..x**{k}], Normal(0,1)), n_samples={nominal:.0e} (BASELINE configs[{0 if name == 'c1' else 1}])",
            "samples/s", k, k, nominal,
            prepare=lambda mc: mc.prepare_integrate(fns, dist),
            launch=lambda pr, n, seed, out, **kw: pr.launch(n, seed, out, **kw),
            units=lambda n_eff: n_eff,
            band=lambda n_eff: (truth, 3.0 * np.sqrt(var / n_eff)),
            blocking=lambda mc, n, seed: mc.integrate(fns, dist, n_samples=n, seed=seed))
    if name == "c3":
        xs = np.linspace(0, 10, 512)
        target = Distribution.from_pdf_table(xs, np.exp(-xs))
        proposal = Distribution.normal(2.0, 3.0)
        fns = moment_functions(4)
        qpdf = lambda x: np.exp(-0.5 * ((x - 2.0) / 3.0) ** 2) / (3.0 * np.sqrt(2 * np.pi))
        mu, var = table_moments(target._x_table.astype(np.float64), target._pdf_table.astype(np.float64), qpdf, 4)
        return Workload(
            name, "integrate_importance_sampling(K=4 moments, target from_pdf_table(512-pt exp(-x) on [0,10]), proposal "
                  "Normal(2,3)), n_samples=1e9 (BASELINE configs[2])",
            "samples/s", 4, 4, 1_000_000_000,
            prepare=lambda mc: mc.prepare_importance_sampling(fns, target, proposal),
            launch=lambda pr, n, seed, out, **kw: pr.launch(n, seed, out, **kw),
            units=lambda n_eff: n_eff,
            band=lambda n_eff: (mu, 3.0 * np.sqrt(var / n_eff)),
            blocking=lambda mc, n, seed: mc.integrate_importance_sampling(fns, target, proposal, n_samples=n, seed=seed))
    if name == "c4":
        target = Distribution.from_pdf(bimodal, support=(-10, 10))
        proposal = Distribution.normal(0.0, 2.0)
        fns = moment_functions(2)
        n_steps, n_burnin, nominal = 10_000, 1_000, 1_048_576

        def band(n_eff, accept=0.6616):
            # independence sampler: integrated autocorrelation time ~ (2 - a) / a at acceptance a; bimodal +-2 unit
            # normals: E x^2 = 5, E x^4 = 43 -> Var(x) = 5, Var(x^2) = 18
            tau = (2.0 - accept) / accept
            return np.array([0.0, 5.0]), 3.0 * np.sqrt(np.array([5.0, 18.0]) * tau / n_eff)

        return Workload(
            name, "integrate_mcmc(K=2, bimodal custom-PDF target on (-10,10), Normal(0,2) independent proposals, "
                  "n_chains=1048576, n_steps=10000, n_burnin=1000) (BASELINE configs[3])",
            "MH steps/s", 2, 3, nominal,
            prepare=lambda mc: mc.prepare_mcmc(fns, target, proposal),
            launch=lambda pr, n, seed, out, **kw: pr.launch(n_steps, n, n_burnin, seed, out, **kw),
            units=lambda n_eff: (n_eff // n_steps) * (n_steps + n_burnin),
            band=band,
            blocking=lambda mc, n, seed: mc.integrate_mcmc(fns, target, proposal, n_steps=n_steps, n_chains=n,
                                                           n_burnin=n_burnin, seed=seed),
            n_steps=n_steps, n_burnin=n_burnin)
    if name == "c5":
        dist = Distribution.beta(2.0, 5.0)
        fns = moment_functions(32)
        truth = np.array([beta25_moment(k) for k in range(1, 33)])
        var = np.array([beta25_moment(2 * k) - beta25_moment(k) ** 2 for k in range(1, 33)])
        return Workload(
            name, "integrate([x**k, k=1..32], Beta(2,5) via its 2048-point CDF table), n_samples=1e10 (BASELINE configs[4])",
            "samples/s", 32, 32, 10_000_000_000,
            prepare=lambda mc: mc.prepare_integrate(fns, dist),
            launch=lambda pr, n, seed, out, **kw: pr.launch(n, seed, out, **kw),
            units=lambda n_eff: n_eff,
            # the 2048-point CDF table has a discretisation bias of ~1e-4 relative: part of the band
            band=lambda n_eff: (truth, 3.0 * np.sqrt(var / n_eff) + truth * 2e-4),
            blocking=lambda mc, n, seed: mc.integrate(fns, dist, n_samples=n, seed=seed))
    raise ValueError(f"unknown config {name!r} (expected c1..c5)")
