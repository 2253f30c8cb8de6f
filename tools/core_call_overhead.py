#!/usr/bin/env python3
"""Per-call cost of the `_core` binding (mcx_core: tables found again by content, modules by payload) beside this package's API on
small calls, where the GPU work is a few microseconds: plain integrands, an importance-sampling call with a 512-point table, an MCMC
call with two 2048-point tables.   python tools/core_call_overhead.py
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import core_payload_bench as cpb  # noqa: E402
from wgpu_montecarlo import Distribution, MonteCarloIntegrator, _core  # noqa: E402


def med(call, reps=40):
    call()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    calls = cpb.golden_calls()
    core = _core.MonteCarloIntegrator()
    mc = MonteCarloIntegrator()
    fns = [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4]
    a = list(calls[1][1]); a[3] = 100_000
    print(json.dumps(dict(call="integrate K=4, n=1e5", core_ms=med(lambda: core.integrate(*a)),
                          api_ms=med(lambda: mc.integrate(fns, Distribution.normal(0.0, 1.0), n_samples=100_000)))))
    b = list(calls[2][1]); b[3] = 100_000
    xs = np.linspace(0, 10, 512)
    target = Distribution.from_pdf_table(xs, np.exp(-xs))
    print(json.dumps(dict(call="integrate_is_tables K=4, 512-point table, n=1e5", core_ms=med(lambda: core.integrate_is_tables(*b)),
                          api_ms=med(lambda: mc.integrate_importance_sampling(fns, target, Distribution.normal(2.0, 3.0), n_samples=100_000)))))
    c = list(calls[3][1]); c[5], c[6], c[7] = 100, 4096, 10
    import math

    bimodal = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    print(json.dumps(dict(call="integrate_mcmc K=2, two 2048-point tables, 4096 chains x 110 steps", core_ms=med(lambda: core.integrate_mcmc(*c)),
                          api_ms=med(lambda: mc.integrate_mcmc(fns[:2], bimodal, Distribution.normal(0.0, 2.0), n_steps=100, n_chains=4096, n_burnin=10)))))


if __name__ == "__main__":
    main()
