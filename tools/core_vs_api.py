#!/usr/bin/env python3
"""BASELINE C2 (K = 4 moments on N(0,1)) through the `_core` binding -- the reference's transpiler text, `pow(x, 2.0)` .. `pow(x, 4.0)`,
translated by libmcx (mcx_wgsl_translate) -- against this package's own API, blocking calls. (The "before" rows of
profiles/r03_core_binding_c2.txt, with every pow(x, k.0) a powf call, were measured at commit 0d76349.)
    python tools/core_vs_api.py
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator, _core, transpile_function  # noqa: E402

fns = [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4]
texts = [transpile_function(f) for f in fns]
for math in ("precise", "default"):
    core = _core.MonteCarloIntegrator(math=math)
    core.integrate(texts, "normal", {"mean": 0.0, "std": 1.0}, 1000, 42)
    for n in (10**7, 10**9):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            core.integrate(texts, "normal", {"mean": 0.0, "std": 1.0}, n, 42)
            ts.append(time.perf_counter() - t0)
        best = min(ts)
        print(f"_core math={math:8s} n={n:.0e}: {best * 1e3:8.3f} ms  {core.integrate(texts, 'normal', {'mean': 0.0, 'std': 1.0}, n, 42)}", flush=True)
mc = MonteCarloIntegrator()
for n in (10**7, 10**9):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = mc.integrate(fns, Distribution.normal(0.0, 1.0), n_samples=n, seed=42)
        ts.append(time.perf_counter() - t0)
    print(f"api   math=default  n={n:.0e}: {min(ts) * 1e3:8.3f} ms  {r.values}", flush=True)
