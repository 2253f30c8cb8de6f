"""Kernel time and values of the reference benchmark integrand (examples/benchmark.py: x / (exp(sin x) + cos(exp x))), sin x and cos 5x on
N(0,1) in the three math modes, n = 1e7 and 1e9. Run on the GPU box: python tools/trig_probe.py -> profiles/r03_trig_pow_accuracy.txt."""
import json
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "wgpu-monte-carlo_amd"))
import numpy as np
from wgpu_montecarlo import Distribution, MonteCarloIntegrator
f_bench = lambda x: x / (math.exp(math.sin(x)) + math.cos(math.exp(x)))
f_sin = lambda x: math.sin(x)
f_cos5 = lambda x: math.cos(5.0 * x)
for m in ("precise", "default", "fast"):
    mc = MonteCarloIntegrator(math=m)
    for n in (10**7, 10**9):
        r = mc.integrate([f_bench, f_sin, f_cos5], Distribution.normal(0.0, 1.0), n_samples=n, seed=42)
        ks = [mc.integrate([f_bench, f_sin, f_cos5], Distribution.normal(0.0, 1.0), n_samples=n, seed=42).meta["kernel_ms"] for _ in range(4)]
        print(json.dumps(dict(math=m, n=n, values=[float(v) for v in r.values], kernel_ms=round(min(ks),4))))
