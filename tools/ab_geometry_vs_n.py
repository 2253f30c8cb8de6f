#!/usr/bin/env python3
"""Physical threads per launch against the call size, for kernels that stage tables (each workgroup copies them into LDS
before it samples: many short-lived workgroups pay that more often). Beta(2,5) K = 4 / K = 32 (72 KiB staged, 1024-thread
workgroups) and importance sampling on a 512-point table (17 KiB, 512 threads)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402

mc = MonteCarloIntegrator()
beta = Distribution.beta(2.0, 5.0)
xs = np.linspace(0, 10, 512)
target = Distribution.from_pdf_table(xs, np.exp(-xs))
cases = [("beta_k4", lambda n: mc.integrate([lambda x, p=p: x**p for p in range(1, 5)], beta, n_samples=n)),
         ("beta_k32", lambda n: mc.integrate([lambda x, p=p: x**p for p in range(1, 33)], beta, n_samples=n)),
         ("is_k4", lambda n: mc.integrate_importance_sampling([lambda x, p=p: x**p for p in range(1, 5)], target,
                                                              Distribution.normal(2.0, 3.0), n_samples=n))]
for name, call in cases:
    for n in (10**6, 10**7, 10**8, 3 * 10**8, 10**9, 3 * 10**9):
        row = dict(case=name, n=n)
        for t in (0, 1 << 20, 1 << 21, 1 << 22):
            mc._engine.set_target_threads(t)
            for _ in range(12):
                r = call(n)
            best = min(call(n).meta["kernel_ms"] for _ in range(8))
            row[f"t{t}"] = round(best, 4)
            row[f"blocks{t}"] = r.meta["n_blocks"]
        print(json.dumps(row), flush=True)
mc._engine.set_target_threads(0)
