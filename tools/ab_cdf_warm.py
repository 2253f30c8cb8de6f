#!/usr/bin/env python3
"""Beta(2,5) K = 4 (and K = 16 / 32), n = 2e9, with a warm device: the CDF sampler's variants on equal terms.
    python tools/ab_cdf_warm.py            # current default
    MCX_NO_DIRECT=1 python tools/ab_cdf_warm.py                      # guided search, ds_read lookups
    MCX_NO_DIRECT=1 MCX_EXTRA_DEFINES="MCX_TBL=" python tools/ab_cdf_warm.py      # round 1: guided search, flat_load lookups
"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402

mc = MonteCarloIntegrator()
for k in (4, 16, 32):
    fns = [lambda x, p=p: x**p for p in range(1, k + 1)]
    dist = Distribution.beta(2.0, 5.0)
    for _ in range(40):
        r = mc.integrate(fns, dist, n_samples=2_000_000_000)
    best = min(mc.integrate(fns, dist, n_samples=2_000_000_000).meta["kernel_ms"] for _ in range(6))
    print(json.dumps(dict(k=k, kernel_ms=round(best, 3), samples_per_s=float("%.4g" % (r.meta["n_eff"] / (best * 1e-3))), lds=r.meta["lds_bytes"],
                          no_direct=os.environ.get("MCX_NO_DIRECT"), extra=os.environ.get("MCX_EXTRA_DEFINES"))), flush=True)
