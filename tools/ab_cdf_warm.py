#!/usr/bin/env python3
"""Moment families on a warm device, n = 2e9: Beta(2,5) through its CDF table (K = 4, 8, 16, 32) and N(0,1) (K = 8, 16, 32).
The kernel's variants on equal terms:
    python tools/ab_cdf_warm.py            # current default
    MCX_AB_RNG=philox python tools/ab_cdf_warm.py
    MCX_NO_DIRECT=1 python tools/ab_cdf_warm.py                      # guided search, ds_read lookups
    MCX_NO_DIRECT=1 MCX_EXTRA_DEFINES="MCX_TBL=" python tools/ab_cdf_warm.py      # round 1: guided search, flat_load lookups
    MCX_EXTRA_DEFINES="MCX_MOMENT_QUAD=0" python tools/ab_cdf_warm.py              # Newton pairs instead of quads
    MCX_EXTRA_DEFINES="MCX_DIRECT_SWAP=0" python tools/ab_cdf_warm.py              # append-and-resolve queue everywhere
"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402

rng = os.environ.get("MCX_AB_RNG", "pcg_ref")
mc = MonteCarloIntegrator(rng=rng)
cases = [("beta25", Distribution.beta(2.0, 5.0), k) for k in (4, 8, 16, 32)] + \
        [("normal01", Distribution.normal(0.0, 1.0), k) for k in (8, 16, 32)]
only = os.environ.get("MCX_AB_ONLY")
for name, dist, k in cases:
    if only and only not in f"{name}-{k}":
        continue
    fns = [lambda x, p=p: x**p for p in range(1, k + 1)]
    for _ in range(40):
        r = mc.integrate(fns, dist, n_samples=2_000_000_000)
    best = min(mc.integrate(fns, dist, n_samples=2_000_000_000).meta["kernel_ms"] for _ in range(6))
    print(json.dumps(dict(dist=name, k=k, rng=rng, kernel_ms=round(best, 3),
                          samples_per_s=float("%.4g" % (r.meta["n_eff"] / (best * 1e-3))), lds=r.meta["lds_bytes"],
                          no_direct=os.environ.get("MCX_NO_DIRECT"), extra=os.environ.get("MCX_EXTRA_DEFINES"))), flush=True)
