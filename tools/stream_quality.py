#!/usr/bin/env python3
"""Statistical quality of the two counter streams: z-scores of the four N(0,1) moments over many seeds.

The reference's stream hashes a 32-bit linear counter (seed + idx*7199369 + iter*15485863 mod 2^32), so one call at
n samples uses n of the 2^32 possible hash inputs and different seeds re-use the same inputs shifted; Philox4x32-10
has a 128-bit counter. For an ideal stream z = (estimate - truth)/sigma is N(0,1): mean z^2 = 1, P(|z| > 3) = 0.27 %.

    python tools/stream_quality.py [--seeds 400] [--n 1e8]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=400)
    ap.add_argument("--n", type=float, default=1e8)
    args = ap.parse_args()
    fns = [lambda x, k=k: x**k for k in range(1, 5)]
    truth = np.array([0.0, 1.0, 0.0, 3.0])
    var = np.array([1.0, 2.0, 15.0, 96.0])
    for rng in ("pcg_ref", "philox"):
        mc = MonteCarloIntegrator(rng=rng)
        prepared = mc.prepare_integrate(fns, Distribution.normal(0.0, 1.0))
        zs = []
        for seed in range(1000, 1000 + args.seeds):
            r = prepared.run(int(args.n), seed)
            zs.append((r.values - truth) / np.sqrt(var / r.meta["n_eff"]))
        z = np.array(zs)
        print(json.dumps(dict(rng=rng, n=int(args.n), seeds=args.seeds, mean_z2=[round(float(v), 3) for v in (z**2).mean(axis=0)],
                              frac_beyond_3sigma=float((np.abs(z) > 3).mean()), max_abs_z=float(np.abs(z).max()),
                              mean_z=[round(float(v), 3) for v in z.mean(axis=0)])), flush=True)


if __name__ == "__main__":
    main()
