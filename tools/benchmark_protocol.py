#!/usr/bin/env python3
"""The reference's own benchmark protocol on MI355X.

examples/benchmark.py:8-68 of the reference times `integrator.integrate([f], Normal(0,1), n)` for
f(x) = x / (exp(sin x) + cos(exp x)) at n in {1e3 .. 1e7} with wall-clock time around the call (one warm-up call
at n = 1000), next to a pure-Python loop and a per-element numpy loop. This script runs the same protocol through
this package (blocking public API, so emission + launch + readback are inside the timing, as in the reference)
and, beside it, the CPU oracle (OpenMP C restatement of the reference kernel) and a vectorised numpy evaluation.

    python tools/benchmark_protocol.py [--max-n 1e9]
"""
import argparse
import json
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
    sys.path.insert(0, str(p))

import oracle  # noqa: E402  (CPU comparison leg only)
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402


def f(x):
    return x / (math.exp(math.sin(x)) + math.cos(math.exp(x)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-n", type=float, default=1e9)
    args = ap.parse_args()
    sizes = [n for n in (1e3, 5e3, 1e4, 5e4, 1e5, 5e5, 1e6, 5e6, 1e7, 1e8, 1e9) if n <= args.max_n]
    mc = MonteCarloIntegrator()
    dist = Distribution.normal(0.0, 1.0)
    t0 = time.perf_counter()
    mc.integrate([f], dist, n_samples=1000)                       # warm-up: includes the one hiprtc compile
    cold_ms = (time.perf_counter() - t0) * 1e3
    rows = []
    for n in sizes:
        n = int(n)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = mc.integrate([f], dist, n_samples=n)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row = dict(n=n, n_eff=res.meta["n_eff"], gpu_call_ms=best * 1e3, gpu_kernel_ms=res.meta["kernel_ms"], value=float(res.values[0]))
        if n <= 1e8:
            t0 = time.perf_counter()
            ref = oracle.integrate([(oracle.FN_BENCH, 0)], oracle.NORMAL, 0.0, 1.0, n_samples=n, seed=42, guard=1)
            row["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
            row["cpu_oracle_threads"] = oracle.num_threads()
            row["abs_diff_vs_oracle"] = abs(ref["sums"][0] / ref["n_eff"] - row["value"])
        if n <= 1e7:
            xs = np.random.default_rng(0).standard_normal(n).astype(np.float32)
            t0 = time.perf_counter()
            float(np.mean(xs / (np.exp(np.sin(xs)) + np.cos(np.exp(xs)))))
            row["numpy_vectorised_1core_ms"] = (time.perf_counter() - t0) * 1e3
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(dict(cold_first_call_ms=cold_ms, note="first call = emission + hiprtc compile (or disk-cache hit) + launch")))


if __name__ == "__main__":
    main()
