#!/usr/bin/env python3
"""Soak: random API calls for a fixed wall time; watches host RSS, free device memory and result sanity.

    python tools/soak.py --seconds 180
"""
import argparse
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402
from wgpu_montecarlo import runtime as rt  # noqa: E402


def rss_mb():
    with open(f"/proc/{os.getpid()}/status") as fh:
        for line in fh:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return 0.0


def free_device_mb():
    import torch

    free, _ = torch.cuda.mem_get_info(0)
    return free / 2**20


def make_fn(c):
    # a new constant gives a new code object key -> a new module; the body stays in the emitter's subset
    return [lambda x, c=c: x * c, lambda x, c=c: x * x + c, lambda x, c=c: math.sin(x * c), lambda x, c=c: math.exp(-x * x * c),
            lambda x, c=c: np.abs(x) ** (c + 0.5) + math.cos(3.0 * x)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    t0 = time.time()
    calls = 0
    first = None
    while time.time() - t0 < args.seconds:
        kind = int(rng.integers(0, 13))
        mc = MonteCarloIntegrator(target_threads=int(rng.choice([256, 4096, 65536])), rng=str(rng.choice(["pcg_ref", "philox"])),
                                  std_error=bool(rng.integers(0, 2)))
        fns = make_fn(float(rng.integers(1, 40)) / 8.0)[: int(rng.integers(1, 6))]
        n = int(rng.choice([1000, 100_000, 3_000_000, 50_000_000]))
        if kind == 0:
            r = mc.integrate(fns, Distribution.normal(float(rng.normal()), 0.5 + float(rng.random())), n_samples=n, seed=calls)
        elif kind == 1:
            r = mc.integrate(fns, Distribution.beta(1.5 + float(rng.random()), 2.0 + float(rng.random()), table_size=int(rng.choice([512, 1001, 2048]))),
                             n_samples=n, seed=calls)
        elif kind == 2:
            r = mc.integrate(fns, Distribution.uniform(-1.0, 2.0 + float(rng.random())), n_samples=n, seed=calls)
        elif kind == 3:
            xs = np.linspace(-4, 4, int(rng.choice([300, 512, 1000])))
            r = mc.integrate_importance_sampling(fns, Distribution.from_pdf_table(xs, np.exp(-0.5 * xs * xs)), Distribution.normal(0.0, 1.5),
                                                 n_samples=n, seed=calls)
        elif kind == 4:
            r = mc.integrate_mcmc(fns, Distribution.normal(0.5, 1.0), Distribution.normal(0.0, 2.0), n_steps=int(rng.choice([10, 200])),
                                  n_chains=int(rng.choice([100, 4096])), n_burnin=int(rng.choice([0, 20])), seed=calls)
        elif kind == 5:
            r = mc.integrate_mcmc(fns, Distribution.normal(0.5, 1.0), Distribution.uniform(-1.0, 1.0), n_steps=int(rng.choice([10, 200])),
                                  n_chains=int(rng.choice([100, 4096])), n_burnin=int(rng.choice([0, 20])), seed=calls,
                                  proposal_kind="random_walk")
        elif kind == 6:
            # the bucket-direct sampler against the guided search on a random table: same draws, same cells
            dist = Distribution.beta(1.2 + 2 * float(rng.random()), 1.5 + 3 * float(rng.random()), table_size=int(rng.choice([300, 1000, 2048, 4000])))
            mc = MonteCarloIntegrator(target_threads=int(rng.choice([256, 65536])), std_error=bool(rng.integers(0, 2)))
            r = mc.integrate(fns, dist, n_samples=n, seed=calls)
            os.environ["MCX_NO_DIRECT"] = "1"
            try:
                g = MonteCarloIntegrator(target_threads=mc._target_threads, std_error=mc._std_error).integrate(fns, dist, n_samples=n, seed=calls)
            finally:
                del os.environ["MCX_NO_DIRECT"]
            assert np.allclose(r.values, g.values, rtol=5e-6, atol=5e-6), (r.values, g.values)
        elif kind == 7:
            # one process, three engines on this GPU (host-sum path of MonteCarloIntegrator(devices=[...])), split launches
            multi = MonteCarloIntegrator(devices=[0, 0, 0], rng=str(rng.choice(["pcg_ref", "philox"])))
            rt.set_max_launch_units(int(rng.choice([0, 1_000_000, 7_000_000])))
            try:
                r = multi.integrate(fns, Distribution.normal(0.0, 1.0), n_samples=n, seed=calls)
                one = MonteCarloIntegrator(rng="pcg_ref" if multi._rng == 0 else "philox").integrate(fns, Distribution.normal(0.0, 1.0), n_samples=n, seed=calls)
            finally:
                rt.set_max_launch_units(0)
            assert np.allclose(r.values, one.values, rtol=1e-8, atol=1e-8), (r.values, one.values)
            for eng in multi._engines[1:]:
                eng.close()
        elif kind == 9:
            # an MCMC call cut into time segments on two streams against the same call in one launch: the same chains
            plain = MonteCarloIntegrator()
            tgt = Distribution.from_pdf(lambda x: math.exp(-0.5 * (x - 0.7) ** 2), support=(-6.0, 6.0), table_size=int(rng.choice([700, 2048])))
            kw = dict(n_steps=int(rng.choice([8, 61, 400, 1501])), n_chains=int(rng.choice([256, 1000, 8192, 70_000])),
                      n_burnin=int(rng.choice([0, 1, 20, 333])), seed=calls)
            one = plain.integrate_mcmc(fns[:2], tgt, Distribution.normal(0.2, 1.7), **kw)
            plain._engine.set_mcmc_segments(int(rng.choice([2, 3, 8])))
            try:
                r = plain.integrate_mcmc(fns[:2], tgt, Distribution.normal(0.2, 1.7), **kw)
            finally:
                plain._engine.set_mcmc_segments(rt.SEGMENTS_AUTO)
            assert r.meta["accept_rate"] == one.meta["accept_rate"], (kw, r.meta["accept_rate"], one.meta["accept_rate"])
            assert np.allclose(r.values, one.values, rtol=5e-6, atol=5e-6), (kw, r.values, one.values)
        elif kind == 11:
            # the plan cache: a repeat call (new lambda objects, same code and captures) must give the first call's result bit
            # for bit; a changed capture must not be answered from the cache; a full-size default-segmented MCMC call
            # against its one-launch form
            scale = float(rng.integers(1, 40)) / 8.0
            family = lambda c: [lambda x: x * c, lambda x: x * x + c]         # closure captures, one code object per slot
            dist = Distribution.normal(0.25, 1.5)
            a = mc.integrate(family(scale), dist, n_samples=n, seed=calls)
            b = mc.integrate(family(scale), dist, n_samples=n, seed=calls)
            r = mc.integrate(family(scale + 1.0), dist, n_samples=n, seed=calls)
            assert np.array_equal(a.values, b.values), (a.values, b.values)
            assert abs(r.values[0] * scale - a.values[0] * (scale + 1.0)) < 1e-5 * (1.0 + abs(a.values[0])) * (scale + 1.0), (a.values, r.values)
            assert abs((r.values[1] - a.values[1]) - 1.0) < 1e-5 * (1.0 + abs(a.values[1])), (a.values, r.values)
        elif kind == 12:
            # blocking calls of every size class through the pinned-result / ticket-polling path against the copy + stream-wait path
            big = int(rng.choice([1000, 3_000_000, 400_000_000]))
            r = mc.integrate(fns, Distribution.normal(0.1, 1.1), n_samples=big, seed=calls)
            os.environ["MCX_POLL_US"] = "0"
            os.environ["MCX_NO_ZERO_COPY"] = "1"
            try:
                plain = rt.Engine(0)                        # a fresh engine reads the two knobs
                other = MonteCarloIntegrator(target_threads=mc._target_threads, rng="pcg_ref" if mc._rng == 0 else "philox",
                                             std_error=mc._std_error)
                other._engine = plain
                other._engines = [plain]
                g = other.integrate(fns, Distribution.normal(0.1, 1.1), n_samples=big, seed=calls)
                plain.close()
            finally:
                del os.environ["MCX_POLL_US"], os.environ["MCX_NO_ZERO_COPY"]
            assert np.array_equal(r.values, g.values), (r.values, g.values)
        elif kind == 10:
            # importance sampling / MH with padded, unclamped cell tables against the clamped lookup
            xs = np.linspace(-3.0, 4.0, int(rng.choice([200, 512, 1500])))
            tab = Distribution.from_pdf_table(xs, np.exp(-0.5 * (xs - 0.5) ** 2))
            prop = [Distribution.normal(0.3, 2.0), Distribution.uniform(-5.0, 6.0), Distribution.exponential(0.5)][int(rng.integers(0, 3))]
            r = mc.integrate_importance_sampling(fns, tab, prop, n_samples=n, seed=calls)
            os.environ["MCX_NO_NOCLAMP"] = "1"
            try:
                g = MonteCarloIntegrator(target_threads=mc._target_threads, rng="pcg_ref" if mc._rng == 0 else "philox",
                                         std_error=mc._std_error).integrate_importance_sampling(fns, tab, prop, n_samples=n, seed=calls)
            finally:
                del os.environ["MCX_NO_NOCLAMP"]
            assert np.allclose(r.values, g.values, rtol=5e-6, atol=5e-6), (r.values, g.values)
        else:
            lap = Distribution.from_pdf(lambda x: math.exp(-abs(x)) / 2, support=(-12.0, 12.0), table_size=int(rng.choice([512, 2048])))
            r = mc.integrate_importance_sampling(fns, Distribution.normal(0.0, 1.0), lap, n_samples=n, seed=calls)
        assert np.all(np.isfinite(r.values)), (kind, r.values)
        calls += 1
        if calls % 100 == 0:
            line = dict(calls=calls, seconds=round(time.time() - t0, 1), rss_mb=round(rss_mb(), 1), free_device_mb=round(free_device_mb(), 1))
            first = first or line
            print(line, flush=True)
    print("soak done:", calls, "calls; rss growth since call 100:", round(rss_mb() - first["rss_mb"], 1), "MB;",
          "device memory change:", round(first["free_device_mb"] - free_device_mb(), 1), "MB")


if __name__ == "__main__":
    main()
