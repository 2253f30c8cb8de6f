#!/usr/bin/env python3
"""Where a blocking integrate() call spends its time at n = 1e9 and n = 1e6: C entry point vs Python API."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402
from wgpu_montecarlo import runtime as rt  # noqa: E402
from wgpu_montecarlo.api import functions_to_hip  # noqa: E402

f1 = lambda x: x
f2 = lambda x: x**2
f3 = lambda x: x**3
f4 = lambda x: x**4


def best(fn, n=30):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


def main():
    mc = MonteCarloIntegrator()
    eng = mc._engine
    fns = [f1, f2, f3, f4]
    dist = Distribution.normal(0.0, 1.0)
    mod = eng.module(functions_to_hip(fns), rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, unit_params=True))
    for n in (10**6, 10**8, 10**9):
        c_min, c_med = best(lambda: eng.integrate(mod, n, 42, 0.0, 1.0))
        k = eng.last_kernel_ms()
        a_min, a_med = best(lambda: mc.integrate(fns, dist, n_samples=n, seed=42))
        print(f"n={n:.0e}: kernel {k:.3f} ms | C call min {c_min:.3f} med {c_med:.3f} ms | Python API min {a_min:.3f} med {a_med:.3f} ms")


if __name__ == "__main__":
    main()


def device_path():
    """Kernel time (HIP events) of PreparedIntegrand.launch on torch streams, one launch at a time."""
    import torch

    mc = MonteCarloIntegrator()
    prepared = mc.prepare_integrate([f1, f2, f3, f4], Distribution.normal(0.0, 1.0))
    dev = torch.device("cuda", 0)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    for label, stream in (("null stream", None), ("side stream", torch.cuda.Stream(device=dev))):
        ts = []
        for i in range(20):
            if stream is None:
                prepared.launch(10**9, 42 + i, out)
            else:
                with torch.cuda.stream(stream):
                    prepared.launch(10**9, 42 + i, out)
            ts.append(mc._engine.last_kernel_ms())
        print(f"prepared.launch on the {label}: kernel min {min(ts):.3f} median {sorted(ts)[10]:.3f} max {max(ts):.3f} ms")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "device":
    device_path()
