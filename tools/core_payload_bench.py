#!/usr/bin/env python3
"""The payloads the reference's own Python half hands to `_core` for the BASELINE configs (tests/golden/boundary_payloads.*: its
transpiler's WGSL, its importance-sampling wrapper text, parameter dicts, float32 tables), replayed at FULL size through this
package's `_core` binding and timed (blocking calls, best of 3), beside the same config through this package's API.
    python tools/core_payload_bench.py [--math default] [--configs c2,c3,c4,c5]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import baseline_configs as bc  # noqa: E402
from wgpu_montecarlo import Distribution, MonteCarloIntegrator, _core  # noqa: E402


def golden_calls():
    gdir = ROOT / "tests" / "golden"
    meta = json.loads((gdir / "boundary_payloads.json").read_text())
    arrays = np.load(gdir / "boundary_payloads.npz")
    calls = []
    for entry in meta:
        args = []
        for a in entry["args"]:
            if isinstance(a, dict) and "array" in a:
                args.append(arrays[a["array"]])
            elif isinstance(a, dict) and "wgsl" in a:
                args.append(list(a["wgsl"]))
            else:
                args.append(a)
        calls.append((entry["method"], args))
    return calls


def best(call, reps=3):
    call()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = call()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--math", default="default")
    ap.add_argument("--configs", default="c2,c3,c4,c5")
    args = ap.parse_args()
    calls = golden_calls()
    core = _core.MonteCarloIntegrator(math=args.math)
    mc = MonteCarloIntegrator()
    index = {"c2": 1, "c3": 2, "c4": 3, "c5": 4}
    for name in args.configs.split(","):
        method, a = calls[index[name]]
        a = list(a)
        ms, got = best(lambda: getattr(core, method)(*a))
        wl = bc.get(name, Distribution)
        api_ms, res = best(lambda: wl.blocking(mc, wl.nominal, 42))
        if name == "c5":
            # the recorded call is the sampler's (K = 2: x, pow(x, 2.0)); BASELINE's K = 32 in the same recorded format
            a32 = list(a)
            a32[0] = [a[0][0]] + [a[0][1].replace("pow(x, 2.0)", f"pow(x, {k}.0)") for k in range(2, 33)]
            ms32, got32 = best(lambda: getattr(core, method)(*a32))
            print(json.dumps(dict(config="c5 K=32", method=method, k=32, core_math=args.math, core_ms=round(ms32, 3),
                                  core_values=[float(v) for v in got32[:4]])), flush=True)
        print(json.dumps(dict(config=name, method=method, k=len(a[0]), core_math=args.math, core_ms=round(ms, 3), core_values=[float(v) for v in got[:4]],
                              api_k=wl.k, api_ms=round(api_ms, 3), api_values=[float(v) for v in res.values[:4]])), flush=True)


if __name__ == "__main__":
    main()
