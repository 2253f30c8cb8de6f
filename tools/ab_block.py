#!/usr/bin/env python3
"""Kernel time of integrate(K powers, Beta(2,5)) for several K and workgroup sizes (MCX_BLOCK) -- the data behind
resolve_block() in csrc/mcx_runtime.cpp. Run on the GPU box: python tools/ab_block.py"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))


def child(k: int, n: int) -> None:
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    mc = MonteCarloIntegrator()
    fns = [lambda x, p=p: x**p for p in range(1, k + 1)]
    best, r = None, None
    for _ in range(4):
        r = mc.integrate(fns, Distribution.beta(2.0, 5.0), n_samples=n)
        best = r.meta["kernel_ms"] if best is None else min(best, r.meta["kernel_ms"])
    print(json.dumps(dict(k=k, kernel_ms=round(best, 3), block=r.meta["block"], lds=r.meta["lds_bytes"])))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), int(float(sys.argv[2])))
    else:
        for k in (4, 8, 12, 16, 24, 32, 48):
            for block in (256, 512, 1024):
                env = dict(os.environ, MCX_BLOCK=str(block))
                out = subprocess.run([sys.executable, __file__, str(k), "2e9"], env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-300:]
                print(f"K={k:3d} block={block:5d} -> {line}", flush=True)
