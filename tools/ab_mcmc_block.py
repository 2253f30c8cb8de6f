#!/usr/bin/env python3
"""Kernel time of C4's MH call for a rank's share of the chains at several workgroup sizes (MCX_BLOCK): the data behind
the chain-count dependent block choice of integrate_mcmc. Run on the GPU box: python tools/ab_mcmc_block.py"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wgpu-monte-carlo_amd"), str(ROOT / "tools")]


def child(chains: int) -> None:
    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    mc = MonteCarloIntegrator()
    wl = bc.get("c4", Distribution)
    best, r = None, None
    for _ in range(3):
        r = wl.blocking(mc, chains, 42)
        best = r.meta["kernel_ms"] if best is None else min(best, r.meta["kernel_ms"])
    print(json.dumps(dict(chains=chains, kernel_ms=round(best, 3), block=r.meta["block"], n_blocks=r.meta["n_blocks"],
                          steps_per_s=float("%.4g" % (chains * 11000 / (best * 1e-3))))))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for chains in (1_048_576, 524_288, 262_144, 131_072, 65_536):
            for block in (256, 512, 1024):
                env = dict(os.environ, MCX_BLOCK=str(block))
                out = subprocess.run([sys.executable, __file__, str(chains)], env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-300:]
                print(f"block={block:5d} -> {line}", flush=True)
