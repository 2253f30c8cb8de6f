#!/usr/bin/env python3
"""C4's chain shards (what each rank of a 2 / 4 / 8-GPU run of BASELINE configs[3] launches): kernel time of the default
MH loop against MCX_MH_AHEAD = 4 / 8 (that many proposals ahead of as many accept tests in the phased loops) and, for reference, the
workgroup sizes. Run on the GPU box: python tools/ab_mh_ahead.py > gpurun_out/r03_mh_ahead4.txt"""
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
    for _ in range(3):
        wl.blocking(mc, chains, 41)                   # leave the idle clock state
    ts, r = [], None
    for i in range(6):
        r = wl.blocking(mc, chains, 42 + i)
        ts.append(r.meta["kernel_ms"])
    print(json.dumps(dict(chains=chains, kernel_ms=round(min(ts), 3), median=round(sorted(ts)[3], 3), block=r.meta["block"],
                          accept=round(r.meta["accept_rate"], 6), values=[round(float(v), 6) for v in r.values],
                          steps_per_s=float("%.4g" % (chains * 11000 / (min(ts) * 1e-3))))))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for chains in (524_288, 262_144, 131_072, 65_536):
            for label, env_add in (("ahead2 (r02)", {"MCX_EXTRA_DEFINES": "MCX_MH_AHEAD=2"}), ("ahead4", {"MCX_EXTRA_DEFINES": "MCX_MH_AHEAD=4"}),
                                   ("ahead8", {"MCX_EXTRA_DEFINES": "MCX_MH_AHEAD=8"}),
                                   ("ahead8 phased", {"MCX_EXTRA_DEFINES": "MCX_MH_AHEAD=8;MCX_MH_PHASED=1"})):
                env = dict(os.environ, **env_add)
                out = subprocess.run([sys.executable, __file__, str(chains)], env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-300:]
                print(f"{label:18s} -> {line}", flush=True)
