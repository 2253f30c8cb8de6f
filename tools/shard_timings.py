#!/usr/bin/env python3
"""What each rank of an N-GPU run launches, timed on ONE GPU: for every BASELINE config and N = 1, 2, 4, 8 the shard
(rank r of N) of the step bench.py would run -- C2 / C3 weak (N x the nominal size, so a shard is the nominal size), C4 /
C5 strong (the nominal size cut N ways) -- as `prepared.launch(..., shard=(r, N))`, main + fold kernel, HIP events around
`reps` back-to-back launches. Not a multi-GPU measurement: it leaves out the all-reduce of K doubles (latency-bound,
overlapped with the next step) and assumes the ranks' GPUs behave like this one. It does show what the sharding itself
costs: the per-shard times a real run cannot beat.

    python tools/shard_timings.py > gpurun_out/r03_shard_timings.txt
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wgpu-monte-carlo_amd"), str(ROOT / "tools")]

import torch  # noqa: E402

import baseline_configs as bc  # noqa: E402
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402

WEAK = {"c2": True, "c3": True, "c4": False, "c5": False}


def time_shard(wl, prepared, n_step, shard, reps):
    out = torch.zeros(wl.rows, dtype=torch.float64, device="cuda")
    for _ in range(3):
        wl.launch(prepared, n_step, 7, out, shard=shard)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(reps):
        wl.launch(prepared, n_step, 100 + j, out, shard=shard)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    mc = MonteCarloIntegrator()
    for name in ("c2", "c3", "c4", "c5"):
        wl = bc.get(name, Distribution)
        prepared = wl.prepare(mc)
        warm = torch.zeros(wl.rows, dtype=torch.float64, device="cuda")
        t_end = time.perf_counter() + 0.15              # leave the idle clock state before the first timing (bench.py: device_prewarm)
        while time.perf_counter() < t_end:
            for _ in range(4):
                wl.launch(prepared, wl.nominal, 7, warm, shard=(0, 1))
            torch.cuda.synchronize()
        base = None
        for world in (1, 2, 4, 8):
            n_step = wl.nominal * (world if WEAK[name] else 1)
            reps = 20 if name in ("c2", "c3") else 6
            ranks = sorted({0, world // 2, world - 1})
            ms = {r: time_shard(wl, prepared, n_step, (r, world), reps) for r in ranks}
            worst = max(ms.values())
            base = base or worst
            launch = mc._engine.last_launch()
            # whole-job throughput if every rank took as long as the slowest shard measured here
            speedup = (world * base / worst) if WEAK[name] else (base / worst)
            print(json.dumps(dict(config=name, scaling="weak" if WEAK[name] else "strong", world=world, step_size=n_step,
                                  shard_ms={str(r): round(v, 4) for r, v in ms.items()}, slowest_shard_ms=round(worst, 4),
                                  expected_speedup_over_1_gpu=round(speedup, 2), block=launch["block"], n_blocks=launch["n_blocks"],
                                  launches=launch["launches"], segments=launch["segments"])), flush=True)


if __name__ == "__main__":
    main()
