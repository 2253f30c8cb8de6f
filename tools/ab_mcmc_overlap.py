#!/usr/bin/env python3
"""Would C4 gain from running as two half-size chain sets on two streams, each cut into S time segments? Emulated with
independent launches of the same total work: 2 streams x S launches of (524 288 chains x 11 000 / S steps) against one launch
of 1 048 576 chains x 11 000 steps."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wgpu-monte-carlo_amd"), str(ROOT / "tools")]
import baseline_configs as bc  # noqa: E402
from wgpu_montecarlo import Distribution, MonteCarloIntegrator  # noqa: E402

mc = MonteCarloIntegrator()
wl = bc.get("c4", Distribution)
prep = wl.prepare(mc)
out = torch.zeros((64, prep.rows), dtype=torch.float64, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(segments, halves, reps=6):
    chains = 1_048_576 // halves
    steps = 10_000 // segments
    burn = 1_000 // segments
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        row = 0
        for s in range(segments):
            for h in range(halves):
                with torch.cuda.stream(streams[h % 2] if halves > 1 else streams[0]):
                    prep.launch(steps, chains, burn, 42 + row, out[row], reduce=False)
                row += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


for _ in range(3):
    run(1, 1, reps=2)
for segments, halves in ((1, 1), (1, 2), (2, 2), (4, 2), (8, 2), (4, 1)):
    print(json.dumps(dict(segments=segments, halves=halves, ms=round(run(segments, halves), 3))), flush=True)
