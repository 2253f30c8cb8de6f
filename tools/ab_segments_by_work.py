import sys, time, json
sys.path[:0]=['/root/repo/wgpu-monte-carlo_amd','/root/repo/tools']
import numpy as np, torch
import baseline_configs as bc
from wgpu_montecarlo import Distribution, MonteCarloIntegrator
mc = MonteCarloIntegrator()
target = Distribution.from_pdf(bc.bimodal, support=(-10, 10)); proposal = Distribution.normal(0.0, 2.0)
fns = bc.moment_functions(2)
prep = mc.prepare_mcmc(fns, target, proposal)
out = torch.zeros(prep.rows, dtype=torch.float64, device="cuda")
def t(chains, steps, burn, seg, reps=8):
    mc._engine.set_mcmc_segments(seg)
    for _ in range(3): prep.launch(steps, chains, burn, 1, out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(reps): prep.launch(steps, chains, burn, 2+j, out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
# warm
for _ in range(20): prep.launch(10000, 1048576, 1000, 1, out)
torch.cuda.synchronize()
for chains in (1048576, 262144, 131072):
    for steps in (50, 150, 400, 1000, 3000, 11000):
        a=t(chains, steps, 0, 0); b=t(chains, steps, 0, 8); c=t(chains, steps, 0, 4)
        print(json.dumps(dict(chains=chains, steps=steps, work=chains*steps, one_launch_ms=round(a,4), seg4_ms=round(c,4), seg8_ms=round(b,4), gain8=round(1-b/a,3), gain4=round(1-c/a,3))), flush=True)
