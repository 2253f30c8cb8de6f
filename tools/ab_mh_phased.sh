#!/bin/bash
# batched independence sampler: per-step phase test + flush counter (MCX_MH_PHASED=0) against burn-in / sampling loops
for v in 1 0 1 0; do
  MCX_EXTRA_DEFINES="MCX_MH_PHASED=$v" python bench.py --config c4 --no-cpu-baseline --no-cold --no-philox --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 phased=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for v in 1 0; do echo "phased=$v"; MCX_EXTRA_DEFINES="MCX_MH_PHASED=$v" python tools/ab_mcmc_block.py 2>/dev/null | grep -v "^$" | tail -12; done
