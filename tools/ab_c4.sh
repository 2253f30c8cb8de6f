#!/bin/bash
# A/B of K3 / K2 lookup variants on the GPU box: tools/ab_c4.sh  (run through gpurun from the repo root)
set -u
mkdir -p gpurun_out
IFS="|" read -r -a VARIANTS <<< "${VARIANTS:-|}"
for defs in "${VARIANTS[@]}"; do
  echo "== MCX_EXTRA_DEFINES='$defs'"
  MCX_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/run_configs.py --only ${ONLY:-C3,C4,C4RW} --repeat ${REPEAT:-6} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'kernel_ms' in d: print('   %-40s kernel %.3f ms  worst err/3sigma %.2f' % (d['config'][:40], d['kernel_ms'], d['worst_err_over_3sigma']))
"
done
