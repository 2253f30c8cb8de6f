#!/bin/bash
# kernel_ms / ms_per_step of the BASELINE configs through bench.py (quick A/B of a kernel change): bash tools/ab_configs.sh [c2,c3,c4,c5] [tag]
CONFIGS=${1:-c2,c3,c4,c5}
TAG=${2:-ab}
mkdir -p gpurun_out
for CFG in ${CONFIGS//,/ }; do
  STEPS=30; [ "$CFG" = c4 ] && STEPS=8; [ "$CFG" = c5 ] && STEPS=8
  timeout -k 10 150 python3 bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-cold --no-philox 2> gpurun_out/${TAG}_$CFG.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG $CFG value %.4g %s  ms_per_step %.4f  kernel_ms %.4f  worst_err/3sigma %.2f' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['kernel_ms'], d['worst_err_over_3sigma']))
" || tail -3 gpurun_out/${TAG}_$CFG.err
done
