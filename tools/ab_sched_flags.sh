#!/bin/bash
# LLVM scheduling strategies for the hiprtc-compiled modules (MCX_EXTRA_FLAGS), every config
run() { c=$1; shift; S=30; [ $c = c4 ] && S=10; [ $c = c5 ] && S=8; env "$@" python bench.py --config $c --no-cpu-baseline --no-cold --no-philox --steps $S --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $*', '%.4g' % d['value'], round(d['ms_per_step'],4))" || echo "$c $* FAILED"; }
for c in c2 c3 c4 c5; do
  run $c MCX_X=0
  run $c MCX_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp"
  run $c MCX_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause"
  run $c MCX_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=iterative-ilp"
  run $c MCX_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=iterative-minreg"
  run $c MCX_EXTRA_DEFINES="MCX_PAIR_LANES=0"
done
