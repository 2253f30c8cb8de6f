#!/bin/bash
# C3: workgroup size and loop variants
run() { echo "== $*"; env "$@" python bench.py --config c3 --no-cpu-baseline --no-cold --no-philox --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
run MCX_X=0
run MCX_BLOCK=512
run MCX_BLOCK=256
run MCX_NO_NOCLAMP=1 MCX_BLOCK=256
run MCX_EXTRA_DEFINES="MCX_UNROLL=2"
run MCX_EXTRA_DEFINES="MCX_UNROLL=4"
run MCX_EXTRA_DEFINES="MCX_FLUSH=256"
run MCX_EXTRA_DEFINES="MCX_FLUSH=64"
