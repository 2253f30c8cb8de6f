#!/bin/bash
# A/B kernel variants on the GPU box: tools/ab_variants.sh "<configs>" "<defines 1>" "<defines 2>" ...
# (MCX_EXTRA_DEFINES is a tuning knob of libmcx: "NAME=VALUE;NAME=VALUE" prepended to the JIT translation unit)
CONFIGS=$1; shift
run() { echo "== ${1:-<default>}"; MCX_EXTRA_DEFINES="$1" timeout -k 10 300 python tools/run_configs.py --only "$CONFIGS" --repeat 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'][:3], 'kernel_ms', round(d['kernel_ms'],3), '%.3e' % d['throughput_kernel'], round(d['worst_err_over_3sigma'],3), d['launch'])"; }
run ""
for v in "$@"; do run "$v"; done
