run() { echo "== $1"; MCX_EXTRA_DEFINES="$1" timeout -k 10 200 python tools/run_configs.py --only C3,C4 --repeat 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'][:3], 'kernel_ms', round(d['kernel_ms'],3), '%.3e' % d['throughput_kernel'], round(d['worst_err_over_3sigma'],3))"; }
run ""
run "MCX_COLD=MCX_DEV"
run "MCX_UNIFORM_TABLES=1"
run "MCX_UNIFORM_TABLES=1;MCX_COLD=MCX_DEV"
