#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench command, SQ / FETCH_SIZE / WRITE_SIZE counters in separate passes
#   (gpurun refuses --pmc combined with sys/hip traces), kernel stats of the BASELINE configs C3-C5.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_stats.err"
echo "bench stats rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq.err"
echo "pmc sq rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.err"
echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.err"
echo "pmc write rc=$?"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/pmc_grbm" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_grbm.err"
echo "pmc grbm rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/configs_stats" -- python3 tools/run_configs.py --repeat 2 > "$OUT/configs_under_rocprof.jsonl" 2> "$OUT/configs_stats.err"
echo "configs stats rc=$?"
find "$OUT" -name "*kernel_stats.csv" -exec sh -c 'echo "== $1"; head -6 "$1"' _ {} \;
