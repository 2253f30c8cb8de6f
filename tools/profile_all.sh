#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   for every BASELINE config c2..c5, through the SAME entry point the driver uses (bench.py --config):
#     kernel-trace stats, then PMC counters in their own passes (gpurun refuses --pmc combined with sys/hip traces;
#     8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# The program is placed directly after `--` (no env / bash -c hop: the profiler has initialised the GPU already), and
# bench.py itself starts no child process under the profiler (--no-cold is implied there, given explicitly anyway).
#   bash tools/profile_all.sh r03 [c2,c3,c4,c5]
set -u
TAG=${1:-r03}
CONFIGS=${2:-c2,c3,c4,c5}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMON="--no-cpu-baseline --no-cold --no-philox"
for CFG in ${CONFIGS//,/ }; do
  STEPS=20; [ "$CFG" = c4 ] && STEPS=6; [ "$CFG" = c5 ] && STEPS=6
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${CFG}_stats" -- python3 bench.py --config $CFG --steps $STEPS --warmup 3 $COMMON > "$OUT/${CFG}_under_rocprof.json" 2> "$OUT/${CFG}_stats.err"
  echo "$CFG stats rc=$?"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/${CFG}_pmc_a" -- python3 bench.py --config $CFG --steps 3 --warmup 1 $COMMON > /dev/null 2> "$OUT/${CFG}_pmc_a.err"
  echo "$CFG pmc a rc=$?"
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${CFG}_pmc_b" -- python3 bench.py --config $CFG --steps 3 --warmup 1 $COMMON > /dev/null 2> "$OUT/${CFG}_pmc_b.err"
  echo "$CFG pmc b rc=$?"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${CFG}_pmc_fetch" -- python3 bench.py --config $CFG --steps 3 --warmup 1 $COMMON > /dev/null 2> "$OUT/${CFG}_pmc_fetch.err"
  echo "$CFG pmc fetch rc=$?"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${CFG}_pmc_write" -- python3 bench.py --config $CFG --steps 3 --warmup 1 $COMMON > /dev/null 2> "$OUT/${CFG}_pmc_write.err"
  echo "$CFG pmc write rc=$?"
done
python3 tools/collect_profiles.py "$TAG" --into "$OUT/summary" && cat "$OUT/summary/${TAG}_pmc_summary.txt"
