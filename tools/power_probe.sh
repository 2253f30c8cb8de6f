#!/bin/bash
# Sample rocm-smi power / clocks while the bench's async step loop runs (GPU box): is the sustained 2.13 GHz a power cap?
set -u
python bench.py --steps 20000 --warmup 5 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
BENCH=$!
sleep 6
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk" | head -6
  echo "--"
  sleep 0.5
done
wait $BENCH
tail -c 300 gpurun_out/power_bench.json
