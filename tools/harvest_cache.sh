#!/bin/bash
# Run on the GPU box (through gpurun): compile every module the GPU tests, the BASELINE configs, the bench and the
# smoke test ask for into gpurun_out/jit_harvest/, which gpurun merges back. Then, in the build container:
#     cp gpurun_out/jit_harvest/*.hsaco wgpu-monte-carlo_amd/wgpu_montecarlo/jit_cache/
# so that a fresh box starts with a warm code-object cache (keys = hash of source + hiprtc version + flags).
set -u
mkdir -p gpurun_out/jit_harvest
export MCX_CACHE_DIR="$GRAFT_REPO_ROOT/gpurun_out/jit_harvest"
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/harvest_tests.log 2>&1; tail -2 gpurun_out/harvest_tests.log
timeout -k 10 300 python tools/run_configs.py --repeat 1 > /dev/null 2>&1
timeout -k 10 300 python tools/run_configs.py --only C4RW,C4D --repeat 1 > /dev/null 2>&1
timeout -k 10 300 python tools/run_configs.py --only C4D --repeat 1 --rng philox > /dev/null 2>&1
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rng philox > /dev/null 2>&1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1
ls gpurun_out/jit_harvest | wc -l; du -sh gpurun_out/jit_harvest
