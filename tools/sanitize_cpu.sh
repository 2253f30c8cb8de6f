#!/bin/bash
# Host-side sanitizer run -- CPU container only, never on the GPU box (GPU sanitizers are not
# available on the pool, and this needs none): builds libmcx and the oracle with AddressSanitizer + UBSan and runs the
# whole `not gpu` test suite against them -- planning, shard arithmetic, table analysis, guide / cell / slope
# construction, source assembly, the LRU code cache, the hiprtc compile path, the C client's planning half, the oracle.
#   bash tools/sanitize_cpu.sh [pytest args]      -> profiles/r02_sanitizer_cpu.log is a captured run of this script
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/wgpu-monte-carlo_amd/csrc" asan
make -C "$ROOT/oracle" asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
echo "sanitizer runtimes: $ASAN_LIB $UBSAN_LIB"
# the instrumented libraries are dlopen'ed by an uninstrumented python: the ASan runtime has to come first in the
# process. Leak checking is off (CPython and hiprtc keep process-lifetime allocations); everything else is fatal.
export LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:strict_string_checks=1:detect_stack_use_after_return=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export MCX_LIBRARY="$ROOT/wgpu-monte-carlo_amd/wgpu_montecarlo/libmcx_asan.so"
export MCX_ORACLE_LIBRARY="$ROOT/oracle/liboracle_asan.so"
export MCX_CACHE_DIR=$(mktemp -d)          # cold code cache: the hiprtc path runs under the sanitizer too
cd "$ROOT"
python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
