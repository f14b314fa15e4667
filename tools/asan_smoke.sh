#!/bin/bash
# AddressSanitizer smoke run of libhq_hip (device code instrumented, gfx950:xnack+).  Build in the build container
# (the apply unit takes ~5 min under ASAN; the .so travels with the snapshot), run on a GPU box.
# Usage: bash tools/asan_smoke.sh [build|run|all] [n_qubits]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
LIB=$REPO/hybridq_amd/csrc/libhq_hip_asan.so
MODE=${1:-all}
FLAGS="--offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g -O2 -std=c++17 -fPIC -DHQ_ASAN -Wno-unused-function -Wno-unused-value"
if [ "$MODE" != run ] && { [ ! -f "$LIB" ] || [ "$MODE" = build ]; }; then
  mkdir -p /tmp/hq_asan
  for u in hq_core hq_apply hq_swap hq_shard hq_state; do
    hipcc $FLAGS -c $REPO/hybridq_amd/csrc/$u.hip -o /tmp/hq_asan/$u.o &
  done
  wait
  hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -shared -fPIC /tmp/hq_asan/*.o -o $LIB
fi
[ "$MODE" = build ] && exit 0
RT=$(dirname $(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1))
hipcc --offload-arch=gfx950 -I $REPO/include $REPO/tools/asan_smoke.cpp -o /tmp/asan_smoke -ldl
export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_LIBRARY_PATH=$RT:/opt/rocm/lib:$LD_LIBRARY_PATH LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so
/tmp/asan_smoke $LIB ${2:-17}
