#!/bin/bash
# Round 5, first lease: the oracle parity files first (they pin U::apply / swap_array), then the files whose subject is the
# kernel code written without hardware in rounds 3-4, then the A/B that decides that code (pipelined loops vs the
# -DHQ_BLOCKED_NOPIPE -DHQ_GEMM_NOPIPE build shipped as libhq_hip_nopipe.so, barrier-free groups, direct first gate, 128 KiB tiles).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r5_first.sh'
set -u
out=gpurun_out/r5_first
mkdir -p "$out"
NOPIPE=$PWD/hybridq_amd/csrc/libhq_hip_nopipe.so
run() { echo "== $*"; timeout 900 "$@" 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}"; }
rocm-smi --showproductname 2>/dev/null | head -8
run python -m pytest -q -m gpu -x tests/test_gpu_parity.py tests/test_gpu_golden.py | tee "$out/oracle_first.txt"
run python -m pytest -q -m gpu -s tests/test_gpu_round4.py tests/test_gpu_determinism.py | tee "$out/round4_determinism.txt"
for rep in 1 2; do
  echo "== rep $rep pipelined / nopipe"
  timeout 300 python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_pipelined.txt"
  HQ_HIP_LIBRARY=$NOPIPE timeout 300 python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_nopipe.txt"
  for g in 0; do
    HQ_BLOCKED_GROUPS=$g timeout 300 python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_groups_$g.txt"
  done
  for g in 1 0; do
    HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=$g timeout 300 python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_direct_groups_$g.txt"
  done
  for d in 0 1; do
    HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=$d timeout 300 python tools/ab_blocked.py 30 complex64 14 2>&1 | tail -2 | tee -a "$out/blocked_big_direct_$d.txt"
  done
done
timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_pipelined.txt"
HQ_HIP_LIBRARY=$NOPIPE timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_nopipe.txt"
run python __graft_entry__.py smoke | tee "$out/smoke.txt"
run python -m pytest -q -m gpu -x tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_depth_parity.py | tee "$out/round23_depth.txt"
