#!/bin/bash
# Round-4 changes that need a GPU to be judged, back to back on one box (one gpurun call; outputs under gpurun_out/ab_round4/):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/ab_round4.sh'
# 1. cache-blocked schedule of the n = 30 benchmark circuit with the barrier-free wave groups (default) and with one workgroup
#    barrier per inner gate (HQ_BLOCKED_GROUPS=0): each is its own process (the library reads its switches once), twice each
#    in alternation so that a drifting clock does not decide
# 2. the same bit for bit: determinism test incl. the GROUPS=0 setting
# 3. rocprofv3 kernel stats of the blocked pass with groups
set -u
out=gpurun_out/ab_round4
mkdir -p "$out"
# 0. LDS reads ahead of the matrix cores in the inner gates (default since the third session of round 4) against the old loop:
#    a second build of the library with -DHQ_BLOCKED_NOPIPE (tools/_ab/, ~1 min of hipcc on the box), same circuit, alternating
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from hybridq_amd import build
os.makedirs('tools/_ab/nopipe', exist_ok=True)
print(build.build(force=True, extra_flags=['-DHQ_BLOCKED_NOPIPE', '-DHQ_GEMM_NOPIPE'], lib=os.path.abspath('tools/_ab/libhq_hip_nopipe.so'), objdir=os.path.abspath('tools/_ab/nopipe')))
PY
for rep in 1 2 3; do
  echo "== rep $rep pipelined (in-tree) / old loop (nopipe build)"
  python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_pipelined.txt"
  HQ_HIP_LIBRARY=$PWD/tools/_ab/libhq_hip_nopipe.so python tools/ab_blocked.py 30 complex64 2>&1 | tail -2 | tee -a "$out/blocked_nopipe.txt"
done
# 0b. the same for the k = 7..10 tile GEMM (B operands one K-step, A operands one step group ahead of the MFMAs)
python tools/sweep_gemm.py 2>&1 | tail -12 | tee "$out/gemm_pipelined.txt"
HQ_HIP_LIBRARY=$PWD/tools/_ab/libhq_hip_nopipe.so python tools/sweep_gemm.py 2>&1 | tail -12 | tee "$out/gemm_nopipe.txt"
for rep in 1 2; do
  for g in 1 0; do
    echo "== rep $rep HQ_BLOCKED_GROUPS=$g"
    HQ_BLOCKED_GROUPS=$g python tools/ab_blocked.py 30 complex64 2>&1 | tail -3 | tee -a "$out/blocked_groups_$g.txt"
  done
  # 1b. the tile movement folded into the first gate of a pass (apply_blocked_direct_kernel, opt-in: 24 of the 30 passes
  #     of the benchmark plan are eligible), with and without the barrier-free groups
  for g in 1 0; do
    echo "== rep $rep HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=$g"
    HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=$g python tools/ab_blocked.py 30 complex64 2>&1 | tail -3 | tee -a "$out/blocked_direct_groups_$g.txt"
  done
done
# 1c. 128 KiB tiles (2^14 amplitudes, one 1024-thread workgroup per CU, HQ_BLOCKED_BIG=1: 25 instead of 28 passes), staged and direct
for rep in 1 2; do
  for d in 0 1; do
    echo "== rep $rep HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=$d tile bits 14"
    HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=$d python tools/ab_blocked.py 30 complex64 14 2>&1 | tail -3 | tee -a "$out/blocked_big_direct_$d.txt"
  done
done
python -m pytest -q -m gpu tests/test_gpu_round4.py -s 2>&1 | tail -8 | tee "$out/direct_parity.txt"
python -m pytest -q -m gpu tests/test_gpu_determinism.py 2>&1 | tail -5 | tee "$out/determinism.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/tools/ab_blocked.py" 30 complex64 > "$OLDPWD/$out/prof.log" 2>&1
HQ_BLOCKED_DIRECT=1 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_direct" -- python "$OLDPWD/tools/ab_blocked.py" 30 complex64 > "$OLDPWD/$out/prof_direct.log" 2>&1
cd "$OLDPWD"
python profiles/extract_stats.py "$out/prof" 2>/dev/null | head -20 | tee "$out/kernel_stats_head.txt"
python profiles/extract_stats.py "$out/prof_direct" 2>/dev/null | head -20 | tee "$out/kernel_stats_direct_head.txt"
