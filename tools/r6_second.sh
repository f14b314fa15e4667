#!/bin/bash
# Round 6, second lease (~35 min): what decides the opt-in kernel variants and refreshes the counters.  A/B of the
# cache-blocked step under every switch (twice, same box), the tile GEMM and role-kernel sweeps in both loop forms, then
# rocprofv3 counter passes (one --pmc set per pass, with --kernel-trace only, as gpurun requires) over the cache-blocked kernel
# in its default form and with pipelined gates + barrier-free groups: MFMA-busy, LDS wait, instruction counters
# (copy into profiles/r06_pmc_blocked_*.txt), and the HBM traffic of the headline kernels (profiles/traffic.json).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r6_second.sh'
set -u
out=gpurun_out/r6_second
mkdir -p "$out"
ab() { echo "-- $*"; env "$@" timeout 300 python tools/ab_blocked.py 30 complex64 ${TB:-13} 2>&1 | tail -2; }
for rep in 1 2; do
  echo "== rep $rep"
  { ab HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=0
    ab HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=0
    ab HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=0
    TB=14 ab HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=0 HQ_BLOCKED_GROUPS=1
    TB=14 ab HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=1; } | tee -a "$out/blocked_ab.txt"
done
echo "== tile GEMM k = 7..10: the K loop hardware has run (default) / operands ahead of the MFMAs"
timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_pipe0.txt"
HQ_GEMM_PIPE=1 timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_pipe1.txt"
echo "== role kernels k = 5, 6 (complex128 k = 6: default / two LDS bases)"
timeout 600 python tools/sweep_k56.py 2>&1 | tail -40 | tee "$out/sweep_k56_twobase0.txt"
HQ_BIG_TWOBASE=1 timeout 600 python tools/sweep_k56.py 2>&1 | tail -40 | tee "$out/sweep_k56_twobase1.txt"
HQ_BLOCKED_SELFCHECK=0 timeout 900 bash tools/pmc_blocked.sh > "$out/pmc_blocked_default.txt" 2>&1
HQ_BLOCKED_SELFCHECK=0 HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=1 timeout 900 bash tools/pmc_blocked.sh > "$out/pmc_blocked_pipe1_groups1.txt" 2>&1
tail -30 "$out/pmc_blocked_default.txt" "$out/pmc_blocked_pipe1_groups1.txt"
timeout 600 python tools/sweep_aux.py 2>&1 | tail -30 | tee "$out/sweep_aux.txt"
