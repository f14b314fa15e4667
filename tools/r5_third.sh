#!/bin/bash
# Round 5, third lease: rocprofv3 counter passes (one --pmc set per pass, with --kernel-trace only, as gpurun requires) over the
# cache-blocked kernel with the pipelined inner gates (default) and with the loops of round 2 (HQ_BLOCKED_PIPE=0
# HQ_BLOCKED_GROUPS=0): MFMA-busy, LDS wait and instruction counters before / after (copy into profiles/r05_pmc_blocked_*.txt),
# then the per-kernel rates that changed statically this round (complex128 k = 6 role kernel, aux kernels).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r5_third.sh'
set -u
out=gpurun_out/r5_third
mkdir -p "$out"
HQ_BLOCKED_SELFCHECK=0 timeout 900 bash tools/pmc_blocked.sh > "$out/pmc_blocked_pipe1_groups1.txt" 2>&1
HQ_BLOCKED_SELFCHECK=0 HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=0 timeout 900 bash tools/pmc_blocked.sh > "$out/pmc_blocked_round2_kernels.txt" 2>&1
tail -30 "$out/pmc_blocked_pipe1_groups1.txt" "$out/pmc_blocked_round2_kernels.txt"
timeout 600 python tools/sweep_k56.py 2>&1 | tail -40 | tee "$out/sweep_k56.txt"
timeout 600 python tools/sweep_aux.py 2>&1 | tail -30 | tee "$out/sweep_aux.txt"
