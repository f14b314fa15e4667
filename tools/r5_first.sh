#!/bin/bash
# Round 5, first lease: the oracle parity files first (they pin U::apply / swap_array), then the files whose subject is the
# kernel code written without hardware in rounds 3-4 (every HQ_BLOCKED_* setting against the oracle, the library's own
# bit-for-bit cross-check of the variants), then the A/B timings that decide that code: all variants are template
# parameters of one build, chosen by environment switches.
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/r5_first.sh'
set -u
out=gpurun_out/r5_first
mkdir -p "$out"
run() { echo "== $*"; timeout 1200 "$@" 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}"; }
ab() { echo "-- $*"; env "$@" timeout 300 python tools/ab_blocked.py 30 complex64 ${TB:-13} 2>&1 | tail -2; }
rocm-smi --showproductname 2>/dev/null | head -8
run python -m pytest -q -m gpu -x tests/test_gpu_parity.py tests/test_gpu_golden.py | tee "$out/oracle_first.txt"
run python -m pytest -q -m gpu -s tests/test_gpu_round4.py tests/test_gpu_determinism.py | tee "$out/round4_determinism.txt"
# the bench line (it carries the A/B of every variant in `blocked_variants` and the self-check status) and the rocprofv3 kernel
# statistics of the same command, before anything longer: a lease that ends early still leaves the record
run python bench.py | tee "$out/bench.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --no-variants > "$OLDPWD/$out/bench_under_rocprof.txt" 2>&1)
db=$(find "$out/prof" -name "*.db" | head -1); [ -n "$db" ] && python profiles/extract_stats.py "$db" "$out/kernel_stats.csv" && head -30 "$out/kernel_stats.csv"
run python __graft_entry__.py smoke | tee "$out/smoke.txt"
for rep in 1 2; do
  echo "== rep $rep"
  { ab HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=0
    ab HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=0
    ab HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=1
    ab HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_GROUPS=0
    TB=14 ab HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=0
    TB=14 ab HQ_BLOCKED_BIG=1 HQ_BLOCKED_DIRECT=1; } | tee -a "$out/blocked_ab.txt"
done
echo "== tile GEMM k = 7..10, operands ahead of the MFMAs (default) / the old K loop"
timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_pipe1.txt"
HQ_GEMM_PIPE=0 timeout 600 python tools/sweep_gemm.py 2>&1 | tail -14 | tee "$out/gemm_pipe0.txt"
run python -m pytest -q -m gpu -x tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_depth_parity.py | tee "$out/round23_depth.txt"
