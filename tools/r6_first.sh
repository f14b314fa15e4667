#!/bin/bash
# Round 6, first lease (~25 min): oracle parity first, then the driver's own bench command (the line is printed right after
# the timed region; extras incl. the A/B of every opt-in kernel variant follow under a 300 s budget), the rocprofv3 kernel
# statistics of the same command, smoke, the tests of the kernel code written without hardware (rounds 3-5), the full suite.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r6_first.sh'
set -u
out=gpurun_out/r6_first
mkdir -p "$out"
run() { echo "== $*"; timeout 900 "$@" 2>&1 | tail -40; echo "rc=${PIPESTATUS[0]}"; }
rocm-smi --showproductname 2>/dev/null | head -8
run python -m pytest -q -m gpu -x tests/test_gpu_parity.py tests/test_gpu_golden.py | tee "$out/oracle_first.txt"
echo "== python bench.py --gpus 1 --steps 20 --warmup 5"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.txt" 2> "$out/bench.err"; echo "rc=$?"; head -c 3000 "$out/bench.txt"; echo
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --extras-seconds 0 > "$OLDPWD/$out/bench_under_rocprof.txt" 2>&1)
db=$(find "$out/prof" -name "*.db" | head -1); [ -n "$db" ] && python profiles/extract_stats.py "$db" "$out/kernel_stats.csv" && head -30 "$out/kernel_stats.csv"
run python __graft_entry__.py smoke | tee "$out/smoke.txt"
run python -m pytest -q -m gpu -s tests/test_gpu_round4.py tests/test_gpu_determinism.py | tee "$out/round4_determinism.txt"
run python -m pytest -q -m gpu -x tests | tee "$out/full_suite.txt"
