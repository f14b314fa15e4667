#!/bin/bash
# Round 5, a lease of ~15 minutes for when little of the round is left: oracle parity, the bench line, the rocprofv3 kernel
# statistics of the same command, the round-4/5 kernel tests.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r5_short.sh'
set -u
out=gpurun_out/r5_short
mkdir -p "$out"
run() { echo "== $*"; timeout 600 "$@" 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}"; }
run python -m pytest -q -m gpu -x tests/test_gpu_parity.py tests/test_gpu_golden.py | tee "$out/oracle_first.txt"
run python bench.py | tee "$out/bench.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --no-variants --no-config-legs > "$OLDPWD/$out/bench_under_rocprof.txt" 2>&1)
db=$(find "$out/prof" -name "*.db" | head -1); [ -n "$db" ] && python profiles/extract_stats.py "$db" "$out/kernel_stats.csv" && head -30 "$out/kernel_stats.csv"
run python -m pytest -q -m gpu -s tests/test_gpu_round4.py | tee "$out/round4.txt"
