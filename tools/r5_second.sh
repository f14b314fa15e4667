#!/bin/bash
# Round 5, second lease: the whole -m gpu suite in the driver's order, the bench line with every leg, and the rocprofv3
# kernel statistics of the same bench command (copy the summary into profiles/r05_*).
#   /usr/local/graft/bin/gpurun --timeout 3300 -- 'bash tools/r5_second.sh'
set -u
out=gpurun_out/r5_second
mkdir -p "$out"
run() { echo "== $*"; timeout 2400 "$@" 2>&1 | tail -40; echo "rc=${PIPESTATUS[0]}"; }
run python -m pytest tests -q -x -m gpu | tee "$out/gpu_suite.txt"
run python bench.py | tee "$out/bench.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --no-variants > "$OLDPWD/$out/bench_under_rocprof.txt" 2>&1)
db=$(find "$out/prof" -name "*.db" | head -1); [ -n "$db" ] && python profiles/extract_stats.py "$db" "$out/kernel_stats.csv" && head -30 "$out/kernel_stats.csv"
