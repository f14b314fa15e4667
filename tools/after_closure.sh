#!/bin/bash
# Everything written after GPU access was closed in round 3 and during round 4, when it stayed closed,
# in the order that isolates a fault fastest.  One gpurun call (then tools/ab_round4.sh for the A/B of the round-4 kernel change):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/after_closure.sh'
# Outputs under gpurun_out/after_closure/ (copy what should be judged into profiles/).
set -u
out=gpurun_out/after_closure
mkdir -p "$out"
run() { echo "== $*"; "$@" 2>&1 | tail -25; echo "rc=${PIPESTATUS[0]}"; }
# 1. the tests whose bodies or expectations changed, and the new files
run python -m pytest -q -m gpu tests/test_gpu_upstream_mirrors.py tests/test_gpu_zz_guard_bands.py | tee "$out/new_tests.txt"
run python -m pytest -q -m gpu tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_golden.py -k \
  "chooses_a_schedule or blocked_matches_oracle or functional_gate_branch or simple_qasm or error_codes or host_pointer or apply_blocked" | tee "$out/adjusted_tests.txt"
run python -m pytest -q -m gpu tests/test_gpu_determinism.py tests/test_gpu_round3.py | tee "$out/determinism_and_round3.txt"
# 1b. round 4: the opt-in cache-blocked kernels (direct first gate, 128 KiB tiles) on the device
run python -m pytest -q -m gpu -s tests/test_gpu_round4.py | tee "$out/round4_kernels.txt"
# 2. the measurements that go with the host-side changes
run python tools/e2e_small_n.py | tee "$out/e2e_small_n.txt"
run python tools/host_overhead.py | tee "$out/host_overhead.txt"
# 3. the whole suite, the smoke test, the bench line
run python -m pytest tests -q -m gpu | tee "$out/gpu_suite.txt"
run python __graft_entry__.py smoke | tee "$out/smoke.txt"
run python bench.py | tee "$out/bench.txt"
# 4. kernel statistics of the bench (copy the summary into profiles/): dominant kernel average x launches must reproduce roofline.frac
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --no-variants --no-config-legs > "$OLDPWD/$out/bench_under_rocprof.txt" 2>&1)
python profiles/extract_stats.py "$out/prof" 2>/dev/null | head -30 | tee "$out/kernel_stats_head.txt"
