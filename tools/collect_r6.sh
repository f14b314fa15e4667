#!/bin/bash
# After a lease (tools/r6_first.sh / r6_second.sh) has run: copy what is to be judged from gpurun_out/ (scratch, not tracked)
# into profiles/ (tracked), named per round.
#   bash tools/collect_r6.sh && git add profiles && git commit -m "profiles: round-6 lease"
set -u
a=gpurun_out/r6_first; b=gpurun_out/r6_second
cpy() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cpy $a/kernel_stats.csv            profiles/r06_kernel_stats.csv
cpy $a/bench_under_rocprof.txt     profiles/r06_bench_under_rocprof.json
cpy $a/oracle_first.txt            profiles/r06_gpu_oracle_parity.txt
cpy $a/round4_determinism.txt      profiles/r06_gpu_round4_determinism.txt
cpy $a/full_suite.txt              profiles/r06_gpu_full_suite.txt
cpy $a/smoke.txt                   profiles/r06_gpu_smoke.txt
if [ -s $a/bench.txt ]; then   # both printed lines; the complete one (last) as the round's bench record
  grep '^{' $a/bench.txt | tail -1 > profiles/r06_v1_bench.json && echo "  profiles/r06_v1_bench.json"
  grep '^{' $a/bench.txt | head -1 > profiles/r06_v1_bench_headline.json
fi
for f in blocked_ab gemm_pipe0 gemm_pipe1 sweep_k56_twobase0 sweep_k56_twobase1 pmc_blocked_default pmc_blocked_pipe1_groups1 sweep_aux; do
  cpy $b/$f.txt profiles/r06_$f.txt
done
