#!/bin/bash
# Round-3 profiling passes (run on the GPU box through gpurun): kernel trace + stats of the default bench line, then
# separate PMC passes (HBM fetch / write bytes, matrix-core busy) over tools/pmc_probe.py (gate kernels incl. the VALU
# kernel Auto now dispatches to, and the one-pass permutation kernel).  Outputs under gpurun_out/prof_r3/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
DB=$(find $OUT/kt -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/profiles/extract_stats.py $DB $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o probe -- python $REPO/tools/pmc_probe.py 30 > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o probe -- python $REPO/tools/pmc_probe.py 30 > $OUT/pmc_mfma.log 2>&1
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $REPO/tools/pmc_summarize.py hbm $F $W $OUT/pmc_hbm_traffic.csv 30
[ -n "$M" ] && python $REPO/tools/pmc_summarize.py mfma $M $OUT/pmc_mfma_busy.csv
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_mfma $OUT/kt
ls -la $OUT
