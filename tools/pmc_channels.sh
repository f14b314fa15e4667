#!/bin/bash
# Per-channel (per TCC instance) memory-side counters of the SAME gate kernel on torch memory, a slow and a fast VMM
# draw (tools/pmc_placement.py): which L2 channels / HBM stacks are hot?  JSON output keeps the instance dimension.
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_chan; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for C in "TCC_EA0_RDREQ" "TCC_EA0_WRREQ" "TCC_EA0_WRREQ_STALL" "TCC_EA0_RDREQ_LEVEL" "TCC_EA0_WRREQ_LEVEL" "TCC_BUBBLE" "TCC_TAG_STALL" "TCC_REQ" "TCC_EA0_RD_UNCACHED_32B TCC_EA0_WR_UNCACHED_32B" "TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format json csv -d $OUT/p$i -o run -- python $REPO/tools/pmc_placement.py > $OUT/p$i.log 2>&1
  echo "PASS $i: $C" | tee -a $OUT/summary.txt
  grep -E "^(torch|slow|fast)" $OUT/p$i.log | tr '\n' '|' | tee -a $OUT/summary.txt; echo | tee -a $OUT/summary.txt
  python $REPO/tools/pmc_channels_parse.py $OUT/p$i 2>&1 | tee -a $OUT/summary.txt
  # keep what travels back small: the JSON only for the last 6 dispatches is what the parser prints
  find $OUT/p$i -name '*.json' -size +8M -delete
done
