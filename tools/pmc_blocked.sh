#!/bin/bash
# rocprofv3 --pmc passes over tools/pmc_blocked.py: where the inner gates of the cache-blocked kernel spend their cycles
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_blocked; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ[C]*_[A-Z0-9_]*" | sort -u > $OUT/avail_sq.txt
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o run -- python $REPO/tools/pmc_blocked.py > $OUT/p$i.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/p$i/*counter_collection.csv')
print('PASS $i: $C')
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if 'apply_blocked' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    for nm in sorted({r['Counter_Name'] for r in rows}):
        vals = [sum(float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == d and r['Counter_Name'] == nm) for d in ids]
        print('   %-34s' % nm, ' '.join('%14.0f' % v for v in vals))
else:
    print('   no counter file:', open('$OUT/p$i.log').read()[-300:].replace('\n', ' | '))
PY
done
