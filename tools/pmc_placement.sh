#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_place; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum" "TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_UC_WRITE_REQ_sum TCP_TCC_RW_WRITE_REQ_sum TCP_TCC_CC_WRITE_REQ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o run -- python $REPO/tools/pmc_placement.py > $OUT/p$i.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/p$i/*counter_collection.csv')
log = [l.strip() for l in open('$OUT/p$i.log') if l.startswith(('torch','slow','fast','draw'))]
print('PASS $i:', ' | '.join(l for l in log if not l.startswith('draw')))
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if 'apply_mfma_kernel<float, 4, 0, 2, true>' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})[-6:]
    names = sorted({r['Counter_Name'] for r in rows})
    for nm in names:
        vals = [sum(float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == d and r['Counter_Name'] == nm) for d in ids]
        print('   %-46s' % nm, ' '.join('%14.0f' % v for v in vals))
else:
    print('   no counter file', open('$OUT/p$i.log').read()[-400:])
PY
done
