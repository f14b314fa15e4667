"""Summarise rocprofv3 --pmc passes over tools/pmc_probe.py into the CSVs kept under profiles/.
  python tools/pmc_summarize.py mfma  <counter_collection.csv> <out.csv>
  python tools/pmc_summarize.py hbm   <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [n] 
Counter handling follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, FETCH_SIZE reports half
of the bytes on gfx950 (x2), GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import sys
from collections import OrderedDict


def load(path):
    rows = OrderedDict()
    for r in csv.DictReader(open(path)):
        key = int(r['Dispatch_Id'])
        d = rows.setdefault(key, {'kernel': r['Kernel_Name'], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return rows


def short(name):
    name = name.replace('void hq::', '')
    return name[:name.index('(')] if '(' in name else name


mode = sys.argv[1]
if mode == 'mfma':
    rows = load(sys.argv[2])
    with open(sys.argv[3], 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatch', 'duration_ms', 'GRBM_GUI_ACTIVE_sum_over_8_XCD', 'clock_GHz', 'SQ_VALU_MFMA_BUSY_CYCLES',
                    'mfma_busy_percent_of_1024_SIMDs', 'SQ_INSTS_VALU_MFMA_MOPS', 'mfma_TFLOPs'])
        for k, d in rows.items():
            if 'apply_' not in d['kernel']:
                continue
            grbm = d.get('GRBM_GUI_ACTIVE', 0.0)
            busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            mops = d.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) + d.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.0)
            ms = d['ns'] / 1e6
            w.writerow([short(d['kernel']), k, round(ms, 3), int(grbm), round(grbm / 8 / d['ns'], 3), int(busy),
                        round(100 * busy / (grbm / 8 * 1024), 1) if grbm else '', int(mops), round(mops * 512 / d['ns'] / 1e3, 1)])
else:
    fetch, write = load(sys.argv[2]), load(sys.argv[3])
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 30
    alg = 16.0 * (1 << n)
    with open(sys.argv[4], 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatch', 'FETCH_SIZE_KiB_raw', 'fetch_bytes_corrected', 'WRITE_SIZE_KiB', 'write_bytes',
                    'hbm_bytes_per_launch', 'ratio_to_algorithmic'])
        for k, d in fetch.items():
            if not any(t in d['kernel'] for t in ('apply_', 'bitperm_', 'swap_', 'exchange_pack', 'permute_bits')) or k not in write:
                continue
            fb = d.get('FETCH_SIZE', 0.0) * 2 * 1024
            wb = write[k].get('WRITE_SIZE', 0.0) * 1024
            w.writerow([short(d['kernel']), k, d.get('FETCH_SIZE', 0.0), fb, write[k].get('WRITE_SIZE', 0.0), wb, fb + wb,
                        round((fb + wb) / alg, 5)])
print('wrote', sys.argv[-1] if mode == 'mfma' else sys.argv[4])
