"""Print, for the last six launches of the gate kernel in a rocprofv3 JSON result (torch, torch, slow, slow, fast, fast),
each counter per instance: min / mean / max over the instances and the per-XCD and per-channel sums when the records
carry dimensions."""
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + '/**/*results.json', recursive=True) or glob.glob(d + '/**/*.json', recursive=True)
if not files:
    print('   no json result in', d)
    sys.exit(0)
doc = json.load(open(files[0]))
root = doc['rocprofiler-sdk-tool'][0] if 'rocprofiler-sdk-tool' in doc else doc
kernels = {k['kernel_id']: k.get('formatted_kernel_name', k.get('kernel_name', '?')) for k in root.get('kernel_symbols', [])}
counters = {}
for c in root.get('counters', []):
    counters[c['id']['handle'] if isinstance(c.get('id'), dict) else c.get('id')] = c
recs = root.get('callback_records', {}).get('counter_collection', []) or root.get('buffer_records', {}).get('counter_collection', [])
print('   records:', len(recs), 'keys of one:', list(recs[0].keys()) if recs else None)
if recs:
    r0 = recs[0]
    print('   sample record (truncated):', json.dumps(r0)[:1500])
sel = []
for r in recs:
    di = r.get('dispatch_data', {}).get('dispatch_info', {})
    name = kernels.get(di.get('kernel_id'), '')
    if 'apply_mfma_kernel<float, 4, 0, 2, true>' in name:
        sel.append(r)
sel = sel[-6:]
labels = ['torch', 'torch', 'slow', 'slow', 'fast', 'fast']
for lab, r in zip(labels, sel):
    per = defaultdict(list)
    for rec in r.get('records', []):
        cid = rec.get('counter_id', {})
        cid = cid.get('handle') if isinstance(cid, dict) else cid
        per[cid].append(rec.get('value'))
    for cid, vals in per.items():
        nm = counters.get(cid, {}).get('name', str(cid))
        vals = [float(v) for v in vals]
        print(f'   {lab:5s} {nm:28s} n={len(vals):4d} sum={sum(vals):.4g} min={min(vals):.4g} mean={sum(vals)/len(vals):.4g} max={max(vals):.4g}')
        if len(vals) in (128, 256):
            per_ch = [sum(vals[i] for i in range(len(vals)) if i % 16 == ch) for ch in range(16)]
            per_x = [sum(vals[x * (len(vals) // 8):(x + 1) * (len(vals) // 8)]) for x in range(8)]
            print('         per channel (i%16):', ' '.join(f'{v:.3g}' for v in per_ch))
            print('         per XCD (block)   :', ' '.join(f'{v:.3g}' for v in per_x))
