"""Developer sweep: VALU butterfly kernel ('direct') vs matrix-core role kernel ('mfma') for k <= 3,
every class of target position.  python tools/sweep_valu_vs_mfma.py [n]"""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
DT = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == 'complex128' else torch.float32
core.set_stream(torch.cuda.current_stream().cuda_stream)
from hybridq_amd.simulation import alloc_planes  # noqa: E402
planes = alloc_planes(n, DT, torch.device('cuda'))
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
for p in range(0, n, 2):
    core.apply_U(planes[0], planes[1], haar_unitary(2, rng), [p])
H = n - 1
cases = [[0], [1], [2], [3], [4], [5], [6], [8], [12], [16], [20], [24], [H],
         [0, 1], [0, 5], [1, 12], [2, 3], [3, 4], [2, 9], [4, 5], [5, 20], [6, 7], [8, 16], [12, 13], [20, 24], [H - 1, H], [3, H],
         [0, 1, 2], [0, 7, 13], [2, 3, 4], [3, 9, 20], [4, 5, 6], [8, 9, 10], [10, 17, 25], [H - 2, H - 1, H]]
tot = {'direct': 0.0, 'mfma': 0.0}
for pos in cases:
    U = haar_unitary(1 << len(pos), rng)
    row = {}
    for mode in ('direct', 'mfma'):
        core.set_apply_mode(mode)
        core.apply_U(planes[0], planes[1], U, pos)
        kern = core.last_kernel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(6):
            core.apply_U(planes[0], planes[1], U, pos)
        e1.record()
        torch.cuda.synchronize()
        row[mode] = (e0.elapsed_time(e1) / 6, kern)
        tot[mode] += row[mode][0]
    core.set_apply_mode('auto')
    d, m = row['direct'][0], row['mfma'][0]
    print(f'k={len(pos)} pos={str(pos):<14} direct {d:6.3f} ms ({row["direct"][1]})  mfma {m:6.3f} ms  mfma/direct {m / d:5.3f}', flush=True)
print('sum', tot, 'placement', os.environ.get('HQ_STATE_ALLOC', 'vmm (tuned)'))

# power / clock under each family (rocm-smi sampled in the middle of a 300-launch loop)
import subprocess
import threading


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.split('GPU[0]')[-1].strip(' :\t') for ln in out.splitlines() if any(k in ln for k in ('Power (W)', 'sclk'))]
        return ' | '.join(keep)
    except Exception as e:  # noqa: BLE001
        return repr(e)


for pos in ([12], [5, 20], [2, 3], [10, 17, 25]):
    U = haar_unitary(1 << len(pos), rng)
    for mode in ('direct', 'mfma'):
        core.set_apply_mode(mode)
        core.apply_U(planes[0], planes[1], U, pos)
        torch.cuda.synchronize()
        box = {}
        t = threading.Thread(target=lambda: box.update(s=smi()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            core.apply_U(planes[0], planes[1], U, pos)
            if i == 60:
                t.start()
        e1.record()
        torch.cuda.synchronize()
        t.join()
        print(f'k={len(pos)} pos={str(pos):<14} {mode:<6} {e0.elapsed_time(e1) / 300:6.3f} ms   [{box.get("s")}]', flush=True)
    core.set_apply_mode('auto')
