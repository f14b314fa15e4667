"""Is the k = 6 kernel power-limited?  Four structurally different ways of overlapping its memory and matrix-core phases
land on the same 4.65-4.75 ms, 84 % MFMA-busy at an effective clock of 2.15 GHz instead of 2.4.  The chip clocks
to its power budget, and data of all zeros toggles far fewer wires: if the same kernel on a ZERO state (and/or with a zero
matrix) runs markedly faster, the limit is power, not the schedule.  Prints ms per call for k = 4, 5, 6, 8 on a dense random
state, on a zero state, and with a zero matrix; samples power / clock through rocm-smi while each loop runs."""
import os
import subprocess
import sys
import threading
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.use_torch_stream()
rng = np.random.default_rng(0)
planes = torch.empty((2, 1 << n), dtype=torch.float32, device='cuda')


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ('Power', 'sclk', 'mclk', 'fclk'))]
        return ' | '.join(k.split('GPU[0]')[-1].strip(' :\t') for k in keep)[:300]
    except Exception as e:  # noqa: BLE001
        return f'rocm-smi failed: {e!r}'


def dense():
    core.init_state(planes[0], planes[1], 'plus')
    for p in range(0, n, 2):
        core.apply_U(planes[0], planes[1], haar_unitary(2, rng), [p])
    core.sync()


def run(label, k, U, reps):
    pos = list(range(8, 8 + k))
    core.apply_U(planes[0], planes[1], U, pos)
    torch.cuda.synchronize()
    box = {}
    t = threading.Thread(target=lambda: box.update(smi=smi()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        core.apply_U(planes[0], planes[1], U, pos)
        if i == reps // 4:
            t.start()
    e1.record()
    torch.cuda.synchronize()
    t.join()
    ms = e0.elapsed_time(e1) / reps
    print(f'k={k} {label:<34} {ms:7.3f} ms  {8.0 * (1 << k) * (1 << n) / ms / 1e9:6.1f} TF  {16 * (1 << n) / ms / 1e6:6.0f} GB/s   [{box.get("smi", "")}]', flush=True)


print('idle:', smi(), flush=True)
for k in (4, 5, 6, 8):
    reps = {4: 600, 5: 500, 6: 400, 8: 100}[k]
    U = haar_unitary(1 << k, rng)
    dense()
    run('dense random state, Haar U', k, U, reps)
    planes.zero_()
    run('ZERO state, Haar U', k, U, reps)
    dense()
    run('dense random state, ZERO U', k, np.zeros_like(U), reps)
