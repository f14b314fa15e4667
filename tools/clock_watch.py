"""Run one gate in a loop for T seconds and log, once per interval, the average kernel time together with
the GPU's clock / power telemetry (sysfs), to see whether the 5.5 vs 6.3 TB/s modes are a DPM state."""
import glob
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

T = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
base = os.path.dirname(dev[0]) if dev else None
try:  # the card this process can see (the host has several)
    import subprocess
    uid = [l.split(':')[-1].strip() for l in subprocess.run(['rocm-smi', '--showuniqueid'], capture_output=True, text=True).stdout.splitlines() if 'Unique ID' in l][0]
    for d in dev:
        if open(os.path.join(os.path.dirname(d), 'unique_id')).read().strip() == uid.replace('0x', ''):
            base = os.path.dirname(d)
except Exception as e:  # noqa: BLE001
    print('card lookup failed', e)


def temps():
    out = []
    for f in sorted(glob.glob(os.path.join(base, 'hwmon', 'hwmon*', 'temp*_input'))):
        try:
            lab = open(f.replace('_input', '_label')).read().strip()
        except Exception:
            lab = os.path.basename(f)
        out.append(f'{lab}={int(open(f).read()) / 1000:.0f}')
    return ' '.join(out)


def cur(name):
    try:
        for line in open(os.path.join(base, name)):
            if '*' in line:
                return line.split(':')[1].replace('*', '').strip()
    except Exception:
        return '?'
    return '?'


def power():
    try:
        p = glob.glob(os.path.join(base, 'hwmon', 'hwmon*', 'power1_average')) + glob.glob(os.path.join(base, 'hwmon', 'hwmon*', 'power1_input'))
        return '%.0fW' % (int(open(p[0]).read()) / 1e6)
    except Exception:
        return '?'


core.use_torch_stream()
N = 1 << n
raw = torch.empty((2, N + 3072), dtype=torch.float32, device='cuda')
re, im = raw[0, :N], raw[1, :N]
core.init_state(re, im, 'plus')
U = haar_unitary(2, np.random.default_rng(0))
print('sysfs', base, flush=True)
t_end = time.time() + T
i = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 300
    for _ in range(reps):
        core.apply_U(re, im, U, [12], n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f't={T - (t_end - time.time()):6.1f}s  {ms:.3f} ms {16.0 * N / ms / 1e9:.3f} TB/s  sclk {cur("pp_dpm_sclk")} mclk {cur("pp_dpm_mclk")} '
          f'power {power()} {temps()}', flush=True)
    i += 1
    if i == 20 and len(sys.argv) > 3:  # optional: a big allocate / free in the middle
        big = torch.empty(int(sys.argv[3]) << 28, dtype=torch.float32, device='cuda')
        del big
        torch.cuda.empty_cache()
        print('--- allocated and freed', sys.argv[3], 'GiB', flush=True)
