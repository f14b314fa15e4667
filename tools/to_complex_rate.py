"""to_complex (hq_to_complex64) on an n = 30 state, tuned placement and torch memory: ms and GB/s (16 * 2^n bytes per call)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from hybridq_amd import core
from hybridq_amd.simulation import alloc_planes
n = 30
core.use_torch_stream()
for place in ('tuned', 'torch'):
    pl = alloc_planes(n, torch.float32, 'cuda', vmm=place == 'tuned')
    core.init_state(pl[0], pl[1], 'plus')
    out = torch.empty(1 << n, dtype=torch.complex64, device='cuda')
    core.to_complex(pl[0], pl[1], out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): core.to_complex(pl[0], pl[1], out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print('VAR', os.environ.get('HQ_TOCOMPLEX_VARIANT', '0'), 'GRID', os.environ.get('HQ_TOCOMPLEX_GRID', '32'), place, f'{ms:.3f} ms {16 * (1 << n) / ms / 1e6:.0f} GB/s', flush=True)
    del pl, out
