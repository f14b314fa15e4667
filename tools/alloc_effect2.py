"""Fresh-process experiments on the placement effect (tools/alloc_effect.py): argv[1] selects the recipe."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

recipe = sys.argv[1]
n = 30
N = 1 << n
pad = 12288 // 4
core.use_torch_stream()
rng = np.random.default_rng(0)
GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([29], haar_unitary(2, rng)),
         ([4, 28], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([27, 29], haar_unitary(4, rng))]


def measure(tag, re, im):
    core.init_state(re, im, 'plus')
    tot = 0.0
    for pos, U in GATES:
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) / 5
    print(f'{recipe:<10} {tag:<40} mean {tot / len(GATES):.3f} ms = {16.0 * (1 << n) / (tot / len(GATES)) / 1e9:.3f} TB/s', flush=True)


free_b, total_b = torch.cuda.mem_get_info()
print(recipe, f'free {free_b / 2**30:.1f} GiB of {total_b / 2**30:.1f}', flush=True)
if recipe == 'plain':
    pass
elif recipe == 'slab_free':  # allocate most of the free memory untouched, free it
    slab = torch.empty(int(0.9 * free_b) // 4, dtype=torch.float32, device='cuda')
    del slab
    torch.cuda.empty_cache()
elif recipe == 'slab_touch':  # ... touched
    slab = torch.empty(int(0.9 * free_b) // 4, dtype=torch.float32, device='cuda')
    slab.zero_()
    torch.cuda.synchronize()
    del slab
    torch.cuda.empty_cache()
elif recipe == 'slab64':
    slab = torch.empty(64 << 28, dtype=torch.float32, device='cuda')
    del slab
    torch.cuda.empty_cache()
elif recipe == 'slab_keep':  # hold 200 GiB, allocate next to it
    slab = torch.empty(200 << 28, dtype=torch.float32, device='cuda')
if recipe == 'separate':
    re = torch.empty(N, dtype=torch.float32, device='cuda')
    im = torch.empty(N + pad, dtype=torch.float32, device='cuda')[pad:]
    measure('two separate allocations', re, im)
else:
    raw = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
    measure('one allocation (alloc_planes layout)', raw[0, :N], raw[1, :N])
    if recipe == 'plain':  # second allocation in the same process, first one still held
        raw2 = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
        measure('second allocation, first still held', raw2[0, :N], raw2[1, :N])
        del raw
        torch.cuda.empty_cache()
        raw3 = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
        measure('third, after freeing the first', raw3[0, :N], raw3[1, :N])
