"""Placement effect, seventh experiment: explicit physical layouts through VMM (granule i of a sequentially
created set mapped at a chosen virtual slot): contiguous, re/im interleaved, shuffled, at several granule sizes."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = 30
N = 1 << n
core.use_torch_stream()
torch.zeros(1, device='cuda')
rng = np.random.default_rng(0)
GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
         ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng))]


def measure(tag, re, im):
    core.init_state(re, im, 'plus')
    out = []
    for pos, U in GATES:
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(4):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 4)
    nrm = core.norm2(re, im)
    m = sum(out) / len(out)
    print(f'{tag:<58} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}] norm {nrm:.6f}', flush=True)


def run(tag, gran, slots_fn):
    per_plane = (4 * N) // gran
    total = 2 * per_plane
    slots = slots_fn(per_plane, total)
    assert sorted(slots) == list(range(total))
    buf = core.DeviceBuffer(total * gran, scattered=gran, va_slots=slots)
    re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(4 * N, (N,), '<f4'), device='cuda')
    measure(f'{tag} granule {gran >> 10} KiB (min {buf.granule_min >> 10} KiB)', re, im)
    del re, im
    buf.free()


r = np.random.default_rng(5)
for gran in (2 << 20, 64 << 20):
    run('identity (re then im, pad 0)', gran, lambda pp, tot: list(range(tot)))
    run('re/im interleaved (phys 2i -> re_i, 2i+1 -> im_i)', gran, lambda pp, tot: [(p // 2) + (p % 2) * pp for p in range(tot)])
    run('shuffled', gran, lambda pp, tot: [int(x) for x in r.permutation(tot)])

    def pair_shuffle(pp, tot):
        order = r.permutation(pp)
        slots = [0] * tot
        for i in range(pp):  # physical pair i backs chunk order[i] of both planes
            slots[2 * i] = int(order[i])
            slots[2 * i + 1] = pp + int(order[i])
        return slots
    run('re/im interleaved, pairs shuffled', gran, pair_shuffle)
    run('im shifted by half a plane (phys i -> re_i, im_(i+pp/2))', gran,
        lambda pp, tot: list(range(pp)) + [pp + (i + pp // 2) % pp for i in range(pp)])
