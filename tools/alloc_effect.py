"""Experiments on the placement effect of rounds 1-2 (why the same gate kernel streams 5.5 TB/s from hipMalloc memory and
6.2-6.4 TB/s from library-mapped granules), folded into one script (they were tools/alloc_effect.py ... alloc_effect8.py;
results: profiles/r02_placement_1..5*.txt).  Fresh process per recipe:
    python tools/alloc_effect.py <experiment 1..8> [arguments of that experiment]
Round 3 added tools/placement_remap.py (same granules, other order) and tools/pmc_channels.sh (per-L2-channel counters)."""
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402


def experiment_1(argv):
    """Why does the same gate kernel stream at 6.3-6.4 TB/s on a 64+ GiB state and at 5.3-5.6 TB/s on the
8 GiB state of n = 30?  Same n = 30 planes placed in allocations of different size / alignment."""
    n = 30
    core.use_torch_stream()
    rng = np.random.default_rng(0)
    GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([29], haar_unitary(2, rng)),
             ([4, 28], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([27, 29], haar_unitary(4, rng))]


    def measure(tag, re, im):
        core.init_state(re, im, 'plus')
        tot = 0.0
        out = []
        for pos, U in GATES:
            core.apply_U(re, im, U, pos, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                core.apply_U(re, im, U, pos, n)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tot += ms
            out.append(f'{ms:.3f}')
        print(f'{tag:<58} re=0x{re.data_ptr():x} im-re=0x{im.data_ptr() - re.data_ptr():x}  mean {tot / len(GATES):.3f} ms = '
              f'{16.0 * (1 << n) / (tot / len(GATES)) / 1e9:.3f} TB/s   [{" ".join(out)}]', flush=True)


    N = 1 << n
    pad = 12288 // 4
    # A: what alloc_planes does today
    raw = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
    measure('A own 8 GiB allocation (alloc_planes)', raw[0, :N], raw[1, :N])
    del raw
    torch.cuda.empty_cache()
    # B..: planes inside a slab of S GiB, at offset off GiB
    for S, off in ((16, 0), (32, 0), (64, 0), (128, 0), (128, 64), (200, 0), (200, 150)):
        try:
            slab = torch.empty(S << 28, dtype=torch.float32, device='cuda')  # S GiB
        except Exception as e:  # noqa: BLE001
            print('slab', S, 'failed', repr(e)[:80])
            continue
        base = (off << 28)
        re = slab[base:base + N]
        im = slab[base + N + pad:base + 2 * N + pad]
        measure(f'B planes inside a {S} GiB slab at +{off} GiB', re, im)
        del slab, re, im
        torch.cuda.empty_cache()
    # C: two separate 4 GiB allocations
    re = torch.empty(N, dtype=torch.float32, device='cuda')
    im = torch.empty(N + pad, dtype=torch.float32, device='cuda')[pad:]
    measure('C two separate allocations', re, im)
    del re, im
    torch.cuda.empty_cache()
    # D: A again (is it the order of allocation / what was freed before?)
    raw = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
    measure('D own 8 GiB allocation again, after the slabs were freed', raw[0, :N], raw[1, :N])


def experiment_2(argv):
    """Fresh-process experiments on the placement effect (tools/alloc_effect.py): argv[1] selects the recipe."""
    recipe = argv[1]
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


def experiment_3(argv):
    """Placement effect, third experiment: physically CONTIGUOUS VRAM (hipDeviceMallocContiguous through
hq_alloc) against the default allocation, fresh process per recipe."""
    recipe = argv[1]
    n = int(argv[2]) if len(argv) > 2 else 30
    N = 1 << n
    pad = 12288
    core.use_torch_stream()
    torch.zeros(1, device='cuda')
    rng = np.random.default_rng(0)
    GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
             ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([n - 3, n - 1], haar_unitary(4, rng))]


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
        print(f'{recipe:<12} n={n} {tag:<46} mean {tot / len(GATES):.3f} ms = {16.0 * (1 << n) / (tot / len(GATES)) / 1e9:.3f} TB/s', flush=True)


    def planes_from(buf, off_re, off_im):
        re = torch.as_tensor(buf.view(off_re, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(off_im, (N,), '<f4'), device='cuda')
        return re, im


    if recipe == 'default':
        raw = torch.empty((2, N + pad // 4), dtype=torch.float32, device='cuda')
        measure('torch.empty (alloc_planes layout)', raw[0, :N], raw[1, :N])
    elif recipe == 'hipmalloc':
        buf = core.DeviceBuffer(8 * N + pad, contiguous=False)
        re, im = planes_from(buf, 0, 4 * N + pad)
        measure('hq_alloc default flags, one buffer', re, im)
    elif recipe == 'contig1':
        buf = core.DeviceBuffer(8 * N + pad, contiguous=True)
        re, im = planes_from(buf, 0, 4 * N + pad)
        measure('hq_alloc CONTIGUOUS, one buffer', re, im)
    elif recipe == 'contig2':
        b0, b1 = core.DeviceBuffer(4 * N, contiguous=True), core.DeviceBuffer(4 * N + pad, contiguous=True)
        re = torch.as_tensor(b0.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(b1.view(pad, (N,), '<f4'), device='cuda')
        measure('hq_alloc CONTIGUOUS, one buffer per plane', re, im)
        print('   re 0x%x im 0x%x' % (re.data_ptr(), im.data_ptr()))


def experiment_4(argv):
    """Placement effect, fourth experiment: with physically contiguous planes the layout is deterministic,
so sweep (a) the distance between the re and im planes and (b) the absolute position (dummy
allocations in front)."""
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
        m = sum(out) / len(out)
        print(f'{tag:<44} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}]', flush=True)


    mode = argv[1]
    if mode == 'pad':
        buf = core.DeviceBuffer(10 * N, contiguous=True)  # 10 GiB: re at 0, im anywhere in [4, 6) GiB
        for pad in (0, 256, 1024, 4096, 8192, 12288, 16384, 20480, 32768, 65536, 98304, 1 << 17, 3 << 16, 1 << 18, 1 << 19, 1 << 20,
                    3 << 19, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 3 << 23, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 3 << 27, 1 << 29,
                    1 << 30, (1 << 30) + 12288, (1 << 30) + (1 << 20)):
            re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
            im = torch.as_tensor(buf.view(4 * N + pad, (N,), '<f4'), device='cuda')
            measure(f'contiguous, im = re + 4 GiB + {pad}', re, im)
    elif mode == 'abs':
        keep = []
        for front in (0, 8, 16, 32, 64, 96, 128, 160, 192, 224):
            while sum(b.nbytes for b in keep) < front << 30:
                keep.append(core.DeviceBuffer(8 << 30, contiguous=True))
            buf = core.DeviceBuffer(8 * N + 12288, contiguous=True)
            re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
            im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
            measure(f'contiguous 8 GiB behind {front} GiB of other buffers', re, im)
            del re, im
            buf.free()


def experiment_5(argv):
    """Placement effect, fifth experiment: physical granules mapped in a SHUFFLED order (VMM)."""
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
        print(f'{tag:<48} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}] norm {nrm:.6f}', flush=True)


    for gran, seed in ((2 << 20, 1), (2 << 20, 0), (16 << 20, 1), (64 << 20, 1), (256 << 20, 1), (1 << 30, 1), (2 << 20, 7)):
        t0 = time.time()
        try:
            buf = core.DeviceBuffer(8 * N + (64 << 20), contiguous=False, scattered=gran, seed=seed)
        except Exception as e:  # noqa: BLE001
            print('granule', gran, 'failed:', repr(e)[:200])
            continue
        t1 = time.time()
        re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
        measure(f'VMM granule {gran >> 20} MiB, seed {seed} (alloc {t1 - t0:.2f} s)', re, im)
        del re, im
        buf.free()


def experiment_6(argv):
    """Placement effect, sixth experiment: hipMalloc and VMM buffers alternating inside ONE process, several
rounds, same six gates: is the VMM advantage a property of the mapping or of the moment?"""
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
        m = sum(out) / len(out)
        print(f'{tag:<44} re=0x{re.data_ptr():x} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}]', flush=True)


    for rnd in range(3):
        p = alloc_planes(n, torch.float32, 'cuda', vmm=True)
        measure(f'round {rnd}: alloc_planes VMM', p[0], p[1])
        del p
        p = alloc_planes(n, torch.float32, 'cuda', vmm=False)
        measure(f'round {rnd}: alloc_planes torch', p[0], p[1])
        del p
        torch.cuda.empty_cache()
        buf = core.DeviceBuffer(8 * N + (64 << 20), contiguous=False, scattered=2 << 20, seed=1)
        re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
        measure(f'round {rnd}: VMM 2 MiB granules shuffled', re, im)
        del re, im
        buf.free()
    # both kinds alive at the same time
    a = alloc_planes(n, torch.float32, 'cuda', vmm=True)
    b = alloc_planes(n, torch.float32, 'cuda', vmm=False)
    for rnd in range(2):
        measure('coexisting: VMM', a[0], a[1])
        measure('coexisting: torch', b[0], b[1])


def experiment_7(argv):
    """Placement effect, seventh experiment: explicit physical layouts through VMM (granule i of a sequentially
created set mapped at a chosen virtual slot): contiguous, re/im interleaved, shuffled, at several granule sizes."""
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


def experiment_8(argv):
    """Placement effect, eighth experiment: which VMM layout is reproducibly fastest?  Granule sizes x layouts,
each twice, n = 30 (and a check at n = 28 / 31 for the best ones)."""
    core.use_torch_stream()
    torch.zeros(1, device='cuda')


    def gates_for(n):
        rng = np.random.default_rng(0)
        return [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
                ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([n - 3, n - 1], haar_unitary(4, rng)),
                ([n - 9, n - 5, n - 2], haar_unitary(8, rng))]


    def measure(tag, re, im, n):
        core.init_state(re, im, 'plus')
        out = []
        for pos, U in gates_for(n):
            core.apply_U(re, im, U, pos, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                core.apply_U(re, im, U, pos, n)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 3)
        nrm = core.norm2(re, im)
        m = sum(out) / len(out)
        print(f'{tag:<50} mean {m:7.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  worst {max(out):.2f} norm {nrm:.6f}', flush=True)
        return m


    def run(tag, n, gran, layout):
        N = 1 << n
        pp = (4 * N) // gran
        tot = 2 * pp
        if layout == 'identity':
            slots = list(range(tot))
        elif layout.startswith('rot'):
            num, den = (int(x) for x in layout[3:].split('/'))
            sh = pp * num // den
            slots = list(range(pp)) + [pp + (i + sh) % pp for i in range(pp)]
        elif layout.startswith('rotg'):
            pass
        elif layout.startswith('shuffle'):
            slots = [int(x) for x in np.random.default_rng(int(layout[7:])).permutation(tot)]
        buf = core.DeviceBuffer(tot * gran, scattered=gran, va_slots=slots)
        re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(4 * N, (N,), '<f4'), device='cuda')
        m = measure(f'n={n} {gran >> 10:6d} KiB {tag}{layout}', re, im, n)
        del re, im
        buf.free()
        return m


    for rep in range(2):
        for gran in (512 << 10, 2 << 20, 8 << 20):
            for layout in ('identity', 'rot1/2', 'rot1/4', 'rot1/8', 'rot3/8', 'rot1/3', 'shuffle1', 'shuffle2'):
                run(f'rep{rep} ', 30, gran, layout)
    for n in (28, 31, 32):
        for layout in ('identity', 'rot1/2', 'shuffle1'):
            run('', n, 2 << 20, layout)
        N = 1 << n
        raw = torch.empty((2, N + 3072), dtype=torch.float32, device='cuda')
        measure(f'n={n} torch.empty (alloc_planes layout)', raw[0, :N], raw[1, :N], n)
        del raw
        torch.cuda.empty_cache()


if __name__ == '__main__':
    if len(sys.argv) < 2 or not sys.argv[1].isdigit() or not 1 <= int(sys.argv[1]) <= 8:
        raise SystemExit(__doc__)
    globals()[f'experiment_{int(sys.argv[1])}']([sys.argv[0]] + sys.argv[2:])
