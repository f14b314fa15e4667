#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X state-vector evolution core.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): n=30 (+log2 N) qubit random
circuit, depth 40 = 20 x (Haar U(2) on every qubit + Haar U(4) on a random perfect
matching), complex64, NO gate fusion -> 900 gate applications at n=30, k in {1,2}.
One "step" = one pass of that circuit over the resident state vector.  The state is
already in HBM when the timed region starts; matrices (<= 128 B each) cross the C ABI
as host pointers exactly like in the reference (simulation.py:637-644).

Prints the JSON line of the contract (rank 0):  value = amplitude updates per second of the
whole job (= gate-applications/s x 2^n), plus gate_apps_per_s, a `roofline` object for the
dominant kernel measured live with HIP events on the library's stream, and (N=1) a
`cpu_baseline` object: the reference's own C++ core (oracle/_ref, "reference") or this
repo's C restatement ("port") timed on the host cores on a bounded sample of the same
circuit.

Order of work (VERDICT r05 next #1a): timed region -> roofline -> the same kernels on a
plain-placement state -> cpu_baseline -> THE LINE IS PRINTED AND FLUSHED.  Everything
else (`extras`: fused / cache-blocked schedules, per-k rates, data-movement primitives,
config-4 / config-5 legs, the full-depth parity block, the A/B of the opt-in kernel
variants) runs afterwards inside a wall-clock budget (--extras-seconds, default 300),
anything that launches kernel code hardware has not yet run does so in a subprocess with
its own timeout, and the complete line (headline + extras) is printed once more as the
LAST line.  Both lines carry identical contract fields; a run killed at any moment after
the timed region has left a parseable line.
"""
import argparse
import json
import os
import sys
import time


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes expose 256 hardware threads but a 16-CPU quota; 256 OpenMP threads on that run the
    reference 15x SLOWER than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


# numpy's BLAS would otherwise spin one thread per hardware thread (256) into a 16-CPU CFS quota:
# the kernel then freezes the WHOLE process for the rest of each 100 ms period, GPU issue included
os.environ.setdefault('OPENBLAS_NUM_THREADS', str(max(1, usable_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--qubits', type=int, default=None, help='qubits per GPU shard + log2(gpus); default 30 per GPU')
    ap.add_argument('--depth', type=int, default=40)
    ap.add_argument('--workload', default='rqc_1q2q', choices=['rqc_1q2q', 'dense_k34', 'dm'],
                    help='rqc_1q2q: BASELINE configs 2/3; dense_k34: config 4; dm: config 5 (noisy circuit as a 2n-qubit state vector)')
    ap.add_argument('--sweep', default=None, help="'A..B': also report gate-apps/s of the rqc_1q2q generator for every n in "
                    'A..B that fits (north_star: n=30..36), short depth, as a `sweep` list in the JSON line')
    ap.add_argument('--sweep-depth', type=int, default=8)
    ap.add_argument('--dtype', default='complex64', choices=['complex64', 'complex128'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true', help='skip the fused (compress=4) variant')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-events', action='store_true', help='skip per-gate HIP events in the timed region')
    ap.add_argument('--overlap', action='store_true', help='N > 1: exchanges in rounds with the independent local gates applied to the pieces as they land (hybridq_amd.dist.overlap_exchanges; default off until measured on xGMI)')
    ap.add_argument('--extras-seconds', type=float, default=300.0, help='wall-clock budget of everything after the first printed line (0: headline only)')
    ap.add_argument('--plain-gates', type=int, default=30, help='gate applications of the headline circuit timed on a plain-placement (torch allocator) state as well')
    ap.add_argument('--variants-min-qubits', type=int, default=26, help='the blocked_variants leg runs from this size on')
    ap.add_argument('--no-variants', action='store_true', help='skip the blocked_variants leg (the cache-blocked step under the opt-in kernel switches of round 4, one subprocess each)')
    ap.add_argument('--no-config-legs', action='store_true', help='skip the short BASELINE config 4 / config 5 legs after the timed region')
    ap.add_argument('--parity-qubits', type=int, default=24, help='size of the parity_check circuit (the CPU reference runs all of it)')
    ap.add_argument('--leg-parity-qubits', type=int, default=16, help='size of the small-n parity runs of the config legs (even)')
    return ap.parse_args()


def dm_workload(nq, depth):
    """BASELINE config 5: an nq-qubit noisy circuit as a 2 nq-qubit state vector through the dm front-end:
    every gate U becomes U on the left copy and conj(U) on the right copy, followed by a depolarizing superoperator (one
    dense NON-unitary 2k-qubit gate).  Returns the gate list on integer labels 0..2nq-1."""
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.dm import depolarizing, to_statevector_circuit
    noisy = []
    for U, qs in rqc_1q2q(nq, depth=depth, seed=nq):
        noisy.append((U, qs))
        noisy.append(depolarizing(qs, 0.01 if len(qs) == 1 else 0.02))
    sv = to_statevector_circuit(noisy)
    labels = sorted({q for _, qs in sv for q in qs})  # (0, q) < (1, q): left copies are the high index bits
    index = {lab: i for i, lab in enumerate(labels)}
    return [(U, tuple(index[q] for q in qs)) for U, qs in sv]


def cpu_baseline(gates, n, seconds, complex_type):
    """Time the reference C++ core (or the port) on a bounded prefix of the same circuit."""
    import ctypes
    import oracle
    from oracle.binding import aligned_empty
    try:
        lib = oracle.load_ref()
    except Exception:
        lib = oracle.load_port()
    ft = np.float32 if complex_type == 'complex64' else np.float64
    try:
        avail = os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE')
    except (ValueError, OSError):
        avail = 16 << 30
    n_cpu = n
    while n_cpu > 20 and 2 * (1 << n_cpu) * np.dtype(ft).itemsize * 1.5 > avail:
        n_cpu -= 1
    if n_cpu != n:  # same generator, fewer qubits (host RAM too small for the full state)
        from hybridq_amd.circuits import rqc_1q2q
        gates = rqc_1q2q(n_cpu, depth=40, seed=n)
    threads = int(os.environ.get('OMP_NUM_THREADS', usable_cpus()))
    try:
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(threads)
    except OSError:
        pass
    planes = aligned_empty((2, 1 << n_cpu), ft)
    planes[:] = 0
    planes[0, 0] = 1
    warm = min(8, max(0, len(gates) - 2))
    _, info = oracle.evolve_reference_protocol(lib, gates, n_cpu, complex_type=complex_type, planes=planes,
                                               warmup_gates=warm, max_seconds=seconds, to_complex=False)
    gps = info['n_gates'] / info['runtime (s)']
    return {
        'value': gps * (1 << n_cpu),
        'unit': 'amplitudes/s',
        'gate_apps_per_s': gps,
        'cores': threads,
        'kind': lib.kind,
        'sample': (f'first {info["n_gates"]} gate applications (after {warm} warm-up) of the same n={n_cpu} '
                   f'depth-40 circuit through the reference driver protocol (swap policy + apply_U), '
                   f'{info["runtime (s)"]:.1f} s, OpenMP threads={threads} = the cgroup CPU quota of the box '
                   f'({os.cpu_count()} hardware threads visible); '
                   + ('reference core built -Ofast -march=haswell (AVX2, LOG2_PACK_SIZE=3: its -march=native build '
                      'segfaults, SURVEY 8c) in the build container and shipped as oracle/_ref'
                      if lib.kind == 'reference' else 'C port of the reference core (oracle/hq_oracle.c, -O3 -march=haswell)')),
    }


def parity_check(complex_type, depth, n=24):
    """Full circuit (same generator, n=24, every gate) on the reference CPU core through the reference driver protocol
    and on the GPU through hybridq_amd: max-norm and L2 relative difference of ALL final amplitudes for the plain, fused
    and blocked GPU paths, the same against a complex128 evolution, and -- gate by gate, at 12 prefixes of the circuit --
    the deepest prefix at which HIP-vs-reference still meets north_star's literal bar (`literal_bar_depth`)."""
    import oracle
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.simulation import EvolutionState, simulate
    try:
        lib = oracle.load_ref()
    except Exception:
        lib = oracle.load_port()
    gates = rqc_1q2q(n, depth=depth, seed=n)
    cps = sorted({max(1, (len(gates) * i) // 12) for i in range(1, 13)})
    t0 = time.perf_counter()
    ref, info = oracle.evolve_reference_protocol(lib, gates, n, complex_type=complex_type, checkpoints=cps)
    t_cpu = time.perf_counter() - t0
    ref_cp = dict(info['checkpoints'])
    ref_cp[len(gates)] = ref  # the loop snapshots BEFORE gate c; the last prefix is the final state
    scale = float(np.abs(ref).max())
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tolerances import BAR, C_MODEL, circuit_tol, rounding_bound, widths  # the same bounds the -m gpu tests assert
    bar = BAR[np.dtype(complex_type)]

    def rel(a, b):  # max-norm and L2, both relative to b
        d = a - b
        return float(np.abs(d).max() / np.abs(b).max()), float(np.linalg.norm(d) / np.linalg.norm(b))

    out = {'n_qubits': n, 'gate_applications': len(gates), 'cpu_kind': lib.kind, 'cpu_seconds': t_cpu,
           'bar': bar, 'norm': 'max|d| / max|psi| (the *_l2 entries: ||d||_2 / ||psi||_2)',
           'rounding_model_constant': C_MODEL,
           'rounding_model_bound_one_f32_evolution': rounding_bound(widths(gates), complex_type),
           'tolerance_two_evolutions': circuit_tol(gates, gates, complex_type),
           'statement': ("pass = (i) every single call meets the bar (the -m gpu per-call tests; here: the first prefix), "
                         "(ii) GPU vs complex128 truth <= 1.15 x the reference's own distance to it at full depth "
                         '(complex64: two float32 evolutions of hundreds of gates differ by accumulated rounding whatever '
                         'the implementation), (iii) GPU vs reference <= tolerance_two_evolutions.  literal_bar_met / '
                         'literal_bar_depth report north_star\'s 1e-6 itself, with no model: where it stops holding, it '
                         'stops holding for the reference against the truth as well')}
    results = {}
    ok = True
    for name, kw in (('per_gate', dict(compress=0)), ('fused_k4', dict(compress=4)), ('blocked', dict(blocked=True))):
        psi = simulate(gates, initial_state='0' * n, complex_type=complex_type, qubits=list(range(n)), **kw).reshape(-1)
        out['max_rel_diff_' + name], out['l2_rel_diff_' + name] = rel(psi, ref)
        out['literal_bar_met_' + name] = bool(out['max_rel_diff_' + name] <= bar)  # north_star's number itself, whatever the model allows
        ok = ok and out['max_rel_diff_' + name] <= out['tolerance_two_evolutions']
        results[name] = psi
    # gate by gate at the prefixes: HIP vs reference, and (complex64) both against the complex128 evolution
    st = EvolutionState(list(range(n)), complex_type=complex_type, initial_state='0' * n)
    st64 = EvolutionState(list(range(n)), complex_type='complex128', initial_state='0' * n) if complex_type == 'complex64' else None
    rows, done, depth_ok = [], 0, 0
    still = True
    for c in cps:
        for U, qs in gates[done:c]:
            st.apply(U, qs)
            if st64 is not None:
                st64.apply(U, qs)
        done = c
        psi = st.to_numpy().reshape(-1)
        row = {'gates': c, 'hip_vs_reference': rel(psi, ref_cp[c])[0], 'model_bound_two_evolutions': circuit_tol(gates[:c], gates[:c], complex_type)}
        if st64 is not None:
            truth = st64.to_numpy().reshape(-1)
            row['hip_vs_f64'] = rel(psi, truth)[0]
            row['reference_vs_f64'] = rel(ref_cp[c], truth)[0]
        still = still and row['hip_vs_reference'] <= bar
        if still:
            depth_ok = c
        rows.append(row)
    del st, st64
    out['prefixes'] = rows
    out['literal_bar_depth'] = depth_ok  # gate applications into the circuit for which HIP-vs-reference <= bar holds at every prefix checked
    out['literal_bar_depth_of'] = len(gates)
    ok = ok and rows[0]['hip_vs_reference'] <= bar
    if complex_type == 'complex64':
        last = rows[-1]
        out['reference_cpu_f32_vs_f64'] = last['reference_vs_f64']
        truth = simulate(gates, initial_state='0' * n, complex_type='complex128', qubits=list(range(n)), compress=0).reshape(-1)
        for name, psi in results.items():
            out['gpu_f32_%s_vs_f64' % name], out['gpu_f32_%s_vs_f64_l2' % name] = rel(psi, truth)
            ok = ok and out['gpu_f32_%s_vs_f64' % name] <= max(bar, 1.15 * out['reference_cpu_f32_vs_f64'])  # (below the bar the ratio is noise)
        # the depth at which the REFERENCE itself leaves the bar against the truth, for scale
        out['reference_vs_f64_leaves_bar_after'] = next((r['gates'] for r in rows if r['reference_vs_f64'] > bar), None)
    out['pass'] = bool(ok)
    out['literal_bar_met'] = bool(all(v for k, v in out.items() if k.startswith('literal_bar_met_')))
    return out


def _ab_blocked(n, complex_type, bits, env, mode='json', timeout=240):
    """One run of tools/ab_blocked.py (plan + 1 warm-up + 3 timed cache-blocked steps of the depth-40 generator circuit at n
    qubits) in its OWN process under `env`: a wedged wave in kernel code that hardware has not run costs this leg, not the
    line.  Never raises."""
    import subprocess
    try:
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'ab_blocked.py'), str(n), complex_type, str(bits), mode]
        r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        return dict(json.loads(line[-1]), env=env) if line else {'error': (r.stderr or r.stdout)[-300:], 'rc': r.returncode}
    except Exception as e:  # noqa: BLE001
        return {'error': repr(e)}


def blocked_variants(n, complex_type, time_left=lambda: 1e9):
    """The cache-blocked step under every opt-in kernel switch, one subprocess each: pipelined inner gates, barrier-free wave
    groups, both, the tile movement folded into the first gate (with / without groups), 128 KiB tiles on one 1024-thread
    workgroup (staged / direct) and -- a planner setting, not a kernel -- one forced low tile bit less.  The library's default
    (the kernels of round 2, the ones the driver's GPU tests have seen) is the `blocked` leg.  Never raises."""
    tb = 13 if complex_type == 'complex64' else 12
    settings = [('pipe', {'HQ_BLOCKED_PIPE': '1'}, tb), ('groups', {'HQ_BLOCKED_GROUPS': '1'}, tb),
                ('pipe_groups', {'HQ_BLOCKED_PIPE': '1', 'HQ_BLOCKED_GROUPS': '1'}, tb),
                ('direct', {'HQ_BLOCKED_DIRECT': '1'}, tb), ('direct_groups', {'HQ_BLOCKED_DIRECT': '1', 'HQ_BLOCKED_GROUPS': '1'}, tb),
                ('big_tiles', {'HQ_BLOCKED_BIG': '1', 'HQ_BLOCKED_GROUPS': '1'}, tb + 1),
                ('big_tiles_direct', {'HQ_BLOCKED_BIG': '1', 'HQ_BLOCKED_DIRECT': '1', 'HQ_BLOCKED_GROUPS': '1'}, tb + 1),
                # one forced low tile bit less = 64-byte runs per plane instead of whole 128-byte lines: 26 instead of 28 passes
                ('low_bits_minus_1', {'HQ_AB_LOW_BITS': str(4 if complex_type == 'complex64' else 3)}, tb)]
    out = {}
    for name, env, bits in settings:
        left = time_left()
        if left < 20:
            out[name] = {'skipped': 'extras budget spent'}
            continue
        out[name] = _ab_blocked(n, complex_type, bits, env, timeout=min(240, left))
    return out


def config_leg(torch, core, state, gates, n, bytes_per_gate, steps):
    """A short leg of another BASELINE config on the state that is already resident: `steps` timed passes of `gates` applied
    one by one (no fusion, as in the headline), HIP events around every call on the library's stream."""
    core.init_state(state.planes[0], state.planes[1], 'basis', 0)
    plan = [(U, [state.map[q] for q in reversed(qs)]) for U, qs in gates]
    kernel_of = []
    for U, pos in plan:  # untimed pass: kernel names, operand uploads
        core.apply_U(state.planes[0], state.planes[1], U, pos, n)
        kernel_of.append(core.last_kernel_desc())
    torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan] for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ in range(steps):
        for i, (U, pos) in enumerate(plan):
            ev[s_][i][0].record()
            core.apply_U(state.planes[0], state.planes[1], U, pos, n)
            ev[s_][i][1].record()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    per = {}
    for s_ in range(steps):
        for kname, (e0, e1) in zip(kernel_of, ev[s_]):
            per.setdefault(kname, []).append(e0.elapsed_time(e1))
    dom = max(per, key=lambda c: float(np.sum(per[c])))
    avg = float(np.mean(per[dom]))
    khist = {}
    for _, pos in plan:
        khist[str(len(pos))] = khist.get(str(len(pos)), 0) + 1
    return {'gate_applications_per_step': len(plan), 'steps': steps, 'ms_per_step': 1e3 * el, 'gate_apps_per_s': len(plan) / el,
            'amplitudes_per_s': len(plan) / el * float(1 << n), 'k_histogram': khist,
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': bytes_per_gate / (avg * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': bytes_per_gate / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 'avg_launch_ms': avg,
                         'launches': len(per[dom]), 'algorithmic_bytes_per_launch': bytes_per_gate,
                         'per_kernel_avg_ms': {c: float(np.mean(v)) for c, v in sorted(per.items())},
                         'per_kernel_launches': {c: len(v) for c, v in sorted(per.items())}}}


def leg_parity(gates, n, complex_type):
    """The leg's generator at a size the CPU reference finishes at once: GPU (gate by gate) vs the reference core."""
    import oracle
    from hybridq_amd.simulation import simulate
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tolerances import BAR, circuit_tol
    try:
        lib = oracle.load_ref()
    except Exception:
        lib = oracle.load_port()
    exp, _ = oracle.evolve_reference_protocol(lib, gates, n, complex_type=complex_type, qubits=list(range(n)))
    psi = simulate(gates, initial_state='0' * n, complex_type=complex_type, qubits=list(range(n)), compress=0, simplify=False).reshape(-1)
    err = float(np.abs(psi - exp).max() / np.abs(exp).max())
    tol = circuit_tol(gates, gates, complex_type)
    return {'n_qubits': n, 'gate_applications': len(gates), 'cpu_kind': lib.kind, 'max_rel_diff': err, 'tolerance': tol,
            'pass': bool(err <= tol), 'literal_bar_met': bool(err <= BAR[np.dtype(complex_type)])}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
        args.gpus = world
    # HQ_BENCH_SHARE_GPU=1 (development): every rank uses GPU 0 and the process group runs on gloo, which
    # exercises the whole N > 1 path of this file on a one-GPU box (the exchange then takes the
    # peer-to-peer transport; RCCL refuses two ranks per device).  The numbers mean nothing.
    share_gpu = os.environ.get('HQ_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from hybridq_amd import core
    from hybridq_amd.circuits import rqc_1q2q, dense_kq

    g = int(np.log2(world))
    assert 1 << g == world, 'number of GPUs must be a power of two'
    n_local = 30 if args.qubits is None else args.qubits - g
    n = n_local + g
    if args.workload == 'rqc_1q2q':
        gates = rqc_1q2q(n, depth=args.depth, seed=n)
        workload_name = f'n={n} random circuit, depth {args.depth}, Haar 1q/2q gates, {args.dtype}, no fusion'
    elif args.workload == 'dense_k34':
        gates = dense_kq(n, n_gates=200, seed=34)
        workload_name = f'n={n}, 200 Haar 3q/4q dense gates, {args.dtype}'
    else:
        # dm_workload above; n must be even: nq = 15 on one GPU (n = 30), 16 on 2 and 4 GPUs, 17 on 8 GPUs (n = 34, 31
        # local qubits per GPU)
        nq = (15 + (g + 1) // 2) if args.qubits is None else args.qubits // 2
        n, n_local = 2 * nq, 2 * nq - g
        depth = args.depth if args.depth != 40 else 10
        gates = dm_workload(nq, depth)
        workload_name = (f'{nq}-qubit noisy circuit (depth {depth}, depolarizing noise after every gate) as an n={n} state '
                         f'vector via hybridq_amd.dm, {args.dtype}, no fusion: k=1..4 gates, superoperators non-unitary')
    ft = np.dtype('float32') if args.dtype == 'complex64' else np.dtype('float64')
    vb = 2 if ft == np.dtype('float32') else 1
    bytes_per_gate = 2 * (1 << n_local) * 2 * ft.itemsize  # read+write both planes (per GPU)

    # the state of a benchmark lives for the whole run: let the placement search of hq_alloc_state use all of its draws unless one
    # is excellent (default early stop: 6.25 TB/s; the search is outside every timed region and reported as `state_placement`)
    os.environ.setdefault('HQ_STATE_GOOD_TBPS', '6.45')
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    sharded_path = world > 1 or os.environ.get('HQ_BENCH_FORCE_SHARDED') == '1'  # env: smoke-test the N>1 code on one GPU
    if world == 1 and sharded_path:
        for var, val in (('RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533')):
            os.environ.setdefault(var, val)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if not sharded_path:
        from hybridq_amd.simulation import EvolutionState
        state = EvolutionState(list(range(n)), complex_type=args.dtype, initial_state='0' * n)
        plan = [(U, qs, [state.map[q] for q in reversed(qs)]) for U, qs in gates]

        kernel_of = [None] * len(plan)  # instantiation each gate dispatches to (from the library)

        def run_step(events=None):
            for i, (U, qs, pos) in enumerate(plan):
                if events is not None:
                    events[i][0].record()
                core.apply_U(state.planes[0], state.planes[1], U, pos, n)
                if events is not None:
                    events[i][1].record()
                elif kernel_of[i] is None:
                    kernel_of[i] = core.last_kernel_desc()

        n_exchanges = n_permutes = 0
    else:
        from hybridq_amd.dist import ShardedEvolution
        sharded = ShardedEvolution(n, complex_type=args.dtype, initial_state='0' * n, overlap=args.overlap)
        # every step applies the SAME logical circuit; the qubit placement it starts from is
        # whatever the previous step left, so each step gets its own (pre-computed) schedule
        pos0 = dict(sharded.pos)
        schedules = []
        for _ in range(max(1, args.warmup) + args.steps):
            schedules.append(sharded.plan(gates))
            sharded.pos = dict(sharded._planned_final_pos)
        pos_after_main = dict(sharded.pos)
        sharded.pos = pos0
        n_exchanges = sum(1 for op in schedules[-1] if op[0] in ('X', 'XP', 'XO'))
        n_permutes = sum(1 for op in schedules[-1] if op[0] in ('P', 'XP'))
        step_no = [0]

        OP_NAMES = {'P': 'permute_bits', 'X': 'exchange', 'XP': 'exchange_with_folded_permutation', 'XO': 'exchange_in_rounds_with_overlapped_gates'}

        class OpTimer:
            """HIP events around every op of the sharded schedule (torch's current stream IS the
            library stream, set above); 'G' ops are labelled with the kernel they dispatched to."""

            def __init__(self):
                self.rows = []

            def start(self, op):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                return (op[0], e0, e1)

            def stop(self, tok):
                tok[2].record()
                self.rows.append((core.last_kernel_desc() if tok[0] == 'G' else OP_NAMES.get(tok[0], tok[0]), tok[1], tok[2]))

        op_timer = OpTimer() if not args.no_events else None

        def run_step(events=None):
            sched = schedules[step_no[0]]
            step_no[0] += 1
            sharded._planned_final_pos = None
            sharded.run(sched, update_map=False, timer=op_timer if step_no[0] > max(1, args.warmup) else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):  # at least one untimed pass (also records kernel names)
        run_step()
    barrier()
    events = None
    if not sharded_path and not args.no_events:
        events = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                   for _ in gates] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(events[s] if events is not None else None)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gate_apps = args.steps * len(gates)
    gps = gate_apps / elapsed
    result = {
        'metric': 'gate-applications/sec + amplitudes/sec, n-qubit random circuit at 1/2/4/8 MI355X',  # BASELINE.json:metric
        'value_is': 'amplitudes/sec = gate-applications/sec x 2^n (gate-applications/sec in gate_apps_per_s)',
        'value': gps * float(1 << n),
        'unit': 'amplitudes/s',
        'gate_apps_per_s': gps,
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'ms_per_gate': 1e3 * elapsed / gate_apps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32' if ft == np.dtype('float32') else 'f64',
        'data': 'synthetic',
        'config': {
            'workload': workload_name,
            'n_qubits': n,
            'gate_applications_per_step': len(gates),
            'state_bytes_per_gpu': 2 * (1 << n_local) * ft.itemsize,
            'parallelism': f'high-qubit shard x{world}' if world > 1 else 'single GPU',
            'exchanges_per_step': n_exchanges,
            'local_permutation_passes_per_step': n_permutes,
        },
    }

    try:
        from hybridq_amd import simulation as _sim
        if _sim.last_placement:
            result['state_placement'] = dict(_sim.last_placement, note='tuned placement of the state planes (hybridq_amd/simulation.py: '
                                             'VMM_MIN_BYTES); HQ_STATE_ALLOC=torch disables it')
    except Exception as e:  # noqa: BLE001
        result['state_placement_error'] = repr(e)
    if rank == 0 and (events is not None or (sharded_path and op_timer is not None)):
        per_class = {}
        if events is not None:
            for s in range(args.steps):
                for kname, (e0, e1) in zip(kernel_of, events[s]):
                    per_class.setdefault(kname, []).append(e0.elapsed_time(e1))
        else:  # sharded: rank 0's local gate kernels (exchange / permutation passes reported on their own)
            other = {}
            for kname, e0, e1 in op_timer.rows:
                (per_class if kname not in OP_NAMES.values() else other).setdefault(kname, []).append(e0.elapsed_time(e1))
            result['in_loop_ms'] = {c: {'launches': len(v), 'avg_ms': float(np.mean(v)), 'total_ms_per_step': float(np.sum(v)) / args.steps}
                                    for c, v in sorted(other.items())}
        total = {c: float(np.sum(v)) for c, v in per_class.items()}
        dom = max(total, key=total.get)
        avg_ms = float(np.mean(per_class[dom]))
        achieved = bytes_per_gate / (avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))  # PMC-measured HBM bytes per launch (profiles/r01_pmc_hbm_traffic.csv)
                if tj.get('n_qubits') == n_local and tj.get('dtype') == args.dtype:
                    traffic = tj.get(dom)
            except Exception:
                traffic = None
        result['roofline'] = {
            'bound': 'hbm',
            'kernel': dom,
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic,
            'traffic_source': (None if traffic is None else
                               'stored PMC measurement (profiles/traffic.json <- profiles/r02_pmc_hbm_traffic.csv, r03_pmc_hbm_traffic.csv: '
                               '2 x FETCH_SIZE + WRITE_SIZE of this kernel at this n, separate rocprofv3 --pmc passes), '
                               'not a counter of this run'),
            'algorithmic_bytes_per_launch': bytes_per_gate,
            'avg_launch_ms': avg_ms,
            'launches': len(per_class[dom]),
            'per_kernel_avg_ms': {c: float(np.mean(v)) for c, v in sorted(per_class.items())},
            'per_kernel_launches': {c: len(v) for c, v in sorted(per_class.items())},
        }
    # ---- the same kernels on a PLAIN-placement state (torch's allocator): the tuned placement is a draw-and-probe search
    # (hq_alloc_state) whose winner differs from box to box; both figures in the driver's line show the spread
    if rank == 0 and not sharded_path and events is not None and args.plain_gates > 0:
        try:
            free_b, _tot = torch.cuda.mem_get_info()
            if free_b > 1.5 * state.planes.numel() * state.planes.element_size():
                pstate = EvolutionState(list(range(n)), complex_type=args.dtype, initial_state='0' * n, placement='plain')
                sample = plan[:args.plain_gates]
                for U, qs, pos in sample:  # untimed pass
                    core.apply_U(pstate.planes[0], pstate.planes[1], U, pos, n)
                torch.cuda.synchronize()
                pev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in sample]
                for (U, qs, pos), (e0, e1) in zip(sample, pev):
                    e0.record()
                    core.apply_U(pstate.planes[0], pstate.planes[1], U, pos, n)
                    e1.record()
                torch.cuda.synchronize()
                per_p = {}
                for kname, (e0, e1) in zip(kernel_of, pev):
                    per_p.setdefault(kname, []).append(e0.elapsed_time(e1))
                dom_p = result['roofline']['kernel'] if result.get('roofline', {}).get('kernel') in per_p else max(per_p, key=lambda c: float(np.sum(per_p[c])))
                avg_p = float(np.mean(per_p[dom_p]))
                result['roofline_plain_placement'] = {
                    'bound': 'hbm', 'kernel': dom_p, 'achieved': bytes_per_gate / (avg_p * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': bytes_per_gate / (avg_p * 1e-3) / 1e9 / HBM_PEAK_GBS, 'avg_launch_ms': avg_p, 'launches': len(per_p[dom_p]),
                    'gate_applications': len(sample), 'all_kernels_frac': bytes_per_gate / (float(np.mean([t for v in per_p.values() for t in v])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'note': 'the first gate applications of the same circuit on a state from torch.empty (no placement search), HIP events per call'}
                if 'roofline' in result:  # the two placements side by side in the one object every reader of the line keeps
                    result['roofline']['plain_placement_frac'] = result['roofline_plain_placement']['frac']
                    result['roofline']['plain_placement_achieved'] = result['roofline_plain_placement']['achieved']
                del pstate
                torch.cuda.empty_cache()
            else:
                result['roofline_plain_placement'] = {'skipped': 'no room for a second state'}
        except Exception as e:  # noqa: BLE001 -- a reported extra
            result['roofline_plain_placement'] = {'error': repr(e)}
    if rank == 0 and not sharded_path and not args.no_cpu_baseline:
        try:
            result['cpu_baseline'] = cpu_baseline(gates, n, args.cpu_seconds, args.dtype)
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
            result['cpu_baseline'] = {'value': None, 'unit': 'amplitudes/s', 'cores': 0, 'kind': 'port',
                                      'sample': f'failed: {e!r}'}
    if sharded_path:
        result['exchange_transport'] = {'transport': getattr(sharded.backend, 'transport', None), 'note': getattr(sharded.backend, 'transport_note', '')}
        try:
            result['exchange_transport'].update(core.shard_info())  # world / rank / transport as the LIBRARY holds them; rccl_ranks_seen = ncclCommCount of its communicator
        except Exception as e:  # noqa: BLE001
            result['exchange_transport']['info_error'] = repr(e)

    # ---- THE LINE: everything the contract names is in `result` now.  Printed and flushed before any extra runs.
    def emit(final=False):
        if rank == 0:
            result['line'] = 'complete' if final else 'headline (the complete line follows as the last line of this run)'
            print(json.dumps(result), flush=True)

    emit()
    t_extras = time.perf_counter()
    skipped = []

    def time_left():
        return args.extras_seconds - (time.perf_counter() - t_extras)

    def budget(name, need):
        """True when `need` seconds (a generous estimate of the leg) are left; under N > 1 rank 0 decides for everybody."""
        ok = time_left() >= need
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cuda')
            dist.broadcast(flag, src=0)
            ok = bool(flag.item())
        if not ok:
            skipped.append(name)
        return ok

    if sharded_path and budget('exchange', 60):
        # everything in this block is a reported extra: a failure here must never cost the headline line
        try:
            # the exchange on its own: every rank sends (G-1)/G of both planes, one
            # distinct chunk per peer, so the per-link figure is chunk bytes / time
            reps = 3
            shard_bytes = 2 * (1 << n_local) * ft.itemsize
            chunk_bytes = shard_bytes // max(world, 1)

            def time_op(op):
                barrier()
                t0_ = time.perf_counter()
                for _ in range(2 * reps):  # an even count leaves the placement where it was
                    sharded.run([op], update_map=False)
                barrier()
                dt = (time.perf_counter() - t0_) / (2 * reps)
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                return dt

            tx = time_op(('X',))
            perm = np.arange(n_local, dtype=np.uint32)
            if n_local >= 2 * g + 2 and g:  # the eviction pattern: two mid qubits swapped into the top-g slots
                a, b = n_local - 1, n_local // 2
                perm[a], perm[b] = b, a
            elif n_local >= 2:
                perm[n_local - 1], perm[n_local - 2] = n_local - 2, n_local - 1
            tp = time_op(('P', perm))
            txp = time_op(('XP', perm))
            # cache-blocked local passes between the exchanges (same circuit; reported separately)
            if n_local >= 14 and not args.no_fused:
                sharded.pos = dict(pos_after_main)
                bsched = []
                for _ in range(1 + args.steps):
                    bsched.append(sharded.plan(gates, blocked=True))
                    sharded.pos = dict(sharded._planned_final_pos)
                sharded.run(bsched[0], update_map=False)
                barrier()
                t0b = time.perf_counter()
                for sc in bsched[1:]:
                    sharded.run(sc, update_map=False)
                barrier()
                elb = (time.perf_counter() - t0b) / args.steps
                if world > 1:
                    tb_ = torch.tensor([elb], dtype=torch.float64, device='cuda')
                    dist.all_reduce(tb_, op=dist.ReduceOp.MAX)
                    elb = float(tb_.item())
                result['blocked'] = {
                    'ms_per_step': 1e3 * elb,
                    'logical_gate_apps_per_s': len(gates) / elb,
                    'logical_amplitudes_per_s': len(gates) / elb * float(1 << n),
                    'blocked_passes_per_step': sum(1 for op in bsched[-1] if op[0] == 'B'),
                    'plain_gates_per_step': sum(1 for op in bsched[-1] if op[0] == 'G'),
                    'exchanges_per_step': sum(1 for op in bsched[-1] if op[0] in ('X', 'XP', 'XO')),
                }
            result['exchange'] = {
                'transport': getattr(sharded.backend, 'transport', None),
                'transport_note': getattr(sharded.backend, 'transport_note', ''),
                'ms_per_exchange': 1e3 * tx,
                'ms_per_exchange_with_folded_permutation': 1e3 * txp,
                'ms_per_permutation_pass_alone': 1e3 * tp,
                'bytes_sent_per_gpu': shard_bytes - chunk_bytes,
                'bytes_per_link': chunk_bytes,
                'GBps_per_link': chunk_bytes / tx / 1e9 if world > 1 else None,
                'GBps_per_link_with_folded_permutation': chunk_bytes / txp / 1e9 if world > 1 else None,
                'GBps_per_gpu_out': (shard_bytes - chunk_bytes) / tx / 1e9 if world > 1 else None,
                'xgmi_link_peak_GBps': 153.0,
                'link_frac_of_peak': chunk_bytes / tx / 1e9 / 153.0 if world > 1 else None,
                'exchanges_per_step': n_exchanges,
                'of_which_with_folded_permutation': n_permutes,
            }
            # what the first record on a real multi-GPU node should show (VERDICT r02 next #1d): every figure below is
            # arithmetic on the sizes, the 153 GB/s per xGMI link of the guide and single-GPU rates measured in
            # profiles/ (r03_perm_rate.txt: pack pass 5.3-5.7 TB/s; r03_v1_bench.json: 6.2-6.4 TB/s gate kernels)
            gate_ms_model = bytes_per_gate / 6.3e12 * 1e3
            pack_ms_model = 2 * shard_bytes / 5.4e12 * 1e3
            link_ms = chunk_bytes / 153e9 * 1e3
            n_local_gates = len(gates)
            result['exchange']['expected'] = {
                'bytes_per_link_per_exchange': chunk_bytes,          # both planes, one peer; the G-1 links of a GPU run in parallel
                'ms_at_153GBps': link_ms,
                'pack_pass_ms_at_5.4TBps': pack_ms_model,            # only exchanges with a folded permutation pack
                'self_chunk_copy_ms_at_5TBps': 2 * chunk_bytes / 5.0e12 * 1e3,
                'ms_per_exchange_model': link_ms,
                'ms_per_exchange_with_folded_permutation_model': [max(link_ms, pack_ms_model), link_ms + pack_ms_model],  # [planes overlap fully, not at all]
                'local_gate_ms_at_6.3TBps': gate_ms_model,
                'ms_per_step_model': n_local_gates * gate_ms_model + (n_exchanges - n_permutes) * link_ms + n_permutes * (link_ms + pack_ms_model),
                'exchange_share_of_step_model': ((n_exchanges - n_permutes) * link_ms + n_permutes * (link_ms + pack_ms_model)) /
                                                max(1e-9, n_local_gates * gate_ms_model + (n_exchanges - n_permutes) * link_ms + n_permutes * (link_ms + pack_ms_model)),
                'note': 'model, not a measurement: compare with ms_per_exchange / GBps_per_link above on a node with xGMI',
            }
            # exchange / compute overlap (--overlap; hybridq_amd.dist.overlap_exchanges): the gates taken along per exchange are
            # budgeted at 1.25 x the modelled transfer time, so at most that much of every exchange hides behind them
            k_att = int(1.25 * link_ms / gate_ms_model)
            hidden = min(link_ms, k_att * gate_ms_model)
            result['exchange']['expected']['overlap'] = {
                'enabled_in_this_run': bool(args.overlap), 'rounds_per_exchange': 4,
                'attached_gates_per_exchange_budget': k_att, 'ms_hidden_per_exchange_model': hidden,
                'ms_per_step_model_with_overlap': result['exchange']['expected']['ms_per_step_model'] - n_exchanges * hidden,
                'exchanges_in_rounds_this_run': sum(1 for op in schedules[-1] if op[0] == 'XO'),
                'note': 'unmeasured on hardware: no multi-GPU node was available to this build'}
        except Exception as e:  # noqa: BLE001
            result['extras_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and budget('fused', 30):
        try:  # a reported extra: a failure here must never cost the headline line
            # The reference's DEFAULT driver setting fuses the circuit into <= 4-qubit gates first
            # (compress=4, simulation.py:314,436-454; untimed there, :519).  Reported separately:
            # same circuit, same state, fewer and larger gates; "logical" rates count the ORIGINAL
            # gate applications.
            from hybridq_amd.fusion import fuse
            # max_n_qubits = 4 is the reference's default; 5 is what the k = 5 matrix-core kernel
            # makes worthwhile on this GPU (a k = 5 pass costs ~10 % more than a k <= 4 pass)
            for width, key in ((4, 'fused'), (5, 'fused_k5')):
                t_f = time.perf_counter()
                fused = fuse(gates, width, complex_type=args.dtype)
                t_fuse = time.perf_counter() - t_f
                fplan = [(U, [state.map[q] for q in reversed(qs)]) for U, qs in fused]
                for U, pos in fplan:
                    core.apply_U(state.planes[0], state.planes[1], U, pos, n)
                barrier()
                t0f = time.perf_counter()
                for _ in range(args.steps):
                    for U, pos in fplan:
                        core.apply_U(state.planes[0], state.planes[1], U, pos, n)
                barrier()
                el = (time.perf_counter() - t0f) / args.steps
                result[key] = {
                    'max_n_qubits': width,
                    'apply_U_calls_per_step': len(fplan),
                    'k_histogram': {str(k): sum(1 for _, p in fplan if len(p) == k) for k in range(1, width + 1)},
                    'ms_per_step': 1e3 * el,
                    'ms_per_call': 1e3 * el / len(fplan),
                    'logical_gate_apps_per_s': len(gates) / el,
                    'logical_amplitudes_per_s': len(gates) / el * float(1 << n),
                    'host_fusion_seconds_untimed': t_fuse,
                }
        except Exception as e:  # noqa: BLE001
            result['fused_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and budget('blocked', 60):
        # Cache-blocked execution (hybridq_amd/blocking.py): many gates per HBM pass through LDS tiles.  Same circuit; the
        # scheduling is host work done before the clock, like fusion; "logical" rates count the ORIGINAL gate applications.
        # Runs in its OWN process (tools/ab_blocked.py, on a plain-placement state of its own as simulate() picks for blocked
        # schedules): since round 3 the inner loops of this kernel family have been reworked without hardware, and although
        # the library's defaults are the loops hardware has run, nothing after the printed line shares a process with the
        # line's own measurements unless it has to.
        tb = 13 if args.dtype == 'complex64' else 12  # 64 KiB of LDS per tile
        leg = _ab_blocked(n, args.dtype, tb, {}, mode='json_full', timeout=max(30, min(300, time_left())))
        if 'error' in leg:
            result['blocked_error'] = leg
        else:
            ms = float(np.mean(leg['ms_per_step']))
            result['blocked'] = dict(leg.pop('stats', {}), tile_bits=tb, ms_per_step=ms, ms_per_step_runs=leg['ms_per_step'],
                                     logical_gate_apps_per_s=len(gates) / (ms * 1e-3), logical_amplitudes_per_s=len(gates) / (ms * 1e-3) * float(1 << n),
                                     host_planning_seconds_untimed=leg.get('plan_seconds'), state_placement='plain (torch allocator)',
                                     kernel=leg.get('kernel'), selfcheck=leg.get('selfcheck'), process='subprocess (tools/ab_blocked.py)')
            if 'no_fusion' in leg:
                nf = leg['no_fusion']
                result['blocked_no_fusion'] = dict(nf, gate_apps_per_s=len(gates) / (nf['ms_per_step'] * 1e-3),
                                                   amplitudes_per_s=len(gates) / (nf['ms_per_step'] * 1e-3) * float(1 << n))
    if rank == 0 and not sharded_path and not args.no_fused and budget('valu_direct_only', 15):
        try:  # a reported extra: a failure here must never cost the headline line
            # the same 900-gate step through the VALU register-butterfly kernels only (no matrix cores):
            # north_star asks for MFMA only where the tile update is a genuine GEMM (k >= 4); the role
            # kernel is the default because it is faster for k <= 3 as well, this is the evidence
            core.set_apply_mode('direct')
            try:
                run_step()
                barrier()
                t0d = time.perf_counter()
                run_step()
                barrier()
                eld = time.perf_counter() - t0d
                kinds = sorted({core.last_kernel()})
            finally:
                core.set_apply_mode('auto')
            result['valu_direct_only'] = {'ms_per_step': 1e3 * eld, 'gate_apps_per_s': len(gates) / eld,
                                          'amplitudes_per_s': len(gates) / eld * float(1 << n),
                                          'hbm_frac_of_peak': len(gates) / eld * bytes_per_gate / 1e9 / HBM_PEAK_GBS,
                                          'last_kernel': kinds[0]}
        except Exception as e:  # noqa: BLE001
            result['valu_direct_only_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and budget('role_kernel_only', 20):
        try:  # a reported extra: the same step through the matrix-core role kernel ONLY (the round-2 default; Auto now sends
            # k <= 3 gates with every target at bit >= 8 to the VALU kernel): same state, same process, back to back
            core.set_apply_mode('mfma')
            try:
                run_step()
                barrier()
                t0r = time.perf_counter()
                run_step()
                barrier()
                elr = time.perf_counter() - t0r
            finally:
                core.set_apply_mode('auto')
            run_step()
            barrier()
            t0a = time.perf_counter()
            run_step()
            barrier()
            ela = time.perf_counter() - t0a
            result['role_kernel_only'] = {'ms_per_step': 1e3 * elr, 'gate_apps_per_s': len(gates) / elr,
                                          'auto_ms_per_step_back_to_back': 1e3 * ela, 'auto_gate_apps_per_s_back_to_back': len(gates) / ela}
        except Exception as e:  # noqa: BLE001
            result['role_kernel_only_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and budget('per_k', 15):
        try:  # a reported extra: a failure here must never cost the headline line
            # one gate of every width on the same resident state (k >= 5 reach the matrix cores through
            # apply_mfma_big_kernel / apply_gemm_kernel): ms per gate, HBM rate and MFMA rate
            from hybridq_amd.circuits import haar_unitary
            rng_k = np.random.default_rng(5)
            per_k = {}
            for k in range(1, (10 if args.dtype == 'complex64' else 9) + 1):
                pos = sorted(int(p) for p in rng_k.permutation(n)[:k])
                U = np.ascontiguousarray(haar_unitary(1 << k, rng_k), dtype=args.dtype)
                core.apply_U(state.planes[0], state.planes[1], U, pos, n)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4 if k <= 8 else 2
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    core.apply_U(state.planes[0], state.planes[1], U, pos, n)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                per_k[str(k)] = {'ms_per_gate': ms, 'kernel': core.last_kernel_desc(), 'positions': pos,
                                 'hbm_GBps': bytes_per_gate / ms / 1e6, 'TFLOPs': 8.0 * (1 << k) * (1 << n) / ms / 1e9}
            result['per_k'] = per_k
        except Exception as e:  # noqa: BLE001
            result['per_k_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and budget('aux', 15):
        try:  # a reported extra
            # the data-movement primitives of the boundary on the same resident state (north_star names the
            # index-swap primitive): algorithmic bytes = one read + one write of what they touch
            tdt = torch.float32 if ft == np.dtype('float32') else torch.float64
            P = (1 << n) * ft.itemsize  # bytes of one plane
            aux = {}

            def time_aux(name, fn, nbytes, reps=4):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                aux[name] = {'ms': ms, 'GBps': nbytes / ms / 1e6, 'frac_of_hbm_peak': nbytes / ms / 1e6 / HBM_PEAK_GBS}

            rng_a = np.random.default_rng(3)
            re_, im_ = state.planes[0], state.planes[1]
            order8 = np.array([0, 1, 5, 6, 7, 2, 3, 4])  # what the reference driver issues (simulation.py:584-594)
            time_aux('swap_pair_s8_reference_order', lambda: (core.swap(re_, order8, n), core.swap(im_, order8, n)), 4 * P)
            for s_ in (12, 13, 14, 16):
                if s_ > n:
                    continue
                pos_ = rng_a.permutation(s_)
                time_aux(f'swap_one_plane_s{s_}', lambda: core.swap(re_, pos_, n), 2 * P)
            out_c = torch.empty(1 << n, dtype=torch.complex64 if tdt == torch.float32 else torch.complex128, device='cuda')
            time_aux('to_complex', lambda: core.to_complex(re_, im_, out_c), 4 * P)
            del out_c
            tmp_ = torch.empty(1 << n, dtype=tdt, device='cuda')
            perm_ = np.arange(n)
            perm_[n - 1], perm_[n // 2] = n // 2, n - 1
            time_aux('permute_bits_one_plane_top_with_mid', lambda: core.permute_bits(re_, tmp_, perm_, n), 2 * P)
            time_aux('permute_bits_one_plane_random_perm_above_bit4', lambda p=np.concatenate([np.arange(4), 4 + rng_a.permutation(n - 4)]):
                     core.permute_bits(re_, tmp_, p, n), 2 * P)
            del tmp_
            time_aux('norm2', lambda: core.norm2(re_, im_), 2 * P)
            time_aux('vdot_with_itself', lambda: core.vdot(re_, im_, re_, im_), 2 * P)
            time_aux('probabilities_k3', lambda: core.probabilities(re_, im_, [3, n // 2, n - 2], n), 2 * P)
            time_aux('init_state', lambda: core.init_state(re_, im_, 'plus'), 2 * P)
            result['aux'] = aux
        except Exception as e:  # noqa: BLE001
            result['aux_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_config_legs and args.workload == 'rqc_1q2q' and n % 2 == 0 and budget('config_legs', 40):
        # BASELINE configs 4 and 5 as short legs of the SAME run (after the timed config-2 region, on the same resident
        # state): gate-apps/s, the dominant kernel and its share of the HBM peak, and the generator's parity at a small n
        for key, make in (('cfg4_dense_k34', lambda nn: dense_kq(nn, n_gates=200, seed=34)),
                          ('cfg5_noisy_dm', lambda nn: dm_workload(nn // 2, 10))):
            try:
                lg = make(n)
                leg = config_leg(torch, core, state, lg, n, bytes_per_gate, max(1, min(args.steps, 2)))
                leg['workload'] = ('n=%d, 200 Haar 3q/4q dense gates (BASELINE configs[3])' % n if key.startswith('cfg4') else
                                   '%d-qubit noisy circuit, depth 10, as an n=%d state vector via hybridq_amd.dm (BASELINE configs[4] on one GPU)' % (n // 2, n))
                if not args.no_cpu_baseline:
                    leg['parity_small_n'] = leg_parity(make(args.leg_parity_qubits), args.leg_parity_qubits, args.dtype)
                result[key] = leg
            except Exception as e:  # noqa: BLE001 -- a reported extra
                result[key + '_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_cpu_baseline and budget('parity_check', 45):
        # max relative difference of the final amplitudes, GPU vs the reference CPU path,
        # on the same generator at the largest n the CPU finishes in seconds.
        try:
            result['parity_check'] = parity_check(args.dtype, args.depth, n=args.parity_qubits)
        except Exception as e:
            result['parity_check'] = {'error': repr(e)}
    if args.sweep and budget('sweep', 30):
        # north_star: gate-applications/s for n = 30..36 random circuits.  Same generator, short depth,
        # every size that fits this job's HBM (single GPU: one state; sharded: two shard buffers per rank).
        try:
            lo, hi = (int(x) for x in args.sweep.split('..'))
            if not sharded_path:
                del state
            else:
                del sharded
            torch.cuda.empty_cache()
            rows = []
            for n_s in range(lo, hi + 1):
                m_s = n_s - g
                need = 2 * (1 << m_s) * ft.itemsize * (2 if sharded_path and world > 1 else 1)
                free_b, _total = torch.cuda.mem_get_info()
                if m_s < 2 * g + 2 or need > 0.92 * free_b:
                    rows.append({'n_qubits': n_s, 'skipped': f'needs {need / 2**30:.0f} GiB per GPU, {free_b / 2**30:.0f} GiB free'})
                    continue
                gs = rqc_1q2q(n_s, depth=args.sweep_depth, seed=n_s)
                if not sharded_path:
                    from hybridq_amd.simulation import EvolutionState
                    st_s = EvolutionState(list(range(n_s)), complex_type=args.dtype, initial_state='0' * n_s)
                    plan_s = [(U, [st_s.map[q] for q in reversed(qs)]) for U, qs in gs]

                    def once():
                        for U, pos in plan_s:
                            core.apply_U(st_s.planes[0], st_s.planes[1], U, pos, n_s)
                    n_x_s = 0
                else:
                    from hybridq_amd.dist import ShardedEvolution
                    st_s = ShardedEvolution(n_s, complex_type=args.dtype, initial_state='0' * n_s)
                    scheds = []
                    for _ in range(2):
                        scheds.append(st_s.plan(gs))
                        st_s.pos = dict(st_s._planned_final_pos)
                    n_x_s = sum(1 for op in scheds[1] if op[0] in ('X', 'XP', 'XO'))
                    it_s = iter(scheds)

                    def once():
                        st_s.run(next(it_s), update_map=False)
                once()
                barrier()
                t0s = time.perf_counter()
                once()
                barrier()
                dt = time.perf_counter() - t0s
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                rows.append({'n_qubits': n_s, 'gate_applications': len(gs), 'exchanges': n_x_s, 'ms': 1e3 * dt,
                             'gate_apps_per_s': len(gs) / dt, 'amplitudes_per_s': len(gs) / dt * float(1 << n_s),
                             'hbm_frac_of_peak_per_gpu': len(gs) / dt * 4 * (1 << m_s) * ft.itemsize / 1e9 / HBM_PEAK_GBS})
                del st_s
                torch.cuda.empty_cache()
            result['sweep'] = {'depth': args.sweep_depth, 'n_gpus': world, 'rows': rows}
        except Exception as e:  # noqa: BLE001 -- a reported extra
            result['sweep_error'] = repr(e)
    if rank == 0 and not sharded_path and not args.no_fused and not args.no_variants and n >= args.variants_min_qubits and budget('blocked_variants', 30):
        # The cache-blocked step under the opt-in kernel switches built without a GPU in rounds 4-5 (the library reads them once,
        # so each runs in its own process): whoever runs this line on hardware gets the A/B with it.  Last: the longest leg.
        result['blocked_variants'] = blocked_variants(n, args.dtype, time_left)
    result['extras'] = {'budget_seconds': args.extras_seconds, 'used_seconds': time.perf_counter() - t_extras, 'skipped_for_budget': skipped}
    if rank == 0:
        # the figures a reader of the LAST two kilobytes of the output should find (records keep the tail of stdout)
        try:
            summ = {'gate_apps_per_s': result['gate_apps_per_s'], 'roofline_frac': result.get('roofline', {}).get('frac'),
                    'roofline_frac_plain_placement': result.get('roofline_plain_placement', {}).get('frac'),
                    'cpu_baseline_gate_apps_per_s': (result.get('cpu_baseline') or {}).get('gate_apps_per_s'),
                    'cpu_baseline_cores': (result.get('cpu_baseline') or {}).get('cores')}
            for key in ('fused', 'fused_k5', 'blocked'):
                if key in result:
                    summ[key + '_ms_per_step'] = result[key].get('ms_per_step')
            if 'blocked' in result:
                summ['blocked_kernel'] = result['blocked'].get('kernel')
            if 'blocked_variants' in result:
                summ['blocked_variants_ms_per_step'] = {k: (float(np.median(v['ms_per_step'])) if 'ms_per_step' in v else v.get('error', v.get('skipped')))
                                                        for k, v in result['blocked_variants'].items()}
                summ['blocked_variants_selfcheck_failures'] = sum(v.get('selfcheck', {}).get('failures', 0) for v in result['blocked_variants'].values() if isinstance(v, dict))
            pc = result.get('parity_check')
            if isinstance(pc, dict) and 'pass' in pc:
                summ['parity_check'] = {k: pc.get(k) for k in ('n_qubits', 'gate_applications', 'cpu_kind', 'pass', 'literal_bar_met', 'literal_bar_depth', 'literal_bar_depth_of',
                                                                'max_rel_diff_per_gate', 'max_rel_diff_fused_k4', 'max_rel_diff_blocked', 'reference_cpu_f32_vs_f64',
                                                                'gpu_f32_per_gate_vs_f64', 'tolerance_two_evolutions', 'bar')}
            for key in ('cfg4_dense_k34', 'cfg5_noisy_dm'):
                if key in result:
                    summ[key] = {'gate_apps_per_s': result[key].get('gate_apps_per_s'), 'roofline_frac': result[key].get('roofline', {}).get('frac'),
                                 'kernel': result[key].get('roofline', {}).get('kernel'), 'parity_small_n_pass': (result[key].get('parity_small_n') or {}).get('pass')}
            summ['errors'] = sorted(k for k in result if k.endswith('_error'))
            result['summary'] = summ
        except Exception as e:  # noqa: BLE001
            result['summary'] = {'error': repr(e)}
    emit(final=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
