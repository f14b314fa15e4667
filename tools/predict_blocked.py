"""A prediction, written down BEFORE the next hardware run: ms per cache-blocked pass of the n = 30 benchmark
plan for every kernel variant, from (a) hardware counters and timings of the round-2 kernel that are on file, (b) static
facts of the variants (assembly, host grouping code) and (c) three stated assumptions.  No GPU needed.
    python tools/predict_blocked.py > profiles/r06_blocked_prediction.txt

(a) on file (MI355X, n = 30, complex64, tile = 2^13 amplitudes = 64 KiB, 2 workgroups of 512 threads per CU):
    profiles/r02_blocked_scaling.txt   ms of ONE pass with G inner k = 3 gates: G0 2.93, G2 3.03, G4 4.18, G8 7.08, G16 13.19
    profiles/r02_pmc_blocked_tab.txt   per inner gate: 2^25 MFMA instructions (256 per tile), SQ_VALU_MFMA_BUSY_CYCLES 2^30 = 32 cycles
                                       each; GRBM_GUI_ACTIVE / 8 XCDs / wall time = 1.98 GHz effective clock under the counters
    profiles/r03_v2_bench.json         the whole plan: 28 passes, 138 inner gates {k=2: 9, k=3: 102, k=4: 27}: 135.5 ms = 4.84 ms / pass
(b) static: awaited LDS reads per MFMA 0.13 (round-2 loops) / 0.03 (pipelined) (tests/test_kernel_schedule.py);
    barrier-free groups remove 77 of 135 per-gate barriers of this plan (tools/blocked_groups_stats.py);
    the direct first gate is eligible in 30 of 30 passes (HQ_BLOCKED_DIRECT=1 in the same tool), 128 KiB tiles need 25-26 passes.
(c) assumptions -- each can be wrong, the run will say:
    A1  matrix-core time of a gate is its MFMA count x 32 cycles / (4 SIMDs x 256 CUs) at the effective clock; what the round-2
        kernel loses on top of it (0.63 ms per k = 3 gate fitted on this plan = 84 % busy; 0.73-0.76 ms = 70-73 % busy as the slope
        of the one-gate-shape scaling run; the counters say 58-73 % inside gates; the fitted figure is used below) is stall time
        proportional to a stall index  S = awaited reads per MFMA + B x barriers per MFMA  (B = 2: a workgroup barrier parks all
        8 waves for about two LDS round trips; B = 0.5 and B = 4 bracket it)
    A2  the HBM stream of a pass (2.93 ms alone) hides behind the gates except BLOCKED_OVERLAP_MS = 1.3 ms (fitted in round 3 on
        this plan); the direct first gate removes the staging LDS round trip of every tile: one LDS write + read of the tile per
        plane less per pass, 2 x 64 KiB per tile at 79 / 244 B/clk/CU (MI355X_MICROARCH.md section LDS) = 0.28 ms per pass if none of it was
        hidden, 0 if all of it was
    A3  128 KiB tiles: the same per-gate cost per amplitude, fewer passes (planner: below), one workgroup per CU instead of two, so
        nothing covers a workgroup's barriers: its barrier term doubles (B -> 2B)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from hybridq_amd.blocking import blocked_stats, plan_blocked  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402

CLOCK = 1.98e9            # effective clock under load (GRBM_GUI_ACTIVE, profiles/r02_pmc_blocked_tab.txt)
PIPES = 4 * 256
MFMA_PER_TILE = {2: 256, 3: 256, 4: 512}   # 16x16x4 f32 MFMAs per 2^13-amplitude tile (k <= 3: 16 x 16 real embedding, k = 4: 32 x 32)
TILES = 1 << 17
MEASURED_INNER_MS = {1: 0.38, 2: 0.63, 3: 0.63, 4: 1.20}  # hybridq_amd/simulation.py: BLOCKED_INNER_MS (fitted on hardware, round 3)
BASE_MS, OVERLAP_MS = 3.0, 1.3
n = 30
gates = rqc_1q2q(n, depth=40, seed=n)
ident = {q: n - 1 - q for q in range(n)}


def floor_ms(k):
    return MFMA_PER_TILE[k] * TILES * 32 / PIPES / CLOCK * 1e3


def plan(tile_bits):
    ops = plan_blocked(gates, ident, n, tile_bits=tile_bits, low_bits=5, complex_type='complex64', seeds=4)  # as bench.py's cache-blocked legs plan
    passes = [[len(g[1]) for g in op[2]] for op in ops if op[0] == 'B']
    return passes, blocked_stats(ops)


def pass_ms(ks, inner):
    t = sum(inner[k] for k in ks)
    return max(BASE_MS, OVERLAP_MS + t)


passes13, st13 = plan(13)
passes14, st14 = plan(14)
n_g = sum(len(p) for p in passes13)
mf = sum(MFMA_PER_TILE.get(k, 0) for p in passes13 for k in p) / 8  # MFMAs per wave and tile over the plan
bar_r2 = sum(1 for p in passes13 for k in p) / mf                     # one barrier per gate
bar_groups = bar_r2 * (135 - 77) / 135
print(__doc__)
print(f'plan at tile 2^13: {len(passes13)} passes, {n_g} inner gates {st13["inner_k_histogram"]};  tile 2^14: {len(passes14)} passes, {sum(len(p) for p in passes14)} inner gates {st14["inner_k_histogram"]}')
print(f'matrix-core floor per inner gate (A1): k<=3 {floor_ms(3):.3f} ms, k=4 {floor_ms(4):.3f} ms;  measured round-2 kernel: k=3 {MEASURED_INNER_MS[3]:.2f} ms ({floor_ms(3) / MEASURED_INNER_MS[3]:.0%} busy), k=4 {MEASURED_INNER_MS[4]:.2f} ms ({floor_ms(4) / MEASURED_INNER_MS[4]:.0%} busy)')
base = sum(pass_ms(p, MEASURED_INNER_MS) for p in passes13)
print(f'check: the measured coefficients on this plan give {base:.1f} ms = {base / len(passes13):.2f} ms / pass;  hardware (profiles/r03_v2_bench.json): 135.5 ms = 4.84 ms / pass  ({base / 135.5 - 1:+.1%})')
print()
print(f'{"variant (environment)":58s} {"B=0.5":>18s} {"B=2":>18s} {"B=4":>18s}')
rows = [('round 2 = default (PIPE=0 GROUPS=0)', 0.13, bar_r2, passes13, 1.0, 0.0),
        ('HQ_BLOCKED_PIPE=1', 0.03, bar_r2, passes13, 1.0, 0.0),
        ('HQ_BLOCKED_GROUPS=1', 0.13, bar_groups, passes13, 1.0, 0.0),
        ('HQ_BLOCKED_PIPE=1 HQ_BLOCKED_GROUPS=1', 0.03, bar_groups, passes13, 1.0, 0.0),
        ('HQ_BLOCKED_DIRECT=1 (+PIPE) GROUPS=1, staging fully exposed', 0.03, bar_groups, passes13, 1.0, 0.28),
        ('HQ_BLOCKED_DIRECT=1 (+PIPE) GROUPS=1, staging was hidden', 0.03, bar_groups, passes13, 1.0, 0.0),
        ('HQ_BLOCKED_BIG=1 (+PIPE) GROUPS=1, 128 KiB tiles', 0.03, bar_groups, passes14, 2.0, 0.0)]
for name, awaited, bars, passes, bar_mult, saved in rows:
    cells = []
    for B in (0.5, 2.0, 4.0):
        s_ref = 0.13 + B * bar_r2
        s = awaited + B * bar_mult * bars
        inner = {k: (floor_ms(k) + (MEASURED_INNER_MS[k] - floor_ms(k)) * s / s_ref) if k in MFMA_PER_TILE else MEASURED_INNER_MS[k] for k in MEASURED_INNER_MS}
        if passes is passes14:  # twice the amplitudes per tile, half the tiles: the same cost per gate and pass
            pass
        total = sum(max(BASE_MS, OVERLAP_MS - saved + sum(inner[k] for k in p)) for p in passes)
        cells.append(f'{total:6.1f} ms {total / len(passes):5.2f}/pass')
    print(f'{name:58s} ' + ' '.join(f'{c:>18s}' for c in cells))
print()
print('HBM floor of the plan: 28 passes x 17.18 GB / 6.3 TB/s = 76 ms; matrix-core floor (A1, every gate at 100 % busy, nothing else): '
      f'{sum(floor_ms(k) if k in MFMA_PER_TILE else MEASURED_INNER_MS[k] for p in passes13 for k in p):.0f} ms.')
print('What would refute the model: PIPE=1 alone gaining < 3 % (then the waits were already covered by the other three waves of the SIMD and A1 is wrong),')
print('GROUPS=1 alone gaining more than PIPE=1 alone (then barriers, not LDS round trips, are the loss: B >> 4), BIG=1 losing to the 64 KiB tiles (A3: barriers uncovered).')
