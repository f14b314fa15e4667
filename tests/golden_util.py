"""Loaders/replayers for the committed golden vectors (tests/golden/*.npz, produced by
tests/golden/make_golden.py from the reference itself)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def apply_cases():
    z = load('calls_apply_U.npz')
    for i in range(int(z['n_cases'])):
        yield z[f'c{i}_in'], z[f'c{i}_out'], z[f'c{i}_U'], z[f'c{i}_pos']


def swap_cases():
    z = load('calls_swap.npz')
    for i in range(int(z['n_cases'])):
        yield int(z['n_qubits']), z[f's{i}_pos'], z[f's{i}_out']


def trace(z, prefix):
    kinds = bytes(z[prefix + 'kinds']).decode()
    for i, kind in enumerate(kinds):
        yield kind, z[f'{prefix}{i}_pos'], (z[f'{prefix}{i}_U'] if kind == 'U' else None)


def replay(z, prefix, n, apply_U, swap):
    """Replay a recorded C-ABI call trace.  The reference issues every swap twice (re then
    im plane, simulation.py:623-630): consecutive 'S' entries with the same positions."""
    calls = list(trace(z, prefix))
    i = 0
    while i < len(calls):
        kind, pos, U = calls[i]
        if kind == 'S':
            assert calls[i + 1][0] == 'S' and (calls[i + 1][1] == pos).all()
            swap(pos)  # both planes
            i += 2
        else:
            apply_U(U, pos)
            i += 1


def simple_qasm_gates(z):
    names = [str(x) for x in z['gate_names']]
    mats = {str(nm): z['matrix_' + str(nm)] for nm in z['matrix_names']}
    gates = []
    for nm, qs in zip(names, z['gate_qubits']):
        qs = tuple(int(q) for q in qs if q >= 0)
        gates.append((mats[nm], qs))
    return gates


def rqc_gates(z, tag):
    return [(z[f'{tag}_U{i}'], tuple(int(q) for q in z[f'{tag}_q{i}'])) for i in range(int(z[f'{tag}_n_gates']))]
