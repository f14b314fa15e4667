"""The failure-safe start-up of the exchange transport (hybridq_amd.dist.HipBackend.setup_exchange) with faults injected,
two ranks over gloo on CPU.  The protocol is the product's; the library calls it makes (core.shard_*, ipc_*) are replaced
by doubles that fail, hang or succeed per rank, and the self-test by one that passes or raises.  Whatever happens on one
rank, BOTH ranks must leave the start-up on the same transport, within the timeout, and say why."""
import os
import socket
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Core:
    """Double of the hybridq_amd.core functions the start-up calls; `fault` = (phase, rank, kind)."""

    def __init__(self, rank, fault):
        self.rank, self.fault, self.calls = rank, fault, []

    def _maybe(self, phase):
        self.calls.append(phase)
        f = self.fault
        if f and f[0] == phase and f[1] == self.rank:
            if f[2] == 'raise':
                raise RuntimeError(f'injected failure in {phase}')
            if f[2] == 'hang':
                time.sleep(3600)

    def shard_load_rccl(self):
        self._maybe('load')

    def shard_unique_id(self):
        self._maybe('uid')
        return b'\x07' * 128

    def shard_init_rccl(self, world, rank, uid):
        assert uid == b'\x07' * 128 and rank == self.rank
        self._maybe('init')

    def shard_free(self):
        self.calls.append('free')

    def ipc_export(self, t):
        self._maybe('export')
        return (b'h%d' % self.rank, 0)

    def ipc_open(self, handle, offset):
        self._maybe('open')
        return 0x1000

    def shard_init_p2p(self, world, rank):
        self._maybe('init_p2p')

    def shard_p2p_register(self, plane, addrs):
        assert len(addrs) == 2
        self.calls.append('register')


def _worker(rank, world, port, want, fault, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['HQ_SHARD_TIMEOUT'] = '3'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hybridq_amd.dist import HipBackend
        be = object.__new__(HipBackend)  # __init__ wants a HIP device; the start-up protocol does not
        from types import SimpleNamespace
        fake_torch = SimpleNamespace(cuda=SimpleNamespace(current_device=lambda: 0, set_device=lambda d: None))
        be.torch, be.dist, be.core = fake_torch, dist, _Core(rank, fault)  # the start-up only asks torch for the device
        be.float_type, be.tdt, be.device = np.dtype('float32'), torch.float32, 'cpu'
        be.transport, be.transport_note, be.placement = want, '', 'plain'
        HipBackend._ipc_mappings.clear()

        def selftest(world_, rank_, buffers=None, timeout=None):
            be.core._maybe('selftest')
        be._rccl_selftest = selftest
        buffers = [torch.zeros((2, 16)), torch.zeros((2, 16))]
        t0 = time.monotonic()
        be.setup_exchange(None, buffers)
        took = time.monotonic() - t0
        with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
            f.write(f'{be.transport}|{be.transport_note}|{took:.2f}|{",".join(be.core.calls)}')
    finally:
        dist.destroy_process_group()


CASES = [
    ('rccl', None, 'rccl'),
    ('rccl', ('load', 1, 'raise'), 'torch'),      # librccl does not load on one rank: nobody enters ncclCommInitRank
    ('rccl', ('uid', 0, 'raise'), 'torch'),
    ('rccl', ('init', 0, 'raise'), 'torch'),      # communicator creation fails on one rank
    ('rccl', ('init', 1, 'hang'), 'torch'),       # ... or never returns: the timeout turns it into a failure
    ('rccl', ('selftest', 1, 'raise'), 'torch'),  # the first real exchange delivers wrong data on one rank
    ('p2p', None, 'p2p'),
    ('p2p', ('export', 0, 'raise'), 'torch'),     # hipIpcGetMemHandle refuses the planes on one rank
    ('p2p', ('open', 1, 'raise'), 'torch'),       # a peer's planes cannot be mapped
    ('torch', None, 'torch'),
]


@pytest.mark.parametrize('want,fault,expected', CASES, ids=[f'{w}-{"-".join(map(str, f)) if f else "ok"}' for w, f, _ in CASES])
def test_both_ranks_leave_the_startup_together(tmp_path, want, fault, expected):
    import torch.multiprocessing as mp
    world = 2
    t0 = time.monotonic()
    mp.spawn(_worker, args=(world, _free_port(), want, fault, str(tmp_path)), nprocs=world, join=True)
    assert time.monotonic() - t0 < 60
    results = [open(os.path.join(str(tmp_path), f'rank{r}.txt')).read().split('|') for r in range(world)]
    transports = [r[0] for r in results]
    assert transports == [expected] * world, results
    for r, (transport, note, took, calls) in enumerate(results):
        assert float(took) < 15, results  # HQ_SHARD_TIMEOUT = 3 s per blocking phase
        calls = calls.split(',')
        if fault and expected == 'torch':
            assert 'unavailable' in note and f'rank {fault[1]}' in note, results  # every rank names the rank that failed
            if want == 'rccl':
                assert 'free' in calls  # the half-made communicator is torn down everywhere
                if fault[0] in ('load', 'uid'):
                    assert 'init' not in calls  # nobody entered the collective creation
        else:
            assert note == ''
