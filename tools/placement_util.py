"""Helpers of the placement experiments (tools/pmc_placement.py, tools/placement_remap.py): an
explicitly laid-out VMM buffer seen by torch, and the gate-application probe.  The PRODUCT's search lives behind the C
ABI (hq_alloc_state); these are for experiments that need control over the layout."""
import numpy as np

from hybridq_amd import core


def _torch():
    import torch
    return torch


class _VmmPlanes:
    """Owner of a VMM-backed buffer seen by torch through ``__cuda_array_interface__``; the physical
    granules go back to the driver when the last tensor aliasing it dies (the virtual range is retired,
    see hq_free)."""

    def __init__(self, nbytes, shape, typestr, granule, shuffle_seed):
        ng = -(-nbytes // granule)
        slots = np.random.default_rng(shuffle_seed).permutation(ng) if shuffle_seed else np.arange(ng)
        self.buf = core.DeviceBuffer(ng * granule, scattered=granule, va_slots=slots)
        self.layout = f'{granule >> 20} MiB granules, ' + (f'shuffled (seed {shuffle_seed})' if shuffle_seed else 'in creation order')
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (self.buf.ptr, False),
                                         'version': 2, 'strides': None}

    def __del__(self):
        try:
            self.buf.free()
        except Exception:
            pass


def _probe_ms(planes, n, float_type):
    """Average time of a few gate applications on `planes` (any content; they are overwritten)."""
    torch = _torch()
    rng = np.random.default_rng(0)
    ct = np.dtype('complex64') if np.dtype(float_type) == np.dtype('float32') else np.dtype('complex128')

    def haar(d):
        q, r = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))
        return (q * (np.diagonal(r) / np.abs(np.diagonal(r)))).astype(ct)

    gates = [([3], haar(2)), ([n // 2], haar(2)), ([n - 1], haar(2)), ([5, n - 3], haar(4))]
    core.init_state(planes[0], planes[1], 'plus')
    for pos, U in gates:
        core.apply_U(planes[0], planes[1], U, pos, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    core.sync()
    e0.record()
    for _ in range(2):
        for pos, U in gates:
            core.apply_U(planes[0], planes[1], U, pos, n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * len(gates))

