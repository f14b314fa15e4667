"""TEST INFRASTRUCTURE: makes CPU torch answer to the small `torch.cuda` vocabulary this repository uses, so that the
`-m gpu` tests -- and the product's own driver code above the C ABI -- can run on a box without a GPU against the host
emulation of the HIP library (tests/emu).  "Device" tensors are ordinary CPU tensors; with HQ_EMU_HOST_IS_DEVICE=1 the
emulated runtime classifies every pointer as device memory, so the library takes its device-pointer paths on them.
Installed only by tests/conftest.py when HQ_EMU_GPU_SUITE=1; nothing under hybridq_amd/ refers to it."""
import contextlib
import ctypes
import time

import numpy as np


def _to_cpu_device(dev):
    import torch
    if dev is None:
        return None
    if isinstance(dev, int):
        return 'cpu'
    if isinstance(dev, str):
        return 'cpu' if dev.startswith('cuda') else dev
    if isinstance(dev, torch.device):
        return torch.device('cpu') if dev.type == 'cuda' else dev
    return dev


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def record_event(self, ev=None):
        return ev or _Event()

    def query(self):
        return True


class _Event:
    def __init__(self, *a, **k):
        self.t = time.perf_counter()

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _from_cuda_array_interface(obj):
    import torch
    cai = obj.__cuda_array_interface__
    dt = np.dtype(cai['typestr'])
    shape = tuple(cai['shape'])
    strides = cai.get('strides')
    ptr = cai['data'][0]
    if strides is None:
        nbytes = int(np.prod(shape)) * dt.itemsize
    else:
        nbytes = sum((s - 1) * st for s, st in zip(shape, strides)) + dt.itemsize
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    flat = np.frombuffer(buf, dtype=np.uint8)
    arr = np.lib.stride_tricks.as_strided(flat.view(dt) if nbytes % dt.itemsize == 0 else np.frombuffer(buf, dtype=dt, count=nbytes // dt.itemsize),
                                          shape=shape, strides=strides or None)
    t = torch.from_numpy(arr)
    t._hq_owner = obj  # the allocation lives as long as the tensor
    return t


class _Owner:
    """Frees an emulated device allocation when the last view of it is gone."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.hq_free(ctypes.c_void_p(self.ptr))
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


_LIB = []
#: "device" tensors of at least this many bytes made by torch.empty(..., device='cuda') live in EMULATED DEVICE MEMORY
#: (hq_alloc of the emulated library: a shared-memory object), so that they can be exported through the emulated HIP IPC to
#: the other ranks of a multi-process test -- the shard buffers of the peer-to-peer exchange
DEVICE_ALLOC_MIN_BYTES = 1 << 12


def _device_empty(torch, shape, dtype):
    import os
    if not _LIB:
        _LIB.append(ctypes.CDLL(os.environ['HQ_HIP_LIBRARY']))
        _LIB[0].hq_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64, ctypes.c_int]
        _LIB[0].hq_free.argtypes = [ctypes.c_void_p]
    lib = _LIB[0]
    itemsize = torch.empty((), dtype=dtype).element_size()
    numel = int(np.prod(shape))
    ptr = ctypes.c_void_p()
    if lib.hq_alloc(ctypes.byref(ptr), ctypes.c_uint64(numel * itemsize), 0) != 0:
        raise MemoryError('emulated hq_alloc failed')
    buf = (ctypes.c_char * (numel * itemsize)).from_address(ptr.value)
    buf._hq_owner = _Owner(lib, ptr.value)  # the numpy view below keeps `buf` alive, `buf` the allocation
    flat = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))
    return flat.view(dtype).reshape(shape)


def install():
    import torch
    if getattr(torch, '_hq_fake_cuda', False):
        return
    torch._hq_fake_cuda = True
    c = torch.cuda
    c.is_available = lambda: True
    c.device_count = lambda: 1
    c.current_device = lambda: 0
    c.set_device = lambda d: None
    c.synchronize = lambda *a, **k: None
    c.empty_cache = lambda: None
    c.mem_get_info = lambda *a, **k: (16 << 30, 16 << 30)
    c.memory_allocated = lambda *a, **k: 0
    c.max_memory_allocated = lambda *a, **k: 0
    c.reset_peak_memory_stats = lambda *a, **k: None
    c.current_stream = lambda *a, **k: _Stream()
    c.Stream = _Stream
    c.Event = _Event
    c.stream = lambda s: contextlib.nullcontext()
    c.device = lambda d: contextlib.nullcontext()

    def wrap_factory(name):
        orig = getattr(torch, name)

        def f(*args, **kw):
            on_device = 'device' in kw and str(kw['device']).startswith('cuda')
            if 'device' in kw:
                kw['device'] = _to_cpu_device(kw['device'])
            kw.pop('pin_memory', None)
            if name == 'empty' and on_device and kw.get('dtype') is not None and not kw.get('requires_grad'):
                shape = args[0] if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else args
                if all(isinstance(d, int) for d in shape):
                    nbytes = int(np.prod(shape)) * torch.empty((), dtype=kw['dtype']).element_size()
                    if nbytes >= DEVICE_ALLOC_MIN_BYTES:
                        return _device_empty(torch, tuple(shape), kw['dtype'])
            if name in ('as_tensor', 'tensor') and args and hasattr(args[0], '__cuda_array_interface__'):
                return _from_cuda_array_interface(args[0])
            return orig(*args, **kw)
        f.__name__ = name
        setattr(torch, name, f)

    for name in ('empty', 'zeros', 'ones', 'full', 'arange', 'randn', 'rand', 'tensor', 'as_tensor', 'empty_like', 'zeros_like',
                 'ones_like', 'randint', 'eye', 'linspace'):
        wrap_factory(name)
    orig_gen = torch.Generator

    def generator(device='cpu'):
        return orig_gen(device=_to_cpu_device(device))
    torch.Generator = generator
    T = torch.Tensor
    T.cuda = lambda self, *a, **k: self.clone()
    T.pin_memory = lambda self, *a, **k: self
    T.is_cuda = property(lambda self: True)
    orig_to = T.to

    def to(self, *args, **kw):
        args = list(args)
        moved = False
        if args and isinstance(args[0], (str, torch.device)) and not isinstance(args[0], torch.dtype):
            was = args[0]
            args[0] = _to_cpu_device(args[0])
            moved = str(was).startswith('cuda')
        if 'device' in kw:
            moved = moved or str(kw['device']).startswith('cuda')
            kw['device'] = _to_cpu_device(kw['device'])
        out = orig_to(self, *args, **kw)
        return out.clone() if moved and out.data_ptr() == self.data_ptr() else out  # a host -> device copy is a copy
    T.to = to
    orig_cpu = T.cpu
    T.cpu = lambda self, *a, **k: orig_cpu(self).clone()  # device -> host is a copy too
