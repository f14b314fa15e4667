"""Device-resident evolution driver: the counterpart of the ``optimize='evolution-hybridq'``
branch of ``_simulate_evolution`` (hybridq/circuit/simulation/simulation.py:372-781,
hot loop :464-678) and of the evolution arguments of ``simulate`` (:59-369).

Protocol kept from the reference
  * qubit labels are sorted; label #x sits at flat-index bit ``n-1-x`` (:512);
  * split real/imaginary planes, 2 x 2^n floats (:491-509);
  * per gate ``pos = [map[q] for q in reversed(gate.qubits)]`` (:633), ``U`` cast to the
    complex type in C order (:637), one ``apply_U`` call (:640-646);
  * ``info['runtime (s)']`` covers the gate loop only (:519,666), the planes are
    interleaved afterwards (:669-675).
Changed for the GPU
  * the state lives in HBM (torch tensors) for the whole loop, calls are asynchronous
    on one HIP stream and there is ONE synchronisation at the end of the loop;
  * the kernels accept any target position, so the reference's "swap the lowest 8
    index bits" policy (:559-630) and the final restore (:655-663) disappear: the
    logical->physical map stays the identity and no extra HBM pass is ever made.
There is no CPU fallback here: if the HIP library is missing the import of
``hybridq_amd.core`` has already failed.
"""
import time
from warnings import warn

import numpy as np

from . import core

_FLOAT_OF = {np.dtype('complex64'): np.dtype('float32'), np.dtype('complex128'): np.dtype('float64')}
MAX_GATE_QUBITS = 10  # kMaxK of the HIP core (apply_U rejects n_pos > 10; reference dot.py:236)


class FunctionalGate:
    """Minimal stand-in for ``hybridq.gate.property.FunctionalGate`` (property.py:732): a gate
    given as ``apply(psi, order) -> (new_psi, new_order)`` acting on the raw split-plane
    state ``psi`` of shape (2,) + (2,)*n (simulation.py:525-554).  Reference FunctionalGate
    objects (anything with ``.qubits`` and ``.apply`` but no ``.matrix``) are accepted too."""

    def __init__(self, qubits, apply, name='FN', on_device=False):
        """``on_device=True``: `apply` receives the state as a torch CUDA tensor of the same shape
        (2,)+(2,)*n -- a VIEW of the planes in HBM -- instead of a numpy copy, so that
        ``hybridq_amd.dot(U, psi, axes_b=..., b_as_complex_array=True, inplace=True)`` (the pattern of the
        reference's test_simulation_2__fn, tests.py:2037-2110) runs in HBM with no PCIe traffic."""
        self.qubits = tuple(qubits)
        self.apply = apply
        self.name = name
        self.on_device = bool(on_device)


def _is_functional(gate):
    if isinstance(gate, (tuple, list)):
        return False
    if callable(getattr(gate, 'apply_device', None)):  # hybridq_amd.functional: stays in HBM
        return True
    return callable(getattr(gate, 'apply', None)) and not hasattr(gate, 'matrix')


def _gate_qubits_matrix(gate):
    """Accept ``(U, qubits)`` pairs or reference-style gate objects exposing
    ``.qubits`` and ``.matrix()`` (``gate.provides(['qubits','matrix'])``, :556)."""
    if isinstance(gate, (tuple, list)) and len(gate) == 2:
        U, qs = gate
        return tuple(qs), np.asarray(U)
    if hasattr(gate, 'qubits') and hasattr(gate, 'matrix'):
        return tuple(gate.qubits), np.asarray(gate.matrix())
    raise RuntimeError(f"'{gate}' not supported")  # simulation.py:648-649


def flatten(circuit):
    """``utils.flatten`` (hybridq/circuit/utils.py:26-42, called at simulation.py:239): container gates -- anything that is
    not a ``(U, qubits)`` pair, provides ``flatten`` and iterates over its gates, the reference's TupleGate duck-typed --
    are replaced by their gates (nested containers too: a superset of the reference's one level)."""
    out = []
    for g in circuit:
        if not isinstance(g, (tuple, list, np.ndarray)) and callable(getattr(g, 'flatten', None)):
            try:
                inner = list(g)
            except TypeError:
                raise RuntimeError(f"'{g}' not supported")  # provides flatten but does not iterate over gates (simulation.py:648-649)
            out.extend(flatten(inner))
        else:
            out.append(g)
    return out


def all_qubits(circuit):
    """Sorted qubit labels (hybridq/circuit/circuit.py:406-451)."""
    qs = {q for g in circuit for q in ((g.qubits or ()) if _is_functional(g) else _gate_qubits_matrix(g)[0])}
    try:
        return sorted(qs)
    except TypeError:  # heterogeneous labels: order by (type name, value) like utils.sort
        return sorted(qs, key=lambda q: (type(q).__name__, q))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('hybridq_amd.simulate needs a HIP device (torch.cuda.is_available() is False)')
    return torch


#: bytes between the end of the re plane and the start of the im plane.  With im = re + 2^n
#: elements both planes map to the same HBM channel/bank for every index (power-of-two
#: distance) and the two streams of every kernel fight for row buffers; an offset of a few
#: 4 KiB pages removes that (measured at n=30, gpurun_out/sweep_pad*.txt: targets at positions
#: >= 22 run 3-7 % faster, pairs of high targets up to 15 %).
PLANE_PAD_BYTES = 12288


#: States of at least this many bytes (both planes) get a TUNED PLACEMENT from the library's own allocator
#: (hq_alloc_state, include/hq_hip.h): the planes are backed by HIP virtual-memory-management granules mapped by the
#: library instead of torch's caching allocator, several placements are drawn, each is probed with a handful of gate
#: applications, the fastest one is kept, and a freed winner stays in a per-size pool for the next state.  Why (MI355X,
#: n = 30, profiles/r02_placement_*.txt): the SAME gate kernels stream 5.50-5.58 TB/s from hipMalloc / torch memory and
#: 5.8-6.4 TB/s from 2-8 MiB granules mapped through the VMM interface, depending on which physical pages the driver hands
#: out.  The search, the probe and the pool live behind the C ABI (round 3; round 2 had them here in Python).
#: HQ_STATE_ALLOC=torch switches the mechanism off, HQ_STATE_TRIES sets the number of draws.
VMM_MIN_BYTES = 1 << 28
#: what the last tuned allocation found (bench.py reports it): hq_state_info's report
last_placement = {}


def alloc_planes(n, torch_dtype, device, vmm=True):
    """(2, 2^n) view of one allocation whose two rows are PLANE_PAD_BYTES further apart than
    2^n elements.  planes[0] / planes[1] are contiguous, 32-byte aligned 1-D tensors.  Large states come from
    hq_alloc_state with a tuned placement (VMM_MIN_BYTES above); small ones, and planes other ranks map through HIP IPC
    (``vmm=False``), from torch's caching allocator."""
    import os
    torch = _torch()
    itemsize = torch.empty((), dtype=torch_dtype).element_size()
    pad = PLANE_PAD_BYTES // itemsize if n >= 12 else 0
    # the row stride stays a multiple of 32 bytes for the tiniest states too (n <= 2 in complex64,
    # n <= 1 in complex128: 2^n elements alone are shorter than the alignment apply_U checks)
    stride = -(-((1 << n) + pad) * itemsize // 32) * 32 // itemsize
    nbytes = 2 * stride * itemsize
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if vmm and nbytes >= VMM_MIN_BYTES and os.environ.get('HQ_STATE_ALLOC', 'vmm') == 'vmm' and dev.index in (None, torch.cuda.current_device()):
        try:
            core.use_torch_stream()  # the probes run on the library stream
            owner = core.StatePlanes(n, np.dtype('float32') if itemsize == 4 else np.dtype('float64'))
            raw = torch.as_tensor(owner, device=dev)
            if raw.data_ptr() != owner.re or owner.stride != stride:
                raise RuntimeError('torch did not alias the planes of hq_alloc_state')
            last_placement.clear()
            last_placement.update(owner.info)
            return raw[:, :1 << n]
        except Exception as e:  # noqa: BLE001 -- the driver refused (fragmented HBM, VMM unavailable): torch's allocator
            warn(f'hybridq_amd: tuned state placement failed ({e!r}); using torch.empty')
            core.state_pool_trim()
    raw = device_empty((2, stride), dtype=torch_dtype, device=device)
    return raw[:, :1 << n]


def device_empty(shape, dtype, device=None):
    """``torch.empty`` on the device that gives the library's idle state pool back before it gives up: hq_free_state
    keeps one tuned placement per state size mapped (8-128 GiB at n = 30-34) where torch's caching allocator cannot see
    it, so a large torch allocation may fail while a pooled state sits idle."""
    torch = _torch()
    try:
        return torch.empty(shape, dtype=dtype, device=device)
    except torch.OutOfMemoryError:
        core.state_pool_trim()
        torch.cuda.empty_cache()
        return torch.empty(shape, dtype=dtype, device=device)


#: States of at least this many bytes are copied to the host in 128 MiB chunks through a ring of page-locked staging
#: buffers (1 GiB) while four threads move finished chunks into the (ordinary) numpy array: 0.19 s instead of 0.67 s for
#: an n = 30 complex64 state (`tensor.cpu()` is bound by one thread's first-touch copy at 12.5 GB/s).
CHUNKED_RETURN_MIN_BYTES = 256 << 20


def _to_host(dev_tensor, out=None):
    """Device tensor -> numpy array (a fresh, ordinary host array, or `out`: a contiguous array of the same size)."""
    torch = _torch()
    flat = dev_tensor.reshape(-1)
    nbytes = flat.numel() * flat.element_size()
    chunk = (128 << 20) // flat.element_size()
    plain = nbytes < CHUNKED_RETURN_MIN_BYTES or flat.numel() % chunk
    stage = None
    if not plain:
        try:
            stage = [torch.empty(chunk, dtype=flat.dtype, pin_memory=True) for _ in range(8)]
        except RuntimeError:  # no page-locked memory to be had: the ordinary copy
            plain = True
    if plain:
        host = dev_tensor.cpu().numpy()
        if out is None:
            return host
        out.reshape(-1)[:] = host.reshape(-1)
        return out
    from concurrent.futures import ThreadPoolExecutor
    depth = len(stage)
    res = np.empty(flat.numel(), dtype=stage[0].numpy().dtype) if out is None else out.reshape(-1)
    done = [torch.cuda.Event() for _ in range(depth)]
    futures = [None] * depth

    def drain(c, s):
        done[s].synchronize()
        res[c * chunk:(c + 1) * chunk] = stage[s].numpy()

    with ThreadPoolExecutor(4) as pool:
        for c in range(flat.numel() // chunk):
            s = c % depth
            if futures[s] is not None:
                futures[s].result()  # the staging buffer is free again
            stage[s].copy_(flat[c * chunk:(c + 1) * chunk], non_blocking=True)
            done[s].record()
            futures[s] = pool.submit(drain, c, s)
        for f in futures:
            if f is not None:
                f.result()
    return res.reshape(tuple(dev_tensor.shape)) if out is None else out


def _from_host(host, dev_tensor):
    """Contiguous numpy array -> (contiguous, 1-D) device tensor of the same size and dtype: the mirror of _to_host
    (threads fill pinned staging buffers, the copies to the device run asynchronously behind them)."""
    torch = _torch()
    flat = dev_tensor.reshape(-1)
    src = np.ascontiguousarray(host).reshape(-1)
    chunk = (128 << 20) // flat.element_size()
    stage = None
    if flat.numel() * flat.element_size() >= CHUNKED_RETURN_MIN_BYTES and flat.numel() % chunk == 0 and flat.is_contiguous():
        try:
            stage = [torch.empty(chunk, dtype=flat.dtype, pin_memory=True) for _ in range(8)]
        except RuntimeError:
            stage = None
    if stage is None:
        dev_tensor.copy_(torch.from_numpy(src).reshape(dev_tensor.shape))
        return
    from concurrent.futures import ThreadPoolExecutor
    depth = len(stage)
    sent = [torch.cuda.Event() for _ in range(depth)]
    used = [False] * depth
    nch = flat.numel() // chunk

    def fill(c, s):
        stage[s].numpy()[:] = src[c * chunk:(c + 1) * chunk]

    with ThreadPoolExecutor(4) as pool:
        pending = {}
        for c in range(min(depth, nch)):
            pending[c] = pool.submit(fill, c, c % depth)
        for c in range(nch):
            s = c % depth
            pending.pop(c).result()
            flat[c * chunk:(c + 1) * chunk].copy_(stage[s], non_blocking=True)
            sent[s].record()
            used[s] = True
            nxt = c + depth
            if nxt < nch:
                sent[s].synchronize()  # the staging buffer has left for the device
                pending[nxt] = pool.submit(fill, nxt, s)
    torch.cuda.current_stream().synchronize()


def prepare_state_planes(initial_state, n, float_type, device, placement='tuned'):
    """Planes for an initial state given as a '01+-' string (hybridq/circuit/simulation/
    utils.py:41-156) or as an array of 2^n amplitudes.  Strings are written by device kernels
    (basis, uniform, and the mixed '01+-' product state); arrays are uploaded.
    ``placement``: 'tuned' (VMM granules, best for the streaming per-gate kernels) or 'plain' (torch's
    allocator: the cache-blocked passes gather 128-byte runs from all over the state and run 7 % FASTER
    from it -- 156 vs 167 ms for the n = 30 benchmark circuit on the same box)."""
    torch = _torch()
    tdt = {np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64}[float_type]
    planes = alloc_planes(n, tdt, device, vmm=placement == 'tuned')
    if isinstance(initial_state, str):
        s = initial_state
        if len(s) == 1:
            s = s * n
        if len(s) != n:
            raise ValueError("'initial_state' has the wrong number of qubits.")
        if any(c not in '01+-' for c in s):
            raise ValueError("'initial_state' may contain only '0', '1', '+', '-'.")
        if all(c in '01' for c in s):
            core.init_state(planes[0], planes[1], 'basis', int(s, 2))  # label x <-> bit n-1-x
            return planes
        if all(c == '+' for c in s):
            core.init_state(planes[0], planes[1], 'plus')
            return planes
        if n >= 2:  # mixed '01+-': one device pass, no 2^n host array (label x <-> bit n-1-x)
            core.init_product_state(planes[0], planes[1], {n - 1 - x: c for x, c in enumerate(s)})
            return planes
        vec = np.ones(1, dtype=np.float64)
        single = {'0': [1, 0], '1': [0, 1], '+': [2**-0.5, 2**-0.5], '-': [2**-0.5, -2**-0.5]}
        for c in s:
            vec = np.kron(vec, np.asarray(single[c]))
        initial_state = vec
    psi = np.asarray(initial_state).reshape(-1)
    if psi.size != 1 << n:
        raise ValueError("'initial_state' has the wrong size.")
    ctype = np.dtype('complex64') if np.dtype(float_type) == np.dtype('float32') else np.dtype('complex128')
    if psi.size * ctype.itemsize >= CHUNKED_RETURN_MIN_BYTES:
        # large arrays: one chunked upload of the complex amplitudes, split into the planes on the device (the host
        # would spend longer extracting .real / .imag than the whole transfer takes)
        dev = device_empty(psi.size, dtype=torch.complex64 if ctype == np.dtype('complex64') else torch.complex128,
                          device=planes.device)
        _from_host(np.ascontiguousarray(psi, dtype=ctype), dev)
        planes[0].copy_(dev.real)
        planes[1].copy_(dev.imag)
        del dev
        return planes
    planes[0].copy_(torch.from_numpy(np.array(psi.real, dtype=float_type, order='C')))
    planes[1].copy_(torch.from_numpy(np.array(psi.imag, dtype=float_type, order='C')))
    return planes


def prepare_state(state, d=2, complex_type='complex64'):
    """``hybridq.circuit.simulation.prepare_state`` (circuit/simulation/utils.py:41-156): the '01+-' product state as a
    complex numpy array of shape (2,)*n.  Built where every state of this package is built -- by the device kernels
    behind :func:`prepare_state_planes` -- and brought back; only qubits (d = 2) exist here."""
    try:
        state = str(state)
    except Exception:  # noqa: BLE001
        raise ValueError("'state' must be convertible to 'str'.")
    try:
        dims = (int(d),) * len(state)
    except TypeError:
        dims = tuple(int(x) for x in d)
    if set(state).difference('+-01'):
        raise ValueError(f"Symbols {set(state).difference('+-01')} are not allowed.")
    if any(x <= 0 for x in dims):
        raise ValueError("All dimensions must be positive")
    if len(dims) != len(state):
        raise ValueError("Number of qubits and dimensions are not consistent.")
    if any(x != 2 for x in dims):
        raise ValueError("Only qubits of dimension 2 are supported.")
    if not state:
        raise ValueError("'state' is empty.")
    n = len(state)
    return EvolutionState(list(range(n)), complex_type=complex_type, initial_state=state, placement='plain').to_numpy().reshape((2,) * n)


class _LazySplitState:
    """What a FunctionalGate on NO qubits receives as `psi` (the reference's MessageGate, extras/gate/gate.py:25-27, prints
    and hands `psi` back): the (2,) + (2,)*n split array, fetched from the device only if the gate actually looks at it --
    at n = 30 a round trip of the state is 16 GiB over PCIe, per message.  Anything but identity / shape / dtype queries
    materialises the host array and behaves like it from then on."""

    def __init__(self, fetch, shape, dtype):
        self._fetch, self._host = fetch, None
        self.shape, self.dtype, self.ndim = tuple(shape), np.dtype(dtype), len(shape)

    def materialize(self):
        if self._host is None:
            self._host = self._fetch()
        return self._host

    @property
    def materialized(self):
        return self._host is not None

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        return self.materialize()[key]

    def __setitem__(self, key, value):
        self.materialize()[key] = value

    def __getattr__(self, name):  # every other ndarray attribute or method
        return getattr(self.materialize(), name)


def _apply_host_functional(gate, order, fetch, store, shape, float_type):
    """The host form of the FunctionalGate branch (simulation.py:525-554): ``gate.apply(psi, order)`` on the split array
    `fetch()` returns, the result written back with `store(array)`.  A gate on no qubits gets the lazy stand-in above and
    costs nothing unless it touches the state."""
    qubits = getattr(gate, 'qubits', None)
    lazy = _LazySplitState(fetch, shape, float_type) if qubits is not None and len(qubits) == 0 else None
    psi = lazy if lazy is not None else fetch()
    new_psi, new_order = gate.apply(psi=psi, order=order)
    if any(x != y for x, y in zip(order, new_order)):  # :552-554
        raise RuntimeError("'order' has changed.")
    if lazy is not None and new_psi is lazy:
        if not lazy.materialized:
            return  # the gate never looked: the state on the device is the state
        new_psi = lazy.materialize()  # looked at, maybe modified in place
    store(new_psi)


class EvolutionState:
    """Split-plane state vector resident in HBM plus the logical->physical qubit map."""

    def __init__(self, qubits, complex_type='complex64', initial_state='0', device=None, placement='tuned'):
        torch = _torch()
        self.complex_type = np.dtype(complex_type)
        if self.complex_type not in _FLOAT_OF:
            warn("optimize=evolution-hybridq only support ['complex64', 'complex128']. "
                 "Using 'complex64'.")  # simulation.py:467-471
            self.complex_type = np.dtype('complex64')
        self.float_type = _FLOAT_OF[self.complex_type]
        self.qubits = list(qubits)
        self.n = len(self.qubits)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.map = {q: self.n - x - 1 for x, q in enumerate(self.qubits)}  # simulation.py:512
        with torch.cuda.device(self.device):
            core.use_torch_stream()
            self.planes = prepare_state_planes(initial_state, self.n, self.float_type, self.device, placement)

    @property
    def planes(self):
        return self._planes

    @planes.setter
    def planes(self, value):
        # the two plane views are made once: indexing the (2, 2^n) tensor anew for every gate costs ~5 us of the ~11 us a
        # Python-level apply_U call takes, which is what bounds the loop below n ~ 22
        self._planes = value
        self._re, self._im = value[0], value[1]

    @property
    def re(self):
        return self._re

    @property
    def im(self):
        return self._im

    def apply(self, U, qubits):
        core.use_torch_stream()
        pos = [self.map[q] for q in reversed(qubits)]  # simulation.py:633
        core.apply_U(self._re, self._im, U, pos, self.n)

    def apply_functional(self, gate):
        """FunctionalGate branch of the loop (simulation.py:525-554): the gate receives the raw
        (2,)+(2,)*n real array and the current qubit order.  Reference functional gates are
        host numpy code, so for them the state makes a D2H/H2D round trip here; gates built with
        ``FunctionalGate(..., on_device=True)`` get a device VIEW of the planes instead, and the
        built-in Projection / Measure (hybridq_amd.functional) run as device kernels."""
        core.use_torch_stream()
        if callable(getattr(gate, 'apply_device', None)):
            gate.apply_device(self)
            return
        torch = _torch()
        order = tuple(q for q, _ in sorted(self.map.items(), key=lambda x: x[1])[::-1])  # :528-530
        if getattr(gate, 'on_device', False):
            view = self.planes.view((2,) + (2,) * self.n)  # aliases the planes: axis a+1 <-> qubit order[a]
            new_psi, new_order = gate.apply(psi=view, order=order)
            if any(x != y for x, y in zip(order, new_order)):  # :552-554
                raise RuntimeError("'order' has changed.")
            same = new_psi is view or (hasattr(new_psi, 'data_ptr') and new_psi.data_ptr() == view.data_ptr() and
                                       tuple(new_psi.shape) == tuple(view.shape) and tuple(new_psi.stride()) == tuple(view.stride()))
            if not same:
                # a new array came back (:545-550) -- or another VIEW of the same storage (permuted axes, a slice starting
                # at the same address): bring it into the planes (device to device)
                new_psi = torch.as_tensor(new_psi, device=self.planes.device).to(self.planes.dtype)
                self.planes.copy_(new_psi.reshape(2, -1))
            return
        def store(new_psi):
            new_psi = np.ascontiguousarray(new_psi, dtype=self.float_type).reshape(2, -1)
            for p in (0, 1):
                _from_host(new_psi[p], self.planes[p])
        _apply_host_functional(gate, order, self.to_split_array, store, (2,) + (2,) * self.n, self.float_type)

    def to_split_array(self):
        """The state on the host in the reference's own working layout: a real array of shape (2,) + (2,)*n, [0] = real
        parts, [1] = imaginary parts (simulation.py:490-497) -- what FunctionalGates receive and what the reference's
        ``simulate(..., return_numpy_array=False)`` returns."""
        core.use_torch_stream()
        core.sync()
        host = np.empty((2, 1 << self.n), dtype=self.float_type)
        for p in (0, 1):  # chunked, threaded copies for large states (see _to_host)
            _to_host(self.planes[p], out=host[p])
        return host.reshape((2,) + (2,) * self.n)

    def __array__(self, dtype=None, copy=None):
        """``np.asarray(state)``: the split array above, so that code written against the reference's
        ``return_numpy_array=False`` result (a (2,) + (2,)*n real array) keeps working on the device-resident state."""
        a = self.to_split_array()
        return a if dtype is None else a.astype(dtype, copy=False)

    def compile(self, circuit, compress=4, blocked=False):
        """Record `circuit` (matrix gates only) against THIS state's planes into a
        :class:`hybridq_amd.core.Program`: fusion / blocking are planned once, every operand
        table is uploaded once, and ``program.run()`` replays the launches with no host work
        per gate (one hipGraph launch from the second run on).  Useful when the same circuit
        is applied repeatedly (parameter sweeps at fixed structure re-compile; shots with
        different initial states only rewrite the planes) and at small n where one gate kernel
        (~10 us) is shorter than the ~11 us a Python-level call costs."""
        gates = _plan_ops(list(circuit), self.qubits, self.n, self.complex_type, compress, blocked)
        if any(_is_functional(g) for g in gates):
            raise ValueError('functional gates cannot be compiled')
        core.use_torch_stream()
        prog = core.Program()
        with prog:
            _execute_ops(self, gates)
        prog.keep_alive(self.planes)
        return prog

    def to_complex(self):
        """Interleave the planes into a complex torch tensor on the device (:669-675)."""
        torch = _torch()
        cdt = {np.dtype('complex64'): torch.complex64, np.dtype('complex128'): torch.complex128}[self.complex_type]
        core.use_torch_stream()
        out = device_empty(1 << self.n, dtype=cdt, device=self.device)
        core.to_complex(self.planes[0], self.planes[1], out)
        return out

    def to_numpy(self):
        """The state as ONE complex numpy array (the reference's return value, simulation.py:669-675).  Large states never
        exist as a 2^n complex tensor in HBM (round 3; at n = 34 planes + complex copy + scratch would not fit 288 GB):
        to_complex runs chunk by chunk -- 128 MiB of interleaved amplitudes at a time into a small ring of device
        staging buffers, each followed by its asynchronous copy into page-locked host memory, while four threads move
        finished chunks into the result array (the chunked return path of _to_host with the interleave fused in)."""
        torch = _torch()
        cdt = {np.dtype('complex64'): torch.complex64, np.dtype('complex128'): torch.complex128}[self.complex_type]
        size = 1 << self.n
        chunk = (128 << 20) // self.complex_type.itemsize
        stage = None
        if size * self.complex_type.itemsize >= CHUNKED_RETURN_MIN_BYTES and size % chunk == 0:
            try:
                stage = [torch.empty(chunk, dtype=cdt, pin_memory=True) for _ in range(8)]
            except RuntimeError:  # no page-locked memory to be had: the one-shot path
                stage = None
        if stage is None:
            out = self.to_complex()
            core.sync()
            return _to_host(out)
        from concurrent.futures import ThreadPoolExecutor
        core.use_torch_stream()
        depth = len(stage)
        dev = [torch.empty(chunk, dtype=cdt, device=self.device) for _ in range(4)]
        res = np.empty(size, dtype=self.complex_type)
        done = [torch.cuda.Event() for _ in range(depth)]
        futures = [None] * depth

        def drain(c, s):
            done[s].synchronize()
            res[c * chunk:(c + 1) * chunk] = stage[s].numpy()

        with ThreadPoolExecutor(4) as pool:
            for c in range(size // chunk):
                s, d = c % depth, c % len(dev)
                if futures[s] is not None:
                    futures[s].result()  # the staging buffer is free again (its device buffer was read 4 chunks ago)
                core.to_complex(self.planes[0][c * chunk:(c + 1) * chunk], self.planes[1][c * chunk:(c + 1) * chunk], dev[d])
                stage[s].copy_(dev[d], non_blocking=True)  # same stream: ordered behind the interleave kernel
                done[s].record()
                futures[s] = pool.submit(drain, c, s)
            for f in futures:
                if f is not None:
                    f.result()
        return res

    def norm2(self):
        core.use_torch_stream()
        return core.norm2(self.planes[0], self.planes[1])


def _functional_qubits(g):
    """Qubits a functional gate declares, or None (``Gate('fn', n_qubits=...)`` of the reference: qubits is None)."""
    qs = getattr(g, 'qubits', None)
    return None if qs is None else tuple(qs)


def _fusion_items(circuit):
    """The circuit as fusion.simplify / fusion.fuse take it: ``(U, qubits)`` pairs, FunctionalGates wrapped as fusion.Opaque."""
    from .fusion import Opaque
    items = []
    for g in circuit:
        if _is_functional(g):
            items.append(Opaque(g, _functional_qubits(g)))
        else:
            qs, U = _gate_qubits_matrix(g)
            items.append((U, qs))
    return items


def _simplify_runs(circuit, remove_id_gates, atol, opts):
    """``utils.simplify`` (fusion.simplify) on the whole circuit.  FunctionalGates take part the way they do in the
    reference's walk (insert_from_left, circuit/utils.py:166-208; fusion.Opaque): they slide past gates on other qubits and
    gates slide past them, nothing cancels against them, one without qubits stops everything.  The resulting gate list is
    the reference's, one for one (tests/test_reference_live_host.py)."""
    from .fusion import Opaque, simplify as _simplify
    if not any(_is_functional(g) for g in circuit):
        gates = _simplify([(U, qs) for qs, U in (_gate_qubits_matrix(g) for g in circuit)], atol=atol, remove_id_gates=remove_id_gates, **opts)
        return [(U, qs) for U, qs in gates]
    out = _simplify(_fusion_items(circuit), atol=atol, remove_id_gates=remove_id_gates, **{k: v for k, v in opts.items() if k != 'native'})
    return [g.obj if isinstance(g, Opaque) else g for g in out]


def _compress_args(compress):
    """(max_n_qubits, keyword arguments for fusion.fuse) of a `compress` argument, validated: the core applies gates of
    up to MAX_GATE_QUBITS qubits, and only these keys of the reference's utils.compress have a counterpart in fusion.fuse.
    ``atol`` is accepted and has no effect, exactly as in the reference: ``commutes_with`` compares with a fixed 1e-5
    whatever it is passed (hybridq/gate/property.py:573; fusion._COMMUTE_ATOL)."""
    comp_kw = {k: v for k, v in compress.items() if k != 'max_n_qubits'} if isinstance(compress, dict) else {}
    comp_n = compress.get('max_n_qubits', 4) if isinstance(compress, dict) else compress
    if comp_n and comp_n > MAX_GATE_QUBITS:
        raise ValueError(f"compress={comp_n}: fused gates are limited to {MAX_GATE_QUBITS} qubits by the HIP core")
    unknown = set(comp_kw) - {'use_matrix_commutation', 'max_n_qubits_matrix', 'atol', 'exclude_qubits'}
    if unknown:
        raise ValueError(f"unsupported 'compress' option(s) {sorted(unknown)}: FunctionalGates are never compressed "
                         "here (the reference's skip_compression default), other skip lists have no counterpart")
    return comp_n, comp_kw


def _plan_ops(circuit, qubits, n, ctype, compress, blocked, reference=True):
    """Turn a circuit into the op list the gate loop executes: fused ``(qubits, U)`` gates
    (simulation.py:436-454), or the 'B'/'G' ops of the cache-blocked planner.  FunctionalGates are
    never fused (skip_compression=[FunctionalGate], :441).  Under ``compress`` the walk is the reference's, functional
    gates included: a gate that shares no qubit with a FunctionalGate slides across it into an earlier layer
    (circuit/utils.py:630-648; fusion.Opaque) -- with non-unitary gates around a renormalising Projection that order
    is part of the RESULT, so it is reproduced exactly.  The cache-blocked planner (no reference counterpart) and
    compress=0 (nothing moves) keep the circuit cut at every FunctionalGate.  ``reference`` (the caller named the schedule:
    'evolution-hybridq' or an explicit ``compress=``): fused gates wider than 4 qubits also get the matrix the reference's
    ``to_matrix_gate`` computes (fusion._layer_matrix_like_reference); this driver's own schedules (choose_schedule) take the
    plain product from the native planner."""
    comp_n, comp_kw = _compress_args(compress)
    use_blocked = bool(blocked) and n >= 14
    pos_of = {q: n - x - 1 for x, q in enumerate(qubits)}  # simulation.py:512 (never permuted here)
    if comp_n and not use_blocked and any(_is_functional(g) for g in circuit):
        from .fusion import Opaque, fuse
        fused = fuse(_fusion_items(circuit), comp_n, complex_type=ctype, reference_matrices=reference, **{k: v for k, v in comp_kw.items() if k != 'native'})
        return [g.obj if isinstance(g, Opaque) else (g[1], g[0]) for g in fused]
    gates, run = [], []

    def flush():
        if not run:
            return
        if use_blocked:
            # many gates per HBM pass (hybridq_amd.blocking); inner fusion replaces `compress`
            from .blocking import plan_blocked
            opts = dict(blocked) if isinstance(blocked, dict) else {}
            opts.setdefault('tile_bits', 13 if ctype == np.dtype('complex64') else 12)  # 64 KiB of LDS
            opts.setdefault('low_bits', 5 if ctype == np.dtype('complex64') else 4)
            opts.setdefault('complex_type', ctype)
            gates.extend(plan_blocked([(U, qs) for qs, U in run], pos_of, n, **opts))
        elif comp_n:
            from .fusion import fuse
            gates.extend((qs, U) for U, qs in fuse([(U, qs) for qs, U in run], comp_n, complex_type=ctype, reference_matrices=reference, **comp_kw))
        else:
            gates.extend(run)
        run.clear()

    for g in circuit:
        if _is_functional(g):
            flush()
            gates.append(g)
        else:
            run.append(_gate_qubits_matrix(g))
    flush()
    return gates


#: Cost model of the schedules (ms per pass over a complex64 state of 2^30 amplitudes on MI355X, tuned
#: placement; profiles/r02_v1_bench.json `per_k`, `blocked`): a pass costs the same for every k <= 4, k = 5 is
#: HBM-bound with the matrix cores at 66 %, k >= 6 is matrix-core bound; a cache-blocked pass costs one HBM
#: round trip plus matrix-core time per inner gate.  Times scale with 2^n and with the element size.
PASS_MS = {1: 2.70, 2: 2.70, 3: 2.70, 4: 2.76, 5: 3.08, 6: 4.70, 7: 10.4, 8: 18.7, 9: 36.1, 10: 73.7}
BLOCKED_BASE_MS = 3.0  # a blocked pass whose gates hide behind the HBM stream (tools/blocked_scaling.py: G <= 2)
BLOCKED_OVERLAP_MS = 1.3  # ... what of the stream does NOT hide behind the gates once they dominate (complex64: the
#                            next tile is prefetched into registers; 0.95 ms for a tile on the 8 lowest bits, 1.3 ms
#                            fitted on the benchmark circuit's tiles; complex128 at n - 1 measures 15 % above the model)
BLOCKED_INNER_MS = {1: 0.38, 2: 0.63, 3: 0.63, 4: 1.20}
TUNED_PLACEMENT_SEARCH_MS = 2500.0  # what alloc_planes' draw-and-probe search costs (n = 30; measured 2.5 s)
#: host time of planning one candidate schedule, per matrix gate of the circuit (measured on the benchmark circuits,
#: n = 16..30: fusion to 4 ~0.04 ms / gate, to 5 ~0.07; the Python blocked planner needed ~0.13 at full search effort).
#: Round 4: the blocked planner runs behind the C ABI (hq_plan_blocked): 0.02 ms / gate at full search effort (17 ms for
#: the 900-gate n = 30 circuit; tools/plan_time.py), so the quick search of round 3 is gone
PLAN_HOST_MS_PER_GATE = {'per_gate': 0.0, 'fused_4': 0.006, 'fused_5': 0.011, 'blocked': 0.011}  # all three planners native (hq_plan_*)
BLOCKED_VS_FUSED5 = 0.55  # modelled time of the cache-blocked plan over the fused-5 plan (0.45 benchmark circuit, 0.63 dense 3q/4q gates)
PREDICTION_SLACK = 0.85  # a plan may come out this much better than predicted (commuting gates fuse further)
LAUNCH_FLOOR_MS = 0.011  # Python -> ctypes -> plan -> launch per call (profiles/r01_program_overhead.txt)


def estimate_ms(ops, n, ctype):
    """Modelled device time of an op list of _plan_ops (functional gates cost nothing here)."""
    scale = 2.0 ** (n - 30) * (2.0 if np.dtype(ctype) == np.dtype('complex128') else 1.0)
    t = 0.0
    for g in ops:
        if _is_functional(g):
            continue
        if isinstance(g[0], str):
            if g[0] == 'B':
                inner = sum(BLOCKED_INNER_MS[len(p)] for _, p in g[2])
                ms = max(BLOCKED_BASE_MS, BLOCKED_OVERLAP_MS + inner)
            else:
                ms = PASS_MS[len(g[2])]
        else:
            ms = PASS_MS[min(len(g[0]), 10)]
        t += max(ms * scale, LAUNCH_FLOOR_MS)  # small states: a launch costs more than the pass
    return t


def _predict_fused_ms(circuit, n, ctype, kmax):
    """What fusion to `kmax` qubits will cost, from the qubit sets alone (blocking._dry_layers: the grouping of fusion.fuse
    without matrix commutation, ~2 ms for 900 gates instead of 40-70): a prediction to decide what is worth planning."""
    from .blocking import _dry_layers
    scale = 2.0 ** (n - 30) * (2.0 if np.dtype(ctype) == np.dtype('complex128') else 1.0)
    run = [('F', None if _functional_qubits(g) is None else frozenset(_functional_qubits(g))) if _is_functional(g)
           else frozenset(_gate_qubits_matrix(g)[0]) for g in circuit]  # FunctionalGates: slid across like in the plans (fusion.Opaque)
    return sum(max(PASS_MS[min(len(layer), 10)] * scale, LAUNCH_FLOOR_MS) for layer in _dry_layers(run, kmax) if not isinstance(layer, tuple))


def choose_schedule(circuit, qubits, n, ctype):
    """Plan the circuit the ways this driver knows -- gate by gate, fused to 4 qubits (the reference's default,
    simulation.py:314), fused to 5, cache-blocked -- and keep the plan with the smallest modelled time.  Host work only,
    outside the timed loop like the reference's own compression (simulation.py:436-454 precede :519), but the caller
    waits for it all the same (at n = 30 planning all four takes 210 ms against a 137 ms loop), so a schedule is only
    planned when its PREDICTED device time plus its planning time beats the best plan in hand: the gate-by-gate plan is
    free, the fused ones are predicted from the qubit sets (exact unless gates commute), the cache-blocked one as
    BLOCKED_VS_FUSED5 of the fused-5 prediction.  Short loops (n <~ 24 of the benchmark circuit) therefore run gate by
    gate at once, n = 25 plans fusion to 4 only, n >= 26 the cache-blocked schedule only.  Returns (ops, info)."""
    cands = {'per_gate': dict(compress=0, blocked=False), 'fused_4': dict(compress=4, blocked=False),
             'fused_5': dict(compress=5, blocked=False)}
    n_matrix = sum(1 for g in circuit if not _is_functional(g))
    cost = {name: PLAN_HOST_MS_PER_GATE[name] * n_matrix for name in cands}
    if n >= 14:
        # full search effort (32 visiting orders and 16 fusion orders per pass): ~5 % less device time than the quick
        # search for 8 ms more host time now that the planner is native
        cands['blocked'] = dict(compress=5, blocked=True)
        cost['blocked'] = PLAN_HOST_MS_PER_GATE['blocked'] * n_matrix
    plans = {'per_gate': _plan_ops(circuit, qubits, n, ctype, 0, False, reference=False)}
    est = {'per_gate': estimate_ms(plans['per_gate'], n, ctype)}
    pred = {}
    if est['per_gate'] > min(c for name, c in cost.items() if name != 'per_gate'):  # else nothing can pay its planning back
        pred['fused_4'] = _predict_fused_ms(circuit, n, ctype, 4)
        pred['fused_5'] = _predict_fused_ms(circuit, n, ctype, 5)
        if 'blocked' in cands:
            pred['blocked'] = BLOCKED_VS_FUSED5 * pred['fused_5']
    # the cache-blocked planner is a randomised greedy search (+-3 % of modelled time over seeds): once 3 % of the predicted
    # loop outweighs three more planning runs (n >= 33 for the benchmark circuit) it plans with four seeds and keeps the best
    if 'blocked' in pred and 0.03 * pred['blocked'] > 3 * cost['blocked']:
        cands['blocked'] = dict(compress=5, blocked={'seeds': 4})
        cost['blocked'] *= 4
    for name in sorted(pred, key=lambda k: pred[k] + cost[k]):
        if cost[name] + PREDICTION_SLACK * pred[name] < min(est.values()):
            kw = cands[name]
            plans[name] = _plan_ops(circuit, qubits, n, ctype, kw['compress'], kw['blocked'], reference=False)
            est[name] = estimate_ms(plans[name], n, ctype)
    best = min(est, key=est.get)
    return plans[best], {'chosen': best, 'modelled_ms': {k: round(v, 4) for k, v in est.items()},
                         'passes': {k: sum(1 for g in v if not _is_functional(g)) for k, v in plans.items()},
                         'predicted_ms': {k: round(v, 4) for k, v in pred.items()},
                         'not_planned': [k for k in cands if k not in plans]}


#: schedules of the last PLAN_CACHE_SIZE distinct circuits (content-addressed; 0 switches the cache off)
PLAN_CACHE_SIZE = 16
_PLAN_CACHE = __import__('collections').OrderedDict()


def _plan_key(circuit, qubits, n, ctype, auto_schedule, compress, blocked):
    """Digest of everything a plan depends on, or None when the circuit holds objects a digest cannot speak for
    (FunctionalGates: user code)."""
    if PLAN_CACHE_SIZE <= 0:
        return None
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    h.update(repr((n, str(ctype), bool(auto_schedule), compress if not isinstance(compress, dict) else sorted(compress.items(), key=str),
                   blocked if not isinstance(blocked, dict) else sorted(blocked.items(), key=str), tuple(qubits),
                   sorted(PLAN_HOST_MS_PER_GATE.items()) if auto_schedule else None)).encode())
    for g in circuit:
        if _is_functional(g):
            return None
        qs, U = _gate_qubits_matrix(g)
        U = np.ascontiguousarray(U)
        h.update(repr((qs, U.dtype.str, U.shape)).encode())
        h.update(U.tobytes())
    return h.digest()


def _own_matrices(gates):
    """The op list with every matrix a private read-only copy.  A plan may carry the CALLER's arrays (compress=0, the
    gate-by-gate plan of the auto-schedule, unfused inner gates of a cache-blocked pass, `gate.matrix()` objects that
    hand out an internal buffer): kept by content digest, an in-place update of those arrays between two calls (a
    parameter scan) would leave new values under the old key."""
    def own(U):
        U = np.array(U, copy=True)
        U.setflags(write=False)
        return U
    out = []
    for g in gates:  # no FunctionalGates here: _plan_key returns None for such circuits
        if isinstance(g[0], str):
            out.append(('B', g[1], [(own(U), p) for U, p in g[2]]) + tuple(g[3:]) if g[0] == 'B' else (g[0], own(g[1]), g[2]) + tuple(g[3:]))
        else:
            out.append((g[0], own(g[1])) + tuple(g[2:]))
    return out


def _execute_ops(state, gates):
    """The gate loop (simulation.py:522-646); returns the number of passes over the state."""
    n = state.n
    n_passes = 0
    core.use_torch_stream()  # once: nothing in the loop but a FunctionalGate (user code) can change torch's current stream
    re, im, qmap, apply_U = state.re, state.im, state.map, core.apply_U
    for g in gates:
        if _is_functional(g):
            state.apply_functional(g)
            core.use_torch_stream()
        elif isinstance(g[0], str):  # ops of the blocked planner, positions already physical
            n_passes += 1
            if g[0] == 'B':
                core.apply_blocked(re, im, g[1], g[2], n)
            else:
                apply_U(re, im, g[1], g[2], n)
        else:
            n_passes += 1
            apply_U(re, im, g[1], [qmap[q] for q in reversed(g[0])], n)  # simulation.py:633
    return n_passes


def simulate(circuit, initial_state=None, final_state=None, optimize='evolution', backend='numpy',
             complex_type='complex64', tensor_only=False, simplify=True, remove_id_gates=True,
             use_mpi=None, atol=1e-8, verbose=False, **kwargs):
    """Evolution-path subset of ``hybridq.circuit.simulation.simulate`` (simulation.py:59).

    `circuit`: iterable of ``(U, qubits)`` or of objects with ``.qubits``/``.matrix()``.
    Only ``optimize in ('evolution', 'evolution-hybridq')`` exists here (plus
    ``'evolution-hip'``: same path with this GPU's fastest settings, ``blocked=True`` /
    ``compress=5``, as defaults); everything the reference routes elsewhere (einsum, tensor
    networks, Clifford) is out of scope.
    Supported kwargs: ``allow_sampling`` / ``sampling_seed`` (stochastic gates = objects with ``.sample()``, as
    at simulation.py:241-256), ``return_info``, ``return_numpy_array`` (default True; False returns the device-resident
    :class:`EvolutionState`, whose ``np.asarray()`` is the reference's (2,) + (2,)*n split array),
    ``max_largest_intermediate`` (default 2**36 amplitudes: one MI355X holds n=34 in
    complex64), ``compress`` (max qubits of a fused gate, default 4 like simulation.py:314;
    0 applies the gates as given; a dict may carry ``max_n_qubits`` plus the keyword
    arguments of ``fusion.fuse``), ``device``, ``devices`` / ``shard_bits`` (N = 2^g ranks of the current
    torch.distributed job hold one shard each: hybridq_amd.dist; strings as initial state, no
    FunctionalGates), ``blocked`` (default False; True or a dict of
    ``blocking.plan_blocked`` options: apply many gates per HBM pass through LDS tiles,
    n >= 14 -- see hybridq_amd/blocking.py).  ``simplify`` / ``remove_id_gates`` / ``atol`` as in the
    reference (simulation.py:290-305): identity gates are dropped and `fusion.simplify` (the
    counterpart of circuit/utils.py:825) slides commuting gates and cancels inverse pairs before
    fusion; a dict passes ``use_matrix_commutation`` / ``max_n_qubits_matrix`` on.
    """
    if isinstance(optimize, str) and optimize.startswith('evolution-einsum'):
        # the reference's numpy.einsum engine for the same evolution (simulation.py:676-760; its own tests ask for it,
        # tests.py:2098-2102): no separate engine here -- the HIP core evaluates it, with the reference core's schedule
        warn(f"optimize={optimize!r}: hybridq_amd has one evolution engine; running on the HIP core.")
        optimize = 'evolution-hybridq'
    if optimize not in ('evolution', 'evolution-hybridq', 'evolution-hip'):
        raise ValueError(f"hybridq_amd only implements optimize='evolution' (got {optimize!r})")
    if tensor_only:  # simulation.py:226-228
        raise ValueError(f"'tensor_only' is not support for optimize={optimize}")
    if use_mpi:  # simulation.py:379-380 warns and carries on: so does this driver (sharding is `devices=`, not MPI)
        warn("Detected MPI but optimize='evolution' does not support MPI.")
    kwargs.setdefault('return_info', False)
    kwargs.setdefault('return_numpy_array', True)
    kwargs.setdefault('max_largest_intermediate', 2**36)
    # 'evolution' is the reference's "best evolution engine available" alias (simulation.py:379-399 picks between
    # its C++ core and einsum); here it -- and the explicit 'evolution-hip' -- means this GPU's own schedule: the cost
    # model below picks between gate-by-gate passes, fusion to width 4 / 5 (a k = 5 pass costs ~15 % more than a
    # k <= 4 pass) and cache-blocked passes (many gates per HBM pass), and `info['schedule']` records the choice.
    # Explicit `compress=` / `blocked=` override it; 'evolution-hybridq' keeps the reference core's own schedule
    # (fusion to 4 qubits, one pass per fused gate).
    auto_schedule = optimize in ('evolution', 'evolution-hip') and 'blocked' not in kwargs and 'compress' not in kwargs
    kwargs.setdefault('compress', 4)  # simulation.py:314
    kwargs.setdefault('device', None)
    if final_state is not None:  # simulation.py:415-418
        warn("'final_state' cannot be specified in optimize='evolution'. Ignoring 'final_state'.")
    if initial_state is None:  # simulation.py:421-423
        raise ValueError("'initial_state' must be specified for optimize='evolution'.")

    circuit = flatten(circuit)  # simulation.py:239
    # Stochastic gates (simulation.py:241-256): anything with a ``.sample()`` method -- the reference's
    # ``StochasticGate`` duck-typed -- is replaced by one draw when `allow_sampling` is set; `sampling_seed`
    # seeds numpy's global generator for the draws and the previous state is restored afterwards.
    if kwargs.pop('allow_sampling', False):
        seed = kwargs.pop('sampling_seed', None)
        saved = np.random.get_state() if seed is not None else None
        if seed is not None:
            np.random.seed(int(seed))
        circuit = [g.sample() if callable(getattr(g, 'sample', None)) and not isinstance(g, (tuple, list)) else g for g in circuit]
        if saved is not None:
            np.random.set_state(saved)
    else:
        kwargs.pop('sampling_seed', None)
    qubits = kwargs.get('qubits') or all_qubits(circuit)
    n = len(qubits)
    n_given = len(circuit)
    if remove_id_gates:  # simulation.py:289-291 (named identity gates; matrix identities go in simplify)
        circuit = [g for g in circuit if getattr(g, 'name', None) != 'I']
    host_ms = {}  # what the caller waits for before the gate loop starts (reported in info['host_ms'])
    if simplify:  # simulation.py:293-305
        from .fusion import single_thread_blas
        _t = time.perf_counter()
        with single_thread_blas():
            circuit = _simplify_runs(circuit, remove_id_gates, atol, simplify if isinstance(simplify, dict) else {})
        host_ms['simplify'] = 1e3 * (time.perf_counter() - _t)
        if not kwargs.get('qubits') and all_qubits(circuit) != qubits:
            raise ValueError("Active qubits have changed after simplification. Forcing stop.")
    if not isinstance(initial_state, str):  # simulation.py:270-281 (a flat vector of 2^n amplitudes is accepted as well)
        shape = np.shape(initial_state)
        if len(shape) > 1 and any(x != 2 for x in shape):
            raise ValueError("Only qubits of dimension 2 are supported.")
        if len(shape) > 1 and len(shape) != n:
            raise ValueError("Wrong number of qubits for initial/final state.")
    elif len(initial_state) not in (1, n):  # simulation.py:264-268
        raise ValueError("Wrong number of qubits for initial/final state.")
    if 2**n > kwargs['max_largest_intermediate']:  # simulation.py:409-412, before any planning work
        raise MemoryError("Memory for the given number of qubits exceeds the 'max_largest_intermediate'.")
    ctype = np.dtype(complex_type) if np.dtype(complex_type) in _FLOAT_OF else np.dtype('complex64')
    # Compress circuit (simulation.py:436-454); untimed, like in the reference (:519)
    schedule_info = None
    from .fusion import single_thread_blas
    _compress_args(kwargs['compress'])  # argument checks, before anything else
    if _wants_shards(kwargs):  # the sharded driver plans for its own local qubit count
        _torch()
        return _simulate_sharded(circuit, qubits, n, ctype, initial_state, kwargs, auto_schedule)
    _t = time.perf_counter()
    # plans are kept by circuit content (qubits + matrices + options): a sampling loop or a parameter scan that comes back
    # with the same circuit pays for its schedule once
    key = _plan_key(circuit, qubits, n, ctype, auto_schedule, kwargs['compress'], kwargs.get('blocked', False))
    hit = _PLAN_CACHE.get(key) if key is not None else None
    if hit is not None:
        _PLAN_CACHE.move_to_end(key)
        gates, schedule_info = hit[0], (dict(hit[1], from_cache=True) if hit[1] is not None else None)
    else:
        with single_thread_blas():  # thousands of tiny matrix products: a threaded BLAS only adds wake-ups
            if auto_schedule:
                gates, schedule_info = choose_schedule(circuit, qubits, n, ctype)
            else:
                gates = _plan_ops(circuit, qubits, n, ctype, kwargs['compress'], kwargs.get('blocked', False))
        if key is not None:
            _PLAN_CACHE[key] = (_own_matrices(gates), dict(schedule_info) if schedule_info is not None else None)
            while len(_PLAN_CACHE) > PLAN_CACHE_SIZE:
                _PLAN_CACHE.popitem(last=False)
    host_ms['plan'] = 1e3 * (time.perf_counter() - _t)
    if schedule_info is not None:
        schedule_info['plan_ms'] = round(host_ms['plan'], 3)
    _torch()
    # Placement of the state: the tuned (VMM draw-and-probe) placement speeds the streaming kernels up by ~10 % but
    # the search itself costs ~2.5 s at n = 30 (8 draws): only worth it when the modelled loop is long enough to
    # win that back; the cache-blocked passes prefer the plain allocator anyway (prepare_state_planes).
    n_blocked = sum(1 for g in gates if not _is_functional(g) and isinstance(g[0], str) and g[0] == 'B')
    tuned = 2 * n_blocked <= len(gates) and 0.1 * estimate_ms(gates, n, ctype) > TUNED_PLACEMENT_SEARCH_MS
    state = EvolutionState(qubits, complex_type=complex_type, initial_state=initial_state,
                           device=kwargs['device'], placement='tuned' if tuned else 'plain')
    info = {}
    core.sync()
    t0 = time.perf_counter()  # simulation.py:519
    n_passes = _execute_ops(state, gates)
    core.sync()  # the ONLY synchronisation of the loop
    t1 = time.perf_counter()  # simulation.py:666
    info['runtime (s)'] = t1 - t0
    info['host_ms'] = {k: round(v, 3) for k, v in host_ms.items()}
    if schedule_info is not None:
        info['schedule'] = schedule_info
    info['n_gates'] = len(gates)  # apply_U calls issued (after fusion)
    info['n_passes'] = n_passes  # passes over the state (blocked passes count once)
    info['n_gates_given'] = n_given
    info['n_qubits'] = n

    if kwargs['return_numpy_array']:
        psi = state.to_numpy().reshape((2,) * n)
    else:
        psi = state
    return (psi, info) if kwargs['return_info'] else psi


def _wants_shards(kwargs):
    """``devices=N`` / ``shard_bits=g`` (N = 2^g): run on the N ranks of the current
    torch.distributed job (one process per GPU).  Without either argument a job with more than one
    rank still runs REPLICATED -- every rank its own full state -- like any single-GPU call."""
    devices, bits = kwargs.get('devices'), kwargs.get('shard_bits')
    if devices is None and bits is None:
        return False
    want = int(devices) if devices is not None else 1 << int(bits)
    if bits is not None and devices is not None and want != 1 << int(bits):
        raise ValueError("'devices' and 'shard_bits' disagree")
    if want == 1:
        return False
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world != want:
        raise RuntimeError(f'devices={want} needs a torch.distributed job of {want} ranks, one process per GPU '
                           f'(python -m torch.distributed.run --nproc-per-node {want} ...); this job has {world}')
    return True


def _simulate_sharded(circuit, qubits, n, ctype, initial_state, kwargs, auto_schedule=False):
    """The sharded counterpart of the gate loop: the state is split by its top log2(world) index
    bits over the ranks (hybridq_amd.dist), gates on global qubits are preceded by an exchange
    (hq_exchange_*), the canonical qubit order is restored at the end like the reference's final
    un-permute (simulation.py:655-663).  Returns, on every rank, the full state as a numpy array
    (``return_numpy_array=True``: small states only) or the :class:`ShardedEvolution` itself."""
    from .dist import ShardedEvolution
    if any(_is_functional(g) for g in circuit):
        raise NotImplementedError('FunctionalGates are not supported on a sharded state')
    if not isinstance(initial_state, str):
        raise NotImplementedError("sharded states start from a '01+-' string")
    gates = [(U, qs) for qs, U in (_gate_qubits_matrix(g) for g in circuit)]
    compress = kwargs['compress']
    comp_n = compress.get('max_n_qubits', 4) if isinstance(compress, dict) else compress
    # one-shot call: the placement search of the shard buffers (seconds per buffer) is not won back by one circuit
    from .dist import HipBackend
    sh = ShardedEvolution(n, complex_type=ctype, initial_state=initial_state, qubits=qubits,
                          backend=HipBackend(_FLOAT_OF[np.dtype(ctype)], placement='plain'))
    # auto schedule: cache-blocked local passes between the exchanges (2.7x the fused stream on one GPU)
    blocked = kwargs.get('blocked', bool(auto_schedule) and sh.m >= 14)
    from .fusion import single_thread_blas
    with single_thread_blas():
        sched = sh.plan(gates, compress=comp_n or 0, blocked=blocked)
    info = {}
    if auto_schedule:
        info['schedule'] = {'chosen': 'blocked' if blocked else f'fused_{comp_n or 0}', 'local_qubits': sh.m}
    sh.backend.sync()
    t0 = time.perf_counter()
    sh.run(sched)
    sh.restore_order()
    sh.backend.sync()
    info['runtime (s)'] = time.perf_counter() - t0
    info['n_gates'] = sum(1 for op in sched if op[0] in ('G', 'B'))
    info['n_passes'] = info['n_gates']
    info['n_exchanges'] = sum(1 for op in sched if op[0] in ('X', 'XP', 'XO'))
    info['n_gates_given'] = len(circuit)
    info['n_qubits'] = n
    info['n_ranks'] = sh.world
    info['exchange_transport'] = getattr(sh.backend, 'transport', None)
    if kwargs['return_numpy_array']:
        psi = sh.state_numpy().reshape((2,) * n).astype(ctype, copy=False)
    else:
        psi = sh
    return (psi, info) if kwargs['return_info'] else psi


def expectation_value(state, op, qubits_order, complex_type='complex64', **kwargs):
    """<state| op |state> through the evolution core: the counterpart of
    ``hybridq.circuit.simulation.expectation_value`` (simulation.py:1125-1216).  `state` is an
    array of shape (2,)*n, `op` a circuit on a subset of `qubits_order`.  As in the reference
    the qubits are mapped to the axes of `state` in SORTED label order (its ``simulate`` sorts
    ``all_qubits()``; ``qubits_order`` is only validated).  op|state> is formed in HBM and the
    inner product is reduced on the device: no state ever returns to the host."""
    state = np.asarray(state)
    n = state.ndim
    qubits_order = list(qubits_order)
    if len(qubits_order) != n:
        raise ValueError("'qubits_order' must have the same number of qubits of 'state'.")
    op = list(op)
    if set(all_qubits(op)).difference(qubits_order):
        raise ValueError("'op' has qubits not included in 'qubits_order'.")
    kwargs.pop('remove_id_gates', None)
    kwargs['return_numpy_array'] = False
    kwargs.pop('return_info', None)
    qubits = all_qubits([(None, (q,)) for q in qubits_order])
    out = simulate(op, initial_state=state, complex_type=complex_type, qubits=qubits, **kwargs)
    ref = EvolutionState(qubits, complex_type=complex_type, initial_state=state, device=out.device)
    val = core.vdot(ref.planes[0], ref.planes[1], out.planes[0], out.planes[1])
    return val.real if abs(val.imag) < 100 * np.finfo(np.float64).eps * max(1.0, abs(val.real)) else val
