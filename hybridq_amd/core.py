"""ctypes binding of libhq_hip.so -- the counterpart of the reference's L2 layer
(hybridq/utils/dot.py:26-71, hybridq/utils/transpose.py:25-58 and the loader
hybridq/utils/utils.py:534-553).

The module-level names mirror the reference so that code written against it reads the
same: ``_log2_pack_size``, ``_dot_core[float dtype]``, ``_to_complex_core[complex
dtype]``, ``_swap_core[dtype]``.  Unlike the reference there is NO silent fallback: if
the HIP library cannot be loaded, importing this module raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'csrc', 'libhq_hip.so')


def load_library(path=None):
    """Load libhq_hip.so.  Search order mirrors hybridq/utils/utils.py:534-553 (bare
    name first so LD_LIBRARY_PATH wins) after the in-tree build."""
    # One HIP runtime per process: torch wheels bundle their own libamdhip64/libhsa-runtime64.
    # Importing torch FIRST makes the dynamic linker resolve this library's libamdhip64.so.7
    # dependency to the copy torch already mapped; loading ours first would put a second
    # runtime (/opt/rocm) in the process that cannot see torch's allocations.
    try:
        import torch  # noqa: F401
    except ImportError:  # standalone use through ctypes (host-pointer path) is still valid
        pass
    cands = [path] if path else [os.environ.get('HQ_HIP_LIBRARY'), _LIB_PATH, 'libhq_hip.so']
    errors = []
    for c in cands:
        if not c:
            continue
        try:
            return ctypes.CDLL(c)
        except OSError as e:  # keep looking
            errors.append(f'{c}: {e}')
    raise ImportError('hybridq_amd: cannot load the HIP core libhq_hip.so (build it with '
                      '`python -m hybridq_amd.build`). Tried:\n  ' + '\n  '.join(errors))


def _define_function(lib, fname, restype, *argtypes):
    func = getattr(lib, fname)
    func.argtypes = argtypes
    func.restype = restype
    return func


_lib = load_library()

_c_types_map = {
    np.dtype('float32'): ctypes.c_float,
    np.dtype('float64'): ctypes.c_double,
    np.dtype('int32'): ctypes.c_int32,
    np.dtype('int64'): ctypes.c_int64,
    np.dtype('uint32'): ctypes.c_uint32,
    np.dtype('uint64'): ctypes.c_uint64,
}

# --- reference boundary (same shapes as dot.py:49-71 / transpose.py:52-58) ----------
_get_log2_pack_size = _define_function(_lib, 'get_log2_pack_size', ctypes.c_uint32)

_dot_core = {
    np.dtype(f'float{b}'): _define_function(_lib, f'apply_U_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint,
                                            ctypes.c_uint) for b in (32, 64)
}

_to_complex_core = {
    np.dtype(f'complex{2 * b}'): _define_function(_lib, f'hq_to_complex{2 * b}', ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_uint64) for b in (32, 64)
}

_swap_core = {
    dt: _define_function(_lib, f'swap_{dt.name}', ctypes.c_int, ctypes.c_void_p,
                         ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint, ctypes.c_uint)
    for dt in _c_types_map
}

# --- extensions ------------------------------------------------------------------
_set_stream = _define_function(_lib, 'hq_set_stream', ctypes.c_int, ctypes.c_void_p)
_sync = _define_function(_lib, 'hq_sync', ctypes.c_int)
_last_error = _define_function(_lib, 'hq_last_error', ctypes.c_char_p)
_last_kernel = _define_function(_lib, 'hq_last_kernel', ctypes.c_char_p)
_last_kernel_desc = _define_function(_lib, 'hq_last_kernel_desc', ctypes.c_char_p)
_device_count = _define_function(_lib, 'hq_device_count', ctypes.c_int)
_set_apply_mode = _define_function(_lib, 'hq_set_apply_mode', ctypes.c_int, ctypes.c_char_p)
_set_log2_pack_size = _define_function(_lib, 'hq_set_log2_pack_size', ctypes.c_int, ctypes.c_uint)
_init_state = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_init_state_float{b}', ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                            ctypes.c_int, ctypes.c_uint64) for b in (32, 64)
}
_init_product = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_init_product_state_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint) for b in (32, 64)
}
_norm2 = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_norm2_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.POINTER(ctypes.c_double)) for b in (32, 64)
}

_permute_bits = {
    b: _define_function(_lib, f'hq_permute_bits_{8 * b}', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint) for b in (4, 8)
}

_probabilities = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_probabilities_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.c_uint, ctypes.POINTER(ctypes.c_double)) for b in (32, 64)
}
_project = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_project_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.c_uint, ctypes.c_uint64, ctypes.c_double) for b in (32, 64)
}

_vdot = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_vdot_float{b}', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.POINTER(ctypes.c_double)) for b in (32, 64)
}

_apply_blocked = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_apply_blocked_float{b}', ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32))
    for b in (32, 64)
}

_shard_unique_id = _define_function(_lib, 'hq_shard_unique_id', ctypes.c_int, ctypes.c_void_p)
_shard_load_rccl = _define_function(_lib, 'hq_shard_load_rccl', ctypes.c_int)
_shard_init_rccl = _define_function(_lib, 'hq_shard_init_rccl', ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p)
_shard_attach_rccl = _define_function(_lib, 'hq_shard_attach_rccl', ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint)
_shard_init_p2p = _define_function(_lib, 'hq_shard_init_p2p', ctypes.c_int, ctypes.c_uint, ctypes.c_uint)
_shard_p2p_register = _define_function(_lib, 'hq_shard_p2p_register', ctypes.c_int, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_void_p))
_shard_info = _define_function(_lib, 'hq_shard_info', ctypes.c_int, ctypes.POINTER(ctypes.c_uint),
                               ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_int))
_shard_free = _define_function(_lib, 'hq_shard_free', ctypes.c_int)
_shard_rccl_selftest = _define_function(_lib, 'hq_shard_rccl_selftest', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_uint64)
_ipc_export = _define_function(_lib, 'hq_ipc_export', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.POINTER(ctypes.c_uint64))
_ipc_open = _define_function(_lib, 'hq_ipc_open', ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                             ctypes.POINTER(ctypes.c_void_p))
_ipc_close = _define_function(_lib, 'hq_ipc_close', ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64)
_exchange = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_exchange_float{b}', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int))
    for b in (32, 64)
}
_exchange_rounds = {
    np.dtype(f'float{b}'): _define_function(_lib, f'hq_exchange_rounds_float{b}', ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint))
    for b in (32, 64)
}
_exchange_round_wait = _define_function(_lib, 'hq_exchange_round_wait', ctypes.c_int, ctypes.c_uint)

_alloc = _define_function(_lib, 'hq_alloc', ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64, ctypes.c_int)
_free = _define_function(_lib, 'hq_free', ctypes.c_int, ctypes.c_void_p)
_alloc_mapped = _define_function(_lib, 'hq_alloc_mapped', ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64,
                                 ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64))
_alloc_scattered = _define_function(_lib, 'hq_alloc_scattered', ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64,
                                    ctypes.c_uint64, ctypes.c_uint64)

_alloc_state = _define_function(_lib, 'hq_alloc_state', ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p))
_free_state = _define_function(_lib, 'hq_free_state', ctypes.c_int, ctypes.c_void_p)
_state_info = _define_function(_lib, 'hq_state_info', ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64)
_state_pool_trim = _define_function(_lib, 'hq_state_pool_trim', ctypes.c_int)

_program_begin = _define_function(_lib, 'hq_program_begin', ctypes.c_int)
_program_end = _define_function(_lib, 'hq_program_end', ctypes.c_int, ctypes.POINTER(ctypes.c_void_p))
_program_size = _define_function(_lib, 'hq_program_size', ctypes.c_int, ctypes.c_void_p)
_program_run = _define_function(_lib, 'hq_program_run', ctypes.c_int, ctypes.c_void_p)
_program_free = _define_function(_lib, 'hq_program_free', ctypes.c_int, ctypes.c_void_p)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_plan_blocked = _define_function(_lib, 'hq_plan_blocked', ctypes.c_int, ctypes.c_uint, ctypes.c_uint, _u32p, _u32p, ctypes.c_void_p,
                                 ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                 ctypes.c_uint, ctypes.c_uint64, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p))
_plan_counts = _define_function(_lib, 'hq_plan_counts', ctypes.c_int, ctypes.c_void_p, _u32p, _u32p, ctypes.POINTER(ctypes.c_uint64),
                                ctypes.POINTER(ctypes.c_uint64), _u32p)
_plan_read = _define_function(_lib, 'hq_plan_read', ctypes.c_int, ctypes.c_void_p, _u32p, _u32p, _u32p, _u32p, _u32p, ctypes.c_void_p)
_plan_free = _define_function(_lib, 'hq_plan_free', ctypes.c_int, ctypes.c_void_p)
_plan_fuse = _define_function(_lib, 'hq_plan_fuse', ctypes.c_int, ctypes.c_uint, ctypes.c_uint, _u32p, _u32p, ctypes.c_void_p, ctypes.c_uint,
                              ctypes.c_int, ctypes.c_uint, ctypes.c_uint64, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p))
_plan_simplify = _define_function(_lib, 'hq_plan_simplify', ctypes.c_int, ctypes.c_uint, ctypes.c_uint, _u32p, _u32p, ctypes.c_void_p,
                                  ctypes.c_double, ctypes.c_int, ctypes.c_uint, ctypes.c_int, _u32p, _u32p)

#: symbols the C header include/hq_hip.h declares (checked by tests/test_abi.py)
EXPORTED = [
    'get_log2_pack_size', 'apply_U_float32', 'apply_U_float64', 'to_complex64', 'to_complex128',
    'swap_float32', 'swap_float64', 'swap_int32', 'swap_int64', 'swap_uint32', 'swap_uint64',
    'hq_set_stream', 'hq_sync', 'hq_set_log2_pack_size', 'hq_last_error', 'hq_device_count',
    'hq_set_apply_mode', 'hq_last_kernel', 'hq_last_kernel_desc', 'hq_to_complex64', 'hq_to_complex128',
    'hq_init_state_float32', 'hq_init_state_float64', 'hq_norm2_float32', 'hq_norm2_float64',
    'hq_init_product_state_float32', 'hq_init_product_state_float64',
    'hq_permute_bits_32', 'hq_permute_bits_64',
    'hq_probabilities_float32', 'hq_probabilities_float64', 'hq_project_float32', 'hq_project_float64',
    'hq_vdot_float32', 'hq_vdot_float64', 'hq_apply_blocked_float32', 'hq_apply_blocked_float64',
    'hq_program_begin', 'hq_program_end', 'hq_program_size', 'hq_program_run', 'hq_program_free',
    'hq_shard_unique_id', 'hq_shard_load_rccl', 'hq_shard_init_rccl', 'hq_shard_attach_rccl', 'hq_shard_init_p2p', 'hq_shard_p2p_register',
    'hq_shard_info', 'hq_shard_free', 'hq_shard_rccl_selftest', 'hq_ipc_export', 'hq_ipc_open', 'hq_ipc_close',
    'hq_exchange_float32', 'hq_exchange_float64', 'hq_exchange_rounds_float32', 'hq_exchange_rounds_float64', 'hq_exchange_round_wait', 'hq_alloc', 'hq_free', 'hq_alloc_mapped', 'hq_alloc_scattered',
    'hq_alloc_state', 'hq_free_state', 'hq_state_info', 'hq_state_pool_trim',
    'hq_plan_blocked', 'hq_plan_counts', 'hq_plan_read', 'hq_plan_free', 'hq_plan_simplify', 'hq_plan_fuse',
    'hq_blocked_selfcheck', 'hq_pointer_info', 'hq_vmm_remap', 'hq_shard_comm_count',
]


class HQError(RuntimeError):
    pass


def last_error():
    return (_last_error() or b'').decode()


def last_kernel():
    return (_last_kernel() or b'').decode()


def last_kernel_desc():
    return (_last_kernel_desc() or b'').decode()


def _check(rc, what):
    if rc:
        raise HQError(f'{what} failed: {last_error()}')


def log2_pack_size():
    return int(_get_log2_pack_size())


_log2_pack_size = log2_pack_size()


def device_count():
    return int(_device_count())


_stream_now = None


def set_stream(stream_handle):
    """`stream_handle`: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).
    Switching streams is ordered on the device (the new stream waits for the work already
    enqueued on the old one), never a host synchronisation."""
    global _stream_now
    h = int(stream_handle)
    if h != _stream_now:
        _check(_set_stream(ctypes.c_void_p(h)), 'hq_set_stream')
        _stream_now = h


def use_torch_stream():
    """Bind the library to torch's CURRENT stream (cheap when it has not changed).  The drivers
    call this at every public entry point so that kernels, torch allocations and copies made by
    the caller share one stream even under ``with torch.cuda.stream(...)``."""
    import torch
    set_stream(torch.cuda.current_stream().cuda_stream)


def sync():
    _check(_sync(), 'hq_sync')


def set_apply_mode(name):
    _check(_set_apply_mode(name.encode()), 'hq_set_apply_mode')


def _ptr(x):
    """Raw address of a numpy array, a torch tensor or an int."""
    if isinstance(x, int):
        return x
    if hasattr(x, 'data_ptr'):
        return x.data_ptr()
    if hasattr(x, 'ctypes'):
        return x.ctypes.data
    raise TypeError(f'cannot take the address of {type(x)}')


_TORCH_DTYPES = {}


def _float_dtype(x):
    if hasattr(x, 'data_ptr'):  # torch tensor
        dt = _TORCH_DTYPES.get(x.dtype)
        if dt is None:
            dt = _TORCH_DTYPES[x.dtype] = np.dtype(str(x.dtype).replace('torch.', ''))
        return dt
    return np.dtype(x.dtype)


def _n_qubits(x):
    size = x.numel() if hasattr(x, 'numel') else x.size
    n = int(size).bit_length() - 1
    if 1 << n != size:
        raise ValueError('plane size is not a power of two')
    return n


def apply_U(psi_re, psi_im, U, pos, n_qubits=None):
    """In-place ``apply_U_float{32,64}`` (include/hq_hip.h) on device tensors (torch) or
    host arrays (numpy).  `U`: 2^k x 2^k complex (row-major), `pos`: k positions with
    pos[0] <-> LSB of U's index (simulation.py:633)."""
    ft = _float_dtype(psi_re)
    ctype = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
    U = np.ascontiguousarray(U, dtype=ctype)
    k = len(pos)
    n = _n_qubits(psi_re) if n_qubits is None else int(n_qubits)
    if U.size != 4**k:
        raise ValueError("'U' and 'pos' are incompatible")
    if k and min(pos) < 0:  # ctypes would wrap a negative position silently; the library checks the upper bound
        raise ValueError("'pos' must be non-negative")
    # a ctypes array built from the k integers: 0.25 us against 2 us for a numpy array plus .ctypes.data_as()
    rc = _dot_core[ft](_ptr(psi_re), _ptr(psi_im), U.ctypes.data, (ctypes.c_uint32 * k)(*pos), n, k)
    _check(rc, 'apply_U')


def swap(array, pos, n_qubits=None):
    """In-place ``swap_<dtype>``: new[x] = old[(x & ~(2^s-1)) | sum_i x_i << pos[i]]."""
    dt = _float_dtype(array)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    n = _n_qubits(array) if n_qubits is None else int(n_qubits)
    rc = _swap_core[dt](_ptr(array), pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, len(pos))
    _check(rc, 'swap')


def to_complex(psi_re, psi_im, out):
    """out[2i] = re[i], out[2i+1] = im[i]; `out` is a complex tensor/array of the same size."""
    ft = _float_dtype(psi_re)
    ctype = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
    size = psi_re.numel() if hasattr(psi_re, 'numel') else psi_re.size
    rc = _to_complex_core[ctype](_ptr(psi_re), _ptr(psi_im), _ptr(out), size)
    _check(rc, 'to_complex')
    return out


def init_state(psi_re, psi_im, kind='basis', basis=0):
    ft = _float_dtype(psi_re)
    n = _n_qubits(psi_re)
    rc = _init_state[ft](_ptr(psi_re), _ptr(psi_im), n, {'basis': 0, 'plus': 1}[kind], int(basis))
    _check(rc, 'init_state')


def init_product_state(psi_re, psi_im, chars_by_position, hi_bits=0):
    """Write the product state whose factor on index bit p is ``chars_by_position[p]`` ('0', '1',
    '+' or '-'; a dict or sequence covering every bit of the FULL index).  The planes hold the
    2^n_local amplitudes whose higher index bits equal ``hi_bits >> n_local`` (0 on one GPU)."""
    ft = _float_dtype(psi_re)
    n_local = _n_qubits(psi_re)
    items = chars_by_position.items() if hasattr(chars_by_position, 'items') else enumerate(chars_by_position)
    m01 = v01 = mm = npm = 0
    for p, ch in items:
        if ch in '01':
            m01 |= 1 << p
            v01 |= int(ch) << p
        elif ch in '+-':
            npm += 1
            if ch == '-':
                mm |= 1 << p
        else:
            raise ValueError("product states are made of '0', '1', '+', '-'")
    rc = _init_product[ft](_ptr(psi_re), _ptr(psi_im), n_local, int(hi_bits), m01, v01, mm, npm)
    _check(rc, 'init_product_state')


def norm2(psi_re, psi_im):
    ft = _float_dtype(psi_re)
    size = psi_re.numel() if hasattr(psi_re, 'numel') else psi_re.size
    out = ctypes.c_double(0.0)
    rc = _norm2[ft](_ptr(psi_re), _ptr(psi_im), size, ctypes.byref(out))
    _check(rc, 'norm2')
    return out.value


def permute_bits(src, dst, perm, n_qubits=None):
    """dst[x] = src[pi(x)], bit i of x -> bit perm[i] of pi(x); out of place, device tensors."""
    dt = _float_dtype(src)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    n = _n_qubits(src) if n_qubits is None else int(n_qubits)
    if len(perm) != n:
        raise ValueError("'perm' must have one entry per index bit")
    rc = _permute_bits[dt.itemsize](_ptr(src), _ptr(dst), perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n)
    _check(rc, 'permute_bits')


def probabilities(psi_re, psi_im, pos, n_qubits=None):
    """Marginal probabilities of the index bits `pos` (bit j of the outcome <-> pos[j])."""
    ft = _float_dtype(psi_re)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    n = _n_qubits(psi_re) if n_qubits is None else int(n_qubits)
    out = np.zeros(1 << len(pos), dtype=np.float64)
    rc = _probabilities[ft](_ptr(psi_re), _ptr(psi_im), n, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                            len(pos), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    _check(rc, 'probabilities')
    return out


def project(psi_re, psi_im, pos, state, scale=1.0, n_qubits=None):
    """psi[x] *= scale where bits pos[j] of x equal bit j of `state`, 0 elsewhere."""
    ft = _float_dtype(psi_re)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    n = _n_qubits(psi_re) if n_qubits is None else int(n_qubits)
    rc = _project[ft](_ptr(psi_re), _ptr(psi_im), n, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                      len(pos), int(state), float(scale))
    _check(rc, 'project')


def vdot(a_re, a_im, b_re, b_im):
    """<a|b> = sum conj(a) b on split planes (double accumulation); returns a Python complex."""
    ft = _float_dtype(a_re)
    size = a_re.numel() if hasattr(a_re, 'numel') else a_re.size
    out = (ctypes.c_double * 2)()
    rc = _vdot[ft](_ptr(a_re), _ptr(a_im), _ptr(b_re), _ptr(b_im), size, out)
    _check(rc, 'vdot')
    return complex(out[0], out[1])


# --- multi-GPU shard exchange (include/hq_hip.h) -----------------------------------------------
def shard_unique_id():
    """128 opaque bytes from ncclGetUniqueId (rank 0 calls this and ships them to the others)."""
    buf = ctypes.create_string_buffer(128)
    _check(_shard_unique_id(buf), 'hq_shard_unique_id')
    return bytes(buf.raw)


def shard_load_rccl():
    """Bind librccl (dlopen + symbols) without creating anything: the part of the RCCL start-up a rank can fail alone."""
    _check(_shard_load_rccl(), 'hq_shard_load_rccl')


def shard_init_rccl(world, rank, unique_id):
    """Collective: creates this rank's RCCL communicator on the current device."""
    buf = ctypes.create_string_buffer(bytes(unique_id), 128)
    _check(_shard_init_rccl(int(world), int(rank), buf), 'hq_shard_init_rccl')


def shard_init_p2p(world, rank):
    _check(_shard_init_p2p(int(world), int(rank)), 'hq_shard_init_p2p')


def shard_p2p_register(local_plane, peer_addresses):
    arr = (ctypes.c_void_p * len(peer_addresses))(*[int(a) for a in peer_addresses])
    _check(_shard_p2p_register(_ptr(local_plane), arr), 'hq_shard_p2p_register')


_shard_comm_count = _define_function(_lib, 'hq_shard_comm_count', ctypes.c_int, ctypes.POINTER(ctypes.c_int))


def shard_info():
    w, r, t, cnt = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_int(0), ctypes.c_int(0)
    _shard_info(ctypes.byref(w), ctypes.byref(r), ctypes.byref(t))
    _check(_shard_comm_count(ctypes.byref(cnt)), 'hq_shard_comm_count')
    return {'world': w.value, 'rank': r.value, 'transport': {0: None, 1: 'rccl', 2: 'p2p'}[t.value], 'rccl_ranks_seen': cnt.value}


def shard_rccl_selftest(src, dst):
    """Grouped ncclSend/ncclRecv of `src` into `dst` with this rank as its own peer (plumbing check)."""
    nbytes = src.numel() * src.element_size()
    _check(_shard_rccl_selftest(_ptr(src), _ptr(dst), nbytes), 'hq_shard_rccl_selftest')


def shard_free():
    _check(_shard_free(), 'hq_shard_free')


def ipc_export(tensor):
    """(64-byte handle, offset) of the allocation holding `tensor` (HIP IPC / dmabuf)."""
    buf = ctypes.create_string_buffer(64)
    off = ctypes.c_uint64(0)
    _check(_ipc_export(_ptr(tensor), buf, ctypes.byref(off)), 'hq_ipc_export')
    return bytes(buf.raw), int(off.value)


def ipc_open(handle, offset):
    """Device address, in THIS process, of the memory another process exported."""
    buf = ctypes.create_string_buffer(bytes(handle), 64)
    out = ctypes.c_void_p(None)
    _check(_ipc_open(buf, int(offset), ctypes.byref(out)), 'hq_ipc_open')
    return int(out.value)


def exchange(src_re, src_im, dst_re, dst_im, perm=None, n_local=None):
    """One qubit exchange (hq_exchange_*).  Returns True if the exchanged shard is in the SRC planes."""
    ft = _float_dtype(src_re)
    m = _n_qubits(src_re) if n_local is None else int(n_local)
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.uint32)
        if len(perm) != m:
            raise ValueError("'perm' must have one entry per local index bit")
        pp = perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    where = ctypes.c_int(0)
    rc = _exchange[ft](_ptr(src_re), _ptr(src_im), _ptr(dst_re), _ptr(dst_im), m, pp, ctypes.byref(where))
    _check(rc, 'exchange')
    return bool(where.value)


def exchange_rounds(src_re, src_im, dst_re, dst_im, perm=None, n_local=None, sub_bits=2):
    """One qubit exchange in 2^sub_bits rounds (hq_exchange_rounds_*): returns (result in the SRC planes?, number of
    rounds); the library stream has not waited for any of them -- exchange_round_wait(s) before touching the pieces of round s."""
    ft = _float_dtype(src_re)
    m = _n_qubits(src_re) if n_local is None else int(n_local)
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.uint32)
        if len(perm) != m:
            raise ValueError("'perm' must have one entry per local index bit")
        pp = perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    where, rounds = ctypes.c_int(0), ctypes.c_uint(0)
    rc = _exchange_rounds[ft](_ptr(src_re), _ptr(src_im), _ptr(dst_re), _ptr(dst_im), m, pp, int(sub_bits), ctypes.byref(where),
                              ctypes.byref(rounds))
    _check(rc, 'exchange_rounds')
    return bool(where.value), int(rounds.value)


def exchange_round_wait(round_index):
    """The library stream waits for round `round_index` of the last exchange_rounds call."""
    _check(_exchange_round_wait(int(round_index)), 'exchange_round_wait')


class DeviceBuffer:
    """Device memory from hq_alloc, visible to torch through ``__cuda_array_interface__``
    (``torch.as_tensor(buf, device='cuda')`` aliases it; the buffer must outlive the tensor --
    `as_tensor` keeps a reference to this object)."""

    def __init__(self, nbytes, contiguous=True, scattered=None, seed=1, va_slots=None):
        """`scattered` = granule size in bytes: physical granules mapped in a shuffled order (VMM);
        with `va_slots` (permutation, one entry per granule) the placement is explicit."""
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p(None)
        if va_slots is not None:
            slots = np.ascontiguousarray(va_slots, dtype=np.uint32)
            gmin = ctypes.c_uint64(0)
            _check(_alloc_mapped(ctypes.byref(p), int(scattered), len(slots), slots.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                 ctypes.byref(gmin)), 'hq_alloc_mapped')
            self.granule_min = int(gmin.value)
        elif scattered:
            _check(_alloc_scattered(ctypes.byref(p), self.nbytes, int(scattered), int(seed)), 'hq_alloc_scattered')
        else:
            _check(_alloc(ctypes.byref(p), self.nbytes, 1 if contiguous else 0), 'hq_alloc')
        self.ptr = int(p.value)
        self.contiguous = bool(contiguous) and not scattered

    def view(self, offset_bytes, shape, typestr):
        """An object torch.as_tensor can wrap: `shape` elements of `typestr` ('<f4', '<f8') at an offset."""
        owner = self

        class _View:
            __cuda_array_interface__ = {'shape': tuple(int(d) for d in shape), 'typestr': typestr,
                                        'data': (owner.ptr + int(offset_bytes), False), 'version': 2, 'strides': None}
            _owner = owner
        return _View()

    def free(self):
        if self.ptr:
            _free(ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


STATE_PLAIN, STATE_NO_SEARCH, STATE_NO_POOL = 1, 2, 4


class StatePlanes:
    """Both planes of an n-qubit state from hq_alloc_state (tuned placement for states >= 256 MiB, pooled per size).
    ``torch.as_tensor(obj, device='cuda')`` aliases it as a (2, stride) tensor whose rows start at the re / im planes
    (`as_tensor` keeps this object alive); the memory goes back through hq_free_state when it dies."""

    def __init__(self, n_qubits, float_dtype, flags=0):
        ft = np.dtype(float_dtype)
        re, im = ctypes.c_void_p(None), ctypes.c_void_p(None)
        _check(_alloc_state(int(n_qubits), 8 * ft.itemsize, int(flags), ctypes.byref(re), ctypes.byref(im)), 'hq_alloc_state')
        self.re, self.im = int(re.value), int(im.value)
        self.n_qubits = int(n_qubits)
        self.stride = (self.im - self.re) // ft.itemsize
        self.info = state_info(self.re)
        self.__cuda_array_interface__ = {'shape': (2, self.stride), 'typestr': '<f4' if ft.itemsize == 4 else '<f8',
                                         'data': (self.re, False), 'version': 2, 'strides': None}

    def free(self):
        if self.re:
            _free_state(ctypes.c_void_p(self.re))
            self.re = self.im = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def state_info(psi_re=None):
    """Placement report of a state (address of its re plane) or of the last hq_alloc_state (None), as a dict."""
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    _check(_state_info(ctypes.c_void_p(psi_re) if psi_re else None, buf, len(buf)), 'hq_state_info')
    return json.loads(buf.value.decode() or '{}')


def state_pool_trim():
    _check(_state_pool_trim(), 'hq_state_pool_trim')


def plan_blocked(n, gates, tile_bits, low_bits, inner_max, min_gates, tries, fusion_orders, elem_bytes, seed, commute_tol):
    """``hq_plan_blocked`` (include/hq_hip.h): the cache-blocked schedule of `gates` = [(U, positions)], positions with
    the matrix's MOST significant qubit first.  Returns (op_kind, op_first_gate, op_tile[n_ops, tile_bits], gate_k,
    gate_positions (flat, most significant first), matrices (list of complex128 arrays))."""
    G = len(gates)
    k = np.fromiter((len(p) for _, p in gates), dtype=np.uint32, count=G)
    pos = np.fromiter((int(x) for _, p in gates for x in p), dtype=np.uint32, count=int(k.sum()))
    U = np.concatenate([np.asarray(u, dtype=np.complex128).reshape(-1) for u, _ in gates]) if G else np.zeros(0, np.complex128)
    handle = ctypes.c_void_p()
    _check(_plan_blocked(n, G, k.ctypes.data_as(_u32p), pos.ctypes.data_as(_u32p), U.ctypes.data, tile_bits, low_bits,
                         255 if inner_max == 'auto' else int(inner_max or 0), min_gates, tries, fusion_orders, elem_bytes,
                         seed, commute_tol, ctypes.byref(handle)), 'hq_plan_blocked')
    try:
        n_ops, n_g, tb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        n_pos, n_el = ctypes.c_uint64(), ctypes.c_uint64()
        _check(_plan_counts(handle, ctypes.byref(n_ops), ctypes.byref(n_g), ctypes.byref(n_pos), ctypes.byref(n_el), ctypes.byref(tb)),
               'hq_plan_counts')
        kind = np.empty(n_ops.value, np.uint32)
        first = np.empty(n_ops.value + 1, np.uint32)
        tile = np.empty((n_ops.value, tb.value), np.uint32)
        gk = np.empty(n_g.value, np.uint32)
        gpos = np.empty(n_pos.value, np.uint32)
        mats = np.empty(n_el.value, np.complex128)
        _check(_plan_read(handle, kind.ctypes.data_as(_u32p), first.ctypes.data_as(_u32p), tile.ctypes.data_as(_u32p),
                          gk.ctypes.data_as(_u32p), gpos.ctypes.data_as(_u32p), mats.ctypes.data), 'hq_plan_read')
    finally:
        _plan_free(handle)
    return kind, first, tile, gk, gpos, mats


def _read_plan(handle):
    n_ops, n_g, tb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    n_pos, n_el = ctypes.c_uint64(), ctypes.c_uint64()
    _check(_plan_counts(handle, ctypes.byref(n_ops), ctypes.byref(n_g), ctypes.byref(n_pos), ctypes.byref(n_el), ctypes.byref(tb)),
           'hq_plan_counts')
    kind = np.empty(n_ops.value, np.uint32)
    first = np.empty(n_ops.value + 1, np.uint32)
    tile = np.empty((n_ops.value, tb.value), np.uint32)
    gk = np.empty(n_g.value, np.uint32)
    gpos = np.empty(n_pos.value, np.uint32)
    mats = np.empty(n_el.value, np.complex128)
    _check(_plan_read(handle, kind.ctypes.data_as(_u32p), first.ctypes.data_as(_u32p), tile.ctypes.data_as(_u32p),
                      gk.ctypes.data_as(_u32p), gpos.ctypes.data_as(_u32p), mats.ctypes.data), 'hq_plan_read')
    return kind, first, tile, gk, gpos, mats


def plan_fuse(n_ids, gates, max_n_qubits, use_matrix_commutation, max_n_qubits_matrix, exclude_mask, commute_tol):
    """``hq_plan_fuse``: the fused gates of `gates` = [(U, integer qubit ids)] as (gate_k, ids flat (sorted, most
    significant first), matrices flat complex128)."""
    G = len(gates)
    k = np.fromiter((len(q) for _, q in gates), dtype=np.uint32, count=G)
    ids = np.fromiter((int(x) for _, q in gates for x in q), dtype=np.uint32, count=int(k.sum()))
    U = np.concatenate([np.asarray(u, dtype=np.complex128).reshape(-1) for u, _ in gates]) if G else np.zeros(0, np.complex128)
    handle = ctypes.c_void_p()
    _check(_plan_fuse(n_ids, G, k.ctypes.data_as(_u32p), ids.ctypes.data_as(_u32p), U.ctypes.data, int(max_n_qubits),
                      int(bool(use_matrix_commutation)), int(max_n_qubits_matrix), int(exclude_mask), float(commute_tol),
                      ctypes.byref(handle)), 'hq_plan_fuse')
    try:
        _, _, _, gk, gpos, mats = _read_plan(handle)
    finally:
        _plan_free(handle)
    return gk, gpos, mats


def plan_simplify(n_ids, gates, atol, use_matrix_commutation, max_n_qubits_matrix, remove_id_gates):
    """``hq_plan_simplify``: indices of the gates of `gates` = [(U, integer qubit ids)] that survive, in their new order."""
    G = len(gates)
    k = np.fromiter((len(q) for _, q in gates), dtype=np.uint32, count=G)
    ids = np.fromiter((int(x) for _, q in gates for x in q), dtype=np.uint32, count=int(k.sum()))
    U = np.concatenate([np.asarray(u, dtype=np.complex128).reshape(-1) for u, _ in gates]) if G else np.zeros(0, np.complex128)
    out = np.empty(max(G, 1), np.uint32)
    cnt = ctypes.c_uint32()
    _check(_plan_simplify(n_ids, G, k.ctypes.data_as(_u32p), ids.ctypes.data_as(_u32p), U.ctypes.data, float(atol),
                          int(bool(use_matrix_commutation)), int(max_n_qubits_matrix), int(bool(remove_id_gates)),
                          out.ctypes.data_as(_u32p), ctypes.byref(cnt)), 'hq_plan_simplify')
    return out[:cnt.value]


def pack_blocked(gates, complex_type='complex64'):
    """[(U, pos), ...] -> (U_all, pos_all, k_all) arrays for apply_blocked."""
    U_all = np.concatenate([np.ascontiguousarray(U, dtype=complex_type).reshape(-1) for U, _ in gates])
    pos_all = np.concatenate([np.asarray(p, dtype=np.uint32).reshape(-1) for _, p in gates])
    k_all = np.asarray([len(p) for _, p in gates], dtype=np.uint32)
    return np.ascontiguousarray(U_all), np.ascontiguousarray(pos_all), k_all


def apply_blocked(psi_re, psi_im, tile_pos, gates=None, n_qubits=None, packed=None):
    """Apply a list of k <= 4 gates whose targets all lie in `tile_pos` in ONE pass over the
    state (device tensors).  `gates`: [(U, pos)] with GLOBAL positions, pos[0] = LSB."""
    ft = _float_dtype(psi_re)
    ctype = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
    tile_pos = np.ascontiguousarray(tile_pos, dtype=np.uint32)
    U_all, pos_all, k_all = packed if packed is not None else pack_blocked(gates, ctype)
    if U_all.dtype != ctype:
        raise ValueError('packed matrices do not match the precision of the planes')
    n = _n_qubits(psi_re) if n_qubits is None else int(n_qubits)
    U32P = ctypes.POINTER(ctypes.c_uint32)
    rc = _apply_blocked[ft](_ptr(psi_re), _ptr(psi_im), n, tile_pos.ctypes.data_as(U32P), len(tile_pos),
                            len(k_all), U_all.ctypes.data, pos_all.ctypes.data_as(U32P), k_all.ctypes.data_as(U32P))
    _check(rc, 'apply_blocked')


_blocked_selfcheck = _define_function(_lib, 'hq_blocked_selfcheck', ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_int))


def blocked_selfcheck():
    """What the library's cross-check of the cache-blocked kernel variants has seen so far in this process
    (include/hq_hip.h: hq_blocked_selfcheck): checks run / failed and which variants are currently on."""
    runs, failures, sw = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _check(_blocked_selfcheck(ctypes.byref(runs), ctypes.byref(failures), ctypes.byref(sw)), 'hq_blocked_selfcheck')
    return {'runs': runs.value, 'failures': failures.value, 'pipe': bool(sw.value & 1), 'groups': bool(sw.value & 2),
            'direct': bool(sw.value & 4), 'big': bool(sw.value & 8)}


class Program:
    """Compiled circuit: ``with Program() as prog: <apply_U / apply_blocked / ... calls>`` records
    the launches instead of running them; ``prog.run()`` replays them (one hipGraph launch from
    the second run on).  Bound to the plane tensors used while recording -- keep them alive."""

    def __init__(self):
        self._handle = ctypes.c_void_p(None)
        self._recording = False
        self._keep = []

    def __enter__(self):
        _check(_program_begin(), 'hq_program_begin')
        self._recording = True
        return self

    def __exit__(self, exc_type, exc, tb):
        rc = _program_end(ctypes.byref(self._handle))
        self._recording = False
        if exc_type is None:
            _check(rc, 'hq_program_end')
        elif self._handle:
            self.free()
        return False

    def keep_alive(self, *objects):
        self._keep.extend(objects)

    def __len__(self):
        return max(0, int(_program_size(self._handle)))

    def run(self):
        _check(_program_run(self._handle), 'hq_program_run')

    def free(self):
        if self._handle:
            _program_free(self._handle)
            self._handle = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
