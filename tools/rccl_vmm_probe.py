"""Does RCCL move bytes out of the memory the product allocates?  (VERDICT r02 missing #1.)

HipBackend.empty_planes gives every shard >= 256 MiB a VMM-mapped buffer (hq_alloc_mapped) whenever the transport is
RCCL, but until now RCCL only ever saw torch-allocator memory.  hipIpcGetMemHandle refuses VMM mappings; this probe
finds out what ncclSend / ncclRecv do with them, on the one GPU we have: a one-rank communicator, a grouped
send/recv with the rank as its own peer (hq_shard_rccl_selftest) between two VMM-backed planes of the product's size
class, the same between VMM and torch memory in both directions, and a one-rank hq_exchange_* with a permutation on
VMM planes against the same call on torch memory."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
import placement_util as sim  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 26  # 2 planes x 256 MiB
core.use_torch_stream()
dev = torch.device('cuda', 0)


def vmm_planes(gran, seed):
    stride = (1 << m) + 12288 // 4
    owner = sim._VmmPlanes(2 * stride * 4, (2, stride), '<f4', gran, seed)
    raw = torch.as_tensor(owner, device=dev)
    assert raw.data_ptr() == owner.buf.ptr
    return raw[:, :1 << m]


uid = core.shard_unique_id()
core.shard_init_rccl(1, 0, uid)
print('communicator: ok', core.shard_info(), flush=True)
a_v, b_v = vmm_planes(8 << 20, 101), vmm_planes(2 << 20, 0)
a_t = torch.empty((2, 1 << m), dtype=torch.float32, device=dev)
b_t = torch.empty_like(a_t)
ref = torch.randn((2, 1 << m), dtype=torch.float32, device=dev)
for name, src, dst in (('torch->torch', a_t, b_t), ('vmm->vmm', a_v, b_v), ('vmm->torch', a_v, b_t), ('torch->vmm', a_t, b_v)):
    src.copy_(ref)
    dst.zero_()
    torch.cuda.synchronize()
    try:
        t0 = time.perf_counter()
        for p in (0, 1):
            core.shard_rccl_selftest(src[p], dst[p])
        core.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = torch.equal(dst, ref)
        print(f'rccl self send/recv {name}: {"OK" if ok else "WRONG DATA"}  {2 * 4 * (1 << m) / dt / 1e9:.1f} GB/s', flush=True)
    except Exception as e:  # noqa: BLE001
        print(f'rccl self send/recv {name}: FAILED {e!r}', flush=True)

core.shard_free()
rng = np.random.default_rng(3)
for perm in (np.concatenate([[1, 0], np.arange(2, m)]), rng.permutation(m), np.roll(np.arange(m), 3)):
    a_t.copy_(ref)
    a_v.copy_(ref)
    w_t = core.exchange(a_t[0], a_t[1], b_t[0], b_t[1], perm, m)
    w_v = core.exchange(a_v[0], a_v[1], b_v[0], b_v[1], perm, m)
    core.sync()
    print('one-rank exchange with permutation on VMM planes == on torch planes:', w_t == w_v and torch.equal(b_t, b_v), flush=True)
