"""Developer soak (run by hand on a GPU box: python tests/soak_gpu.py; not collected by pytest): random apply_U calls (n 6..22, k 1..10, both precisions, non-unitary matrices) and random low-bit
swaps (s 1..18, four element types) on the GPU against the CPU oracle.  SEED=<int> selects the stream.
Test infrastructure only: imports oracle/."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle
from hybridq_amd import core
from hybridq_amd.circuits import haar_unitary
seed=int(os.environ.get('SEED','0')); rng=np.random.default_rng(seed)
port=oracle.load_port()
core.use_torch_stream()
bad=0
# apply_U differential: random n, k, positions, dtype, non-unitary matrices
for t in range(300):
    n=int(rng.integers(6,23)); k=int(rng.integers(1,min(n,10)+1)); ft=np.float32 if t%3 else np.float64
    pos=rng.permutation(n)[:k].astype(np.uint32)
    U=(rng.standard_normal((1<<k,1<<k))+1j*rng.standard_normal((1<<k,1<<k))).astype(np.complex64 if ft==np.float32 else np.complex128)/np.sqrt(1<<k)
    from oracle.binding import aligned_empty
    pl=aligned_empty((2,1<<n),ft); pl[:]=rng.standard_normal((2,1<<n)).astype(ft)
    d=torch.from_numpy(pl.copy()).cuda()
    assert port.apply_U(pl[0],pl[1],U,pos)==0
    core.apply_U(d[0],d[1],U,pos,n); core.sync()
    err=np.abs(d.cpu().numpy()-pl).max()/np.abs(pl).max()
    tol=(4e-6 if ft==np.float32 else 4e-14)*max(1,k-5)
    if not err<tol: bad+=1; print('apply_U FAIL',n,k,list(pos),ft.__name__,err,core.last_kernel_desc())
# swap differential
for t in range(200):
    dt=[np.float32,np.float64,np.int32,np.int64][t%4]; s=int(rng.integers(1,19)); n=s+int(rng.integers(0,5)); n=min(n,24)
    s=min(s,n)
    pos=rng.permutation(s)
    a=rng.integers(0,2**31-1,1<<n).astype(dt)
    exp=oracle.swap_numpy(a,pos)
    tt=torch.from_numpy(a.copy()).cuda(); core.swap(tt,pos,n); core.sync()
    if not (tt.cpu().numpy()==exp).all(): bad+=1; print('swap FAIL',dt.__name__,s,n,list(pos))
print('soak seed',seed,'failures',bad)
