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
# round 3: arbitrary bit permutations (out of place), the exchange pack (one rank, both planes), in-place swaps through the tile
# kernel incl. SPLIT mode, against index arithmetic on the device
def src_index(idx, perm):
    y=torch.zeros_like(idx)
    for i,p in enumerate(perm): y|=((idx>>i)&1)<<int(p)
    return y
core.shard_free()
for t in range(int(os.environ.get('PERM_TRIALS','150'))):
    dt=[torch.float32,torch.float64,torch.int32,torch.int64][t%4]; n=int(rng.integers(10,26))
    kind=t%5
    perm=np.arange(n)
    if kind==0: perm=rng.permutation(n)
    elif kind==1: lo=int(rng.integers(0,6)); perm[lo:]=lo+rng.permutation(n-lo)
    elif kind==2: perm=perm[::-1].copy()
    elif kind==3:
        ev=sorted(rng.permutation(n)[:int(rng.integers(1,4))]); perm=np.array([b for b in range(n) if b not in ev]+list(ev))
    else: perm=np.roll(perm,int(rng.integers(1,n)))
    idx=torch.arange(1<<n,device='cuda',dtype=torch.int64)
    data=(idx%1000003).to(dt); dst=torch.empty_like(data)
    core.permute_bits(data,dst,perm,n); core.sync()
    if not torch.equal(dst,data[src_index(idx,perm)]): bad+=1; print('permute_bits FAIL',dt,n,list(perm))
    if n>=12 and dt in (torch.float32,torch.float64):
        two=torch.stack([data[:1<<(n-1)],-data[:1<<(n-1)]]).contiguous(); out=torch.zeros_like(two)
        p2=rng.permutation(n-1); core.exchange(two[0],two[1],out[0],out[1],p2,n-1); core.sync()
        y=src_index(idx[:1<<(n-1)],p2)
        if not (torch.equal(out[0],two[0][y]) and torch.equal(out[1],two[1][y])): bad+=1; print('exchange pack FAIL',dt,n,list(p2))
    s=int(rng.integers(8,min(n,17)+1)); pos=rng.permutation(s) if t%2 else np.roll(np.arange(s),1+t%3)
    full=np.concatenate([pos,np.arange(s,n)]); exp=data[src_index(idx,full)]
    core.swap(data,pos,n); core.sync()
    if not torch.equal(data,exp): bad+=1; print('swap(tile) FAIL',dt,n,s,list(pos))
    del idx,data,dst,exp
# round 3: the state allocator (search, pool, trim) in a loop: every state must work and come back intact
for t in range(int(os.environ.get('ALLOC_TRIALS','6'))):
    n=26+t%2
    os.environ['HQ_STATE_TRIES']=str(1+t%3)
    st=core.StatePlanes(n,np.float32,flags=(core.STATE_NO_POOL if t%3==2 else 0))
    pl=torch.as_tensor(st,device='cuda')[:,:1<<n]
    core.init_state(pl[0],pl[1],'plus'); U=haar_unitary(4,rng)
    core.apply_U(pl[0],pl[1],U,[3,n-1],n); nrm=core.norm2(pl[0],pl[1])
    if abs(nrm-1)>1e-4: bad+=1; print('alloc_state FAIL',n,st.info,nrm)
    del pl; st.free()
    if t%2: core.state_pool_trim()
print('soak seed',seed,'failures',bad)
