"""Developer stress of the schedules (run by hand on a GPU box: python tests/stress_schedules.py; not collected by pytest): random circuits (n 14..22, mixed generators, '01+-' initial strings, both
precisions) through optimize='evolution' (cost-model choice), blocked=True and a small-tile blocked plan, against the
independent complex128 tensordot evolution of the oracle.  SEED=<int>; NMAX=<int> / TRIALS=<int> shrink it for the host emulation
(`NMAX=17 TRIALS=24 python tests/emu/run_emulated.py tests/stress_schedules.py`).  Test infrastructure only: imports oracle/."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle
from tolerances import circuit_tol
from hybridq_amd.circuits import random_dense, rqc_1q2q
from hybridq_amd.simulation import simulate
rng=np.random.default_rng(int(os.environ.get('SEED','0')))
bad=0
NMAX=int(os.environ.get('NMAX','22'))
for trial in range(int(os.environ.get('TRIALS','48'))):
    n=int(rng.integers(min(14,NMAX-3),NMAX+1)); ct='complex64' if trial%3 else 'complex128'
    kind=trial%4
    if kind==0: gates=rqc_1q2q(n,depth=int(rng.integers(4,24)),seed=int(rng.integers(1<<30)))
    elif kind==1: gates=random_dense(n,int(rng.integers(10,120)),kmax=int(rng.integers(2,6)),seed=int(rng.integers(1<<30)),unitary=True)
    else: gates=rqc_1q2q(n,depth=int(rng.integers(3,12)),seed=int(rng.integers(1<<30)))+random_dense(n,int(rng.integers(5,40)),kmax=4,seed=int(rng.integers(1<<30)),unitary=True)
    init=''.join(rng.choice(list('01+-'),size=n))
    exp=oracle.evolve_tensordot(gates,n,initial_state=init,qubits=list(range(n)))
    for opt in ('evolution', dict(blocked=True), dict(blocked={'tile_bits':12 if ct=='complex64' else 11})):
        kw=dict(optimize=opt) if isinstance(opt,str) else dict(opt)
        psi,info=simulate(gates,initial_state=init,complex_type=ct,qubits=list(range(n)),return_info=True,**kw)
        err=np.abs(psi.reshape(-1)-exp).max()/np.abs(exp).max()
        tol=circuit_tol(gates,complex_type=ct)
        if not err<tol:
            bad+=1; print('FAIL',trial,n,ct,opt,err,tol,info.get('schedule'))
print('stress done, failures:',bad)
