"""End-to-end wall time of simulate() on the benchmark circuit (n = 28, 30): planning + allocation + gate loop,
with the state left on the device and returned as a numpy array."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hybridq_amd.circuits import rqc_1q2q
from hybridq_amd.simulation import simulate
for n in (28,30):
    g=rqc_1q2q(n,depth=40,seed=n)
    for rn in (False, True):
        for rep in range(2):
            t=time.perf_counter()
            psi,info=simulate(g,initial_state='0'*n,optimize='evolution',return_numpy_array=rn,return_info=True,qubits=list(range(n)))
            torch.cuda.synchronize()
            print(n, 'numpy' if rn else 'device', rep, 'wall %.3f s' % (time.perf_counter()-t), 'loop %.3f s' % info['runtime (s)'])
            del psi
