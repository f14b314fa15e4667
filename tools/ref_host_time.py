"""Build container only (needs /root/reference): host time of the reference's simplify + compress + to_matrix_gate against
fusion.simplify + fusion.fuse on the n = 30 depth-40 benchmark circuit (same 109 fused gates).
    cd /tmp && LD_LIBRARY_PATH=/root/repo/oracle/_ref python /root/repo/tools/ref_host_time.py"""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden')
import make_golden as mg
mg.install_stubs(); sys.path.insert(0, mg.REF)
from hybridq.circuit import Circuit, utils
from hybridq.gate import Gate
from hybridq.gate import property as pr
from hybridq_amd.circuits import rqc_1q2q
from hybridq_amd.fusion import simplify, fuse, single_thread_blas
n=30
g=rqc_1q2q(n, depth=40, seed=n)
c=Circuit(Gate('MATRIX', qubits=list(qs), U=U) for U,qs in g)
for rep in range(2):
    t=time.time(); cs=utils.simplify(c, remove_id_gates=True, atol=1e-8, verbose=False); t1=time.time()-t
    t=time.time(); layers=utils.compress(cs, 4, verbose=False, skip_compression=[pr.FunctionalGate]); t2=time.time()-t
    t=time.time(); fused=[utils.to_matrix_gate(l, complex_type='complex64') for l in layers]; t3=time.time()-t
    print('reference: simplify %.2f s, compress %.2f s, to_matrix_gate %.2f s -> %d gates'%(t1,t2,t3,len(fused)))
    with single_thread_blas():
        t=time.time(); s2=simplify(g); t4=time.time()-t
        t=time.time(); f2=fuse(s2, 4, complex_type='complex64'); t5=time.time()-t
    print('here:      simplify %.3f s, fuse %.3f s -> %d gates'%(t4,t5,len(f2)))
