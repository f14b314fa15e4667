"""Developer check: which OpenMP thread count gives the reference core its best rate on this host."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle.binding import aligned_empty  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
lib = oracle.load_ref() if oracle.have_ref() else oracle.load_port()
gomp = ctypes.CDLL('libgomp.so.1')
gates = rqc_1q2q(n, 40, seed=n)
print('cpus', os.cpu_count(), 'lib', lib.kind)
for thr in (2, 4, 8, 12, 16, 24):
    if thr > os.cpu_count():
        continue
    gomp.omp_set_num_threads(thr)
    planes = aligned_empty((2, 1 << n), np.float32)
    planes[:] = 0
    planes[0, 0] = 1
    _, info = oracle.evolve_reference_protocol(lib, gates, n, complex_type='complex64', planes=planes,
                                               warmup_gates=8, max_seconds=6, to_complex=False)
    print(thr, 'threads:', round(info['n_gates'] / info['runtime (s)'], 2), 'gate-apps/s', flush=True)
