"""hybridq_amd -- MI355X (gfx950) state-vector evolution core for HybridQ.

Drop-in replacement of the reference's C++/OpenMP "evolution" backend
(/root/reference/include/*, loaded through hybridq/utils/dot.py and transpose.py) as a
hand-written HIP library with the same C ABI, plus the host-side driver that mirrors
``hybridq.circuit.simulation.simulate(..., optimize='evolution')``.

Importing the package loads ``csrc/libhq_hip.so`` and raises if it is missing: there
is no CPU fallback in the product path.
"""
from . import core  # noqa: F401  (raises ImportError if the HIP library is absent)

__version__ = '0.1.0'
