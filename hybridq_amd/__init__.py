"""hybridq_amd -- MI355X (gfx950) state-vector evolution core for HybridQ.

Drop-in replacement of the reference's C++/OpenMP "evolution" backend
(/root/reference/include/*, loaded through hybridq/utils/dot.py and transpose.py) as a
hand-written HIP library with the same C ABI, plus the host-side driver that mirrors
``hybridq.circuit.simulation.simulate(..., optimize='evolution')``.

Submodules are imported on first use (so that ``python -m hybridq_amd.build`` can
(re)build the library without loading a stale one).  ``hybridq_amd.core`` loads
``csrc/libhq_hip.so`` and raises ImportError if it is missing or incomplete: there is
no CPU fallback anywhere in the product path.
"""
import importlib

__version__ = '0.1.0'
_SUBMODULES = ('core', 'simulation', 'circuits', 'build', 'dist', 'fusion', 'dot', 'transpose', 'dm', 'qasm', 'functional', 'blocking', 'aligned')


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module(f'{__name__}.{name}')
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
