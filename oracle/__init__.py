"""CPU oracle for the HybridQ evolution hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``hybridq_amd`` (the product) never does.

Contents
--------
``hq_oracle.c``      C restatement of apply_U / swap / to_complex (same C ABI as
                     the reference, include/python_U.cpp:127-154,
                     include/python_swap.cpp:68-99).
``_ref/``            the reference's own C++ core compiled by ``oracle/Makefile``
                     from /root/reference/include (git-ignored build output).
``binding.py``       ctypes loader for either library.
``evolution.py``     numpy restatement of the reference's driver protocol
                     (hybridq/circuit/simulation/simulation.py:491-675) and an
                     independent float64 tensordot evolution.
"""
from .binding import OracleLib, load_port, load_ref, have_ref  # noqa: F401
from .evolution import (  # noqa: F401
    evolve_reference_protocol,
    evolve_tensordot,
    apply_gate_numpy,
    swap_numpy,
)
