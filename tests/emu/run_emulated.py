"""TEST INFRASTRUCTURE: run a script of this repository (bench.py, tools/*.py, __graft_entry__.py smoke) on a box WITHOUT a GPU
against the host emulation of the HIP library, with CPU torch standing for the device (fake_cuda.py):

    python tests/emu/run_emulated.py bench.py --qubits 14 --depth 4 --steps 1 --parity-qubits 12
    python tests/emu/run_emulated.py __graft_entry__.py smoke

Checks code paths and index arithmetic end to end; every time and rate such a run prints is meaningless."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), HERE]
import emu_util  # noqa: E402

os.environ['HQ_HIP_LIBRARY'] = emu_util.emu_library()
os.environ['HQ_EMU_HOST_IS_DEVICE'] = '1'
os.environ.setdefault('HQ_RCCL_LIBRARY', emu_util.emu_rccl_library())
os.environ['HQ_EMU_GPU_SUITE'] = '1'
import fake_cuda  # noqa: E402

fake_cuda.install()

# children the script starts as `python <file of this repository> ...` (bench.py's blocked_variants leg: one
# tools/ab_blocked.py process per switch setting) run under this launcher too -- the product files know nothing of it
import subprocess  # noqa: E402

_run = subprocess.run


def _run_emulated(cmd, *a, **kw):
    if isinstance(cmd, (list, tuple)) and len(cmd) >= 2 and cmd[0] == sys.executable and str(cmd[1]).endswith('.py') \
            and os.path.abspath(cmd[1]).startswith(ROOT + os.sep) and os.path.abspath(cmd[1]) != os.path.abspath(__file__):
        cmd = [cmd[0], os.path.abspath(__file__)] + list(cmd[1:])
    return _run(cmd, *a, **kw)


subprocess.run = _run_emulated
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name='__main__')
