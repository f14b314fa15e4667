"""TEST INFRASTRUCTURE: worker processes of `-m gpu` tests call ``emu_boot.maybe_install()`` first, so that a run of the suite
against the host emulation (HQ_EMU_GPU_SUITE=1, tests/conftest.py) reaches them too: the environment (library path, pointer
classification) is inherited, the torch.cuda stand-in has to be installed per process."""
import os
import sys


def maybe_install():
    if os.environ.get('HQ_EMU_GPU_SUITE') != '1':
        return False
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, 'emu')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu_util
    os.environ.setdefault('HQ_HIP_LIBRARY', emu_util.emu_library())
    os.environ['HQ_EMU_HOST_IS_DEVICE'] = '1'
    os.environ.setdefault('HQ_RCCL_LIBRARY', emu_util.emu_rccl_library())
    import fake_cuda
    fake_cuda.install()
    return True
