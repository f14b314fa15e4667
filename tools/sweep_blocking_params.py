"""Developer sweep of the blocked planner's parameters on the n=30 depth-40 circuit."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.blocking import blocked_stats, plan_blocked  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import EvolutionState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gates = rqc_1q2q(n, 40, seed=n)
state = EvolutionState(list(range(n)), complex_type='complex64', initial_state='0' * n)
TBS = [int(x) for x in os.environ.get('SWEEP_TB', '12,13').split(',')]
LBS = [int(x) for x in os.environ.get('SWEEP_LB', '3,4,5,6').split(',')]
IMS = [int(x) for x in os.environ.get('SWEEP_IM', '0,2,3,4').split(',')]
for tb in TBS:
    for lb in LBS:
        for im in IMS:
            for mg in (3,):
                ops = plan_blocked(gates, state.map, n, tile_bits=tb, low_bits=lb, inner_max=im, min_gates=mg)
                packed = [('B', op[1], core.pack_blocked(op[2])) if op[0] == 'B' else op for op in ops]

                def run():
                    for op in packed:
                        if op[0] == 'G':
                            core.apply_U(state.planes[0], state.planes[1], op[1], op[2], n)
                        else:
                            core.apply_blocked(state.planes[0], state.planes[1], op[1], packed=op[2], n_qubits=n)

                run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 2 * 1e3
                st = blocked_stats(ops)
                print(f'tb={tb} low={lb} inner_max={im} passes={st["blocked_passes"]} plain={st["plain_gates"]} '
                      f'inner={st["inner_gates"]} {st["inner_k_histogram"]}  {ms:7.1f} ms', flush=True)
