import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem
A = P.laplace_2d(500, 400); n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
A_loop, halo, moved = loopback_problem(A, 600)
op = DeviceCSR(A_loop, comm_init=(0, 1, rccl_ids(1)[0], L.default_rccl_path()), halo=halo)
op.begin(L.PIPE_PR, b, x0, 40); print(op.schedule())
op.iterate(4)
try: op.sync()
except Exception as e: print('ERR')
op.close()
