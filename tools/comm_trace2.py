import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem
which = sys.argv[1]
A = P.laplace_2d(500, 400) if which.startswith('lap') else P.banded_ex2b(400_000, 7)
n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
uid, path = rccl_ids(1)
if which.endswith('loop'):
    A_loop, halo, moved = loopback_problem(A, 600 if which.startswith('lap') else 9)
    op = DeviceCSR(A_loop, comm_init=(0, 1, uid, path), halo=halo)
else:
    op = DeviceCSR(A, comm_init=(0, 1, uid, path))
op.begin(L.PIPE_PR, b, x0, 40); print(op.schedule())
op.iterate(12)
try:
    op.sync()
except Exception as e:
    print('ERR', e)
op.close()
