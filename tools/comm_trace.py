import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem
A = P.banded_ex2b(400_000, 7); n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
for gather in ('1', '0'):
    uid, path = rccl_ids(1)
    A_loop, halo, moved = loopback_problem(A, 9)
    op = DeviceCSR(A_loop, comm_init=(0, 1, uid, path), halo=halo, knobs={'PRCG_GATHER': gather, 'PRCG_FUSED_COMM': '0'})
    op.begin(L.PIPE_PR, b, x0, 40); op.iterate(30); op.sync(); print(op.schedule()); op.close()
