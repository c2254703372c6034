import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids
A = P.banded_ex2b(1_250_000, 7); n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
op = DeviceCSR(A, comm_init=(0, 1, rccl_ids(1)[0], L.default_rccl_path())) if sys.argv[1] == 'comm' else DeviceCSR(A)
op.begin(L.PIPE_PR, b, x0, 400); op.iterate(300); op.sync(); op.close()
