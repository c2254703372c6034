import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids
A = P.laplace_2d(500, 400)
n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
uid, path = rccl_ids(1)
op = DeviceCSR(A, comm_init=(0, 1, uid, path))
def timing(tag):
    op.begin(L.PIPE_PR, b, x0, 1001)
    t0 = time.perf_counter(); op.iterate(600)
    try:
        op.sync(); err = ''
    except Exception as e:
        err = 'TIMEOUT'
    print(tag, '%.1f us/iteration' % ((time.perf_counter() - t0) / 600 * 1e6), err, flush=True)
timing('fresh')
for tag, variant, inv in (('after PIPE_PR+hist', L.PIPE_PR, None), ('after PIPE_P+hist', L.PIPE_P, None), ('after PIPE_PR jacobi+hist', L.PIPE_PR, 1 / A.diagonal()),
                          ('after get_vector', None, None)):
    if variant is not None:
        op.begin(variant, b, x0, 60, x_true=xt, inv_diag=inv, hist_mask=15)
        op.iterate(59); op.sync(); op.history()
    else:
        op.begin(L.PIPE_PR, b, x0, 60); op.iterate(10); op.get_vector('x'); op.get_vector('w')
    timing(tag)
