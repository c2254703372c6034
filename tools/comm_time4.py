import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem
mode = sys.argv[1]
A = P.laplace_2d(500, 400)
n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
def timing(op, tag):
    op.begin(L.PIPE_PR, b, x0, 1001)
    t0 = time.perf_counter(); op.iterate(600)
    try:
        op.sync(); err = ''
    except Exception as e:
        err = 'TIMEOUT'
    print(mode, tag, op.schedule()['gather'], '%.1f us/iteration' % ((time.perf_counter() - t0) / 600 * 1e6), err, flush=True)
uid, path = rccl_ids(1)
op = DeviceCSR(A, comm_init=(0, 1, uid, path))
timing(op, 'first comm alone')
if mode == 'two_comms':
    uid2, path = rccl_ids(1)
    op2 = DeviceCSR(A, comm_init=(0, 1, uid2, path))
    timing(op, 'first comm, second alive')
    timing(op2, 'second comm')
if mode == 'loop':
    A_loop, halo, moved = loopback_problem(A, 600)
    uid2, path = rccl_ids(1)
    op2 = DeviceCSR(A_loop, comm_init=(0, 1, uid2, path), halo=halo)
    timing(op, 'first comm, loop alive')
    timing(op2, 'loop (send/recv + allreduce)')
if mode == 'loop_only':
    A_loop, halo, moved = loopback_problem(A, 600)
    op.close()
    uid2, path = rccl_ids(1)
    op2 = DeviceCSR(A_loop, comm_init=(0, 1, uid2, path), halo=halo)
    timing(op2, 'loop (send/recv + allreduce)')
    op3 = DeviceCSR(A_loop, comm_init=(0, 1, rccl_ids(1)[0], path), halo=halo, knobs={'PRCG_FUSED_COMM': '0'})
    timing(op3, 'loop two-kernel')
