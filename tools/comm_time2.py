import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids
mode = sys.argv[1]
A = P.laplace_2d(500, 400)
n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
extra = []
if 'idle' in mode:
    extra = [DeviceCSR(A) for _ in range(int(mode.split('idle')[1] or 1))]
if 'used' in mode:
    extra = [DeviceCSR(A)]
    extra[0].begin(L.PIPE_PR, b, x0, 100); extra[0].iterate(99); extra[0].sync()
uid, path = rccl_ids(1)
op = DeviceCSR(A, comm_init=(0, 1, uid, path))
for rep in range(3):
    hm = 15 if ('hist' in mode and rep == 0) else 0
    op.begin(L.PIPE_PR, b, x0, 1601, x_true=xt, hist_mask=hm)
    t0 = time.perf_counter(); op.iterate(800 if hm == 0 else 60)
    try:
        op.sync(); err = ''
    except Exception as e:
        err = 'TIMEOUT'
    dt = time.perf_counter() - t0
    print(mode, 'session', rep, 'hist', hm, '%.1f us/iteration' % (dt / (800 if hm == 0 else 60) * 1e6), err, flush=True)
