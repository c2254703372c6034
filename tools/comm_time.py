import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids
which, knobs = sys.argv[1], dict(kv.split('=') for kv in sys.argv[2:])
A = {'lap': lambda: P.laplace_2d(500, 400), 'band': lambda: P.banded_ex2b(400_000, 7), 's3_8th': lambda: P.banded_ex2b(1_250_000, 7),
     's1': lambda: P.laplace_2d(1000, 1000), 'lap3d': lambda: P.laplace_3d(108, 108, 108)}[which]()
n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
uid, path = rccl_ids(1)
op = DeviceCSR(A, comm_init=(0, 1, uid, path), knobs=knobs)
op.begin(L.PIPE_PR, b, x0, 1601)
s = op.schedule()
op.iterate(400); op.sync()
t0 = time.perf_counter(); op.iterate(1200)
tq = time.perf_counter() - t0
try:
    op.sync(); err = ''
except Exception as e:
    err = 'TIMEOUT'
dt = time.perf_counter() - t0
print(which, knobs, os.environ.get('PRCG_DEBUG_NOWAIT', ''), 'fused_comm' if s['fused_comm'] else 'two-kernel', 'gather' if s['gather'] else 'allreduce',
      '%.1f us/iteration (host enqueue %.1f)' % (dt / 1200 * 1e6, tq / 1200 * 1e6), err, flush=True)
