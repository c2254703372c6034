import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem
knobs = dict(kv.split('=') for kv in sys.argv[1:])
A = P.banded_ex2b(1_250_000, 7); n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
A_loop, halo, moved = loopback_problem(A, 9)
op = DeviceCSR(A_loop, comm_init=(0, 1, rccl_ids(1)[0], L.default_rccl_path()), halo=halo, knobs=knobs)
op.begin(L.PIPE_PR, b, x0, 1601); s = op.schedule()
op.iterate(400); op.sync()
t0 = time.perf_counter(); op.iterate(1200); tq = time.perf_counter() - t0; op.sync(); dt = time.perf_counter() - t0
print('s3_8th with loopback halo (boundary tiles)', knobs, 'fused_comm' if s['fused_comm'] else 'two-kernel', '%.1f us/iteration (host enqueue %.1f)' % (dt / 1200 * 1e6, tq / 1200 * 1e6))
