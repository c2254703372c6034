"""What one launch of the one-launch pipelined iteration costs beyond its tiles: us per iteration of a band (ex2b, 15
diagonals) of k x 262,144 rows = k tiles per resident wave (4096 waves), k = 1 .. 6 -- intercept = kernel boundary +
prologue + first-tile latency, slope = one round of tiles.  usage: fixed_cost.py [KNOB=val ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

knobs = dict(kv.split('=') for kv in sys.argv[1:])
for k in (1, 2, 3, 4, 5, 6, 8):
    n = 262144 * k
    A = P.banded_ex2b(n, 7)
    b, x0, _ = P.reference_rhs(A, n)
    op = DeviceCSR(A, knobs=knobs)
    iters = 2000
    op.begin(L.PIPE_PR, b, x0, iters + 401)
    op.iterate(400); op.sync()
    t0 = time.perf_counter(); op.iterate(iters); op.sync(); dt = time.perf_counter() - t0
    lay = op.layout()
    print(f'n = {n:8d}  tiles {lay["tiles"].shape[0]:6d}  grid {lay["grid"]:5d} x {lay["waves_per_block"]} waves   {dt / iters * 1e6:7.2f} us/iteration', flush=True)
    op.close()
