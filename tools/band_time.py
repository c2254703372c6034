"""us per iteration of the one-launch pipelined iteration on an ex2b band with a CONSTANT diagonal (kappa = 1: a pattern operator)
   against the same band through the stream geometries: band_time.py n [KNOB=val ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

n = int(sys.argv[1])
knobs = dict(kv.split('=') for kv in sys.argv[2:])
A = P.banded_ex2b(n, 7, kappa=1.0)
b, x0, _ = P.reference_rhs(A, n)
for extra in ({}, {'PRCG_WIN_PAT': '0'}):
    kn = dict(knobs, **extra)
    op = DeviceCSR(A, knobs=kn)
    iters = 600
    op.begin(L.PIPE_PR, b, x0, iters + 201)
    op.iterate(200); op.sync()
    t0 = time.perf_counter(); op.iterate(iters); op.sync(); dt = time.perf_counter() - t0
    s = op.schedule(); lay = op.layout()
    print(f'n = {n} {kn}: pattern {s["pattern"]} geometry {lay["geometry"]} grid {lay["grid"]} x {lay["waves_per_block"]}   '
          f'{dt / iters * 1e6:8.2f} us/iteration   {64 * n / (dt / iters) * 1e-12:5.2f} TB/s on 64 B per row', flush=True)
    op.close()
