"""us per iteration of the one-launch pipelined iteration on a constant-coefficient Laplacian of a given grid:
   stencil_time.py nx ny [nz] [KNOB=val ...]   (what the far neighbours of a 3-D stencil cost: compare 216 216 216 with 3175 3174)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

dims = [int(a) for a in sys.argv[1:] if '=' not in a]
knobs = dict(kv.split('=') for kv in sys.argv[1:] if '=' in kv)
A = P.laplace_3d(*dims) if len(dims) == 3 else P.laplace_2d(*dims)
n = A.shape[0]
b, x0, _ = P.reference_rhs(A, n)
op = DeviceCSR(A, knobs=knobs)
iters = 600
op.begin(L.PIPE_PR, b, x0, iters + 201)
op.iterate(200); op.sync()
t0 = time.perf_counter(); op.iterate(iters); op.sync(); dt = time.perf_counter() - t0
s = op.schedule(); lay = op.layout()
print(f'{dims} n = {n} nnz = {A.nnz} {knobs}: pattern {s["pattern"]} geometry {lay["geometry"]} grid {lay["grid"]} x {lay["waves_per_block"]}   '
      f'{dt / iters * 1e6:8.2f} us/iteration   {64 * n / (dt / iters) * 1e-12:5.2f} TB/s on 64 B per row', flush=True)
op.close()
