"""us per iteration of the one-launch pipelined iteration on S3 (the ex2b band, n = 1e7 unless given), for the timing builds of
tools/decomp_win.sh:   s3_time.py [n] [KNOB=val ...]   (PRCG_VALDICT=0: the plain-values twin, the general-CSR leg of bench.py)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

args = [a for a in sys.argv[1:] if '=' not in a]
n = int(args[0]) if args else 10_000_000
knobs = dict(kv.split('=') for kv in sys.argv[1:] if '=' in kv)
A = P.WORKLOADS['s3']['make']() if n == 10_000_000 else P.banded_ex2b(n, 7)
b, x0, _ = P.reference_rhs(A, n)
op = DeviceCSR(A, knobs=knobs)
iters = 300
op.begin(L.PIPE_PR, b, x0, iters + 101)
op.iterate(100); op.sync()
t0 = time.perf_counter(); op.iterate(iters); op.sync(); dt = time.perf_counter() - t0
s = op.schedule(); lay = op.layout()
print(f'n = {n} {knobs}: dictionary {s["value_dict"]} geometry {lay["geometry"]} grid {lay["grid"]} x {lay["waves_per_block"]}   '
      f'{dt / iters * 1e6:8.2f} us/iteration   operator {op.operator_bytes() * 1e-9:.3f} GB', flush=True)
op.close()
