"""Is the speed of the one-launch iteration a property of the ALLOCATION?  Several operators of the same matrix alive at once in one
process (different physical memory each), each timed:  placement_probe.py <workload> [instances=4] [iters=100]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
wl = P.WORKLOADS[sys.argv[1]]
opt = dict(a.split('=') for a in sys.argv[2:])
inst, iters = int(opt.pop('instances', 4)), int(opt.pop('iters', 100))
A = wl['make'](); n = A.shape[0]
b, x0, _ = P.reference_rhs(A, n)
ops = []
for i in range(inst):
    op = DeviceCSR(A, knobs=opt)
    ops.append(op)
    out = []
    for o in ops:                       # every instance alive so far, timed again
        o.begin(L.PIPE_PR, b, x0, iters + 52)
        o.iterate(50); o.sync()
        t0 = time.perf_counter(); o.iterate(iters); o.sync()
        out.append((time.perf_counter() - t0) / iters * 1e6)
    print(f'{sys.argv[1]}: {i + 1} instance(s) alive:', ' '.join(f'{v:.1f}' for v in out), flush=True)
for o in ops:
    o.close()
