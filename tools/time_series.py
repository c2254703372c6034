"""us per iteration of the one-launch pipelined iteration in consecutive chunks of one long run (clock / power state over time):
   time_series.py <workload> [chunk=100] [chunks=30] [idle_ms=0] [KNOB=val ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
wl = P.WORKLOADS[sys.argv[1]]
opt = dict(a.split('=') for a in sys.argv[2:])
chunk, chunks, idle = int(opt.pop('chunk', 100)), int(opt.pop('chunks', 30)), float(opt.pop('idle_ms', 0))
A = wl['make'](); n = A.shape[0]
b, x0, _ = P.reference_rhs(A, n)
op = DeviceCSR(A, knobs=opt)
op.begin(L.PIPE_PR, b, x0, chunk * chunks + 2)
out = []
for c in range(chunks):
    if idle and c == chunks // 2:
        time.sleep(idle * 1e-3)
    t0 = time.perf_counter(); op.iterate(chunk); op.sync(); out.append((time.perf_counter() - t0) / chunk * 1e6)
print(sys.argv[1], 'chunk', chunk, ' '.join(f'{v:.0f}' for v in out), 'finite', bool(np.isfinite(op.get_scalars(chunk * chunks)[L.S_NU])), flush=True)
op.close()
