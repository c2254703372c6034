import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
gb = float(sys.argv[1]); keep = int(sys.argv[2])
ball = None
if gb > 0:
    ball = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device='cuda'); ball.zero_(); torch.cuda.synchronize()
    if not keep:
        del ball; torch.cuda.empty_cache()
wl = P.WORKLOADS['s4b']; A = wl['make'](); n = A.shape[0]; b, x0, _ = P.reference_rhs(A, n)
out = []
for i in range(2):
    op = DeviceCSR(A); op.begin(L.PIPE_PR, b, x0, 152); op.iterate(50); op.sync()
    t0 = time.perf_counter(); op.iterate(100); op.sync(); out.append((time.perf_counter() - t0) / 100 * 1e6); op.close()
print('ballast GB', gb, 'kept' if keep else 'freed', ' '.join(f'{v:.1f}' for v in out), flush=True)
