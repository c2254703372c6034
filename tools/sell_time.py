"""us per iteration of the one-launch pipelined iteration on one of the Queen_4147 stand-ins, one operator, several knob sets:
   sell_time.py <workload> [iters=K] [variant=pipe_pr_cg] CFG [CFG ...]     CFG = KNOB=val,KNOB=val   or   -   (defaults)
Generates the matrix once.  Prints one JSON line per configuration.  Under `rocprofv3 --pmc ...` every configuration issues
exactly warm + K launches of the iteration kernel (tools/sell_pmc.py splits the dispatches accordingly)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

args = sys.argv[1:]
if args[0].startswith('fem:'):          # fem:<m> -- the s4b stand-in on m^3 nodes (sizes between the committed workloads)
    _m = int(args[0][4:])
    wl = {'make': lambda: P.fem_like_3d(_m), 'desc': f'FEM-like stand-in on {_m}^3 nodes'}
else:
    wl = P.WORKLOADS[args[0]]
iters, warm, variant, prof, prewarm = 100, 30, 'PIPE_PR', 0, 0
cfgs = []
for a in args[1:]:
    if a.startswith('iters='):
        iters = int(a[6:])
    elif a.startswith('warm='):
        warm = int(a[5:])
    elif a.startswith('prof='):
        prof = int(a[5:])
    elif a.startswith('prewarm='):
        prewarm = int(a[8:])
    elif a.startswith('variant='):
        variant = a[8:]
    else:
        cfgs.append({} if a == '-' else dict(kv.split('=') for kv in a.split(',')))
t0 = time.perf_counter()
A = wl['make']()
n, nnz = A.shape[0], int(A.nnz)
b, x0, _ = P.reference_rhs(A, n)
print(f'# {wl["desc"]}: n = {n} nnz = {nnz}, generated in {time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
for knobs in cfgs:
    t0 = time.perf_counter()
    op = DeviceCSR(A, knobs=knobs)
    setup = time.perf_counter() - t0
    if prewarm:
        op.begin(getattr(L, variant), b, x0, prewarm + 1); op.iterate(prewarm); op.sync()
    op.begin(getattr(L, variant), b, x0, warm + iters + 2)
    op.iterate(warm); op.sync()
    if prof:
        op.set_profiling(prof)
    t0 = time.perf_counter(); op.iterate(iters); op.sync(); dt = time.perf_counter() - t0
    s = op.schedule()
    opb = op.operator_bytes()
    moved = opb + 64 * n
    fin = bool(np.isfinite(op.get_scalars(warm + iters)[L.S_NU]))
    print(json.dumps({'workload': args[0], 'knobs': knobs, 'us_per_iteration': dt / iters * 1e6, 'its_per_s': iters / dt,
                      'operator_bytes': opb, 'bytes_per_nnz': opb / nnz, 'moved_GB': moved * 1e-9,
                      'moved_TBps': moved / (dt / iters) * 1e-12, 'frac_of_8TBps': moved / (dt / iters) * 1e-12 / 8.0,
                      'sliced': s.get('sliced_rows'), 'window': s['window'], 'fused': s['fused'], 'finite': fin,
                      'launches': warm + iters, 'setup_s': setup}), flush=True)
    op.close()
