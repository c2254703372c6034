"""One rank's share of S3 on 8 GPUs (n = 1.25e6, loopback halo) on ONE GPU: us per iteration of the plain one-launch
schedule and of the multi-rank schedules (direct peer exchange; RCCL chains).  usage: slice_comm_time.py [n=rows] [only=plain|nohalo|halo] [KNOB=val ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: its HIP runtime is the process's runtime, as in bench.py)
from new_cg_variants_amd import problems as P, _lib as L, partition
from new_cg_variants_amd.device import DeviceCSR

knobs = dict(kv.split('=') for kv in sys.argv[1:])
only = knobs.pop('only', None)        # only=plain | nohalo | halo: just that configuration (for a kernel trace)
n_rows = int(knobs.pop('n', 1_250_000))   # n=5000000: one half of S3
A = P.banded_ex2b(n_rows, 7); n = A.shape[0]
b, x0, xt = P.reference_rhs(A, n)
A_loop, halo, moved = partition.loopback_problem(A, 7)


def uid():
    u = np.zeros(128, dtype=np.uint8)
    L.check(None, L.lib().prcg_comm_unique_id(L.default_rccl_path().encode(), L.ptr(u)))
    return u.tobytes()


def run(name, op, iters=600):
    op.begin(L.PIPE_PR, b, x0, iters + 401); s = op.schedule()
    op.iterate(400); op.sync()
    t0 = time.perf_counter(); op.iterate(iters); tq = time.perf_counter() - t0; op.sync(); dt = time.perf_counter() - t0
    kind = 'peer exchange' if s['peer'] else ('rccl one-launch' if s['fused_comm'] else ('one launch' if s['fused'] else 'two-kernel'))
    print(f'{name:34s} {kind:16s} {dt / iters * 1e6:7.1f} us/iteration   host enqueue {tq / iters * 1e6:5.1f} us', flush=True)
    op.close()


if only in (None, 'plain'):
    run('plain (no communicator)', DeviceCSR(A, knobs=knobs))
if only in (None, 'nohalo'):
    op = DeviceCSR(A, comm_init=(0, 1, uid(), L.default_rccl_path()), knobs=knobs)
    partition.connect_peer_exchange(op, 0, lambda o: [o])
    run('no halo, peer exchange', op)
if only in (None, 'halo'):
    op = DeviceCSR(A_loop, comm_init=(0, 1, uid(), L.default_rccl_path()), halo=halo, knobs=knobs)
    partition.connect_peer_exchange(op, 0, lambda o: [o])
    run('loopback halo, peer exchange', op)
if only is None and n_rows <= 1_250_000:
    run('loopback halo, RCCL one-launch', DeviceCSR(A_loop, comm_init=(0, 1, uid(), L.default_rccl_path()), halo=halo, knobs=dict(knobs, PRCG_FUSED_COMM='1')))
if only is None:
    run('loopback halo, RCCL two-kernel', DeviceCSR(A_loop, comm_init=(0, 1, uid(), L.default_rccl_path()), halo=halo, knobs=dict(knobs, PRCG_FUSED_COMM='0')))
