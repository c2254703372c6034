"""Soak runs of the one-launch schedules (a rare protocol race shows up as a bounded-wait error or a non-finite residual):
   soak.py [iterations]   -- S3 plain schedule, one eighth of S3 through the peer exchange with a loopback halo, one eighth of S2 the same."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from new_cg_variants_amd import problems as P, _lib as L, partition
from new_cg_variants_amd.device import DeviceCSR

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100000


def uid():
    u = np.zeros(128, dtype=np.uint8)
    L.check(None, L.lib().prcg_comm_unique_id(L.default_rccl_path().encode(), L.ptr(u)))
    return u.tobytes()


def soak(name, op, b, x0):
    # the residual stagnates at rounding level long before; what is checked: no error, finite scalars, steady rate
    chunk = 20000
    op.begin(L.PIPE_PR, b, x0, chunk + 2)
    t0 = time.perf_counter()
    done = 0
    while done < iters:
        op.begin(L.PIPE_PR, b, x0, chunk + 2)
        op.iterate(chunk)
        op.sync()                                   # raises on a bounded-wait timeout
        nu = op.get_scalars(chunk)[L.S_NU]
        assert np.isfinite(nu), (name, done, nu)
        done += chunk
    dt = time.perf_counter() - t0
    print(f'{name}: {done} iterations, {dt / done * 1e6:.1f} us per iteration, last nu {nu:.3e}', flush=True)
    op.close()


A = P.WORKLOADS['s3']['make'](); b, x0, _ = P.reference_rhs(A, A.shape[0])
soak('S3, one-launch schedule', DeviceCSR(A), b, x0)
for wl, k, cut in (('s3_8th', 7, None), ('s2_8th', 216 * 216, 13 * 216 * 216)):
    A = P.WORKLOADS[wl]['make'](); b, x0, _ = P.reference_rhs(A, A.shape[0])
    A_loop, halo, _ = partition.loopback_problem(A, k, cut=cut)
    op = DeviceCSR(A_loop, comm_init=(0, 1, uid(), L.default_rccl_path()), halo=halo)
    assert partition.connect_peer_exchange(op, 0, lambda o: [o])
    soak(f'{wl}, peer exchange with loopback halo', op, b, x0)
