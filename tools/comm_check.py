"""One GPU: the one-launch schedule with a 1-rank RCCL communicator (deferred form) vs the plain one."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
from test_distributed import rccl_ids, loopback_problem

for name, nmat in (('band', P.banded_ex2b(400_000, 7)), ('lap2d', P.laplace_2d(500, 400))):
    A = nmat
    n = A.shape[0]
    b, x0, xt = P.reference_rhs(A, n)
    uid, path = rccl_ids(1)
    plain = DeviceCSR(A)
    comm = DeviceCSR(A, comm_init=(0, 1, uid, path))
    two = DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    for variant, inv in ((L.PIPE_PR, None), (L.PIPE_P, None), (L.PIPE_PR, 1 / A.diagonal())):
        hs = []
        for op in (plain, comm, two):
            op.begin(variant, b, x0, 60, x_true=xt, inv_diag=inv, hist_mask=15)
            s = op.schedule()
            op.iterate(59); op.sync()
            hs.append((op.history(), op.get_vector('x'), s))
        print(name, variant, 'prec' if inv is not None else 'none', 'schedules:', [(h[2]['fused'], h[2]['fused_comm'], h[2]['gather']) for h in hs])
        for q in hs[0][0]:
            d1 = np.max(np.abs(hs[1][0][q][:10] - hs[0][0][q][:10]) / np.abs(hs[0][0][q][:10]))
            d2 = np.max(np.abs(hs[2][0][q][:10] - hs[0][0][q][:10]) / np.abs(hs[0][0][q][:10]))
            print('   ', q, 'comm-vs-plain k<10: %.2e   two-kernel-vs-plain: %.2e' % (d1, d2))
    # loopback halo: boundary tiles + ghosts through the merged gather
    A_loop, halo, moved = loopback_problem(A, 9 if name == 'band' else 600)
    uid, path = rccl_ids(1)
    loop = DeviceCSR(A_loop, comm_init=(0, 1, uid, path), halo=halo)
    for variant in (L.PIPE_PR,):
        outs = []
        for op in (plain, loop):
            op.begin(variant, b, x0, 60, x_true=xt, hist_mask=15)
            s = op.schedule()
            op.iterate(59); op.sync()
            outs.append((op.history(), s))
        print(name, 'loopback schedule', {k: outs[1][1][k] for k in ('fused', 'fused_comm', 'gather', 'comm')})
        for q in outs[0][0]:
            print('   ', q, 'loopback-vs-plain k<10: %.2e' % np.max(np.abs(outs[1][0][q][:10] - outs[0][0][q][:10]) / np.abs(outs[0][0][q][:10])))
    # timing
    for label, op in (('plain fused', plain), ('comm deferred', comm), ('two-kernel', two), ('loopback deferred', loop)):
        op.begin(L.PIPE_PR, b, x0, 2001)
        op.iterate(500); op.sync()
        t0 = time.perf_counter(); op.iterate(1500); op.sync(); dt = time.perf_counter() - t0
        print('   ', label, '%.1f us/iteration' % (dt / 1500 * 1e6))
    for op in (plain, comm, two, loop):
        op.close()
