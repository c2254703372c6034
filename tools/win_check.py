"""Quick GPU check of the window kernels against SciPy and against the CSR-adaptive kernels."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR

rng = np.random.default_rng(3)
cases = {'band300k': P.banded_ex2b(300_000, 7), 'lap2d': P.laplace_2d(400, 300), 'lap3d': P.laplace_3d(60, 50, 40),
         'band_small': P.banded_ex2b(20000, 7), 'lap_tiny': P.laplace_2d(64, 48)}
for name, A in cases.items():
    n = A.shape[0]
    x = rng.standard_normal(n)
    ref = A @ x
    for knobs in ({}, {'PRCG_VALDICT': '0'}, {'PRCG_WIN': '0'}):
        op = DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        y, ms = op.matvec(x, reps=3)
        WU, ms2 = op.matmat2(np.stack([x, -2.0 * x], axis=1), reps=3)
        ok = np.array_equal(y, ref) and np.array_equal(WU[:, 0], ref) and np.array_equal(WU[:, 1], A @ (-2.0 * x))
        print(name, knobs, 'window' if s['window'] else 'classic', 'dict' if s['value_dict'] else 'plain', 'col_bytes', s['col_bytes'],
              'OK' if ok else 'MISMATCH', f'{ms*1e3:.1f} us {ms2*1e3:.1f} us', flush=True)
        if not ok:
            bad = np.nonzero(y != ref)[0]
            print('  first bad rows', bad[:10], 'count', bad.size)
        # fused pipelined iterations vs two-kernel schedule
        b, x0, xt = P.reference_rhs(A, n)
        op2 = DeviceCSR(A, knobs=dict(knobs, PRCG_FUSED='0'))
        hs = []
        for o in (op, op2):
            o.begin(L.PIPE_PR, b, x0, 12, hist_mask=1)
            o.iterate(11); o.sync()
            hs.append(o.history()['updated_residual_2_norm'])
        dev = np.max(np.abs(hs[0][:8] - hs[1][:8]) / hs[1][:8])
        print('   fused vs two-kernel, rel dev of the residual history on k<8:', f'{dev:.2e}', 'fused' if op.schedule()['fused'] else '')
        op.close(); op2.close()
