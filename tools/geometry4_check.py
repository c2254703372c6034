import sys, numpy as np
sys.path.insert(0, '/root/repo')
from new_cg_variants_amd import problems as P, _lib as L
from new_cg_variants_amd.device import DeviceCSR
rng = np.random.default_rng(0)
for name, A in (('lap3d 70^3', P.laplace_3d(70, 70, 70)), ('lap3d 216x40x30', P.laplace_3d(216, 40, 30)), ('lap2d 1000x300', P.laplace_2d(1000, 300))):
    x = rng.standard_normal(A.shape[0])
    for knobs in ({'PRCG_WIN_ROWS': '64'}, {'PRCG_WIN_ROWS': '64', 'PRCG_VALDICT': '0'}):
        op = DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        y, _ = op.matvec(x)
        WU, _ = op.matmat2(np.stack([x, -x[::-1]], axis=1))
        ok = np.array_equal(y, A @ x) and np.array_equal(WU[:, 1], A @ (-x[::-1]))
        b, x0, xt = P.reference_rhs(A, A.shape[0])
        op.begin(L.PIPE_PR, b, x0, 40, hist_mask=1); op.iterate(39); op.sync(); h1 = op.history()['updated_residual_2_norm']
        op.close()
        ref = DeviceCSR(A, knobs={k: v for k, v in knobs.items() if k != 'PRCG_WIN_ROWS'})
        ref.begin(L.PIPE_PR, b, x0, 40, hist_mask=1); ref.iterate(39); ref.sync(); h2 = ref.history()['updated_residual_2_norm']
        ref.close()
        print(name, knobs, 'window', s['window'], 'col_bytes', s['col_bytes'], 'bit-exact products', ok, 'history rel dev', float(np.max(np.abs(h1 - h2) / h2)))
