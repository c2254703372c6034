"""GPU (MI355X): the HIP path through the C-ABI against the oracle and the golden
fixtures.  Run with ``pytest -m gpu``.

Bars (stated here, used below):
  * SpMV / SpMM with short rows: BIT-EXACT vs SciPy's csr_matvec (same left-to-right
    order, no FMA);
  * teacher-forced single iteration from every stored reference state: vectors and
    scalars <= 1e-12 relative (north_star's tolerance);
  * free-running histories: the first k at which the recurrence residual leaves 1e-12 of the reference's is
    MEASURED per run, printed, and must not come earlier than PREFIX_FLOOR (k = 7 bcsstk03 / 15 nos7; BASELINE.md
    section 2 has 9 / 16 for one re-ordering of the reference's own sums).  Beyond it any change of summation order diverges on these
    ill-conditioned problems: the rest is held to convergence-level agreement, ONE
    rule: the paper's two statistics (figure_gen.py:86-89) must lie inside the spread the reference's
    own algorithm shows when only the order of its summations changes, computed in the test with
    six summation orders of oracle/ne_oracle.py (margins: 2 % of the iteration count, 0.5 decades);
  * against the oracle run with the device's own reduction order (tests/device_order.py): the whole
    free-running trajectory of the ONE-LAUNCH schedule (the default) and of the two-kernel schedule, all four
    inner products of every iteration bit-exact, <= 1e-13 on every recorded norm.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import all_runs, golden_state, load_run
from device_order import OneLaunchTree, device_dot
from oracle import mp_oracle
from oracle import ne_oracle as orc

pytestmark = pytest.mark.gpu

FOUR = ['error_A_norm', 'residual_2_norm', 'error_2_norm', 'updated_residual_2_norm']


@pytest.fixture(scope='module')
def amd():
    import new_cg_variants_amd.cg_variants as cgv
    import new_cg_variants_amd.callbacks as cbs
    from new_cg_variants_amd import _lib, device, problems
    return dict(cgv=cgv, cbs=cbs, L=_lib, device=device, problems=problems)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    s = np.linalg.norm(b)
    d = np.linalg.norm(a - b)
    return d / s if s > 0 else d


def srel(a, b):
    return abs(a - b) / abs(b) if b != 0 else abs(a)


# ---------------------------------------------------------------------------------------
# SpMV / SpMM
# ---------------------------------------------------------------------------------------
def test_spmv_bitexact_on_golden_matrices(amd, matrices):
    for name in sorted({r[0] for r in all_runs()}):
        A, z = matrices[name]
        op = amd['device'].DeviceCSR(A)
        for i in range(z['spmv_x'].shape[0]):
            y, _ = op.matvec(z['spmv_x'][i])
            assert np.array_equal(y, z['spmv_y'][i]), name
        RS = np.stack([z['spmv_x'][0], z['spmv_x'][1]], axis=1)
        WU, _ = op.matmat2(RS)
        assert np.array_equal(WU[:, 0], z['spmv_y'][0]) and np.array_equal(WU[:, 1], z['spmv_y'][1])
        op.close()


@pytest.mark.parametrize('name', ['s1_small', 's3_small'])
def test_spmv_bitexact_on_synthetic(amd, name):
    A = amd['problems'].WORKLOADS[name]['make']()
    rng = np.random.default_rng(11)
    x = rng.standard_normal(A.shape[0])
    op = amd['device'].DeviceCSR(A)
    y, _ = op.matvec(x, reps=3)
    assert np.array_equal(y, A @ x)
    op.close()


def test_spmv_ragged_empty_and_long_rows(amd):
    """Edge cases: empty rows, rows of exactly/over a tile, one very long row, unsorted
    indices, an all-empty matrix, a 1x1 matrix."""
    rng = np.random.default_rng(5)
    n = 6000
    lens = rng.integers(0, 30, size=n)
    lens[10] = 509
    lens[11] = 510
    lens[12] = 4000
    lens[300:900] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(indptr[-1])
    indices = rng.integers(0, n, size=nnz).astype(np.int32)      # unsorted, with duplicates
    data = rng.standard_normal(nnz)
    A = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    x = rng.standard_normal(n)
    ref = A @ x                                                  # scipy: sequential per row
    op = amd['device'].DeviceCSR(A)
    y, _ = op.matvec(x)
    short = lens <= 509
    assert np.array_equal(y[short], ref[short])                  # tile path: bit-exact
    scale = abs(A) @ np.abs(x)
    assert np.all(np.abs(y[~short] - ref[~short]) <= 1e-14 * scale[~short])   # wave-reduced rows
    RS = np.stack([x, -2.0 * x], axis=1)
    WU, _ = op.matmat2(RS)
    assert np.array_equal(WU[short, 0], ref[short]) and np.array_equal(WU[short, 1], -2.0 * ref[short])
    op.close()
    for tiny in (sp.csr_matrix((4, 4)), sp.csr_matrix(np.array([[2.5]]))):
        op = amd['device'].DeviceCSR(tiny.astype(np.float64))
        v = np.arange(1.0, tiny.shape[0] + 1)
        y, _ = op.matvec(v)
        assert np.array_equal(y, tiny @ v)
        op.close()


def test_spmv_column_encodings_agree(amd):
    """The CSR-adaptive kernels (PRCG_WIN=0; operators that are no band or stencil use them anyway)
    stream the column indices as 8- or 16-bit offsets from each tile's smallest
    column when every tile's columns span < 256 resp. < 65536, else as the int32 it was given.
    All are the same indices: bit-identical products.  Covers: banded (8-bit), wide random
    (int32 fallback), a matrix whose tiles mix spans (fallback), and the knobs that force the
    wider encodings."""
    rng = np.random.default_rng(9)
    P = amd['problems']
    cases = {'banded': P.banded_ex2b(300_000, 7)}
    n = 200_000
    lens = rng.integers(1, 12, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = rng.integers(0, n, size=int(indptr[-1])).astype(np.int32)
    cases['wide'] = sp.csr_matrix((rng.standard_normal(cols.size), cols, indptr), shape=(n, n))
    near = (np.repeat(np.arange(n), lens) + rng.integers(-50, 50, size=cols.size)).clip(0, n - 1).astype(np.int32)
    near[indptr[n // 2]:indptr[n // 2] + 5] = [0, n - 1, 7, n - 3, 11]          # one tile with a huge span
    cases['mixed'] = sp.csr_matrix((rng.standard_normal(cols.size), near, indptr), shape=(n, n))
    for name, A in cases.items():
        x = rng.standard_normal(A.shape[0])
        ref = A @ x
        outs = []
        for knobs in (None, {'PRCG_WIN': '0'}, {'PRCG_WIN': '0', 'PRCG_COL8': '0'}, {'PRCG_WIN': '0', 'PRCG_COL16': '0'}):
            op = amd['device'].DeviceCSR(A, knobs=knobs)
            y, _ = op.matvec(x)
            RS = np.stack([x, 0.5 * x], axis=1)
            WU, _ = op.matmat2(RS)
            outs.append((y, WU))
            op.close()
        assert np.array_equal(outs[0][0], ref), name
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), name
        assert np.array_equal(outs[0][1][:, 0], ref) and np.array_equal(outs[0][1][:, 1], A @ (0.5 * x)), name


def test_spmv_value_dictionary_is_lossless(amd):
    """CSR-adaptive kernels (PRCG_WIN=0; the window kernels' dictionary: tests/test_gpu_configs.py).
    Where every tile holds <= 64 distinct values (stencils, constant off-diagonals) the device
    streams 1-byte dictionary indices instead of the doubles; the dictionary entries ARE the
    doubles, so products must be bit-identical to SciPy and to the plain stream.  Covers: the
    ex2b band (8-bit columns + dictionary), a 5-point stencil (16-bit columns + dictionary),
    quantised random values, one tile with too many distinct values (whole class falls back)
    and signed zeros (distinct bit patterns)."""
    rng = np.random.default_rng(10)
    P = amd['problems']
    n = 120_000
    lens = rng.integers(1, 12, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    near = (np.repeat(np.arange(n), lens) + rng.integers(-50, 50, size=int(indptr[-1]))).clip(0, n - 1).astype(np.int32)
    quant = rng.integers(-3, 4, size=near.size) / 8.0
    quant[::7] = -0.0
    spoiled = quant.copy()
    spoiled[indptr[n // 3]:indptr[n // 3] + 200] = rng.standard_normal(200)      # > 64 distinct values in one tile
    cases = {'banded': (P.banded_ex2b(300_000, 7), True, 1), 'stencil': (P.laplace_2d(400, 300), True, 2),
             'quantised': (sp.csr_matrix((quant, near, indptr), shape=(n, n)), True, 1),
             'spoiled': (sp.csr_matrix((spoiled, near, indptr), shape=(n, n)), False, 1)}
    for name, (A, expect, col_bytes) in cases.items():
        x = rng.standard_normal(A.shape[0])
        ref = A @ x
        outs = []
        for knobs in ({'PRCG_WIN': '0'}, {'PRCG_WIN': '0', 'PRCG_VALDICT': '0'}):
            op = amd['device'].DeviceCSR(A, knobs=knobs)
            sched = op.schedule()
            assert sched['value_dict'] == (expect and 'PRCG_VALDICT' not in knobs), (name, knobs, sched)
            assert sched['col_bytes'] == col_bytes, (name, sched)
            y, _ = op.matvec(x)
            WU, _ = op.matmat2(np.stack([x, -3.0 * x], axis=1))
            outs.append((y, WU))
            op.close()
        assert np.array_equal(outs[0][0], ref), name
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), name
        assert np.array_equal(outs[0][1][:, 0], ref) and np.array_equal(outs[0][1][:, 1], A @ (-3.0 * x)), name
    # whole solves with and without the dictionary: same vectors from the same products; the inner
    # products of the one-launch schedule are summed per wave over the tiles it processes, and the two
    # kernels run different grids (different register footprints), so the histories agree to rounding
    # on a prefix rather than bit for bit
    A = P.banded_ex2b(200_000, 7)
    b, x0, _ = P.reference_rhs(A, A.shape[0])
    hist = []
    for knobs in (None, {'PRCG_VALDICT': '0'}, {'PRCG_WIN': '0'}, {'PRCG_WIN': '0', 'PRCG_VALDICT': '0'}):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        for variant in (amd['L'].PIPE_PR, amd['L'].HS):
            op.begin(variant, b, x0, 12, hist_mask=amd['L'].HIST_BITS['updated_residual_2_norm'])
            op.iterate(11)
            op.sync()
            hist.append(op.history()['updated_residual_2_norm'])
        op.close()
    for i in range(2, 8):
        np.testing.assert_allclose(hist[i % 2][:8], hist[i][:8], rtol=1e-12)


def test_spmv_full_size_properties(amd):
    """S1 at full size (n=1e6): bit-exact vs SciPy, plus size-independent properties:
    A*1 = row sums, symmetry x'(Ay) = y'(Ax), linearity."""
    A = amd['problems'].laplace_2d(1000, 1000)
    n = A.shape[0]
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    op = amd['device'].DeviceCSR(A)
    Ax, _ = op.matvec(x)
    Ay, _ = op.matvec(y)
    assert np.array_equal(Ax, A @ x)
    ones, _ = op.matvec(np.ones(n))
    assert np.array_equal(ones, np.asarray(A.sum(axis=1)).ravel())
    assert abs(x @ Ay - y @ Ax) <= 1e-11 * np.linalg.norm(x) * np.linalg.norm(Ay)
    Axy, _ = op.matvec(2.0 * x + y)
    assert rel(Axy, 2.0 * Ax + Ay) <= 1e-15
    op.close()


# ---------------------------------------------------------------------------------------
# teacher-forced single steps from the reference's stored iterates
# ---------------------------------------------------------------------------------------
VARIANT_OF = {'hs_cg': 'HS', 'hs_pcg': 'HS', 'pipe_pr_cg': 'PIPE_PR', 'pipe_pr_pcg': 'PIPE_PR',
              'pipe_p_cg': 'PIPE_P', 'pipe_p_pcg': 'PIPE_P', 'pipe_pr_m_cg': 'PIPE_PR_M',
              'pipe_p_m_cg': 'PIPE_P_M', 'pr_pcg': 'PR', 'm_pcg': 'M',
              'cg_cg': 'CG_CG', 'cg_pcg': 'CG_CG', 'gv_cg': 'GV', 'gv_pcg': 'GV'}
FORCED = [r for r in all_runs() if r[1] in VARIANT_OF and len(load_run(*r)['state_ks'])]


@pytest.mark.parametrize('matrix,method,prec', FORCED)
def test_teacher_forced_single_step(amd, matrices, matrix, method, prec):
    L = amd['L']
    A, z = matrices[matrix]
    run = load_run(matrix, method, prec)
    ks = set(int(k) for k in run['state_ks'])
    variant = getattr(L, VARIANT_OF[method])
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    op = amd['device'].DeviceCSR(A)
    n = A.shape[0]
    max_iter = int(run['max_iter'])
    op.begin(variant, z['b'], np.zeros(n), max_iter, inv_diag=inv_diag)
    checked, worst = 0, 0.0
    for k in sorted(ks):
        if k + 1 not in ks:
            continue
        g0, g1 = golden_state(run, k), golden_state(run, k + 1)
        for f, v in g0.items():
            if np.ndim(v) == 1:
                try:
                    op.set_vector(f, v)
                except L.PrcgError:
                    pass            # vector not part of this variant's device state
        sc = np.zeros(L.NUM_SCALARS)
        sc[L.S_MU], sc[L.S_NU] = float(g0['mu']), float(g0['nu'])
        sc[L.S_DELTA], sc[L.S_GAMMA] = float(g0.get('dl', 0.0)), float(g0.get('gm', 0.0))
        op.set_scalars(k, sc)
        op.set_iteration(k)
        op.iterate(1)
        op.sync()
        got_sc = op.get_scalars(k + 1)
        for f, v in g1.items():
            if np.ndim(v) == 1:
                try:
                    got = op.get_vector(f)
                except L.PrcgError:
                    continue
                err = rel(got, v)
            elif f in ('mu', 'nu', 'dl', 'gm'):
                slot = {'mu': L.S_MU, 'nu': L.S_NU, 'dl': L.S_DELTA, 'gm': L.S_GAMMA}[f]
                if f in ('dl', 'gm') and VARIANT_OF[method] == 'HS':
                    continue
                err = srel(got_sc[slot], float(v))
            elif f == 'eta':
                err = srel(got_sc[L.S_DELTA], float(v))
            elif f == 'beta':
                err = srel(op.get_coefficients(k + 1)[1], float(v))
            elif f == 'alpha':
                err = srel(got_sc[L.S_NU] / got_sc[L.S_MU], float(v))
            else:
                continue
            worst = max(worst, err)
            assert err <= 1e-12, (matrix, method, prec, k, f, err)
        checked += 1
    op.close()
    assert checked >= 2
    print(f'{matrix}/{method}/{prec}: {checked} forced steps, worst rel. deviation {worst:.2e}')


# ---------------------------------------------------------------------------------------
# free-running: prefix vs the reference, whole trajectory vs the device-ordered oracle,
# convergence statistics vs the reference
# ---------------------------------------------------------------------------------------
# Published known answers: iterations to relative A-norm error 1e-5 and log10 of the best
# relative error (reference: numerical_experiments/figures/convergence_table_data.tex,
# columns hs, cg, m, pr, gv, pipe_pr_m, pipe_pr).
PUBLISHED = {
    ('bcsstk03', 'None', 'hs_cg'): (364, -14.55), ('bcsstk03', 'None', 'pr_pcg'): (380, -14.43),
    ('bcsstk03', 'None', 'pipe_pr_m_cg'): (492, -12.65), ('bcsstk03', 'None', 'pipe_pr_cg'): (411, -12.96),
    ('nos7', 'None', 'hs_cg'): (2869, -9.01), ('nos7', 'None', 'pipe_pr_cg'): (2899, -7.24),
    ('bcsstk03', 'jacobi', 'hs_pcg'): (118, -14.10), ('bcsstk03', 'jacobi', 'pipe_pr_pcg'): (121, -13.50),
    ('nos7', 'jacobi', 'hs_pcg'): (67, -8.91), ('nos7', 'jacobi', 'pipe_pr_pcg'): (67, -9.41),
    ('bcsstk03', 'None', 'cg_cg'): (439, -14.49), ('bcsstk03', 'None', 'gv_cg'): (598, -6.86),
    ('bcsstk03', 'jacobi', 'cg_pcg'): (118, -14.11), ('bcsstk03', 'jacobi', 'gv_pcg'): (120, -9.48),
    ('nos7', 'jacobi', 'cg_pcg'): (67, -9.21), ('nos7', 'jacobi', 'gv_pcg'): (67, -6.41),
}

# Free-running prefix on which 1e-12 holds.  It is set by how fast these ill-conditioned
# problems amplify ANY change of summation order (about 30x per iteration on bcsstk03): SURVEY.md 7.2 /
# BASELINE.md section 2 measured k <= 8 (bcsstk03) / k <= 15 (nos7) for ONE re-ordering (pairwise sums) of the
# reference's own inner products.  The test MEASURES the first k at which this run's recurrence residual leaves
# 1e-12, prints it, and requires it to be no earlier than a floor (a few iterations of margin below what
# re-orderings of the reference itself show: the depth depends on the reduction tree in use by an iteration or two).
PREFIX_FLOOR = {'bcsstk03': 7, 'nos7': 15}      # first k beyond 1e-12 must be >= this (entries 0..floor-1 hold 1e-12); measured on
                                                # MI355X: 8 / 16 for every unpreconditioned variant -- SURVEY.md 7.2(2)'s depths (k <= 8 / 15 with
                                                # its own re-ordering leaving 1e-12 at 9 / 16) to within one iteration


def first_k_beyond(got, ref, tol=1e-12):
    """first index at which |got - ref| / |ref| exceeds tol (len(ref) if never)"""
    with np.errstate(all='ignore'):
        bad = ~(np.abs(got - ref) <= tol * np.abs(ref))
    bad &= ~(np.isnan(got) & np.isnan(ref))
    return int(np.argmax(bad)) if bad.any() else len(ref)


def _interleaved(a, b):
    return float(sum(np.dot(a[i::4], b[i::4]) for i in range(4)))


# six orders of the SAME inner product (test infrastructure): the reference's BLAS ddot, NumPy's pairwise sum,
# back to front, the device's reduction tree, four interleaved partial sums, the exactly rounded sum
SUMMATION_ORDERS = [
    ('blas', np.dot), ('pairwise', lambda a, b: float(np.sum(a * b))), ('reversed', lambda a, b: float(np.dot(a[::-1], b[::-1]))),
    ('device', device_dot), ('interleaved', _interleaved), ('exact', lambda a, b: __import__('math').fsum(a * b)),
]
FREE = [r for r in all_runs() if r[1] in VARIANT_OF and not (r[0] == 'nos7' and r[1] in ('pr_pcg', 'pipe_p_cg', 'cg_cg', 'gv_cg'))]
# (the four excluded nos7 runs are 1000-iteration stubs that never reach 1e-5; they serve the forced steps)


@pytest.mark.parametrize('matrix,method,prec', FREE)
def test_free_running_against_reference(amd, matrices, matrix, method, prec):
    A, z = matrices[matrix]
    run = load_run(matrix, method, prec)
    max_iter = int(run['max_iter'])
    cbs = [getattr(amd['cbs'], q) for q in FOUR]
    kw = {}
    if method.endswith('pcg'):
        kw['preconditioner'] = (lambda v: (1 / A.diagonal()) * v) if prec == 'jacobi' else (lambda v: v)
    out = getattr(amd['cgv'], method)(A, z['b'], np.zeros(A.shape[0]), max_iter, callbacks=cbs,
                                      x_true=z['x_true'], **kw)
    assert out['name'] == str(run['name']) and out['max_iter'] == max_iter
    prefix = PREFIX_FLOOR.get(matrix, 5)
    measured = first_k_beyond(out['updated_residual_2_norm'], run['hist_updated_residual_2_norm'])
    assert measured >= prefix, f'{matrix}/{method}/{prec}: recurrence residual leaves 1e-12 at k={measured}, floor {prefix}'
    for q in FOUR:
        assert out[q].shape == (max_iter,)
        ref = run['hist_' + q]
        if q == 'updated_residual_2_norm':
            # the residual history proper: 1e-12 relative
            np.testing.assert_allclose(out[q][:prefix], ref[:prefix], rtol=1e-12, atol=0,
                                       err_msg=f'{matrix}/{method}/{prec}/{q}')
        else:
            # b - A x and x - x_true are differences of O(|b|), O(|x|) quantities: a 1e-16
            # perturbation of x shows up as 1e-16 * |b| in them, so the bar is 1e-12
            # relative to the initial value (plus 1e-10 relative)
            # (two entries fewer than the recurrence residual: x carries the divergence of r one update further -- measured on
            #  MI355X, nos7 pipe_pr_cg: the A-norm error leaves 1e-10 at k = 14 while the residual holds 1e-12 up to k = 15)
            np.testing.assert_allclose(out[q][:prefix - 2], ref[:prefix - 2], rtol=1e-10, atol=1e-12 * ref[0],
                                       err_msg=f'{matrix}/{method}/{prec}/{q}')
    its, acc = orc.convergence_summary(out['error_A_norm'])
    ref_its, ref_acc = int(run['iters_to_1e-5']), float(run['log10_min_rel_error_A'])
    q = 'updated_residual_2_norm'
    nxt = [float(abs(out[q][k] - run['hist_' + q][k]) / run['hist_' + q][k]) for k in range(prefix, min(prefix + 3, max_iter))]
    # the spread of the same two statistics over re-orderings of the reference's own inner products
    prec_fn = (lambda v: (1 / A.diagonal()) * v) if prec == 'jacobi' else None
    sp_its, sp_acc = [], []
    for name, dot in SUMMATION_ORDERS:
        o = getattr(orc, method)(A, z['b'], np.zeros(A.shape[0]), max_iter, preconditioner=prec_fn, callbacks=['error_A_norm'],
                                 x_true=z['x_true'], dot=dot)
        i_, a_ = orc.convergence_summary(o['error_A_norm'])
        if name == 'blas':
            assert (i_, a_) == (ref_its, ref_acc), 'the oracle in the reference order must reproduce the fixture'
        sp_its.append(i_)
        sp_acc.append(a_)
    never = [i for i in sp_its if i == 0]
    pub = PUBLISHED.get((matrix, prec, method))
    print(f'{matrix}/{method}/{prec}: its {its} (fixture {ref_its}, spread {min(sp_its)}..{max(sp_its)}'
          f'{", published " + str(pub[0]) if pub else ""}), log10 min err {acc:.2f} (fixture {ref_acc:.2f}, spread '
          f'{min(sp_acc):.2f}..{max(sp_acc):.2f}); first k with |r_k| beyond 1e-12 of the reference: {measured} (floor {prefix}); '
          f'rel. deviation at k={prefix}..: ' + ' '.join(f'{v:.1e}' for v in nxt))
    if never and len(never) < len(sp_its):
        return      # "never reaches 1e-5" for some orders, reaches it for others: the statistic is undefined here
    lo, hi = min(sp_its), max(sp_its)
    assert lo - max(2, 0.02 * lo) <= its <= hi + max(2, 0.02 * hi), (its, sp_its)
    assert min(sp_acc) - 0.5 <= acc <= max(sp_acc) + 0.5, (acc, sp_acc)


@pytest.mark.parametrize('matrix,flavour,method,max_iter', [
    ('bcsstk03', 'pr', 'pipe_pr_cg', 1250), ('nos7', 'pr', 'pipe_pr_cg', 3000),
    ('bcsstk03', 'p', 'pipe_p_cg', 600), ('bcsstk03', 'pr_m', 'pipe_pr_m_cg', 600)])
def test_whole_trajectory_against_device_ordered_oracle(amd, matrices, matrix, flavour, method, max_iter):
    """Same algorithm, same reduction tree, same square on both sides: the inner products
    of EVERY iteration of the free-running solve must agree bit for bit (this is what
    catches races and indexing slips far beyond the prefix)."""
    L = amd['L']
    A, z = matrices[matrix]
    n = A.shape[0]
    want = []
    getattr(orc, method)(A, z['b'], np.zeros(n), max_iter, dot=device_dot, square=lambda a: a * a,
                         tap=lambda st: want.append((st.mu, st.dl, st.gm, st.nu)))
    want = np.array(want)
    # two-kernel schedule: its inner products come from the update kernel, whose reduction
    # tree tests/device_order.py reproduces (the fused one-launch schedule sums per tile)
    op = amd['device'].DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    op.begin(getattr(L, VARIANT_OF[method]), z['b'], np.zeros(n), max_iter)
    op.iterate(max_iter - 1)
    op.sync()
    got = np.array([op.get_scalars(k)[[L.S_MU, L.S_DELTA, L.S_GAMMA, L.S_NU]] for k in range(max_iter)])
    op.close()
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    first_bad = int(np.argmin(same.all(axis=1))) if not same.all() else -1
    print(f'{matrix}/{method}: {max_iter} iterations, all four inner products bit-exact: {bool(same.all())}')
    assert same.all(), f'first mismatch at k={first_bad}: got {got[first_bad]} want {want[first_bad]}'


@pytest.mark.parametrize('matrix,method,max_iter', [
    ('bcsstk03', 'pipe_pr_cg', 1250), ('nos7', 'pipe_pr_cg', 3000), ('bcsstk03', 'pipe_p_cg', 600), ('nos7', 'pipe_pr_m_cg', 600),
    ('s3_small', 'pipe_pr_cg', 300), ('lap2d_300', 'pipe_pr_cg', 200), ('lap3d_40', 'pipe_p_m_cg', 150),
    ('lap3d_40_chunked', 'pipe_pr_cg', 150), ('lap2d_300_chunked', 'pipe_pr_m_cg', 100)])
def test_whole_trajectory_of_the_one_launch_schedule(amd, matrices, matrix, method, max_iter):
    """The strongest parity test, on the schedule that ships (one launch per iteration: what bench.py times and every
    solve uses by default; PRCG_SMALL=0 keeps the one-workgroup solver of tiny systems out of the way).  The oracle
    runs with the launch's own reduction tree (tests/device_order.py: OneLaunchTree, rebuilt from prcg_debug_layout):
    all four inner products of EVERY iteration bit for bit -- whether an iteration's partials were summed by the next
    launch's prologue (free-running prcg_iterate) or by the reduction launch (a recorder in between) -- and all four
    recorder histories of every k to rounding of one norm (the iterates are then identical; the norms are summed in
    another order by the recorder kernels)."""
    L = amd['L']
    if matrix in matrices:
        A, z = matrices[matrix]
    else:
        # tens to hundreds of workgroups: the XCD remap, several tiles per wave and (stencils) 128-row tiles with two rows per lane
        P = amd['problems']
        # (..._chunked: the XCD-chunked tile order that 3-D stencils with far plane neighbours get by themselves, forced here)
        A = {'s3_small': lambda: P.WORKLOADS['s3_small']['make'](), 'lap2d_300': lambda: P.laplace_2d(300, 200),
             'lap3d_40': lambda: P.laplace_3d(40, 40, 40)}[matrix.replace('_chunked', '')]()
        b_, _, xt_ = P.reference_rhs(A, A.shape[0])
        z = {'b': b_, 'x_true': xt_}
    n = A.shape[0]
    knobs = {'PRCG_SMALL': '0'}
    if matrix.endswith('_chunked'):
        knobs['PRCG_WIN_ORDER'] = '1'
    op = amd['device'].DeviceCSR(A, knobs=knobs)
    variant = getattr(L, VARIANT_OF[method])
    # (a) all four recorders: a reduction launch after every iteration
    op.begin(variant, z['b'], np.zeros(n), max_iter, x_true=z['x_true'], hist_mask=15)
    s = op.schedule()
    if not (s['fused'] and s['window']):
        pytest.skip(f'{matrix} is no window operator: {s}')
    op.iterate(max_iter - 1)
    op.sync()
    got = np.array([op.get_scalars(k)[[L.S_MU, L.S_DELTA, L.S_GAMMA, L.S_NU]] for k in range(max_iter)])
    hist = op.history()
    tree = OneLaunchTree(op.layout())
    assert tree.chunked == matrix.endswith('_chunked')
    # (b) no recorder but the recurrence residual: partials summed by the next launch's prologue, calls of any length
    op.begin(variant, z['b'], np.zeros(n), max_iter, hist_mask=1)
    assert op.schedule()['fused'] and not op.schedule()['small']
    for chunk in (1, 2, 37, max_iter):
        op.iterate(min(chunk, max_iter - 1 - op.k))
    op.sync()
    got_b = np.array([op.get_scalars(k)[[L.S_MU, L.S_DELTA, L.S_GAMMA, L.S_NU]] for k in range(max_iter)])
    hist_b = op.history()
    op.close()
    want = []
    ref = getattr(orc, method)(A, z['b'], np.zeros(n), max_iter, dot=tree.dot, dot0=device_dot, square=lambda a: a * a,
                               callbacks=FOUR, x_true=z['x_true'], tap=lambda st: want.append((st.mu, st.dl, st.gm, st.nu)))
    want = np.array(want)
    for name, g in (('with recorders', got), ('free-running', got_b)):
        same = (g == want) | (np.isnan(g) & np.isnan(want))
        first_bad = int(np.argmin(same.all(axis=1))) if not same.all() else -1
        assert same.all(), f'{name}: first mismatch at k={first_bad}: got {g[first_bad]} want {want[first_bad]}'
    assert np.array_equal(hist['updated_residual_2_norm'], hist_b['updated_residual_2_norm'], equal_nan=True)
    worst = {}
    for q in FOUR:
        with np.errstate(all='ignore'):
            dev = np.abs(hist[q] - ref[q]) / np.abs(ref[q])
        dev = dev[np.isfinite(dev)]
        worst[q] = float(dev.max()) if dev.size else 0.0
    print(f'{matrix}/{method}: {max_iter} iterations on the one-launch schedule ({tree.grid} workgroups x {tree.wpb} waves): all four '
          f'inner products bit-exact in both modes; recorder histories worst rel. deviation ' +
          ', '.join(f'{q} {v:.1e}' for q, v in worst.items()))
    # sums of squares in another order: a few ulp; e'Ae has mixed signs near convergence: its bound is looser
    assert worst['updated_residual_2_norm'] <= 1e-13 and worst['residual_2_norm'] <= 1e-13 and worst['error_2_norm'] <= 1e-13, worst
    assert worst['error_A_norm'] <= 1e-11, worst


@pytest.mark.parametrize('matrix,variant,prec', [
    ('bcsstk03', 'PIPE_PR', None), ('nos7', 'PIPE_PR', None), ('nos7', 'PIPE_PR_M', None),
    ('bcsstk03', 'PIPE_P', None), ('nos7', 'PIPE_P_M', None),
    ('bcsstk03', 'PIPE_PR', 'jacobi'), ('nos7', 'PIPE_PR', 'jacobi'), ('494_bus', 'PIPE_P', 'jacobi'),
    ('bcsstk14', 'PIPE_PR_M', 'jacobi'), ('nos4', 'PIPE_P_M', 'jacobi')])
def test_fused_and_two_kernel_schedules_agree(amd, matrices, matrix, variant, prec):
    """One GPU runs every pipelined flavour, with or without Jacobi, as ONE launch per iteration (SpMM with
    the next vector update fused into its row epilogue; u -- and w in the 'pr' flavours -- never stored).
    Same arithmetic per element as the two-kernel schedule, different summation order of the inner
    products: single steps from identical state agree to 1e-12, vectors bit for bit; derived w, u (and
    w~, u~) equal what the two-kernel schedule holds."""
    L = amd['L']
    A, z = matrices[matrix]
    n = A.shape[0]
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    ops = [amd['device'].DeviceCSR(A, knobs={'PRCG_FUSED': f, 'PRCG_SMALL': '0'}) for f in ('1', '0')]
    for op in ops:
        op.begin(getattr(L, variant), z['b'], np.zeros(n), 64, inv_diag=inv_diag)
    assert ops[0].schedule()['fused'] and not ops[1].schedule()['fused']
    stored = ['x', 'r', 'p', 's'] + (['rt', 'st'] if prec else [])
    if variant in ('PIPE_P', 'PIPE_P_M'):
        stored += ['w'] + (['wt'] if prec else [])
    derived = [v for v in (['w', 'u'] + (['wt', 'ut'] if prec else [])) if v not in stored]
    worst = 0.0
    for k in range(40):
        # teacher-force the fused engine with the two-kernel engine's state, step both
        st = {v: ops[1].get_vector(v) for v in stored}
        sc = ops[1].get_scalars(k)
        for v, a in st.items():
            ops[0].set_vector(v, a)
        ops[0].set_scalars(k, sc)
        ops[0].set_iteration(k)
        for op in ops:
            op.iterate(1)
        for v in stored + derived:
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v), equal_nan=True), (k, v)
        a, b = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
        worst = max(worst, float(np.max(np.abs(a - b) / np.abs(b))))
        assert np.array_equal(ops[0].get_coefficients(k + 1), ops[1].get_coefficients(k + 1))
    assert worst <= 1e-12, worst
    for op in ops:
        op.close()
    print(f'{matrix}/{variant}/{prec}: fused vs two-kernel, 40 forced steps, worst scalar deviation {worst:.2e}')


@pytest.mark.parametrize('variant,source,prec,knobs', [
    ('CG_CG', 'bcsstk03', None, {}), ('CG_CG', 'nos7', 'jacobi', {}), ('CG_CG', 's3_small', None, {'PRCG_VALDICT': '0'}),
    ('CG_CG', 'lap3d_20', 'jacobi', {}), ('GV', 'bcsstk03', None, {}), ('GV', 'nos7', None, {}), ('GV', 's3_small', None, {}),
    ('GV', 's1_small', None, {'PRCG_VALDICT': '0'})])
def test_chronopoulos_gear_and_ghysels_vanroose_in_one_launch(amd, matrices, variant, source, prec, knobs):
    """cg_cg / cg_pcg / gv_cg on a window operator run ONE launch per iteration: the launch of iteration k+1 closes iteration k in
    its prologue (b, mu, a from that launch's partials: cg_cg.py:64,67-68) and applies its p, s (u) update while forming its
    window -- the new residual r - a (w + b s) (gv: the new w - a (t + b u)), cg_cg.py:66,60 / gv_cg.py:79,67.  Two checks:
    (i) the deferred form against the same kernels driven one iteration per call (every call closes its iteration with the
    update launch): vectors, scalars, coefficients and histories of 60 iterations BIT FOR BIT -- deferring changes no
    rounding; (ii) against the two-launch schedule (PRCG_CG_ONE=0) on forced single steps: vectors bit for bit, scalars to
    1e-12 (another grid, another summation order of eta and nu)."""
    L = amd['L']
    if source in matrices:
        A, z = matrices[source]
        b, x_true = z['b'], z['x_true']
    else:
        A = amd['problems'].laplace_3d(20, 20, 20) if source == 'lap3d_20' else amd['problems'].WORKLOADS[source]['make']()
        b, _, x_true = amd['problems'].reference_rhs(A, A.shape[0])
    n = A.shape[0]
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    var = getattr(L, variant)
    names = ['x', 'r', 'p', 's', 'w'] + (['u'] if variant == 'GV' else []) + (['rt'] if prec else [])
    iters = 60
    runs = []
    for chunks in ([1] * iters, [iters], [7, 1, 30, 22]):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        op.begin(var, b, np.zeros(n), iters + 1, x_true=x_true, inv_diag=inv_diag, hist_mask=1)
        s = op.schedule()
        if not (s['window'] and s['fused']):
            op.close()
            pytest.skip(f'{source} is no window operator')
        for c in chunks:
            op.iterate(c)
        op.sync()
        runs.append(({v: op.get_vector(v) for v in names}, np.array([op.get_scalars(k) for k in range(iters + 1)]),
                     np.array([op.get_coefficients(k) for k in range(1, iters + 1)]), op.history()['updated_residual_2_norm']))
        op.close()
    for other in runs[1:]:
        for v in names:
            assert np.array_equal(other[0][v], runs[0][0][v], equal_nan=True), v
        assert np.array_equal(other[1][:, :5], runs[0][1][:, :5], equal_nan=True)
        assert np.array_equal(other[2][:, :2], runs[0][2][:, :2], equal_nan=True)
        assert np.array_equal(other[3], runs[0][3], equal_nan=True)
    # (ii) forced steps against the two-launch schedule
    ops = [amd['device'].DeviceCSR(A, knobs=dict(knobs, PRCG_CG_ONE=f)) for f in ('1', '0')]
    for op in ops:
        op.begin(var, b, np.zeros(n), 48, inv_diag=inv_diag)
    stored = ['x', 'r', 'p', 's', 'w'] + (['u'] if variant == 'GV' else []) + (['rt'] if prec else [])
    worst = 0.0
    for k in range(30):
        st = {v: ops[1].get_vector(v) for v in stored}
        sc = ops[1].get_scalars(k)
        for v, a in st.items():
            ops[0].set_vector(v, a)
        ops[0].set_scalars(k, sc)
        ops[0].set_iteration(k)
        for op in ops:
            op.iterate(1)
        for v in ('x', 'r', 'w'):
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v), equal_nan=True), (k, v)
        for v in ('p', 's') + (('u',) if variant == 'GV' else ()):          # inherit b's rounding
            np.testing.assert_allclose(ops[0].get_vector(v), ops[1].get_vector(v), rtol=1e-11, atol=1e-300, err_msg=f'{k} {v}')
        a, c = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
        nz = c != 0
        worst = max(worst, float(np.max(np.abs(a[nz] - c[nz]) / np.abs(c[nz]))))
    assert worst <= 1e-12, worst
    for op in ops:
        op.close()
    print(f'{source}/{variant}/{prec}: one launch per iteration, deferred update bit-identical to per-call closing; vs two launches worst scalar deviation {worst:.1e}')


@pytest.mark.parametrize('variant', ['CG_CG', 'GV'])
@pytest.mark.parametrize('source,prec,knobs', [
    ('bcsstk03', None, {}), ('nos7', 'jacobi', {}), ('494_bus', 'jacobi', {}), ('s3_small', None, {'PRCG_VALDICT': '0'}),
    ('s1_small', 'jacobi', {}), ('lap3d_20', None, {})])
def test_chronopoulos_gear_and_ghysels_vanroose_in_two_launches(amd, matrices, source, prec, knobs, variant):
    """cg_cg / cg_pcg and gv_cg / gv_pcg on a window operator: the product launch forms its window as the new residual
    r - a s (gv: the new w, w - a u), times d with Jacobi, writes x, r, r~, w (gv: also w~, t) and the partials of eta,
    nu; the p, s (s~, u) update sums them itself (cg_cg.py:59-68, gv_cg.py:65-81).  Same arithmetic per element as the
    four-launch schedule (PRCG_FUSED=0), other summation order of eta and nu: forced single steps agree bit for bit
    in the vectors the product launch writes and to 1e-12 in the scalars (p, s, s~, u inherit b's rounding);
    free-running solves agree at convergence level."""
    L = amd['L']
    from oracle import ne_oracle as orc
    if source in matrices:
        A, z = matrices[source]
        b, x_true = z['b'], z['x_true']
    else:
        A = amd['problems'].laplace_3d(20, 20, 20) if source == 'lap3d_20' else amd['problems'].WORKLOADS[source]['make']()
        b, _, x_true = amd['problems'].reference_rhs(A, A.shape[0])
    n = A.shape[0]
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    ops = [amd['device'].DeviceCSR(A, knobs=dict(knobs, PRCG_FUSED=f, PRCG_CG_ONE='0')) for f in ('1', '0')]      # (the TWO-launch schedule)
    var = getattr(L, variant)
    for op in ops:
        op.begin(var, b, np.zeros(n), 64, inv_diag=inv_diag)
    assert ops[0].schedule()['fused'] and ops[0].schedule()['window'] and not ops[1].schedule()['fused']
    exact = ['x', 'r', 'w'] + (['rt'] if prec else []) + (['wt'] if prec and variant == 'GV' else [])
    loose = ['p', 's'] + (['u'] + (['st'] if prec else []) if variant == 'GV' else [])
    worst = 0.0
    for k in range(40):
        st = {v: ops[1].get_vector(v) for v in exact + loose}
        sc = ops[1].get_scalars(k)
        for v, a in st.items():
            ops[0].set_vector(v, a)
        ops[0].set_scalars(k, sc)
        ops[0].set_iteration(k)
        for op in ops:
            op.iterate(1)
        a, c = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
        if not np.all(np.isfinite(c)) or np.any(c[[0, 1, 3, 4]] == 0):
            break
        for v in exact:
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v)), (k, v)
        for v in loose:
            want = ops[1].get_vector(v)          # (b differs in its last bit: elements that cancel feel it relative to the vector)
            np.testing.assert_allclose(ops[0].get_vector(v), want, rtol=1e-12, atol=1e-13 * np.max(np.abs(want)), err_msg=f'{k} {v}')
        worst = max(worst, float(np.max(np.abs(a[[0, 1, 3, 4]] - c[[0, 1, 3, 4]]) / np.abs(c[[0, 1, 3, 4]]))))
    assert k >= 8 and worst <= 1e-12, (k, worst)
    total = 400 if source in ('bcsstk03',) else 150
    hist = []
    for op in ops:
        op.begin(var, b, np.zeros(n), total + 1, x_true=x_true, inv_diag=inv_diag,
                 hist_mask=L.HIST_UPDATED_RESIDUAL_2_NORM | L.HIST_ERROR_A_NORM)
        for chunk in (1, 7, 1, total - 9):
            op.iterate(chunk)
        op.sync()
        hist.append(op.history())
        op.close()
    for q in hist[1]:
        np.testing.assert_allclose(hist[0][q][:6], hist[1][q][:6], rtol=1e-11, err_msg=q)
    ia, aa = orc.convergence_summary(hist[0]['error_A_norm'])
    ib, ab = orc.convergence_summary(hist[1]['error_A_norm'])
    assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 1.0, ((ia, aa), (ib, ab))
    print(f'{source}/{variant}/{prec}: two launches vs four, forced steps, worst scalar deviation {worst:.2e}; '
          f'free running {total} iterations: its-to-1e-5 {ia} vs {ib}, log10 min error {aa:.2f} vs {ab:.2f}')


@pytest.mark.parametrize('method', ['hs_pcg', 'cg_pcg', 'gv_pcg', 'pr_pcg', 'm_pcg', 'pipe_pr_pcg', 'pipe_p_pcg',
                                    'pipe_pr_m_pcg', 'pipe_p_m_pcg'])
def test_host_callback_preconditioner(amd, matrices, method):
    """`method(A, b, x0, max_iter, preconditioner=callable)` with a preconditioner that is NOT a diagonal scaling
    (a tridiagonal solve): the callable runs on the host wherever the reference calls `preconditioner(...)`, the rest
    of the iteration on the device.  Against the oracle with the same callable: the first iterations to 1e-11
    (different summation orders of the inner products), the convergence statistics beyond."""
    from oracle import ne_oracle as orc
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A, z = matrices['nos7']
    n = A.shape[0]
    T = sp.diags([A.diagonal(-1), A.diagonal(), A.diagonal(1)], [-1, 0, 1], format='csc')   # SPD tridiagonal part
    lu = spla.splu(T)
    calls = []

    def prec(v):
        calls.append(1)
        return lu.solve(np.asarray(v, dtype=np.float64))
    max_iter = 260
    cbs = [amd['cbs'].updated_residual_2_norm, amd['cbs'].error_A_norm]
    out = getattr(amd['cgv'], method)(A, z['b'], np.zeros(n), max_iter, callbacks=cbs, x_true=z['x_true'], preconditioner=prec)
    per_it = (len(calls) - 2) / (max_iter - 1)                                   # (two probe calls of the wrapper)
    ref = getattr(orc, method)(A, z['b'], np.zeros(n), max_iter, preconditioner=lambda v: lu.solve(v),
                               callbacks=['updated_residual_2_norm', 'error_A_norm'], x_true=z['x_true'])
    for q in ('updated_residual_2_norm', 'error_A_norm'):
        np.testing.assert_allclose(out[q][:6], ref[q][:6], rtol=1e-11, err_msg=f'{method}/{q}')
    ia, aa = orc.convergence_summary(out['error_A_norm'])
    ib, ab = orc.convergence_summary(ref['error_A_norm'])
    assert ib > 0 and abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 1.5, ((ia, aa), (ib, ab))
    assert 0.9 <= per_it <= 2.3, per_it
    print(f'{method}: tridiagonal preconditioner on the host, {per_it:.2f} applications per iteration; '
          f'its-to-1e-5 {ia} (oracle {ib}), log10 min error {aa:.2f} ({ab:.2f})')


@pytest.mark.parametrize('method', ['pipe_pr_pcg', 'pipe_p_pcg'])
def test_host_callback_preconditioner_tilde_vectors(amd, matrices, method):
    """In a host-callback session w~ and u~ are what the caller's function returned for w and u
    (pipe_pr_cg.py:178-182): a foreign callback must see exactly those vectors (not d*u with a diagonal that
    was never uploaded), and teacher forcing must be able to overwrite them."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A, z = matrices['nos7']
    n = A.shape[0]
    T = sp.diags([A.diagonal(-1), A.diagonal(), A.diagonal(1)], [-1, 0, 1], format='csc')
    lu = spla.splu(T)
    seen = []

    def spy(**env):
        if env['k'] in (0, 3, 7):
            seen.append((env['k'], env['u_k'].copy(), env['ut_k'].copy(), env['w_k'].copy(), env['wt_k'].copy()))
    getattr(amd['cgv'], method)(A, z['b'], np.zeros(n), 9, callbacks=[spy], preconditioner=lambda v: lu.solve(np.asarray(v, dtype=np.float64)))
    assert len(seen) == 3
    for k, u, ut, w, wt in seen:
        assert np.any(ut != 0.0) and np.any(wt != 0.0), k
        np.testing.assert_array_equal(ut, lu.solve(u), err_msg=f'{method}: u~ at k={k}')
        if method == 'pipe_pr_pcg' and k > 0:          # the 'pr' flavours recompute w = A r~, w~ = M^-1 w every iteration
            np.testing.assert_array_equal(wt, lu.solve(w), err_msg=f'{method}: w~ at k={k}')
    # teacher forcing reaches the stored vectors
    L = amd['L']
    op = amd['device'].DeviceCSR(A)
    op.begin(L.PIPE_PR, z['b'], np.zeros(n), 8, preconditioner=lambda v: lu.solve(np.asarray(v, dtype=np.float64)))
    v = np.random.default_rng(3).standard_normal(n)
    op.set_vector('ut', v)
    np.testing.assert_array_equal(op.get_vector('ut'), v)
    op.set_vector('wt', 2 * v)
    np.testing.assert_array_equal(op.get_vector('wt'), 2 * v)
    op.close()


@pytest.mark.parametrize('source,variant,prec,knobs', [
    ('bcsstk03', 'PR', None, {}), ('nos7', 'PR', 'jacobi', {}), ('494_bus', 'M', 'jacobi', {}), ('nos4', 'M', None, {}),
    ('s3_small', 'PR', None, {}), ('s3_small', 'M', None, {'PRCG_VALDICT': '0'}), ('s1_small', 'PR', 'jacobi', {}),
    ('lap3d_20', 'PR', None, {'PRCG_VALDICT': '0'}),
    # the PACKED state of the unpreconditioned iteration (pairs (z, zs) and (p, x), 16-byte accesses only; the default on pattern
    # tiles -- lap3d_20 with its dictionary): band, pattern tiles, 2-byte window indices with plain values, a paper matrix
    ('s3_small', 'PR', None, {'PRCG_PR_PACK': '1'}), ('lap3d_20', 'PR', None, {}), ('lap3d_20', 'PR', None, {'PRCG_PR_PACK': '0'}),
    ('lap3d_20', 'M', None, {'PRCG_VALDICT': '0', 'PRCG_PR_PACK': '1'}), ('nos7', 'PR', None, {'PRCG_PR_PACK': '1'})])
def test_one_launch_predict_and_recompute(amd, matrices, source, variant, prec, knobs):
    """pr_cg / m_cg (pr_pcg, m_pcg) on a window operator run ONE launch per iteration: the staged window of the new
    direction is formed as (r~ - a s~) + b p_old while it is parked (pr_cg.py:148,151), s = A p follows, the row's own
    x, r, r~, p, s, s~ and the five partials come out of the same lane.  Same arithmetic per element as the four-launch
    schedule (PRCG_FUSED=0: update kernel, product, two reductions), different summation order of the inner
    products: forced single steps from identical state agree bit for bit in every vector and to 1e-12 in the
    scalars; free-running solves (iterate calls of any length, recorders in between) agree at convergence level."""
    L = amd['L']
    from oracle import ne_oracle as orc
    if source in matrices:
        A, z = matrices[source]
        b, x_true = z['b'], z['x_true']
    else:
        A = amd['problems'].laplace_3d(20, 20, 20) if source == 'lap3d_20' else amd['problems'].WORKLOADS[source]['make']()
        b, _, x_true = amd['problems'].reference_rhs(A, A.shape[0])
    n = A.shape[0]
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    ops = [amd['device'].DeviceCSR(A, knobs=dict(knobs, PRCG_FUSED=f)) for f in ('1', '0')]
    for op in ops:
        op.begin(getattr(L, variant), b, np.zeros(n), 64, inv_diag=inv_diag)
    assert ops[0].schedule()['fused'] and ops[0].schedule()['window'] and not ops[1].schedule()['fused']
    stored = ['x', 'r', 'p', 's'] + (['rt', 'st'] if prec else [])
    worst = 0.0
    for k in range(40):
        st = {v: ops[1].get_vector(v) for v in stored}
        sc = ops[1].get_scalars(k)
        for v, a in st.items():
            ops[0].set_vector(v, a)
        ops[0].set_scalars(k, sc)
        ops[0].set_iteration(k)
        for op in ops:
            op.iterate(1)
        for v in stored:
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v), equal_nan=True), (k, v)
        a, c = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
        if not np.all(np.isfinite(c)) or np.any(c == 0):
            break                       # converged to breakdown (0/0): nothing left to compare
        worst = max(worst, float(np.max(np.abs(a - c) / np.abs(c))))
        assert np.array_equal(ops[0].get_coefficients(k + 1), ops[1].get_coefficients(k + 1), equal_nan=True)
    assert k >= 8 and worst <= 1e-12, (k, worst)
    # free running, chunked, with the error recorder in between
    total = 400 if source in ('bcsstk03', 'nos4') else 150
    hist = []
    for op in ops:
        op.begin(getattr(L, variant), b, np.zeros(n), total + 1, x_true=x_true, inv_diag=inv_diag, hist_mask=L.HIST_UPDATED_RESIDUAL_2_NORM | L.HIST_ERROR_A_NORM)
        for chunk in (1, 7, 1, total - 9):
            op.iterate(chunk)
        op.sync()
        hist.append(op.history())
        op.close()
    for q in hist[1]:
        np.testing.assert_allclose(hist[0][q][:6], hist[1][q][:6], rtol=1e-11, err_msg=q)
    ia, aa = orc.convergence_summary(hist[0]['error_A_norm'])
    ib, ab = orc.convergence_summary(hist[1]['error_A_norm'])
    assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 1.0, ((ia, aa), (ib, ab))
    print(f'{source}/{variant}/{prec}: one launch vs four, 40 forced steps, worst scalar deviation {worst:.2e}; '
          f'free running {total} iterations: its-to-1e-5 {ia} vs {ib}, log10 min error {aa:.2f} vs {ab:.2f}')


@pytest.mark.parametrize('source,prec,knobs', [
    ('s3_small', None, {}), ('s3_small', 'jacobi', {}), ('s3_small', None, {'PRCG_VALDICT': '0'}),
    ('s1_small', None, {}), ('lap3d_20', 'jacobi', {}), ('lap3d_20', None, {'PRCG_VALDICT': '0'}),
    ('bcsstk03', None, {}), ('nos7', 'jacobi', {}), ('s3_small', None, {'PRCG_WIN': '0'})])
def test_hestenes_stiefel_without_reduction_launches(amd, matrices, source, prec, knobs):
    """hs_cg / hs_pcg on one GPU: two launches per iteration on window operators (update; product whose staged
    window IS the new direction p = z + b p_old, hs_cg.py:57-62), three on CSR-adaptive tiles, the inner products
    summed by the following launch's workgroups in the order of the separate reduction kernel.  Same arithmetic, same
    summation order as the five-launch schedule (PRCG_FUSED=0): every vector, scalar and coefficient of a
    free-running solve must agree BIT FOR BIT, whatever the chunking of the iterate calls (pending partials across
    calls, recorders in between)."""
    L = amd['L']
    if source in matrices:
        A, z = matrices[source]
        b = z['b']
    else:
        A = amd['problems'].laplace_3d(20, 20, 20) if source == 'lap3d_20' else amd['problems'].WORKLOADS[source]['make']()
        b = amd['problems'].reference_rhs(A, A.shape[0])[0]
    n = A.shape[0]
    inv_diag = (1 / A.diagonal()) if prec == 'jacobi' else None
    # (PRCG_SMALL=0: the one-workgroup solver of small systems has its own summation order)
    ops = [amd['device'].DeviceCSR(A, knobs=dict(knobs, PRCG_FUSED=f, PRCG_SMALL='0')) for f in ('1', '0')]
    total = 60
    for op, mask in zip(ops, (1, 1)):
        op.begin(L.HS, b, np.zeros(n), total + 1, inv_diag=inv_diag, hist_mask=mask)
    assert ops[0].schedule()['fused'] and not ops[1].schedule()['fused']
    expect_window = knobs.get('PRCG_WIN') != '0'      # (bcsstk03 and nos7 are narrow enough for window tiles too)
    assert ops[0].schedule()['window'] == expect_window
    k = 0
    for chunk in (1, 7, 1, 12, 2, 37):
        for op in ops:
            op.iterate(chunk)
        k += chunk
        for v in ['x', 'r', 'p', 's'] + (['rt'] if prec else []):
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v), equal_nan=True), (k, v)
        for kk in range(k - chunk, k + 1):
            assert np.array_equal(ops[0].get_scalars(kk)[:5], ops[1].get_scalars(kk)[:5], equal_nan=True), (k, kk)
            if kk >= 1:
                assert np.array_equal(ops[0].get_coefficients(kk)[:2], ops[1].get_coefficients(kk)[:2], equal_nan=True), (k, kk)
    h0, h1 = ops[0].history(), ops[1].history()
    for q in h1:
        assert np.array_equal(h0[q], h1[q], equal_nan=True), q
    for op in ops:
        op.close()


@pytest.mark.parametrize('variant', ['PIPE_PR', 'HS'])
@pytest.mark.parametrize('matrix', ['bcsstk03', 'nos7', 'bcsstk14', 'model_48_8_3'])
def test_one_workgroup_solver_for_small_systems(amd, matrices, matrix, variant):
    """n <= 4096 and nothing but the recurrence residual recorded: the whole solve runs in one
    launch of one workgroup ((r,s) -- for Hestenes-Stiefel the direction p -- in LDS, matrix in LDS when it fits:
    bcsstk14's does not).  The pipelined predict-and-recompute family (k_small_pipe_pr) and Hestenes-Stiefel (k_small_hs:
    BASELINE config 1 is bcsstk03 under hs_cg).
    Per element the arithmetic is the one of the multi-launch schedules: from identical state
    vectors agree bit for bit, inner products to 1e-12; and it is deterministic."""
    import time
    L = amd['L']
    A, z = matrices[matrix]
    n = A.shape[0]
    VAR = getattr(L, variant)
    small = amd['device'].DeviceCSR(A)                                  # default: one-workgroup solver
    multi = amd['device'].DeviceCSR(A, knobs={'PRCG_SMALL': '0'})       # one (HS: two) launch(es) per iteration
    for op in (small, multi):
        op.begin(VAR, z['b'], np.zeros(n), 64)
    assert small.schedule()['small'] == (matrix != 'bcsstk14') and not multi.schedule()['small']
    worst = 0.0
    for k in range(30):
        for v in ('x', 'r', 'p', 's'):
            small.set_vector(v, multi.get_vector(v))
        small.set_scalars(k, multi.get_scalars(k))
        small.set_iteration(k)
        small.iterate(1)
        multi.iterate(1)
        for v in ('x', 'r', 'p', 's'):
            if variant == 'HS' and v in ('p', 's') and small.schedule()['small']:
                # Hestenes-Stiefel forms p with b = nu_k / nu_k1, nu_k summed INSIDE the iteration -- in the workgroup's order here,
                # in the launches' order there: p and s = A p inherit b's last bit
                # (absolute floor: an entry that cancels to a small fraction of its terms)
                ref_v = multi.get_vector(v)
                np.testing.assert_allclose(small.get_vector(v), ref_v, rtol=1e-12, atol=1e-13 * np.abs(ref_v).max(), err_msg=f'{k} {v}')
            else:
                assert np.array_equal(small.get_vector(v), multi.get_vector(v)), (k, v)
        idx = [0, 3, 4] if variant == 'HS' else [0, 1, 2, 3, 4]
        a, b = small.get_scalars(k + 1)[idx], multi.get_scalars(k + 1)[idx]
        worst = max(worst, float(np.max(np.abs(a - b) / np.abs(b))))
    assert worst <= 1e-12, worst
    # free-running: reproducible, and the same history whether run in one call or in pieces
    iters = 3000
    runs, times = [], []
    for chunks in ((iters,), (1000, 1, 1999)):
        small.begin(VAR, z['b'], np.zeros(n), iters + 1, hist_mask=1)
        t0 = time.perf_counter()
        for c in chunks:
            small.iterate(c)
        small.sync()
        times.append(time.perf_counter() - t0)
        runs.append(small.history()['updated_residual_2_norm'])
    assert np.array_equal(runs[0], runs[1], equal_nan=True)
    multi.begin(VAR, z['b'], np.zeros(n), iters + 1, hist_mask=1)
    t0 = time.perf_counter()
    multi.iterate(iters)
    multi.sync()
    dt_multi = time.perf_counter() - t0
    prefix = PREFIX_FLOOR.get(matrix, 5)
    np.testing.assert_allclose(runs[0][:prefix], multi.history()['updated_residual_2_norm'][:prefix], rtol=1e-12)
    print(f'{matrix} (n={n}) {variant}: {iters} iterations in one launch {times[0] * 1e3:.2f} ms '
          f'({times[0] / iters * 1e6:.2f} us/iteration) vs one launch per iteration {dt_multi * 1e3:.2f} ms '
          f'({dt_multi / iters * 1e6:.2f} us/iteration); forced-step scalar deviation {worst:.1e}')
    small.close()
    multi.close()


@pytest.mark.parametrize('name', ['bcsstk16', 'bcsstk18', 'bcsstm25', 'band60k', 'lap200'])
def test_few_workgroup_solver_for_mid_size_systems(amd, name):
    """Systems too large for the one-workgroup solver and too small to fill the chip (the paper's bcsstk16 / 18, bcsstm25;
    up to 131,072 rows), nothing but the recurrence residual recorded: the whole solve runs in ONE launch of a few
    co-operating workgroups (csrc/prcg_medium.hip: rows in slices, x and p in registers, (r,s) through a double-buffered
    exchange array and an LDS window, one all-to-all of partial sums per iteration).  Per element the arithmetic is the one
    of the multi-launch schedules: from identical state the vectors agree bit for bit, the inner products to 1e-12; the
    run is deterministic and does not depend on how it is cut into prcg_iterate calls.
    Opt-in (PRCG_MEDIUM=1): measured on MI355X a grid-wide hand-off (drained sc1 stores, flag, poll) costs ~5 us and an
    iteration needs two of them -- bcsstm25 10 us, bcsstk18 24 us per iteration against 9 / 8 us with one launch per
    iteration; the test prints both."""
    import os
    import time
    from conftest import GOLDEN
    L, P = amd['L'], amd['problems']
    if name == 'band60k':
        A = P.banded_ex2b(60_000, 7)
    elif name == 'lap200':
        A = P.laplace_2d(200, 200)
    else:
        z = np.load(os.path.join(GOLDEN, f'tablemat_{name}.npz'))
        nz = int(z['n'])
        A = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(nz, nz))
    n = A.shape[0]
    b, x0, _ = P.reference_rhs(A, n)
    med = amd['device'].DeviceCSR(A, knobs={'PRCG_MEDIUM': '1'})      # (opt-in: see below)
    multi = amd['device'].DeviceCSR(A)
    for op in (med, multi):
        op.begin(L.PIPE_PR, b, x0, 64)
    assert med.schedule()['medium'] and not med.schedule()['small'], med.schedule()
    assert not multi.schedule()['medium']
    worst = 0.0
    for k in range(12):
        for v in ('x', 'r', 'p', 's'):
            med.set_vector(v, multi.get_vector(v))
        med.set_scalars(k, multi.get_scalars(k))
        med.set_iteration(k)
        med.iterate(1)
        multi.iterate(1)
        for v in ('x', 'r', 'p', 's'):
            assert np.array_equal(med.get_vector(v), multi.get_vector(v)), (k, v)
        a, c = med.get_scalars(k + 1)[:5], multi.get_scalars(k + 1)[:5]
        worst = max(worst, float(np.max(np.abs(a - c) / np.abs(c))))
        assert np.allclose(med.get_coefficients(k + 1)[:2], multi.get_coefficients(k + 1)[:2], rtol=1e-12, atol=0)
    assert worst <= 1e-12, worst
    iters = 2000
    runs, times = [], []
    for chunks in ((iters,), (700, 1, 2, 1297)):
        med.begin(L.PIPE_PR, b, x0, iters + 1, hist_mask=1)
        t0 = time.perf_counter()
        for c in chunks:
            med.iterate(c)
        med.sync()
        times.append(time.perf_counter() - t0)
        runs.append(med.history()['updated_residual_2_norm'])
    assert np.array_equal(runs[0], runs[1], equal_nan=True)
    multi.begin(L.PIPE_PR, b, x0, iters + 1, hist_mask=1)
    t0 = time.perf_counter()
    multi.iterate(iters)
    multi.sync()
    dt_multi = time.perf_counter() - t0
    ref = multi.history()['updated_residual_2_norm']
    np.testing.assert_allclose(runs[0][:5], ref[:5], rtol=1e-12)
    # same convergence: the residual reduction reached after the run, within the noise of a different summation order
    with np.errstate(all='ignore'):
        got_red, ref_red = np.nanmin(runs[0]) / runs[0][0], np.nanmin(ref) / ref[0]
    assert abs(np.log10(got_red) - np.log10(ref_red)) <= 1.5, (got_red, ref_red)
    print(f'{name} (n={n}, nnz={A.nnz}): {iters} iterations in one launch of {med.schedule()} {times[0] * 1e3:.2f} ms '
          f'({times[0] / iters * 1e6:.2f} us/iteration) vs one launch per iteration {dt_multi * 1e3:.2f} ms '
          f'({dt_multi / iters * 1e6:.2f} us/iteration); forced-step scalar deviation {worst:.1e}')
    med.close()
    multi.close()


def test_device_results_are_reproducible(amd, matrices):
    A, z = matrices['nos7']
    cbs = [amd['cbs'].updated_residual_2_norm]
    a = amd['cgv'].pipe_pr_cg(A, z['b'], np.zeros(729), 2000, callbacks=cbs)
    b = amd['cgv'].pipe_pr_cg(A, z['b'], np.zeros(729), 2000, callbacks=cbs)
    assert np.array_equal(a['updated_residual_2_norm'], b['updated_residual_2_norm'])


def test_foreign_callback_sees_reference_locals(amd, matrices):
    """Any callable in callbacks= is honoured with the reference's local names."""
    A, z = matrices['bcsstk03']
    seen = []

    def spy(**kw):
        seen.append((kw['k'], float(np.linalg.norm(kw['r_k'])), kw['output']['name'], 'x_true' in kw['kwargs']))
    out = amd['cgv'].pipe_pr_cg(A, z['b'], np.zeros(112), 30, callbacks=[spy, amd['cbs'].updated_residual_2_norm],
                                x_true=z['x_true'])
    assert [s[0] for s in seen] == list(range(30))
    np.testing.assert_allclose([s[1] for s in seen], out['updated_residual_2_norm'], rtol=1e-14)
    assert seen[0][2] == 'pipe_pr_cg' and seen[0][3]


def test_breakdown_propagates_nan_not_exception(amd):
    """Zero right-hand side: 0/0 at the first step; histories fill with nan, no error
    (the reference behaves the same way; figure_gen.py:89 uses nanmin)."""
    A = amd['problems'].laplace_2d(16, 16)
    out = amd['cgv'].pipe_pr_cg(A, np.zeros(256), np.zeros(256), 6, callbacks=[amd['cbs'].updated_residual_2_norm])
    h = out['updated_residual_2_norm']
    assert h[0] == 0.0 and np.all(np.isnan(h[1:]))


# ---------------------------------------------------------------------------------------
# full-size workloads: size-independent properties + oracle agreement on a short run
# ---------------------------------------------------------------------------------------
def test_s1_full_size_short_run_against_device_ordered_oracle(amd):
    A = amd['problems'].laplace_2d(1000, 1000)
    n = A.shape[0]
    b, x0, x_true = amd['problems'].reference_rhs(A, n)
    iters = 12
    want = orc.pipe_pr_cg(A, b, x0, iters, callbacks=['updated_residual_2_norm', 'error_2_norm'],
                          x_true=x_true, dot=device_dot, square=lambda a: a * a)
    got = amd['cgv'].pipe_pr_cg(A, b, x0, iters, callbacks=[amd['cbs'].updated_residual_2_norm,
                                                            amd['cbs'].error_2_norm], x_true=x_true)
    np.testing.assert_allclose(got['updated_residual_2_norm'], want['updated_residual_2_norm'], rtol=1e-13)
    np.testing.assert_allclose(got['error_2_norm'], want['error_2_norm'], rtol=1e-12)
    # and against the reference-order oracle (BLAS dot) within the stated tolerance on this prefix
    ref = orc.pipe_pr_cg(A, b, x0, iters, callbacks=['updated_residual_2_norm'], x_true=x_true)
    np.testing.assert_allclose(got['updated_residual_2_norm'], ref['updated_residual_2_norm'], rtol=1e-12)


def test_s3_banded_cg_monotone_error_and_residual_identity(amd):
    """ex2b-style banded matrix (n=2e5): A-norm error decreases monotonically for HS-CG,
    and the true residual equals the recurrence residual to rounding for both variants."""
    P = amd['problems']
    n = 200_000
    A = P.banded_ex2b(n, 7)
    b, x0, x_true = P.reference_rhs(A, n)
    cbs = [amd['cbs'].error_A_norm, amd['cbs'].residual_2_norm, amd['cbs'].updated_residual_2_norm]
    for fn in (amd['cgv'].hs_cg, amd['cgv'].pipe_pr_cg):
        out = fn(A, b, x0, 40, callbacks=cbs, x_true=x_true)
        e = out['error_A_norm']
        assert np.all(np.diff(e[:25]) < 0), fn.__name__
        assert e[25] < 0.2 * e[0]
        want = getattr(orc, fn.__name__)(A, b, x0, 40, callbacks=['error_A_norm'], x_true=x_true)
        np.testing.assert_allclose(e[:20], want['error_A_norm'][:20], rtol=1e-6)
        np.testing.assert_allclose(e, want['error_A_norm'], rtol=1e-2)
        np.testing.assert_allclose(out['residual_2_norm'][:20], out['updated_residual_2_norm'][:20], rtol=1e-6)


# ---------------------------------------------------------------------------------------
# driver-level drop-ins
# ---------------------------------------------------------------------------------------
def test_figure_run_reproduces_the_reference_table_row(amd, tmp_path):
    """The figure_gen-compatible runner on the bcsstk03 fixture: saved dicts load like the
    reference's, and the table statistics land on the published row within its own spread."""
    import os
    from conftest import GOLDEN
    from new_cg_variants_amd.experiments import figure_run as fr
    A = fr.load_matrix(os.path.join(GOLDEN, 'matrix_bcsstk03.npz'))
    trials = fr.run_matrix(A, 1250, 'bcsstk03', None, fr.TABLE_METHODS, out=str(tmp_path))
    saved = np.load(tmp_path / 'bcsstk03_None' / 'pipe_pr_pcg.npy', allow_pickle=True).item()
    assert saved['name'] == 'pipe_pr_pcg' and saved['max_iter'] == 1250
    assert set(saved) >= {'error_A_norm', 'residual_2_norm', 'error_2_norm', 'updated_residual_2_norm'}
    # published row: hs 364 / m 425 / pr 380 / pipe_pr_m 492 / pipe_pr 411; -14.55 -14.40 -14.43 -12.65 -12.96
    published = {'hs_pcg': (364, -14.55), 'cg_pcg': (439, -14.49), 'm_pcg': (425, -14.40), 'pr_pcg': (380, -14.43),
                 'gv_pcg': (598, -6.86), 'pipe_pr_m_pcg': (492, -12.65), 'pipe_pr_pcg': (411, -12.96)}
    for m, (its_pub, acc_pub) in published.items():
        its, acc = fr.summarize(trials[m])
        assert abs(its - its_pub) <= 0.10 * its_pub, (m, its, its_pub)
        assert abs(acc - acc_pub) <= 1.5, (m, acc, acc_pub)
    row = fr.table_row('bcsstk03', A, None, trials)
    assert row.startswith('\\texttt{bcsstk03} & - & 112 & 640&') and row.rstrip().endswith('\\\\')


# re-orderings of the DEVICE's inner products (other grids = other partial sums, other kernels): what the oracle's
# SUMMATION_ORDERS are for the small matrices, at sizes the oracle cannot iterate 4000 times in a test
DEVICE_ORDERS = [{}, {'PRCG_WIN_GRID_PER_CU': '2'}, {'PRCG_WIN_GRID_PER_CU': '3'}, {'PRCG_WIN_GRID_PER_CU': '6'},
                 {'PRCG_WIN': '0'}]


def test_ex2b_driver_matches_the_published_petsc_errors(amd, capfd, monkeypatch):
    """The reference's PETSc run (n=650000, k=32, rho=.95, kappa=1e6, off=1e-4, 4000 iterations,
    config_info/slurm-864568.out:129,186,205) printed
        cg         Norm of error 1.60099e-07 iterations 4000
        pipeprcg   Norm of error 3.24332e-07 iterations 4000
        pipeprcg_0 Norm of error 8.94408e-05 iterations 4000
    The same command line through the device, under five orders of its own inner products: the published
    value must lie inside the spread of the device's values (+- 0.5 decades, the margin of the convergence rule
    of test_free_running_against_reference) -- PETSc's VecDot order is one more such order."""
    from new_cg_variants_amd.experiments import ex2b
    base = '-n 650000 -rho 0.95 -kappa 1e6 -k 32 -off_value 1e-4 -pc_type none -num_repeat 1 ' \
           '-ksp_norm_type none -ksp_max_it 4000'
    published = {'-ksp_type cg': 1.60099e-07, '-ksp_type pipeprcg': 3.24332e-07,
                 '-ksp_type pipeprcg -recompute_q 0': 8.94408e-05}
    report = []
    for flags, err_pub in published.items():
        errs = []
        for knobs in DEVICE_ORDERS:
            with monkeypatch.context() as mp:
                for key, val in knobs.items():
                    mp.setenv(key, val)
                ex2b.main((base + ' ' + flags).split())
            lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith('Norm of error ')]
            assert len(lines) == 1 and lines[0].endswith(' iterations 4000'), lines
            errs.append(float(lines[0].split()[3]))
        lg = np.log10(errs)
        report.append(f'ex2b {flags}: Norm of error {errs[0]:g} (PETSc run published {err_pub:g}); over {len(errs)} orders '
                      f"of the device's sums {min(errs):.3e}..{max(errs):.3e}")
        assert lg.min() - 0.5 <= np.log10(err_pub) <= lg.max() + 0.5, (flags, errs, err_pub)
    print('\n'.join(report))


@pytest.mark.parametrize('name', ['pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'])
def test_scaling_mirror_against_mp_oracle(amd, name):
    """`sol, t = variant(comm, A, b, max_iter)` (MP/scaling_tests.py:71) for all five files of
    MP/cg_variants/, on one rank: the device result vs the MP oracle (pinned bitwise against the
    reference) on the same sparse operator -- x after 25 iterations, to 1e-11 relative (the inner
    products are summed in a different order; 25 iterations of this well-conditioned problem do
    not amplify that), and the reference's timing dict."""
    from new_cg_variants_amd import scaling
    A = amd['problems'].laplace_2d(48, 40)
    n = A.shape[0]
    b = A @ (np.ones(n) / np.sqrt(n))

    class Whole:
        def matvec_local(self, V):
            return A @ V
    comm = mp_oracle.SingleRankComm()
    x_ref, _ = getattr(mp_oracle, name)(comm, Whole(), b.copy(), 25)
    x, t = getattr(scaling, name)(scaling.SelfComm(), A, b.copy(), 25)
    assert set(t) >= {'tot'} and t['tot'] > 0
    assert rel(x, x_ref) <= 1e-11, (name, rel(x, x_ref))


def test_scaling_mirror_takes_the_reference_drivers_dense_column_block(amd):
    """MP/scaling_tests.py:31-57 builds the diagonal model problem as a dense n x (n/size) column
    block and calls `variant(comm, A, b, max_iter)` with it.  The same call, same arguments, on one
    rank: x after 40 iterations against the fixture written from the reference itself."""
    import os
    from conftest import GOLDEN
    from new_cg_variants_amd import scaling
    z = np.load(os.path.join(GOLDEN, 'mp_model_problem.npz'))
    n = 1024
    lam = mp_oracle.model_problem_eigs(n)
    A = np.zeros((n, n))
    A[np.arange(n), np.arange(n)] = lam                  # scaling_tests.py:51-54 with size = 1
    b = lam / np.sqrt(n)                                 # :57
    comm = mp_oracle.SingleRankComm()
    op = mp_oracle.DenseColumnBlock(comm, A)
    for name in ('pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'):
        # 5 iterations: nothing has been amplified yet
        x5, _ = getattr(scaling, name)(scaling.SelfComm(), A, b.copy(), 5)
        want5, _ = getattr(mp_oracle, name)(comm, op, b.copy(), 5)
        assert rel(x5, want5) <= 1e-11, (name, rel(x5, want5))
        # 40 iterations against the reference's own output.  This model problem (kappa = 1e6) amplifies
        # ANY change of summation order: permuting the order of the reference's ddot moves x by 2e-4
        # (pipe_pr_cg) / 9e-5 (hs_cg) at iteration 40, one iteration more or less by 1e-2.
        x, t = getattr(scaling, name)(scaling.SelfComm(), A, b.copy(), 40)
        assert rel(x, z[f'{name}_n1024_it40_x']) <= 1e-3, (name, rel(x, z[f'{name}_n1024_it40_x']))


def test_scaling_tests_driver_matches_the_published_errors(amd, tmp_path, monkeypatch, capfd):
    """The reference's own run of `scaling_tests.py 12288 1500` (scaling_experiments_mpi4py/data/,
    BASELINE.md / SURVEY.md section 6) ended with errors hs 1.10e-7, cg_cg 2.08e-6, gv 5.74e-5,
    pr 2.26e-7, pipe_pr 4.01e-7.  The same command line through the device must print the same lines, save
    the same dicts, and end with an error inside the spread the MP oracle shows for that variant under the six
    SUMMATION_ORDERS of its inner products (+- 0.5 decades: the one convergence rule of this file); the
    published value must lie in that band too (it is the 'blas' order on the reference's machine)."""
    from new_cg_variants_amd.experiments import scaling_tests
    monkeypatch.chdir(tmp_path)
    n, its = 12288, 1500
    res = scaling_tests.main([str(n), str(its), 'unit'])
    out = capfd.readouterr().out
    published = {'hs_cg': 1.10e-7, 'cg_cg': 2.08e-6, 'gv_cg': 5.74e-5, 'pr_cg': 2.26e-7, 'pipe_pr_cg': 4.01e-7}
    lam = mp_oracle.model_problem_eigs(n)

    class Diagonal:
        def matvec_local(self, V):
            return lam * V if V.ndim == 1 else lam[:, None] * V
    comm = mp_oracle.SingleRankComm()
    for name, err_pub in published.items():
        assert f'{name} error: ' in out
        saved = np.load(tmp_path / 'data' / str(n) / f'{name}_unit.npy', allow_pickle=True).item()
        assert set(saved) == {'error', 'timings'} and saved['timings']['tot'] > 0
        spread = []
        for _, dot in SUMMATION_ORDERS:
            x, _t = getattr(mp_oracle, name)(comm, Diagonal(), lam / np.sqrt(n), its, dot=dot)
            spread.append(np.log10(np.linalg.norm(np.ones(n) / np.sqrt(n) - x)))
        lo, hi = min(spread) - 0.5, max(spread) + 0.5
        print(f'scaling_tests {name}: error {saved["error"]:.3e} (published {err_pub:.2e}; oracle under '
              f'{len(spread)} summation orders {10 ** min(spread):.2e}..{10 ** max(spread):.2e}), '
              f'{its / saved["timings"]["tot"]:.0f} it/s')
        assert lo <= np.log10(saved['error']) <= hi, (name, saved['error'], spread)
        assert lo <= np.log10(err_pub) <= hi, (name, err_pub, spread)
    assert res.keys() == published.keys()


@pytest.mark.parametrize('matrix,method,prec', [('bcsstk03', 'gv_cg', 'None'), ('bcsstk03', 'gv_pcg', 'jacobi'), ('nos7', 'gv_pcg', 'jacobi')])
def test_ghysels_vanroose_residual_replacement_hook_on_the_device(amd, matrices, matrix, method, prec):
    """`gv_cg(A, b, x0, max_iter, w_replace=predicate, ...)` (gv_cg.py:9,69-71; gv_pcg :93,156-158): the predicate is the
    caller's code and is called on the host with the reference's keywords between the x, r, w update and the product
    t = A w~ of every iteration; True replaces w by A r on the device.  Against the reference-generated fixture (pinned
    with the same predicate): the histories agree on the prefix like any free-running comparison (the replacement at
    k = 3 and k = 7 lies inside it); the predicate sees r_ = the previous call's r, x / w / r of the current iteration."""
    import os
    from conftest import GOLDEN
    A, z = matrices[matrix]
    n = A.shape[0]
    fx = np.load(os.path.join(GOLDEN, f'wreplace_{matrix}_{method}_{prec}.npz'))
    max_iter = int(fx['max_iter'])
    seen = []

    def pred(**kw):
        fl = kw['wk_replace_flags']
        fl['calls'] = fl.get('calls', 0) + 1
        if fl['calls'] <= 12:
            seen.append((kw['k'], kw['r'].copy(), kw['r_'].copy(), kw['x'].copy(), kw['w'].copy(), kw['u'].copy()))
        return kw['k'] % 7 == 0 or fl['calls'] in (3, 40)
    kw = {'preconditioner': (lambda v: (1 / A.diagonal()) * v)} if prec == 'jacobi' else {}
    cbs = [getattr(amd['cbs'], q) for q in FOUR]
    out = getattr(amd['cgv'], method)(A, z['b'], np.zeros(n), max_iter, w_replace=pred, callbacks=cbs, x_true=z['x_true'], **kw)
    assert [s[0] for s in seen] == list(range(1, 13))
    assert np.array_equal(seen[0][2], z['b'])                                   # r_ of the first call: r_0 = b - A 0
    for a, c in zip(seen[:-1], seen[1:]):
        assert np.array_equal(c[2], a[1])                                       # r_ = the previous iteration's r
    np.testing.assert_allclose(np.linalg.norm(seen[0][1]), fx['hist_updated_residual_2_norm'][1], rtol=1e-12)
    ref = fx['hist_updated_residual_2_norm']
    got = out['updated_residual_2_norm']
    first = first_k_beyond(got, ref, 1e-11)
    print(f'{matrix}/{method}/{prec} with w_replace: recurrence residual within 1e-11 of the reference for k < {first} (replacements at k = 3, 7)')
    assert first >= (7 if prec == "None" else 4), first
    # without a predicate that fires the same call is the plain method (the keyword is accepted and unused)
    plain = getattr(amd['cgv'], method)(A, z['b'], np.zeros(n), 40, w_replace=lambda **kw: False, callbacks=cbs, x_true=z['x_true'], **kw)
    base = getattr(amd['cgv'], method)(A, z['b'], np.zeros(n), 40, callbacks=cbs, x_true=z['x_true'], **kw)
    np.testing.assert_allclose(plain['updated_residual_2_norm'][:8], base['updated_residual_2_norm'][:8], rtol=1e-11)
