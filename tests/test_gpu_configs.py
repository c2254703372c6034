"""GPU (MI355X): the BASELINE.json configurations at the sizes they are benchmarked at, and the
row-per-lane window kernels (csrc/prcg_win.hip) against SciPy and against the CSR-adaptive kernels.

Bars: every matrix product on rows that fit a tile is BIT-EXACT vs SciPy's csr_matvec (same
left-to-right order, no FMA); rows summed by a whole wave (longer than a tile) within
1e-14 * sum|a_ij x_j|; short pipelined runs vs the oracle with the reference's summation order
(oracle/ne_oracle.py, BLAS ddot) to 1e-12 and vs the oracle run with the device's reduction tree
(tests/device_order.py) as the two-kernel schedule produces it, bit for bit.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN
from device_order import device_dot
from oracle import ne_oracle as orc

FOUR_HISTS = 15


@pytest.fixture(scope='module')
def amd():
    import new_cg_variants_amd.cg_variants as cgv
    import new_cg_variants_amd.callbacks as cbs
    from new_cg_variants_amd import _lib, device, partition, problems
    return dict(cgv=cgv, cbs=cbs, L=_lib, device=device, problems=problems, partition=partition)


def products_bitexact(op, A, x, label):
    ref = A @ x
    y, _ = op.matvec(x)
    assert np.array_equal(y, ref), f'{label}: SpMV differs from scipy in {np.count_nonzero(y != ref)} rows'
    WU, _ = op.matmat2(np.stack([x, -0.5 * x], axis=1))
    assert np.array_equal(WU[:, 0], ref) and np.array_equal(WU[:, 1], A @ (-0.5 * x)), label


# ---------------------------------------------------------------------------------------
# window kernels: which operators qualify, edge cases, agreement with the CSR-adaptive path
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_window_kernels_selected_for_bands_and_stencils_only(amd):
    P = amd['problems']
    rng = np.random.default_rng(21)
    n = 50_000
    lens = rng.integers(1, 12, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    wide = sp.csr_matrix((rng.standard_normal(indptr[-1]), rng.integers(0, n, size=indptr[-1]).astype(np.int32), indptr),
                         shape=(n, n))
    expect = {'band': (P.banded_ex2b(100_000, 7), True, 1), 'lap2d': (P.laplace_2d(300, 200), True, 0),
              'lap3d': (P.laplace_3d(40, 30, 20), True, 0), 'wide': (wide, False, 4),
              'tiny': (P.laplace_2d(7, 8), False, 1), 'fem': (P.fem_like_3d(12, 3), False, 2)}
    for name, (A, window, col_bytes) in expect.items():
        op = amd['device'].DeviceCSR(A)
        s = op.schedule()
        assert s['window'] == window, (name, s)
        assert s['sliced_rows'] == (name == 'fem'), (name, s)      # lane-per-row slices: medium-length rows that are no window operator
        if window:
            assert s['col_bytes'] == col_bytes, (name, s)
        x = rng.standard_normal(A.shape[0])
        products_bitexact(op, A, x, name)
        op.close()
        off = amd['device'].DeviceCSR(A, knobs={'PRCG_WIN': '0'})
        assert not off.schedule()['window']
        products_bitexact(off, A, x, name + ' (PRCG_WIN=0)')
        off.close()


@pytest.mark.gpu
def test_window_kernels_ragged_empty_unsorted_and_dictionary_edges(amd):
    """Rows of 0..15 entries inside a band (row lengths differ inside a tile), runs of empty rows, unsorted
    and duplicate column indices, signed zeros / inf / nan values, a tile with more than 256 distinct
    values (the operator then streams plain doubles), and a row longer than a window tile (the operator
    then falls back to the CSR-adaptive kernels)."""
    rng = np.random.default_rng(22)
    n = 30_000
    lens = rng.integers(0, 16, size=n)
    lens[500:900] = 0
    lens[-70:] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(indptr[-1])
    rows = np.repeat(np.arange(n), lens)
    cols = (rows + rng.integers(-9, 10, size=nnz)).clip(0, n - 1).astype(np.int32)      # unsorted, duplicates
    quant = rng.integers(-4, 5, size=nnz) / 16.0
    quant[::11] = -0.0
    A = sp.csr_matrix((quant, cols, indptr), shape=(n, n))
    x = rng.standard_normal(n)
    x[17] = 0.0
    ref = A @ x
    for knobs, want_dict in (({}, True), ({'PRCG_VALDICT': '0'}, False)):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        assert s['window'] and s['value_dict'] == want_dict, s
        products_bitexact(op, A, x, f'ragged {knobs}')
        op.close()
    # non-finite values: same bits as scipy wherever scipy's are defined (nan != nan: compare as bit patterns)
    B = A.copy()
    B.data[5] = np.inf
    B.data[50] = np.nan
    B.data[77] = -np.inf
    op = amd['device'].DeviceCSR(B)
    y, _ = op.matvec(x)
    with np.errstate(all='ignore'):
        refB = B @ x
    assert np.array_equal(y, refB, equal_nan=True)
    op.close()
    # > 256 distinct values in one tile: dictionary refused, plain stream, same products
    C = A.copy()
    C.data[indptr[2000]:indptr[2000] + 700] = rng.standard_normal(700)
    op = amd['device'].DeviceCSR(C)
    s = op.schedule()
    assert s['window'] and not s['value_dict'], s
    products_bitexact(op, C, x, 'dictionary overflow')
    op.close()
    # one row longer than a window tile: not a window operator any more
    lens2 = lens.copy()
    lens2[1234] = 1500
    indptr2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
    rows2 = np.repeat(np.arange(n), lens2)
    cols2 = (rows2 + rng.integers(-9, 10, size=indptr2[-1])).clip(0, n - 1).astype(np.int32)
    D = sp.csr_matrix((rng.standard_normal(indptr2[-1]), cols2, indptr2), shape=(n, n))
    op = amd['device'].DeviceCSR(D)
    assert not op.schedule()['window']
    y, _ = op.matvec(x)
    refD = D @ x
    short = lens2 <= 500
    assert np.array_equal(y[short], refD[short])
    assert np.all(np.abs(y[~short] - refD[~short]) <= 1e-14 * (abs(D) @ np.abs(x))[~short])
    op.close()


@pytest.mark.gpu
def test_window_kernels_on_randomised_bands_and_stencils(amd):
    """Seeded sweep over what decides the window geometry: row count not a multiple of the tile, half
    bandwidths from 1 to 40 (2 .. 4 pages, then past the 8-bit geometry), random gaps inside the band, 2-D / 3-D
    grids of odd sizes (128-row tiles, 8- and 12-page windows), value pools small enough for the dictionary
    or not.  Every product bit-exact vs SciPy, whichever kernel family the planner picks."""
    rng = np.random.default_rng(77)
    P = amd['problems']
    seen = set()
    for case in range(14):
        if case < 9:
            n = int(rng.integers(64, 9000))
            k = [1, 2, 5, 7, 9, 11, 17, 30, 40][case]
            rows = np.arange(n)
            offs = np.arange(-k, k + 1)
            keep = rng.random((n, offs.size)) < [1.0, 0.8, 0.4][case % 3]
            keep[:, k] = rng.random(n) < 0.9                              # some rows without a diagonal entry
            J = rows[:, None] + offs[None, :]
            keep &= (J >= 0) & (J < n)
            pool = rng.standard_normal([3, 40, 5000, 40][case % 4])
            V = pool[rng.integers(0, pool.size, size=J.shape)]
            A = sp.csr_matrix((V[keep], J[keep].astype(np.int32), np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32)),
                              shape=(n, n))
        elif case < 12:
            nx, ny = int(rng.integers(9, 140)), int(rng.integers(9, 90))
            A = P.laplace_2d(nx, ny)
        else:
            A = P.laplace_3d(int(rng.integers(5, 30)), int(rng.integers(5, 30)), int(rng.integers(5, 20)))
        x = rng.standard_normal(A.shape[0])
        sizes = {}
        for share in ('1', '0'):           # tiles reading shared stream images / every tile its own
            op = amd['device'].DeviceCSR(A, knobs={'PRCG_WIN_SHARE': share})
            s = op.schedule()
            seen.add((s['window'], s['col_bytes'], s['value_dict']))
            products_bitexact(op, A, x, f'case {case}: n={A.shape[0]} nnz={A.nnz} share={share} {s}')
            sizes[share] = op.operator_bytes()
            op.close()
        assert 0 < sizes['1'] <= sizes['0'] + 64, sizes
    assert {w for w, _, _ in seen} == {True, False} or all(w for w, _, _ in seen)
    assert {(True, 1, True), (True, 2, True)} <= seen, seen


@pytest.mark.gpu
@pytest.mark.parametrize('workload,parts', [('s3_small', 3), ('s1_small', 2)])
def test_row_blocks_with_ghost_columns_reassemble_the_global_product(amd, workload, parts):
    """Row-block partition + localisation (ghost columns behind the owned ones) on the device without a
    communicator: prcg_spmv_ext takes the ghost entries from the caller.  Interior and boundary tiles of
    the window kernels (and of the CSR-adaptive kernels) must give the global product bit for bit."""
    P, part = amd['problems'], amd['partition']
    A = P.WORKLOADS[workload]['make']()
    n = A.shape[0]
    x = np.random.default_rng(23).standard_normal(n)
    ref = A @ x
    offsets = part.even_offsets(n, parts)
    for knobs in ({}, {'PRCG_WIN': '0'}):
        got = np.empty(n)
        for r in range(parts):
            lo, hi = int(offsets[r]), int(offsets[r + 1])
            A_local, ghost_ids = part.localize(A[lo:hi], lo, hi)
            op = amd['device'].DeviceCSR(A_local, knobs=knobs)
            assert op.schedule()['window'] == (not knobs)
            got[lo:hi] = op.matvec_ext(np.concatenate([x[lo:hi], x[ghost_ids]]))
            op.close()
        assert np.array_equal(got, ref), (workload, knobs)


# ---------------------------------------------------------------------------------------
# BASELINE.json configs at bench size
# ---------------------------------------------------------------------------------------
def short_pipelined_run_matches_the_oracles(amd, A, iters=12):
    n = A.shape[0]
    b, x0, x_true = amd['problems'].reference_rhs(A, n)
    cbs = amd['cbs']
    # one-launch schedule (what bench.py times) vs the reference-order oracle
    ref = orc.pipe_pr_cg(A, b, x0, iters, callbacks=['updated_residual_2_norm', 'error_2_norm'], x_true=x_true)
    got = amd['cgv'].pipe_pr_cg(A, b, x0, iters, callbacks=[cbs.updated_residual_2_norm, cbs.error_2_norm], x_true=x_true)
    np.testing.assert_allclose(got['updated_residual_2_norm'], ref['updated_residual_2_norm'], rtol=1e-12)
    np.testing.assert_allclose(got['error_2_norm'], ref['error_2_norm'], rtol=1e-12)
    # two-kernel schedule vs the oracle with the device's own reduction tree: every inner product bit for bit
    L = amd['L']
    want = []
    orc.pipe_pr_cg(A, b, x0, iters, dot=device_dot, square=lambda a: a * a,
                   tap=lambda st: want.append((st.mu, st.dl, st.gm, st.nu)))
    op = amd['device'].DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    op.begin(L.PIPE_PR, b, x0, iters)
    op.iterate(iters - 1)
    op.sync()
    have = np.array([op.get_scalars(k)[[L.S_MU, L.S_DELTA, L.S_GAMMA, L.S_NU]] for k in range(iters)])
    op.close()
    assert np.array_equal(have, np.array(want)), 'two-kernel schedule vs device-ordered oracle'
    return got


@pytest.mark.gpu
def test_s3_full_size_as_benchmarked(amd):
    """S3 = ex2b banded, n = 1e7, 149,999,944 nonzeros: exactly the operator bench.py times, with the
    encodings it runs (window kernels, 1-byte window indices, value dictionary on and off)."""
    P = amd['problems']
    A = P.WORKLOADS['s3']['make']()
    n = A.shape[0]
    assert (n, A.nnz) == (10_000_000, 149_999_944)
    x = np.random.default_rng(31).standard_normal(n)
    ones = np.ones(n)
    for knobs in ({}, {'PRCG_VALDICT': '0'}):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        assert s['window'] and s['col_bytes'] == 1 and s['value_dict'] == (not knobs), s
        y1, _ = op.matvec(ones)
        assert np.array_equal(y1, A @ ones)                      # A*1 = row sums (sequential, as scipy)
        products_bitexact(op, A, x, f's3 {knobs}')
        op.close()
    short_pipelined_run_matches_the_oracles(amd, A)


@pytest.mark.gpu
def test_s2_full_size(amd):
    """S2 = 7-point Laplacian 216^3 (BASELINE.json configs[3]): far neighbours at +-46656.  Pattern tiles by default; with
    PRCG_WIN_PAT=0 or plain values 64-row tiles of six pages and 2-byte window indices (geometry 4); PRCG_WIN_ROWS=128: the
    12-page 128-row geometry."""
    P = amd['problems']
    A = P.WORKLOADS['s2']['make']()
    n = A.shape[0]
    assert (n, A.nnz) == (216 ** 3, 70_263_936)
    x = np.random.default_rng(32).standard_normal(n)
    ones = np.ones(n)
    for knobs, geom in (({}, 5), ({'PRCG_WIN_SWEEP': '0'}, 5), ({'PRCG_VALDICT': '0'}, 4), ({'PRCG_WIN_PAT': '0'}, 4), ({'PRCG_WIN_PAT': '0', 'PRCG_WIN_ROWS': '128'}, 3),
                        ({'PRCG_WIN': '0'}, -1)):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        assert s['window'] == ('PRCG_WIN' not in knobs), s
        assert op.layout()['geometry'] == geom, (knobs, op.layout()['geometry'])
        assert (op.layout()['sweep_waves'] > 0) == (knobs == {}), knobs          # 1e7 rows on full grid planes: sweep table by default
        y1, _ = op.matvec(ones)
        assert np.array_equal(y1, A @ ones)
        products_bitexact(op, A, x, f's2 {knobs}')
        op.close()
    short_pipelined_run_matches_the_oracles(amd, A)


@pytest.mark.gpu
@pytest.mark.parametrize('workload', ['s4', 's4b_80', 's4b', 's4c'])
def test_irregular_standins_at_bench_size(amd, workload):
    """BASELINE.json configs[4] (Queen_4147, irregular degrees) cannot be fetched: its labelled stand-ins at the size
    bench.py --workload s4 / s4b / s4c runs them (s4b, s4c: Queen_4147's own size, 4.1 M rows; s4b_80: rounds 2-3).  SpMV bit-exact on rows that fit a tile, rows summed
    by a whole wave within 1e-14 * sum|a x|; the nnz-balanced 8-way row-block split (what 8 ranks would own)
    reassembles the global product on the device."""
    P, part = amd['problems'], amd['partition']
    A = P.WORKLOADS[workload]['make']()
    n = A.shape[0]
    x = np.random.default_rng(33).standard_normal(n)
    ref = A @ x
    lens = np.diff(A.indptr)
    op = amd['device'].DeviceCSR(A)
    cap = 256 * op.schedule()['tile_steps'] - 3
    y, _ = op.matvec(x)
    fits = lens <= cap
    assert fits.sum() >= n - 1000
    assert np.array_equal(y[fits], ref[fits])
    if (~fits).any():
        scale = (abs(A[np.nonzero(~fits)[0]]) @ np.abs(x))
        assert np.all(np.abs(y[~fits] - ref[~fits]) <= 1e-14 * scale)
    WU, _ = op.matmat2(np.stack([x, 2.0 * x], axis=1))
    assert np.array_equal(WU[fits, 0], ref[fits]) and np.array_equal(WU[fits, 1], (A @ (2.0 * x))[fits])
    op.close()
    offsets = part.nnz_balanced_offsets(A.indptr, 8)
    nnz_of = [int(A.indptr[offsets[r + 1]] - A.indptr[offsets[r]]) for r in range(8)]
    assert max(nnz_of) <= 1.15 * np.mean(nnz_of)
    got = np.empty(n)
    for r in range(8):
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        A_local, ghost_ids = part.localize(A[lo:hi], lo, hi)
        dev = amd['device'].DeviceCSR(A_local)
        got[lo:hi] = dev.matvec_ext(np.concatenate([x[lo:hi], x[ghost_ids]]))
        dev.close()
    assert np.array_equal(got[fits], ref[fits])
    assert np.all(np.abs(got - ref) <= 1e-14 * (abs(A) @ np.abs(x)))
    # a short solve: HS-CG and the pipelined variant against the oracle.  The stand-ins are weakly diagonally dominant
    # (condition number ~1e4, so that a benchmark run of hundreds of iterations stays finite): a free-running history leaves
    # 1e-9 of the oracle's after a handful of iterations, like every ill-conditioned system (tests/test_gpu_parity.py) --
    # the first five entries are held (s4's history then jumps by factors of 100 from step to step: nothing to compare)
    b, x0, x_true = P.reference_rhs(A, n)
    for name in ('hs_cg', 'pipe_pr_cg'):
        want = getattr(orc, name)(A, b, x0, 8, callbacks=['updated_residual_2_norm'], x_true=x_true)
        got_h = getattr(amd['cgv'], name)(A, b, x0, 8, callbacks=[amd['cbs'].updated_residual_2_norm], x_true=x_true)
        np.testing.assert_allclose(got_h['updated_residual_2_norm'][:5], want['updated_residual_2_norm'][:5], rtol=1e-9,
                                   atol=1e-11 * want['updated_residual_2_norm'][0])
        assert np.all(np.isfinite(got_h['updated_residual_2_norm']))


# ---------------------------------------------------------------------------------------
# the Queen_4147 loader (MatrixMarket, symmetric, lower triangle stored -- the layout of the
# reference's matrices/*.mtx, read as numerical_experiments/figure_gen.py:350 does)
# ---------------------------------------------------------------------------------------
def test_queen_loader_reads_a_symmetric_matrixmarket_file(monkeypatch):
    import scipy.io
    from new_cg_variants_amd import problems
    path = os.path.join(GOLDEN, 'tiny_symmetric.mtx')
    monkeypatch.setenv('QUEEN_4147_MTX', path)
    monkeypatch.delitem(problems.WORKLOADS, 'queen', raising=False)
    problems._register_queen()
    wl = problems.WORKLOADS['queen']
    A = wl['make']()
    want = sp.csr_matrix(scipy.io.mmread(path))                  # figure_gen.py:350
    assert wl['n'] == 40 and A.shape == (40, 40) and A.dtype == np.float64
    assert A.nnz == want.nnz == 2 * 126 - 40                     # the stored triangle mirrored
    assert abs(A - A.T).max() == 0 and abs(A - want).max() == 0 and A.has_sorted_indices
    rows = wl['make'](rows=(10, 25))
    assert rows.shape == (15, 40) and abs(rows - A[10:25]).max() == 0
    monkeypatch.delitem(problems.WORKLOADS, 'queen', raising=False)


@pytest.mark.gpu
def test_queen_workload_runs_on_the_device(amd, monkeypatch):
    problems = amd['problems']
    monkeypatch.setenv('QUEEN_4147_MTX', os.path.join(GOLDEN, 'tiny_symmetric.mtx'))
    monkeypatch.delitem(problems.WORKLOADS, 'queen', raising=False)
    problems._register_queen()
    A = problems.WORKLOADS['queen']['make']()
    n = A.shape[0]
    x = np.random.default_rng(34).standard_normal(n)
    op = amd['device'].DeviceCSR(A)
    products_bitexact(op, A, x, 'queen stand-in file')
    op.close()
    b, x0, x_true = problems.reference_rhs(A, n)
    want = orc.pipe_pr_cg(A, b, x0, 30, callbacks=['updated_residual_2_norm', 'error_A_norm'], x_true=x_true)
    got = amd['cgv'].pipe_pr_cg(A, b, x0, 30, callbacks=[amd['cbs'].updated_residual_2_norm, amd['cbs'].error_A_norm],
                                x_true=x_true)
    np.testing.assert_allclose(got['updated_residual_2_norm'][:12], want['updated_residual_2_norm'][:12], rtol=1e-12)
    assert got['error_A_norm'][25] < 1e-8 * got['error_A_norm'][0]
    monkeypatch.delitem(problems.WORKLOADS, 'queen', raising=False)


@pytest.mark.gpu
def test_window_formation_launches_on_randomised_spd_bands_and_grids(amd):
    """The launches that FORM their staged window from old vectors (Hestenes-Stiefel, predict-and-recompute,
    Chronopoulos-Gear, Ghysels-Vanroose) and the one-launch pipelined iteration, on operators that decide the window
    geometry the awkward way: row counts that are no multiple of the tile, 1 .. 30 half bandwidths with random gaps,
    grids of odd sizes, pages that reach the end of the vectors.  30 iterations of every family, with and without
    Jacobi, against the same solve with one kernel per step (PRCG_FUSED=0): same arithmetic per element, so the
    iterates agree to rounding; and nothing may fault (every window source carries spare entries behind its end)."""
    L = amd['L']
    P = amd['problems']
    rng = np.random.default_rng(2024)
    cases = []
    for k, n in ((1, 67), (2, 200), (5, 1999), (7, 4097), (12, 777), (30, 2500)):
        offs = np.arange(1, k + 1)
        keep = rng.random((n, k)) < 0.7
        rows = np.repeat(np.arange(n), k).reshape(n, k)
        cols = rows + offs[None, :]
        ok = keep & (cols < n)
        vals = rng.standard_normal((n, k))
        U = sp.csr_matrix((vals[ok], (rows[ok], cols[ok])), shape=(n, n))
        S = U + U.T
        A = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0 + rng.random(n))).tocsr()
        A.sort_indices()
        cases.append((f'band k={k} n={n}', A))
    cases.append(('lap2d 37x29', P.laplace_2d(37, 29)))
    cases.append(('lap3d 9x11x7', P.laplace_3d(9, 11, 7)))
    worst = 0.0
    for name, A in cases:
        n = A.shape[0]
        b = A @ (np.ones(n) / np.sqrt(n))
        for prec in (None, 1 / A.diagonal()):
            ops = [amd['device'].DeviceCSR(A, knobs={'PRCG_FUSED': f, 'PRCG_SMALL': '0'}) for f in ('1', '0')]
            for variant in ('HS', 'PR', 'M', 'CG_CG', 'GV', 'PIPE_PR', 'PIPE_P_M'):
                xs = []
                for op in ops:
                    op.begin(getattr(L, variant), b, np.zeros(n), 32, inv_diag=prec, hist_mask=1)
                    op.iterate(7)
                    op.iterate(23)
                    op.sync()
                    xs.append((op.get_vector('x'), op.history()['updated_residual_2_norm'], op.schedule()['fused']))
                assert xs[0][2] == ops[0].schedule()['window'] or variant.startswith('PIPE') or variant == 'HS', (name, variant)
                assert not xs[1][2]
                assert np.all(np.isfinite(xs[1][0])), (name, variant)
                scale = np.max(np.abs(xs[1][0]))
                dev = float(np.max(np.abs(xs[0][0] - xs[1][0])) / scale)
                worst = max(worst, dev)
                assert dev <= 1e-9, (name, variant, prec is not None, dev)
                np.testing.assert_allclose(xs[0][1][:8], xs[1][1][:8], rtol=1e-10, err_msg=f'{name} {variant}')
            for op in ops:
                op.close()
    print(f'{len(cases)} operators x 7 variants x 2 preconditioners: worst deviation of x after 30 iterations {worst:.1e}')


@pytest.mark.gpu
def test_short_window_source_is_refused_not_faulted(amd):
    """Round 2's memory fault (a window page of the last tile read past the end of a source vector allocated without
    spare entries) as an error code: with PRCG_DEBUG_SHORT_SOURCES=1 (tests only) a Hestenes-Stiefel session
    allocates r with exactly n entries; the product launch that would stage it must return PRCG_EINVAL."""
    L = amd['L']
    A = amd['problems'].WORKLOADS['s3_small']['make']()
    n = A.shape[0]
    b, x0, _ = amd['problems'].reference_rhs(A, n)
    op = amd['device'].DeviceCSR(A, knobs={'PRCG_DEBUG_SHORT_SOURCES': '1'})
    op.begin(L.HS, b, x0, 8)
    assert op.schedule()['window'] and op.schedule()['fused']
    with pytest.raises(L.PrcgError) as exc:
        op.iterate(1)
    assert exc.value.code == L.EINVAL and 'launch refused' in str(exc.value)
    op.close()
    op = amd['device'].DeviceCSR(A)                     # the same session with the regular allocation runs
    op.begin(L.HS, b, x0, 8)
    op.iterate(5)
    op.sync()
    assert np.isfinite(op.get_scalars(5)[L.S_NU])
    op.close()


@pytest.mark.gpu
def test_sliced_row_kernels_on_randomised_medium_rows(amd):
    """The lane-per-row kernels over 64-row slices (prcg_sell.hip; FEM-like operators, config 5): ragged rows of 28..125
    nonzeros (within a quarter of each other: the padding bound), unsorted and duplicate indices, empty rows, +-0 / inf / nan values, a row block with ghost columns -- the
    products are scipy's csr_matvec bit for bit, and bit for bit what the CSR-adaptive kernels (PRCG_SELL=0) give."""
    rng = np.random.default_rng(77)
    for trial in range(4):
        n = int(rng.integers(3000, 20000))
        lo_len = (64, 100, 28, 81)[trial]
        lens = rng.integers(lo_len, lo_len + lo_len // 4 + 1, size=n)      # (slices are padded to their longest row: <= 25 % overhead)
        lens[rng.integers(0, n, size=n // 100)] = 0
        if trial == 3:
            lens[:] = 81                                      # every row full: no padding, no masked slot
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(n), lens)
        cols = np.clip(rows + rng.integers(-2000, 2001, size=rows.size), 0, n - 1).astype(np.int32)
        vals = rng.standard_normal(rows.size)
        if trial == 1:
            vals[rng.integers(0, vals.size, size=50)] = 0.0
            vals[rng.integers(0, vals.size, size=50)] = -0.0
            vals[rng.integers(0, vals.size, size=5)] = np.inf
            vals[rng.integers(0, vals.size, size=5)] = np.nan
        A = sp.csr_matrix((vals, cols, indptr), shape=(n, n))
        A.has_canonical_format = False
        x = rng.standard_normal(n)
        with np.errstate(all='ignore'):
            ref = A @ x
            ref2 = A @ (2.0 * x[::-1])
        op = amd['device'].DeviceCSR(A)
        s = op.schedule()
        assert s['sliced_rows'] and not s['window'], s
        y, _ = op.matvec(x)
        assert np.array_equal(y, ref, equal_nan=True), trial
        WU, _ = op.matmat2(np.stack([x, 2.0 * x[::-1]], axis=1))
        assert np.array_equal(WU[:, 0], ref, equal_nan=True) and np.array_equal(WU[:, 1], ref2, equal_nan=True), trial
        op.close()
        off = amd['device'].DeviceCSR(A, knobs={'PRCG_SELL': '0'})
        assert not off.schedule()['sliced_rows']
        assert np.array_equal(off.matvec(x)[0], ref, equal_nan=True)
        off.close()
    # a row block with ghost columns: interior slices and slices touching ghosts
    A = amd['problems'].fem_like_3d(14, 3)
    n = A.shape[0]
    x = rng.standard_normal(n)
    got = np.empty(n)
    offsets = amd['partition'].even_offsets(n, 3)
    for r in range(3):
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        A_local, ghost_ids = amd['partition'].localize(A[lo:hi], lo, hi)
        dev = amd['device'].DeviceCSR(A_local)
        assert dev.schedule()['sliced_rows']
        got[lo:hi] = dev.matvec_ext(np.concatenate([x[lo:hi], x[ghost_ids]]))
        dev.close()
    assert np.array_equal(got, A @ x)


@pytest.mark.gpu
def test_sliced_rows_with_sorting_windows(amd):
    """SELL-C-sigma (operators whose row lengths vary: BASELINE config 5's "irregular-degree" stress): the rows of every
    window of sigma consecutive rows are sorted by length before they are cut into 64-row slices, so a slice's lanes hold
    rows of similar length from anywhere in the window.  Every row is still summed left to right by one lane: products
    bit-exact vs scipy's csr_matvec for every window size, solver steps bit-exact in the vectors against the CSR-adaptive
    kernels; without sorting this operator pads by 30 % and would not qualify."""
    L, P = amd['L'], amd['problems']
    rng = np.random.default_rng(21)
    A = P.fem_irregular_3d(22)
    n = A.shape[0]
    lens = np.diff(A.indptr)
    assert lens.max() >= 2.5 * lens.mean() or lens.min() * 3 <= lens.mean()
    x = rng.standard_normal(n)
    ref, ref2 = A @ x, A @ (3.0 * x[::-1])
    for knobs in ({'PRCG_SELL_WINDOW': '0'}, {'PRCG_SELL_SIGMA': '256'}, {'PRCG_SELL_SIGMA': '1024', 'PRCG_SELL_PLANES': '0'}, {'PRCG_SELL_NT': '1', 'PRCG_SELL_WINDOW': '0'}):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        assert op.schedule()['sliced_rows'] and op.schedule()['sorted_windows'], knobs
        assert np.array_equal(op.matvec(x)[0], ref), knobs
        WU, _ = op.matmat2(np.stack([x, 3.0 * x[::-1]], axis=1))
        assert np.array_equal(WU[:, 0], ref) and np.array_equal(WU[:, 1], ref2), knobs
        op.close()
    off = amd['device'].DeviceCSR(A, knobs={'PRCG_SELL_SIGMA': '64'})
    assert not off.schedule()['sliced_rows']                 # consecutive rows: 30 % padding, refused
    off.close()
    # a row block with ghost columns (interior windows and windows of rows touching ghosts are sorted apart)
    got = np.empty(n)
    offsets = amd['partition'].nnz_balanced_offsets(A.indptr, 3)
    for r in range(3):
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        A_local, ghost_ids = amd['partition'].localize(A[lo:hi], lo, hi)
        dev = amd['device'].DeviceCSR(A_local)
        assert dev.schedule()['sliced_rows']
        got[lo:hi] = dev.matvec_ext(np.concatenate([x[lo:hi], x[ghost_ids]]))
        dev.close()
    assert np.array_equal(got, ref)
    # forced steps of the pipelined iteration (one launch) and of Hestenes-Stiefel against the CSR-adaptive kernels
    b, x0, _ = P.reference_rhs(A, n)
    for variant, stored in (('PIPE_PR', ['x', 'r', 'p', 's']), ('HS', ['x', 'r', 'p', 's'])):
        ops = [amd['device'].DeviceCSR(A, knobs={'PRCG_SELL': f, 'PRCG_SELL_WINDOW': '0'}) for f in ('1', '0')]
        for op in ops:
            op.begin(getattr(L, variant), b, x0, 20)
        worst = 0.0
        for k in range(8):
            st = {v: ops[1].get_vector(v) for v in stored}
            for v, a in st.items():
                ops[0].set_vector(v, a)
            ops[0].set_scalars(k, ops[1].get_scalars(k))
            ops[0].set_iteration(k)
            for op in ops:
                op.iterate(1)
            for v in stored:
                assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v)), (variant, k, v)
            a, c = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
            nz = c != 0
            worst = max(worst, float(np.max(np.abs(a[nz] - c[nz]) / np.abs(c[nz]))))
        assert worst <= 1e-12, (variant, worst)
        for op in ops:
            op.close()


@pytest.mark.gpu
def test_sliced_rows_with_window_codes(amd):
    """WINDOW codes (prcg_plan.h; k_sell_win): for slices of consecutive rows whose columns fit 64 granules of 16 consecutive
    entries -- an assembled 3-D matrix in natural ordering -- the kernel stages those entries in LDS and every nonzero reads its
    operand there instead of gathering it from memory.  Same lane, same left-to-right sum: products bit-exact vs scipy's
    csr_matvec and vs the delta-code kernels, with one code per run of three columns and with one per nonzero, empty rows,
    rows whose runs are stored in descending order, nontemporal stream loads, a launch with more waves than slices; operators
    that do not qualify (sorting windows, scattered columns) keep the delta codes."""
    P = amd['problems']
    rng = np.random.default_rng(5)
    A = P.fem_like_3d(14, 3).tolil()
    for r in rng.integers(0, A.shape[0], size=40):
        A.rows[r], A.data[r] = [], []                       # empty rows (a whole slice of them too)
    for r in range(640, 704):
        A.rows[r], A.data[r] = [], []
    A = A.tocsr()
    for r in range(0, A.shape[0], 5):                      # the runs of some rows in descending order
        lo, hi = A.indptr[r], A.indptr[r + 1]
        k = (hi - lo) // 3
        o = (np.arange(k)[::-1][:, None] * 3 + np.arange(3)[None, :]).ravel()
        A.indices[lo:hi] = A.indices[lo:hi][o]; A.data[lo:hi] = A.data[lo:hi][o]
    A.has_sorted_indices = False
    n = A.shape[0]
    x = rng.standard_normal(n)
    ref, ref2 = A @ x, A @ (3.0 * x[::-1])
    for knobs, window in (({}, True), ({'PRCG_SELL_RUNS': '0'}, True), ({'PRCG_SELL_NT': '1'}, True), ({'PRCG_SELL_GRID_PER_CU': '1'}, True),
                          ({'PRCG_SELL_WINDOW': '0'}, False), ({'PRCG_SELL_SIGMA': '256'}, False)):
        op = amd['device'].DeviceCSR(A, knobs=knobs)
        s = op.schedule()
        assert s['sliced_rows'] and s['window_codes'] == window, (knobs, s)
        assert np.array_equal(op.matvec(x)[0], ref), knobs
        WU, _ = op.matmat2(np.stack([x, 3.0 * x[::-1]], axis=1))
        assert np.array_equal(WU[:, 0], ref) and np.array_equal(WU[:, 1], ref2), knobs
        op.close()
    # a footprint of 49..64 granules a slice: the kernels' larger window (16 page loads), one code per run of three
    W = P.block_band_3dof(1500, 120)
    xw = rng.standard_normal(W.shape[0])
    op = amd['device'].DeviceCSR(W, knobs={'PRCG_SELL_SIGMA': '64'})
    assert op.schedule()['window_codes']
    assert np.array_equal(op.matvec(xw)[0], W @ xw)
    WU, _ = op.matmat2(np.stack([xw, 3.0 * xw[::-1]], axis=1))
    assert np.array_equal(WU[:, 0], W @ xw) and np.array_equal(WU[:, 1], W @ (3.0 * xw[::-1]))
    op.close()
    # a window of 24 granules where the slices need up to 33: the planner cuts them (their other lanes idle)
    op = amd['device'].DeviceCSR(A, knobs={'PRCG_SELL_WINDOW': '24', 'PRCG_SELL_SIGMA': '64', 'PRCG_SELL_MAX_OVERHEAD_PCT': '600'})
    assert op.schedule()['window_codes']
    assert np.array_equal(op.matvec(x)[0], ref)
    WU, _ = op.matmat2(np.stack([x, 3.0 * x[::-1]], axis=1))
    assert np.array_equal(WU[:, 0], ref) and np.array_equal(WU[:, 1], ref2)
    op.close()
    # few slices, many waves (most waves of the launch own nothing)
    small = P.fem_like_3d(6, 3)
    op = amd['device'].DeviceCSR(small)
    assert op.schedule()['window_codes']
    xs = rng.standard_normal(small.shape[0])
    assert np.array_equal(op.matvec(xs)[0], small @ xs)
    op.close()
    # irregular row lengths: consecutive rows pad by 30 %, a sorting window by 2 % -- but a sorted slice's rows are no neighbours
    # and gather from memory: consecutive rows with window codes are preferred up to 1.32 x the sorted bytes (prcg_plan.h)
    C = P.fem_irregular_3d(12)
    xc = rng.standard_normal(C.shape[0])
    for knobs, window in (({}, True), ({'PRCG_SELL_WINDOW': '0'}, False)):
        op = amd['device'].DeviceCSR(C, knobs=knobs)
        s = op.schedule()
        assert s['sliced_rows'] and s['window_codes'] == window and s['sorted_windows'] == (not window), (knobs, s)
        assert np.array_equal(op.matvec(xc)[0], C @ xc), knobs
        op.close()
    # the pipelined iteration, free-running: window codes against delta codes, every vector bit for bit (same launches, same sums)
    L = amd['L']
    B = P.fem_like_3d(16, 3)
    b, x0, _ = P.reference_rhs(B, B.shape[0])
    # (the same slices in both: consecutive rows -- left to itself the planner would sort this small grid's rows when it has no window codes)
    ops = [amd['device'].DeviceCSR(B, knobs={'PRCG_SELL_WINDOW': f, 'PRCG_SELL_SIGMA': '64'}) for f in ('1', '0')]
    for variant, inv_diag in ((L.PIPE_PR, None), (L.PIPE_P_M, 1 / B.diagonal()), (L.PR, None)):
        for op in ops:
            op.begin(variant, b, x0, 30, inv_diag=inv_diag)
            op.iterate(25); op.sync()
        assert ops[0].schedule()['window_codes'] and not ops[1].schedule()['window_codes']
        for v in ('x', 'r', 'p', 's'):
            assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v)), v
        assert np.array_equal(ops[0].get_scalars(25), ops[1].get_scalars(25))
    for op in ops:
        op.close()


@pytest.mark.gpu
@pytest.mark.parametrize('window', ['1', '0'])
@pytest.mark.parametrize('variant,prec', [('PIPE_PR', None), ('PIPE_P_M', 'jacobi'), ('HS', None), ('PR', 'jacobi'), ('CG_CG', None)])
def test_sliced_row_kernels_run_every_schedule_of_the_tile_kernels(amd, variant, prec, window):
    """Same row epilogues as the CSR-adaptive family: forced single steps of the solver variants on a FEM-like operator
    agree with the CSR-adaptive kernels (PRCG_SELL=0) bit for bit in every vector and to 1e-12 in the scalars (the inner
    products are summed slice by slice instead of tile by tile) -- with window codes (operands from LDS) and with delta
    codes (gathers)."""
    L = amd['L']
    A = amd['problems'].fem_like_3d(16, 3)
    n = A.shape[0]
    b, x0, x_true = amd['problems'].reference_rhs(A, n)
    inv_diag = (1 / A.diagonal()) if prec else None
    ops = [amd['device'].DeviceCSR(A, knobs={'PRCG_SELL': f, 'PRCG_SELL_WINDOW': window}) for f in ('1', '0')]
    for op in ops:
        op.begin(getattr(L, variant), b, x0, 40, inv_diag=inv_diag)
    assert ops[0].schedule()['sliced_rows'] and not ops[1].schedule()['sliced_rows']
    assert ops[0].schedule()['window_codes'] == (window == '1')
    pipelined = variant.startswith('PIPE')
    stored = ['x', 'r', 'p', 's'] + (['rt', 'st'] if (prec and variant != 'HS') else (['rt'] if prec else []))
    if variant == 'CG_CG':
        stored = ['x', 'r', 'p', 's', 'w']
    if variant == 'PIPE_P_M':
        stored += ['w'] + (['wt'] if prec else [])
    worst = 0.0
    for k in range(12):
        st = {v: ops[1].get_vector(v) for v in stored}
        sc = ops[1].get_scalars(k)
        for v, a in st.items():
            ops[0].set_vector(v, a)
        ops[0].set_scalars(k, sc)
        ops[0].set_iteration(k)
        for op in ops:
            op.iterate(1)
        for v in stored:
            if variant == 'CG_CG' and v in ('p', 's'):
                # Chronopoulos-Gear: b = nu_k / nu_k1 with nu_k summed by the product launch itself (slice by slice here,
                # tile by tile there): p and s inherit b's rounding
                np.testing.assert_allclose(ops[0].get_vector(v), ops[1].get_vector(v), rtol=1e-11, atol=1e-16, err_msg=f'{k} {v}')     # (an entry that cancels to ~1e-4 of its terms)
            else:
                assert np.array_equal(ops[0].get_vector(v), ops[1].get_vector(v), equal_nan=True), (k, v)
        a, c = ops[0].get_scalars(k + 1)[:5], ops[1].get_scalars(k + 1)[:5]
        nz = c != 0
        worst = max(worst, float(np.max(np.abs(a[nz] - c[nz]) / np.abs(c[nz]))))
    assert worst <= 1e-12, worst
    assert pipelined == ops[0].schedule()['fused'] or not pipelined
    for op in ops:
        op.close()


@pytest.mark.gpu
def test_64_row_tiles_for_3d_stencils(amd):
    """PRCG_WIN_ROWS=64 on 3-D stencils: 64-row tiles with up to eight pages and 2-byte window indices (geometry 4; the default
    keeps 128-row tiles, which measured within 4 % on S2 and 25 % better on S1).  Their stream images repeat with the period
    of a grid plane, the launch picks a wave count that is a multiple of it, and the row cache applies: products bit-exact
    vs SciPy, solves identical to the default geometry's to rounding."""
    L, P = amd['L'], amd['problems']
    rng = np.random.default_rng(5)
    for A in (P.laplace_3d(64, 48, 40), P.laplace_3d(216, 30, 20)):
        n = A.shape[0]
        x = rng.standard_normal(n)
        b, x0, _ = P.reference_rhs(A, n)
        hist = []
        for knobs in ({'PRCG_WIN_ROWS': '64'}, {'PRCG_WIN_ROWS': '64', 'PRCG_VALDICT': '0'}, {'PRCG_WIN_PAT': '0'}):
            op = amd['device'].DeviceCSR(A, knobs=knobs)
            s = op.schedule()
            assert s['window'] and s['col_bytes'] == 2, s
            if 'PRCG_WIN_ROWS' in knobs:
                assert op.layout()['rows_per_tile'] == 64 and op.layout()['geometry'] == 4, op.layout()
            products_bitexact(op, A, x, f'lap3d {knobs}')
            op.begin(L.PIPE_PR, b, x0, 30, hist_mask=1)
            op.iterate(29)
            op.sync()
            hist.append(op.history()['updated_residual_2_norm'])
            op.close()
        for h in hist[:2]:
            np.testing.assert_allclose(h, hist[2], rtol=1e-10)


@pytest.mark.gpu
def test_pattern_tiles_for_constant_coefficient_stencils(amd):
    """Constant-coefficient stencils run as PATTERN tiles (geometry 5; csrc/prcg_plan.h: plan_window_patterns): no per-nonzero
    stream, per tile one pattern record (slot offsets + values) and, where rows are incomplete, their slot masks.  Products
    bit-exact vs SciPy -- grid edges, rows of a last partial tile, inf / nan / signed zeros in x, an anisotropic stencil with
    four values, a 9-point stencil, a row block with ghost columns (rows not ascending in local numbering); every solver
    family agrees with the stream geometries (PRCG_WIN_PAT=0) from identical state: vectors bit for bit."""
    L, P, partition = amd['L'], amd['problems'], amd['partition']
    rng = np.random.default_rng(77)
    T = lambda k, c: sp.diags([-c * np.ones(k - 1), 2 * c * np.ones(k), -c * np.ones(k - 1)], [-1, 0, 1])
    nx, ny, nz = 33, 21, 17
    aniso = (sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), T(nx, 1.0))) + sp.kron(sp.eye(nz), sp.kron(T(ny, 0.25), sp.eye(nx))) +
             sp.kron(T(nz, 3.0), sp.eye(ny * nx))).tocsr()
    aniso.sort_indices()
    n1, n2 = 130, 77
    S1 = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
    S2 = sp.diags([np.ones(n2 - 1), np.ones(n2), np.ones(n2 - 1)], [-1, 0, 1])
    nine = (-sp.kron(S2, S1) + 9.0 * sp.eye(n1 * n2)).tocsr()
    nine.sort_indices()
    cases = [('lap3d 70^3', P.laplace_3d(70, 70, 70)), ('lap3d 216x31x9', P.laplace_3d(216, 31, 9)), ('lap2d 1000x77', P.laplace_2d(1000, 77)),
             ('lap2d 37x29', P.laplace_2d(37, 29)), ('aniso', aniso), ('nine-point', nine)]
    for name, A in cases:
        n = A.shape[0]
        op = amd['device'].DeviceCSR(A)
        s = op.schedule()
        assert s['window'] and s['pattern'] and s['value_dict'], (name, s)
        assert op.layout()['geometry'] == 5 and op.layout()['rows_per_tile'] == 64, op.layout()
        x = rng.standard_normal(n)
        products_bitexact(op, A, x, name)
        xs = x.copy()
        xs[rng.integers(0, n, 40)] = np.inf
        xs[rng.integers(0, n, 40)] = np.nan
        xs[rng.integers(0, n, 200)] = -0.0
        xs[rng.integers(0, n, 200)] = 0.0
        with np.errstate(invalid='ignore'):
            ref = A @ xs
        y, _ = op.matvec(xs)
        assert np.array_equal(y, ref, equal_nan=True) and np.array_equal(np.signbit(y), np.signbit(ref)), name
        # solver families from identical state against the stream geometries
        b, x0, _ = P.reference_rhs(A, n)
        other = amd['device'].DeviceCSR(A, knobs={'PRCG_WIN_PAT': '0'})
        assert not other.schedule()['pattern']
        for variant, prec in (('PIPE_PR', None), ('PIPE_P', 1 / A.diagonal()), ('HS', None), ('PR', 1 / A.diagonal()), ('CG_CG', None), ('GV', None)):
            hs = []
            for o in (op, other):
                o.begin(getattr(L, variant), b, x0, 26, inv_diag=prec, hist_mask=1)
                o.iterate(25)
                o.sync()
                hs.append((o.get_vector('x'), o.history()['updated_residual_2_norm']))
            # same rows, same order of the row sums; the inner products are summed over other tile shapes: rounding only
            np.testing.assert_allclose(hs[0][1], hs[1][1], rtol=1e-9, err_msg=f'{name} {variant}')
            assert np.max(np.abs(hs[0][0] - hs[1][0])) <= 1e-9 * np.max(np.abs(hs[1][0])), (name, variant)
        op.close(); other.close()
    # a rank's row block (ghost columns numbered behind the owned ones: its rows are not ascending): block products with the
    # caller's ghost entries reassemble the global product
    A = P.laplace_3d(40, 32, 24)
    n = A.shape[0]
    x = rng.standard_normal(n)
    ref = A @ x
    offsets, blocks = partition.split_serial(A, 3, offsets=np.array([0, 40 * 32 * 8, 40 * 32 * 16, n]))
    for r, (A_loc, ghosts, halo) in enumerate(blocks):
        op = amd['device'].DeviceCSR(A_loc)
        assert op.schedule()['pattern'], (r, op.schedule())
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        y = op.matvec_ext(np.concatenate([x[lo:hi], x[ghosts]]))
        assert np.array_equal(y, ref[lo:hi]), r
        op.close()


@pytest.mark.gpu
def test_sweep_tables_keep_shared_pages_in_lds(amd):
    """Stencils on full grid planes (no ghost columns) get a SWEEP table (csrc/prcg_plan.h: plan_sweep_tiles): a wave's consecutive
    tiles are the same rows of consecutive planes and the pages they share are not loaded again.  Products bit-exact vs SciPy
    whether the launch runs the waves the carry bits assume (small and big workgroups) or not (another grid: every page is
    loaded), solves identical to the row-order pattern tiles (PRCG_WIN_SWEEP=0) to rounding; every solver family runs on it."""
    L, P = amd['L'], amd['problems']
    rng = np.random.default_rng(9)
    for name, A in (('lap3d 70^3', P.laplace_3d(70, 70, 70)), ('lap3d 128x64x64', P.laplace_3d(128, 64, 64)), ('lap2d 2048x512', P.laplace_2d(2048, 512))):
        n = A.shape[0]
        x = rng.standard_normal(n)
        b, x0, _ = P.reference_rhs(A, n)
        hist = {}
        force = {'PRCG_WIN_SWEEP': '2', 'PRCG_SWEEP_WAVES': '2048'}          # (by default only operators of 5e6 rows and more: S2 in test_s2_full_size)
        for tag, knobs in (('sweep', force), ('sweep, small workgroups', dict(force, PRCG_WIN_BIG='0')),
                           ('sweep table, other grid', dict(force, PRCG_WIN_GRID_PER_CU='3')), ('row order', {})):
            op = amd['device'].DeviceCSR(A, knobs=knobs)
            s, lay = op.schedule(), op.layout()
            assert s['pattern'], (name, tag, s)
            assert (lay['sweep_waves'] > 0) == bool(knobs), (name, tag, lay['sweep_waves'])
            products_bitexact(op, A, x, f'{name} {tag}')
            for variant in ('PIPE_PR', 'HS', 'PR', 'CG_CG'):
                op.begin(getattr(L, variant), b, x0, 22, hist_mask=1)
                op.iterate(21)
                op.sync()
                hist[(tag, variant)] = op.history()['updated_residual_2_norm']
                if tag == 'sweep' and variant == 'PIPE_PR':
                    lay2 = op.layout()
                    assert lay2['grid'] * lay2['waves_per_block'] == lay2['sweep_waves'], lay2['grid']       # the carry bits are live
            op.close()
        for (tag, variant), h in hist.items():
            np.testing.assert_allclose(h, hist[('row order', variant)], rtol=1e-9, err_msg=f'{name} {tag} {variant}')


@pytest.mark.gpu
def test_placement_of_the_session_vectors_changes_no_bit(amd):
    """place_session_vectors (prcg_engine.cpp): a pipelined session of 262,144 rows and more allocates its (x,p), (r,s) and second
    (r,s) arrays several times, times every placement with the iteration's byte mix against the operator's own stream and keeps
    the fastest (the price of the row stores depends on where the written arrays lie: profiles/r04_sweeps.md L).  Where the
    vectors lie changes no number: free-running solves with the choice off, with three and with eight placements agree bit
    for bit in every vector and scalar -- window kernels (a band) and sliced rows with window codes (FEM-like); a second session
    on the same handle keeps the placement."""
    L, P = amd['L'], amd['problems']
    for A in (P.banded_ex2b(400_000, 7), P.fem_like_3d(46, 3)):
        n = A.shape[0]
        assert n >= 262_144
        b, x0, _ = P.reference_rhs(A, n)
        ref = None
        for place in ('0', '3', '8'):
            op = amd['device'].DeviceCSR(A, knobs={'PRCG_PLACE': place})
            for session in range(2):
                op.begin(L.PIPE_PR, b, x0, 40)
                op.iterate(30); op.sync()
                got = [op.get_vector(v) for v in ('x', 'r', 'p', 's')] + [op.get_scalars(30)]
                if ref is None:
                    ref = got
                for g, r_ in zip(got, ref):
                    assert np.array_equal(g, r_, equal_nan=True), (place, session)
            op.close()
