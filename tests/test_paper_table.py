"""The reference's end result, redone on the device: the paper's convergence table.

numerical_experiments/figure_gen.py:341-363 runs, for every (matrix, max_iter, preconditioner)
of its list, the seven variants of the published table and reduces each error_A_norm history to
"iterations to a relative error of 1e-5" and "log10 of the best relative error"
(:84-89); the result is committed as figures/convergence_table_data.tex.  The fixture
tests/golden/paper_convergence_table.json holds those published numbers (43 rows: every
matrix the reference ships), tests/golden/tablemat_*.npz the CSR arrays
(tests/golden/make_golden.py table).

The statistics are not bit-level quantities: the reference's own numbers move by a few percent
under any change of summation order (tests/test_gpu_parity.py, PUBLISHED).  Measured on MI355X
(profiles/r01_paper_table.txt): iteration counts within 3.9 % of the published ones in every
cell where the statistic is well defined, accuracies within 0.85 decades (1.41 for gv_pcg).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN

TABLE = os.path.join(GOLDEN, 'paper_convergence_table.json')
CAP = 60000     # bcsstk18 (1.75 M iterations in the reference) and bcsstm25 (130 k) are cut here:
                # their iteration counts (42.5 k, 10-12.7 k) are still reached, the accuracy column is not


def rows():
    with open(TABLE) as f:
        return json.load(f)


def test_table_fixture_is_complete():
    rs = rows()
    assert len(rs) == 43
    for r in rs:
        z = np.load(os.path.join(GOLDEN, f"tablemat_{r['matrix']}.npz"))
        n = int(z['n'])
        A = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n))
        assert (n, A.nnz) == (r['n'], r['nnz'])
        assert A.has_sorted_indices and abs(A - A.T).max() == 0
        assert len(r['iters']) == len(r['log10_min_rel_error_A']) == len(r['columns']) == 7


def inside_oracle_spread(matrix, prec, method, its):
    """The one convergence rule of tests/test_gpu_parity.py for a cell that is further than 6 % from the published
    count: iterations-to-1e-5 inside the spread the reference's own loop shows under six summation orders of its inner
    products (+- 2 %).  Computed only for such cells (the oracle needs seconds per run on the small matrices)."""
    from oracle import ne_oracle as orc
    from test_gpu_parity import SUMMATION_ORDERS
    row = next(r for r in rows() if r['matrix'] == matrix and r['preconditioner'] == prec)
    if row['n'] > 5000:
        return False
    z = np.load(os.path.join(GOLDEN, f'tablemat_{matrix}.npz'))
    n = int(z['n'])
    A = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n))
    x_true = np.ones(n) / np.sqrt(n)
    b = A @ x_true
    d = 1 / A.diagonal()
    fn = (lambda v: d * v) if prec == 'jacobi' else (lambda v: v)
    counts = []
    for _, dot in SUMMATION_ORDERS:
        o = getattr(orc, method)(A, b, np.zeros(n), min(row['max_iter'], CAP), preconditioner=fn, callbacks=['error_A_norm'],
                                 x_true=x_true, dot=dot)
        counts.append(orc.convergence_summary(o['error_A_norm'])[0])
    lo, hi = min(counts), max(counts)
    print(f'{matrix}/{prec}/{method}: {its} iterations on the device; the oracle under six summation orders {lo}..{hi}')
    return min(counts) > 0 and lo - max(2, 0.02 * lo) <= its <= hi + max(2, 0.02 * hi)


@pytest.mark.gpu
def test_device_reproduces_the_published_convergence_table():
    from new_cg_variants_amd.experiments import figure_run as fr
    lines = []
    res = fr.run_published_table(TABLE, GOLDEN, max_iter_cap=CAP, report=lines.append)
    print('\n'.join(lines))
    assert len(res) == 43
    bad = []
    cells = worst_its = worst_acc = 0
    for r in res:
        for j, m in enumerate(r['columns']):
            its, pub_its, pub_acc = r['iters'][j], r['pub_iters'][j], r['pub_acc'][j]
            gv = m == 'gv_pcg'
            # "iterations to 1e-5" is ill defined when the best accuracy a variant ever attains is
            # itself about 1e-5 (gv_pcg on bcsstk16, bcsstm24, nos2, nos7 ...): a hair decides
            # between "reached at iteration N" and "never"
            well_defined = pub_its is not None and pub_acc < -5.6
            if well_defined:
                cells += 1
                dev = abs(its - pub_its) / pub_its
                worst_its = max(worst_its, dev)
                if abs(its - pub_its) > max(3, 0.06 * pub_its) and not inside_oracle_spread(r['matrix'], r['preconditioner'], m, its):
                    bad.append((r['matrix'], r['preconditioner'], m, 'its', its, pub_its))
            elif pub_its is None:
                if its != 0 and r['acc'][j] is not None and r['acc'][j] < -5.6:
                    bad.append((r['matrix'], r['preconditioner'], m, 'converged although the reference never does', its))
            if r['acc'][j] is not None:
                acc = r['acc'][j]
                if np.isneginf(acc):
                    # error exactly zero (bcsstm21, a diagonal matrix solved in 3 steps)
                    ok = pub_acc < -15.0
                else:
                    worst_acc = max(worst_acc, abs(acc - pub_acc))
                    # attained accuracy of the pipelined variants is set by amplified rounding errors: the
                    # reference re-run in the build container lands 2.42 decades from its own published
                    # value on nos7 / pipe_pr (fixture -9.66 vs published -7.24), and a
                    # mere permutation of its dot products moves it over -7.0 .. -9.7.
                    # The bar for those columns is that self-reproducibility of the reference, 2.5 decades;
                    # the non-pipelined columns are held to 1.25.
                    ok = abs(acc - pub_acc) <= (2.5 if (gv or m.startswith('pipe_')) else 1.25)
                if not ok:
                    bad.append((r['matrix'], r['preconditioner'], m, 'acc', acc, pub_acc))
    print(f'{cells} iteration cells compared, worst relative deviation {worst_its:.3f}; '
          f'worst accuracy deviation {worst_acc:.2f} decades')
    assert cells >= 280
    assert not bad, bad
