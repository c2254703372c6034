"""figure_gen-compatible experiment runner (no plotting, no LaTeX build).

The reference's numerical_experiments/figure_gen.py does, per (matrix, max_iter,
preconditioner): set up the problem (:31-34), run every variant with the four history
callbacks (:37,:59), np.save the returned dict (:60) and write one row of the paper's
convergence table (:63-115).  This module does the same with the device solvers:

    python -m new_cg_variants_amd.experiments.figure_run --matrix ../matrices/bcsstk03.mtx \
        --max-iter 1250 [--jacobi] [--methods hs_pcg,pr_pcg,pipe_pr_pcg] [--out ./data]

`--matrix` takes a MatrixMarket file or one of this repo's fixtures (tests/golden/matrix_*.npz).
The saved dicts load with `np.load(..., allow_pickle=True).item()` exactly like the reference's.
"""
import argparse
import os

import numpy as np
import scipy.io
import scipy.sparse as sp

from .. import cg_variants as cgv
from ..callbacks import error_2_norm, error_A_norm, print_k, residual_2_norm, updated_residual_2_norm

# the reference's method list (figure_gen.py:346-348), all on the device
DEVICE_METHODS = ['hs_pcg', 'cg_pcg', 'm_pcg', 'gv_pcg', 'pipe_p_m_pcg', 'pipe_pr_m_pcg', 'pr_pcg', 'pipe_p_pcg',
                  'pipe_pr_pcg']
# the columns of the paper's table (figure_gen.py:360)
TABLE_METHODS = ['hs_pcg', 'cg_pcg', 'm_pcg', 'pr_pcg', 'gv_pcg', 'pipe_pr_m_pcg', 'pipe_pr_pcg']


def load_matrix(path):
    if path.endswith('.npz'):
        z = np.load(path)
        n = int(z['n'])
        return sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n))
    return sp.csr_matrix(scipy.io.mmread(path))              # figure_gen.py:350


def run_matrix(A, max_iter, title, preconditioner=None, methods=DEVICE_METHODS, out='./data', progress=False):
    """One (matrix, preconditioner) experiment: returns {method: trial dict} and saves each."""
    N = A.shape[0]
    x_true = np.ones(N) / np.sqrt(N)                         # figure_gen.py:32
    b = A @ x_true                                           # :33
    x0 = np.zeros(N)                                         # :34
    callbacks = [error_A_norm, residual_2_norm, error_2_norm, updated_residual_2_norm]
    if progress:
        callbacks.append(print_k(10))
    prec = cgv.Jacobi(A) if preconditioner == 'jacobi' else (lambda v: v)
    folder = os.path.join(out, f'{title}_{preconditioner}')
    os.makedirs(folder, exist_ok=True)
    trials = {}
    for name in methods:
        trial = getattr(cgv, name)(A, b, x0, max_iter, callbacks=callbacks, x_true=x_true, preconditioner=prec)
        np.save(os.path.join(folder, name), trial, allow_pickle=True)
        trials[name] = trial
    return trials


def summarize(trial, tol=1e-5):
    """(iterations to relative A-norm error <= tol, log10 of the best relative error) --
    the two statistics of the paper's table (figure_gen.py:84-89); 0 iterations = never."""
    rel = trial['error_A_norm'] / trial['error_A_norm'][0]
    with np.errstate(all='ignore'):
        return int(np.argmin(rel > tol)), float(np.log10(np.nanmin(rel)))


def table_row(matrix_name, A, preconditioner, trials, methods=TABLE_METHODS):
    """One LaTeX row in the layout of figures/convergence_table_data.tex: name, preconditioner,
    n, nnz, then iterations per method, then log10 accuracy per method; entries that are
    >10 % slower than the first column (or never converge) resp. lose >10 % of its accuracy
    are wrapped in \\tableemph (figure_gen.py:106-111)."""
    stats = [summarize(trials[m]) for m in methods]
    its0, acc0 = stats[0]
    cells_it, cells_acc = [], []
    for its, acc in stats:
        slow = its == 0 or its > 1.1 * its0
        cells_it.append(('\\tableemph' if slow else '') + '{' + (str(its) if its else '-') + '}')
        cells_acc.append(('\\tableemph' if acc > 0.9 * acc0 else '') + '{' + f'{acc:1.2f}' + '}')
    name = '\\texttt{' + matrix_name.replace('_', '\\_') + '}'
    prec = 'Jac.' if preconditioner == 'jacobi' else '-'
    return f'{name} & {prec} & {A.shape[0]} & {A.nnz}' + ''.join('& ' + c for c in cells_it) + \
        ''.join('&' + c for c in cells_acc) + '\\\\ \n'


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--matrix', required=True)
    ap.add_argument('--max-iter', type=int, required=True)
    ap.add_argument('--jacobi', action='store_true')
    ap.add_argument('--methods', default=','.join(DEVICE_METHODS))
    ap.add_argument('--out', default='./data')
    args = ap.parse_args()
    A = load_matrix(args.matrix)
    title = os.path.splitext(os.path.basename(args.matrix))[0].replace('matrix_', '')
    prec = 'jacobi' if args.jacobi else None
    methods = [m for m in args.methods.split(',') if m]
    trials = run_matrix(A, args.max_iter, title, prec, methods, args.out)
    for m in methods:
        its, acc = summarize(trials[m])
        print(f'{title:12s} {str(prec):7s} {m:16s} iterations to 1e-5: {its:6d}   log10 min rel. A-norm error: {acc:7.2f}')
    if all(m in trials for m in TABLE_METHODS):
        row = table_row(title, A, prec, trials)
        with open(os.path.join(args.out, f'{title}_{prec}', 'convergence.txt'), 'w') as f:
            f.write(row)
        print(row, end='')


if __name__ == '__main__':
    main()
