"""figure_gen-compatible experiment runner (no plotting, no LaTeX build).

The reference's numerical_experiments/figure_gen.py does, per (matrix, max_iter,
preconditioner): set up the problem (:31-34), run every variant with the four history
callbacks (:37,:59), np.save the returned dict (:60) and write one row of the paper's
convergence table (:63-115).  This module does the same with the device solvers:

    python -m new_cg_variants_amd.experiments.figure_run --matrix ../matrices/bcsstk03.mtx \
        --max-iter 1250 [--jacobi] [--methods hs_pcg,pr_pcg,pipe_pr_pcg] [--out ./data]

`--matrix` takes a MatrixMarket file or one of this repo's fixtures (tests/golden/matrix_*.npz).
`--table tests/golden/paper_convergence_table.json --matrix-dir tests/golden` redoes every row of
the published convergence table beside the published numbers.
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


def run_matrix(A, max_iter, title, preconditioner=None, methods=DEVICE_METHODS, out='./data', progress=False,
               callbacks=None):
    """One (matrix, preconditioner) experiment: returns {method: trial dict} and saves each
    (out=None: nothing is written).  callbacks=None: the reference's four histories."""
    N = A.shape[0]
    x_true = np.ones(N) / np.sqrt(N)                         # figure_gen.py:32
    b = A @ x_true                                           # :33
    x0 = np.zeros(N)                                         # :34
    callbacks = list(callbacks) if callbacks is not None else \
        [error_A_norm, residual_2_norm, error_2_norm, updated_residual_2_norm]
    if progress:
        callbacks.append(print_k(10))
    prec = cgv.Jacobi(A) if preconditioner == 'jacobi' else (lambda v: v)
    if out is not None:
        folder = os.path.join(out, f'{title}_{preconditioner}')
        os.makedirs(folder, exist_ok=True)
    trials = {}
    for name in methods:
        trial = getattr(cgv, name)(A, b, x0, max_iter, callbacks=callbacks, x_true=x_true, preconditioner=prec)
        if out is not None:
            np.save(os.path.join(folder, name), trial, allow_pickle=True)
        trials[name] = trial
    return trials


def run_published_table(table_json, matrix_dir, max_iter_cap=None, only=None, report=print):
    """Redo the paper's convergence table (figures/convergence_table_data.tex; figure_gen.py:84-115,
    :341-363) on the device, row by row, beside the published numbers.

    table_json: rows {matrix, preconditioner, max_iter, columns, iters, log10_min_rel_error_A}
    (tests/golden/paper_convergence_table.json); matrix_dir holds `tablemat_<name>.npz` or
    `<name>.mtx`.  A row whose max_iter exceeds max_iter_cap is run for max_iter_cap iterations
    (its accuracy column is then not comparable and comes back as None).
    Returns [{matrix, preconditioner, columns, iters, acc, pub_iters, pub_acc, capped}]."""
    import json
    with open(table_json) as f:
        rows = json.load(f)
    out = []
    cache = {}
    for row in rows:
        name, prec = row['matrix'], row['preconditioner']
        if only and name not in only:
            continue
        if name not in cache:
            cache.clear()
            path = os.path.join(matrix_dir, f'tablemat_{name}.npz')
            cache[name] = load_matrix(path if os.path.exists(path) else os.path.join(matrix_dir, f'{name}.mtx'))
        A = cache[name]
        max_iter = row['max_iter']
        capped = max_iter_cap is not None and max_iter > max_iter_cap
        trials = run_matrix(A, min(max_iter, max_iter_cap) if capped else max_iter, name,
                            'jacobi' if prec == 'jacobi' else None, row['columns'], out=None, callbacks=[error_A_norm])
        stats = [summarize(trials[m]) for m in row['columns']]
        res = dict(matrix=name, preconditioner=prec, columns=row['columns'], capped=capped,
                   iters=[s[0] for s in stats], acc=[None if capped else s[1] for s in stats],
                   pub_iters=row['iters'], pub_acc=row['log10_min_rel_error_A'])
        out.append(res)
        if report:
            report(f"{name:13s} {prec:6s} its  dev " + ' '.join(f'{v:6d}' for v in res['iters']))
            report(f"{'':13s} {'':6s}      pub " + ' '.join(f'{v:6d}' if v else '     -' for v in res['pub_iters']))
            if not capped:
                report(f"{'':13s} {'':6s} acc  dev " + ' '.join(f'{v:6.2f}' for v in res['acc']))
                report(f"{'':13s} {'':6s}      pub " + ' '.join(f'{v:6.2f}' for v in res['pub_acc']))
    return out


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
    ap.add_argument('--matrix')
    ap.add_argument('--max-iter', type=int)
    ap.add_argument('--table', help='JSON of the published table: redo every row (see run_published_table)')
    ap.add_argument('--matrix-dir', default='.', help='with --table: where tablemat_<name>.npz / <name>.mtx live')
    ap.add_argument('--cap', type=int, default=None, help='with --table: cap on max_iter per row')
    ap.add_argument('--jacobi', action='store_true')
    ap.add_argument('--methods', default=','.join(DEVICE_METHODS))
    ap.add_argument('--out', default='./data')
    args = ap.parse_args()
    if args.table:
        run_published_table(args.table, args.matrix_dir, args.cap)
        return
    if not args.matrix or not args.max_iter:
        ap.error('--matrix and --max-iter are required (or --table)')
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
