#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE, in the build container).

    python tests/golden/make_golden.py          # everything
    python tests/golden/make_golden.py table    # only the published convergence table + its matrices
    python tests/golden/make_golden.py mp       # only the mpi4py-variant fixtures

What it does
  1. imports the reference (read-only mount at /root/reference) -- the serial
     ``numerical_experiments`` package directly, the mpi4py scaling variants under
     a single-rank stand-in for ``mpi4py`` (mpi4py itself is not installed);
  2. runs the reference and ``oracle/`` side by side on bcsstk03 / nos7 / the
     diagonal model problem and REQUIRES bitwise-equal results (this is what pins
     the oracle);
  3. writes small ``.npz`` fixtures: the CSR arrays, right-hand sides, SpMV known
     answers, full histories, sampled per-iteration states for teacher-forced
     single-step checks and the paper's convergence statistics.

Only data is written: inputs and the reference's outputs.  No reference source
text ends up in the repo.  The fixtures are what travels to the GPU box; the
reference never does.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import scipy
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/predict_and_recompute'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, 'numerical_experiments'))

import cg_variants as ref_cg            # the reference (NE/cg_variants/__init__.py)
import callbacks as ref_cb              # the reference (NE/callbacks/__init__.py)
from oracle import ne_oracle as orc
from oracle import mp_oracle as mporc

FOUR = ['error_A_norm', 'residual_2_norm', 'error_2_norm', 'updated_residual_2_norm']
REF_CALLBACKS = [getattr(ref_cb, q) for q in FOUR]


def load_matrix(name):
    # exactly the reference's loading line, NE/figure_gen.py:350
    return sp.csr_matrix(scipy.io.mmread(f'{REF}/matrices/{name}.mtx'))


def problem(A):
    # NE/figure_gen.py:31-34
    N = A.get_shape()[0]
    x_true = np.ones(N) / np.sqrt(N)
    b = A @ x_true
    x0 = np.zeros(N)
    return b, x0, x_true


STATE_VECS = ['x', 'r', 'p', 's', 'w', 'u', 'rt', 'st', 'wt', 'ut']
STATE_SCAL = ['nu', 'mu', 'dl', 'gm', 'eta', 'alpha', 'beta']
REF_NAMES = {'x': 'x_k', 'r': 'r_k', 'p': 'p_k', 's': 's_k', 'w': 'w_k', 'u': 'u_k',
             'rt': 'rt_k', 'st': 'st_k', 'wt': 'wt_k', 'ut': 'ut_k',
             'nu': 'nu_k', 'mu': 'mu_k', 'dl': 'del_k', 'gm': 'gam_k', 'eta': 'eta_k', 'alpha': 'a_k', 'beta': 'b_k'}


def ref_state_grabber(store, ks):
    """A reference-style callback (called as callback(**locals())) that copies the
    iterate at the requested k."""
    ks = set(ks)

    def grab(**kw):
        k = kw['k']
        if k in ks:
            d = {}
            for mine, theirs in REF_NAMES.items():
                if theirs in kw:
                    v = kw[theirs]
                    d[mine] = np.array(v, dtype=np.float64, copy=True)
            store[k] = d
    return grab


def run_pair(matrix, A, method, max_iter, prec_name, state_ks):
    """Run reference and oracle; insist on bitwise equality; return fixture dict."""
    b, x0, x_true = problem(A)
    ref_fn = getattr(ref_cg, method)
    orc_fn = getattr(orc, method)
    kwargs = {}
    okw = {}
    if prec_name == 'jacobi':
        kwargs['preconditioner'] = lambda x: (1 / A.diagonal()) * x      # NE/figure_gen.py:43
        okw['preconditioner'] = orc.jacobi(A)
    states = {}
    cbs = REF_CALLBACKS + [ref_state_grabber(states, state_ks)]
    with np.errstate(all='ignore'):
        ref_out = ref_fn(A, b, x0, max_iter, callbacks=cbs, x_true=x_true, **kwargs)
    ostates = {}

    def tap(st):
        if st.k in states:
            ostates[st.k] = st.clone()
    my_out = orc_fn(A, b, x0, max_iter, callbacks=FOUR, x_true=x_true, tap=tap, **okw)

    assert ref_out['name'] == my_out['name'], (ref_out['name'], my_out['name'])
    for q in FOUR:
        same = np.array_equal(ref_out[q], my_out[q], equal_nan=True)
        assert same, f'{matrix}/{method}/{prec_name}: oracle history {q} differs from the reference'
    for k, d in states.items():
        for f, v in d.items():
            mine = getattr(ostates[k], f)
            if mine is None:
                continue
            assert np.array_equal(np.asarray(mine, dtype=np.float64), v, equal_nan=True), (matrix, method, k, f)

    fx = {'max_iter': np.int64(max_iter), 'name': np.array(ref_out['name'])}
    for q in FOUR:
        fx['hist_' + q] = ref_out[q]
    its, acc = orc.convergence_summary(ref_out['error_A_norm'])
    fx['iters_to_1e-5'] = np.int64(its)
    fx['log10_min_rel_error_A'] = np.float64(acc)
    fx['state_ks'] = np.array(sorted(states), dtype=np.int64)
    for k, d in states.items():
        for f, v in d.items():
            fx[f'state{k}_{f}'] = v
    tag = f'{matrix}_{method}_{prec_name}'
    np.savez_compressed(os.path.join(HERE, f'run_{tag}.npz'), **fx)
    print(f'  {tag:44s} max_iter={max_iter:5d} its={its:5d} log10min={acc:7.2f} states={len(states)}  [oracle == reference, bitwise]')


def w_replace_goldens():
    """gv_cg / gv_pcg with a residual-replacement predicate (gv_cg.py:9,69-71): the default never fires, so the hook is
    pinned with one that does -- every 7th iteration and at its 3rd and 40th call (counted in the wk_replace_flags storage)."""
    def make_pred():
        # (decisions depend on k and on the call count kept in wk_replace_flags only: a run whose inner products are summed
        #  in another order takes the same decisions)
        def pred(**kw):
            fl = kw['wk_replace_flags']
            fl['calls'] = fl.get('calls', 0) + 1
            assert kw['r'].shape == kw['r_'].shape == kw['w'].shape
            return kw['k'] % 7 == 0 or fl['calls'] in (3, 40)
        return pred
    for matrix, method, prec_name, max_iter in (('bcsstk03', 'gv_cg', 'None', 700), ('bcsstk03', 'gv_pcg', 'jacobi', 200), ('nos7', 'gv_pcg', 'jacobi', 120)):
        A = load_matrix(matrix)
        b, x0, x_true = problem(A)
        kwargs, okw = {}, {}
        if prec_name == 'jacobi':
            kwargs['preconditioner'] = lambda x: (1 / A.diagonal()) * x
            okw['preconditioner'] = orc.jacobi(A)
        with np.errstate(all='ignore'):
            ref_out = getattr(ref_cg, method)(A, b, x0, max_iter, w_replace=make_pred(), callbacks=REF_CALLBACKS, x_true=x_true, **kwargs)
            my_out = getattr(orc, method)(A, b, x0, max_iter, w_replace=make_pred(), callbacks=FOUR, x_true=x_true, **okw)
        fx = {'max_iter': np.int64(max_iter)}
        for q in FOUR:
            assert np.array_equal(ref_out[q], my_out[q], equal_nan=True), f'{matrix}/{method}: oracle with w_replace differs from the reference in {q}'
            fx['hist_' + q] = ref_out[q]
        its, acc = orc.convergence_summary(ref_out['error_A_norm'])
        fx['iters_to_1e-5'] = np.int64(its)
        fx['log10_min_rel_error_A'] = np.float64(acc)
        np.savez_compressed(os.path.join(HERE, f'wreplace_{matrix}_{method}_{prec_name}.npz'), **fx)
        print(f'  w_replace {matrix}/{method}/{prec_name}: max_iter={max_iter} its={its} log10min={acc:.2f}  [oracle == reference, bitwise]')


def pairs(ks):
    """teacher forcing needs state k and k+1"""
    out = set()
    for k in ks:
        out.add(k)
        out.add(k + 1)
    return sorted(out)


def matrix_fixture(name, A):
    b, x0, x_true = problem(A)
    rng = np.random.default_rng(20261003)
    X = rng.standard_normal((3, A.shape[0]))
    Y = np.stack([A @ X[i] for i in range(3)])
    np.savez_compressed(os.path.join(HERE, f'matrix_{name}.npz'),
                        n=np.int64(A.shape[0]), indptr=A.indptr, indices=A.indices, data=A.data,
                        b=b, x_true=x_true, diag=A.diagonal(), spmv_x=X, spmv_y=Y)
    print(f'matrix {name}: n={A.shape[0]} nnz={A.nnz} rows {np.diff(A.indptr).min()}..{np.diff(A.indptr).max()}')


def mp_goldens():
    """MP variants under a single-rank stand-in for mpi4py."""
    class _FakeComm:
        def Get_size(self): return 1
        def Get_rank(self): return 0
        def Barrier(self): pass
        def Allreduce(self, send, recv, op=None):
            recv[0][...] = send[0]
    import time
    mpi = types.ModuleType('mpi4py')
    mpi.MPI = types.SimpleNamespace(DOUBLE='d', SUM='sum', Wtime=time.perf_counter)
    sys.modules['mpi4py'] = mpi

    def load(fname):
        spec = importlib.util.spec_from_file_location('ref_mp_' + fname, f'{REF}/scaling_experiments_mpi4py/cg_variants/{fname}.py')
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return getattr(mod, fname)

    fx = {}
    for n, iters in ((1024, 40), (1024, 400), (12288, 1500)):
        lam = mporc.model_problem_eigs(n)                    # MP/scaling_tests.py:31-36
        b = lam / np.sqrt(n)                                 # :57 (after the diagonal fill)
        for vname in ('pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'):
            if n == 12288 and vname not in ('pipe_pr_cg', 'hs_cg'):
                continue        # the two variants of SURVEY row a8 get the full-size run; the others n=1024
            if n == 12288:
                # the reference's dense n x n operator (1.2 GB) is affordable once
                A = np.zeros((n, n)); A[np.arange(n), np.arange(n)] = lam
            else:
                A = np.diag(lam)
            with np.errstate(all='ignore'):
                x_ref, t_ref = load(vname)(_FakeComm(), A, b.copy(), iters)
            comm = mporc.SingleRankComm()
            x_orc, t_orc = getattr(mporc, vname)(comm, mporc.DenseColumnBlock(comm, A), b.copy(), iters)
            bitwise = np.array_equal(x_ref, x_orc)
            err = np.linalg.norm(np.ones(n) / np.sqrt(n) - x_ref)   # MP/scaling_tests.py:81
            print(f'  MP {vname:11s} n={n:6d} iters={iters:5d} error={err:.4e} oracle bitwise={bitwise}')
            assert bitwise, 'MP oracle differs from the reference'
            assert set(t_ref.keys()) >= {'tot'} and t_orc.keys() == {'tot'}
            if n <= 1024:
                fx[f'{vname}_n{n}_it{iters}_x'] = x_ref
            fx[f'{vname}_n{n}_it{iters}_error'] = np.float64(err)
            del A
    np.savez_compressed(os.path.join(HERE, 'mp_model_problem.npz'), **fx)


# The reference's experiment list: (matrix, max_iter, preconditioner), NE/figure_gen.py:247-339,
# restricted to the matrices the reference ships under matrices/.
PAPER_RUNS = [
    ('model_48_8_3', 110, None), ('model_48_8_3', 200, 'jacobi'),
    ('bcsstk03', 250, 'jacobi'), ('bcsstk14', 800, 'jacobi'), ('bcsstk15', 830, 'jacobi'),
    ('bcsstk16', 320, 'jacobi'), ('bcsstk18', 2700, 'jacobi'), ('bcsstk27', 380, 'jacobi'),
    ('bcsstk03', 1250, None), ('bcsstk14', 25000, None), ('bcsstk15', 35000, None),
    ('bcsstk16', 900, None), ('bcsstk18', 1750000, None), ('bcsstk27', 2300, None),
    ('nos1', 900, 'jacobi'), ('nos2', 11000, 'jacobi'), ('nos3', 350, 'jacobi'), ('nos4', 120, 'jacobi'),
    ('nos5', 350, 'jacobi'), ('nos6', 130, 'jacobi'), ('nos7', 200, 'jacobi'),
    ('nos1', 4500, None), ('nos2', 45000, None), ('nos3', 400, None), ('nos4', 150, None),
    ('nos5', 600, None), ('nos6', 2400, None), ('nos7', 7000, None),
    ('bcsstm19', 1100, None), ('bcsstm20', 700, None), ('bcsstm21', 10, None), ('bcsstm22', 85, None),
    ('bcsstm23', 10000, None), ('bcsstm24', 45000, None), ('bcsstm25', 130000, None),
    ('494_bus', 2500, None), ('662_bus', 1200, None), ('685_bus', 950, None), ('1138_bus', 5000, None),
    ('494_bus', 500, 'jacobi'), ('662_bus', 350, 'jacobi'), ('685_bus', 350, 'jacobi'), ('1138_bus', 1300, 'jacobi'),
]
PAPER_COLUMNS = ['hs_pcg', 'cg_pcg', 'm_pcg', 'pr_pcg', 'gv_pcg', 'pipe_pr_m_pcg', 'pipe_pr_pcg']   # NE/figure_gen.py:360


def paper_table():
    """The published convergence table (NE/figures/convergence_table_data.tex: iterations to a
    relative A-norm error of 1e-5 and log10 of the best relative error, seven variants per
    row) as JSON, plus the CSR arrays of every matrix it needs -- so that the GPU box can
    redo the paper's table without the reference."""
    import json
    import re
    published = {}
    for line in open(f'{REF}/numerical_experiments/figures/convergence_table_data.tex'):
        line = line.strip()
        if not line:
            continue
        cells = [c.strip() for c in line.rstrip('\\ ').split('&')]
        name = re.match(r'\\texttt\{(.*)\}', cells[0]).group(1).replace('\\_', '_')
        vals = [re.sub(r'\\tableemph|[{}]', '', c) for c in cells[4:]]
        assert len(vals) == 14, line
        prec = 'jacobi' if cells[1].startswith('Jac') else 'None'
        published[(name, prec)] = dict(
            n=int(cells[2]), nnz=int(cells[3]),
            iters=[None if v == '-' else int(v) for v in vals[:7]],
            log10_min_rel_error_A=[float(v) for v in vals[7:]])
    rows = []
    done = set()
    for name, max_iter, prec in PAPER_RUNS:
        A = load_matrix(name)
        pub = published[(name, str(prec))]
        assert A.shape[0] == pub['n'] and A.nnz == pub['nnz'], (name, A.shape, A.nnz, pub)
        assert A.has_sorted_indices and A.indices.dtype == np.int32
        if name not in done:
            np.savez_compressed(os.path.join(HERE, f'tablemat_{name}.npz'), n=np.int64(A.shape[0]),
                                indptr=A.indptr, indices=A.indices, data=A.data)
            done.add(name)
        rows.append(dict(matrix=name, preconditioner=str(prec), max_iter=max_iter, columns=PAPER_COLUMNS, **pub))
    with open(os.path.join(HERE, 'paper_convergence_table.json'), 'w') as f:
        json.dump(rows, f, indent=1)
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.startswith('tablemat_'))
    print(f'paper table: {len(rows)} rows, {len(done)} matrices, {total/1e6:.2f} MB')


def main():
    print('numpy', np.__version__, 'scipy', scipy.__version__)
    if sys.argv[1:] == ['w_replace']:
        w_replace_goldens()
        return
    if sys.argv[1:] == ['table']:
        paper_table()
        return
    if sys.argv[1:] == ['mp']:
        mp_goldens()
        return
    mats = {m: load_matrix(m) for m in ('bcsstk03', 'nos7', 'nos4', '494_bus', 'bcsstk14', 'bcsstm22', 'model_48_8_3')}
    for m, A in mats.items():
        assert A.has_sorted_indices and A.indices.dtype == np.int32
        matrix_fixture(m, A)

    # (matrix, method, max_iter, preconditioner, sampled k for teacher forcing)
    # max_iter values are the reference's: NE/figure_gen.py:253,263,279,289
    dense_b = list(range(0, 21)) + [50, 100, 200, 364, 400, 800, 1200]
    sparse_n = [0, 1, 2, 3, 5, 8, 15, 30, 60, 150, 560, 1500, 2900, 5000, 6990]
    few_n = [0, 1, 5, 30, 150, 700]
    plan = [
        ('bcsstk03', 'hs_cg', 1250, None, pairs(dense_b)),
        ('bcsstk03', 'pipe_pr_cg', 1250, None, pairs(dense_b)),
        ('bcsstk03', 'hs_pcg', 1250, None, []),
        ('bcsstk03', 'pipe_pr_pcg', 1250, None, []),
        ('bcsstk03', 'pr_pcg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'm_pcg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'pipe_p_cg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'pipe_pr_m_cg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'pipe_p_m_cg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'cg_cg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'gv_cg', 1250, None, pairs([0, 1, 10, 100, 600])),
        ('bcsstk03', 'hs_pcg', 250, 'jacobi', pairs([0, 1, 10, 100, 200])),
        ('bcsstk03', 'pr_pcg', 250, 'jacobi', pairs([0, 1, 10, 100, 200])),
        ('bcsstk03', 'pipe_pr_pcg', 250, 'jacobi', pairs([0, 1, 10, 100, 200])),
        ('bcsstk03', 'pipe_p_pcg', 250, 'jacobi', pairs([0, 1, 10, 100])),
        ('bcsstk03', 'cg_pcg', 250, 'jacobi', pairs([0, 1, 10, 100, 200])),
        ('bcsstk03', 'gv_pcg', 250, 'jacobi', pairs([0, 1, 10, 100, 200])),
        ('nos7', 'cg_cg', 1000, None, pairs(few_n)),
        ('nos7', 'gv_cg', 1000, None, pairs(few_n)),
        ('nos7', 'cg_pcg', 200, 'jacobi', pairs([0, 1, 10, 66, 150])),
        ('nos7', 'gv_pcg', 200, 'jacobi', pairs([0, 1, 10, 66, 150])),
        # further matrices of the paper's list (max_iter: NE/figure_gen.py:247-339), histories + a few states
        ('nos4', 'hs_pcg', 150, None, pairs([0, 5, 60])),
        ('nos4', 'pipe_pr_pcg', 150, None, pairs([0, 5, 60])),
        ('nos4', 'pipe_pr_pcg', 120, 'jacobi', pairs([0, 5, 60])),
        ('494_bus', 'hs_pcg', 2500, None, pairs([0, 7, 900])),
        ('494_bus', 'pipe_pr_pcg', 2500, None, pairs([0, 7, 900])),
        ('494_bus', 'pr_pcg', 500, 'jacobi', pairs([0, 7, 300])),
        ('494_bus', 'pipe_pr_pcg', 500, 'jacobi', pairs([0, 7, 300])),
        ('bcsstk14', 'hs_pcg', 800, 'jacobi', pairs([0, 9, 400])),
        ('bcsstk14', 'pipe_pr_pcg', 800, 'jacobi', pairs([0, 9, 400])),
        ('bcsstk14', 'gv_pcg', 800, 'jacobi', pairs([0, 9])),
        ('bcsstm22', 'hs_pcg', 85, None, pairs([0, 20])),
        ('bcsstm22', 'pipe_pr_pcg', 85, None, pairs([0, 20])),
        ('model_48_8_3', 'hs_pcg', 110, None, pairs([0, 30])),
        ('model_48_8_3', 'pipe_pr_pcg', 110, None, pairs([0, 30])),
        ('model_48_8_3', 'pipe_pr_pcg', 200, 'jacobi', pairs([0, 30])),
        ('nos7', 'hs_cg', 7000, None, pairs(sparse_n)),
        ('nos7', 'pipe_pr_cg', 7000, None, pairs(sparse_n)),
        ('nos7', 'pr_pcg', 1000, None, pairs(few_n)),
        ('nos7', 'pipe_p_cg', 1000, None, pairs(few_n)),
        ('nos7', 'hs_pcg', 200, 'jacobi', pairs([0, 1, 10, 66, 150])),
        ('nos7', 'pipe_pr_pcg', 200, 'jacobi', pairs([0, 1, 10, 66, 150])),
    ]
    for matrix, method, max_iter, prec, ks in plan:
        run_pair(matrix, mats[matrix], method, max_iter, prec, [k for k in ks if k < max_iter])
    mp_goldens()
    paper_table()
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.npz'))
    print(f'fixtures: {total/1e6:.2f} MB')


if __name__ == '__main__':
    main()
