"""Deterministic synthetic operators and right-hand sides (S1..S4).

Every generator can produce just the row block ``[lo, hi)`` (global column ids), so a
rank never materialises more than its share.  No RNG except the labelled S4 stand-in.
"""
import numpy as np
import scipy.sparse as sp


def _rows(n, rows):
    lo, hi = (0, n) if rows is None else rows
    return int(lo), int(hi)


def _assemble(I, J, V, valid, n_local, n):
    counts = valid.sum(axis=1)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    if indptr[-1] < 2**31:
        indptr = indptr.astype(np.int32)
    return sp.csr_matrix((V[valid], J[valid].astype(np.int32), indptr), shape=(n_local, n))


def laplace_2d(nx, ny, rows=None):
    """S1 -- 5-point Laplacian on an nx x ny grid, natural ordering, diag 4 / off -1.
    nx=ny=1000: n=1,000,000, nnz=4,996,000."""
    n = nx * ny
    lo, hi = _rows(n, rows)
    I = np.arange(lo, hi, dtype=np.int64)[:, None]
    ix = I % nx
    off = np.array([-nx, -1, 0, 1, nx], dtype=np.int64)[None, :]
    J = I + off
    valid = (J >= 0) & (J < n)
    valid[:, 1] &= (ix[:, 0] > 0)
    valid[:, 3] &= (ix[:, 0] < nx - 1)
    V = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), J.shape)
    return _assemble(I, J, V, valid, hi - lo, n)


def laplace_3d(nx, ny, nz, rows=None):
    """S2 -- 7-point Laplacian, natural ordering, diag 6 / off -1.
    216^3: n=10,077,696, nnz=70,263,936."""
    n = nx * ny * nz
    lo, hi = _rows(n, rows)
    I = np.arange(lo, hi, dtype=np.int64)[:, None]
    ix = (I % nx)[:, 0]
    iy = ((I // nx) % ny)[:, 0]
    off = np.array([-nx * ny, -nx, -1, 0, 1, nx, nx * ny], dtype=np.int64)[None, :]
    J = I + off
    valid = (J >= 0) & (J < n)
    valid[:, 1] &= (iy > 0)
    valid[:, 2] &= (ix > 0)
    valid[:, 4] &= (ix < nx - 1)
    valid[:, 5] &= (iy < ny - 1)
    V = np.broadcast_to(np.array([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]), J.shape)
    return _assemble(I, J, V, valid, hi - lo, n)


def banded_ex2b(n, k=7, kappa=1e6, rho=0.95, off_value=1e-4, rows=None):
    """S3 -- the banded SPD model matrix of the reference's PETSc driver
    (scaling_experiments_petsc/ex2b.c:86-96): constant off-diagonals within half
    bandwidth k, diag_i = 1 + (i/(n-1)) (kappa-1) rho^(n-1-i).  Values kappa, rho,
    off_value as in strong_scaling_tests.py:49-56.
    n=10,000,000, k=7: nnz=149,999,944."""
    lo, hi = _rows(n, rows)
    I = np.arange(lo, hi, dtype=np.int64)[:, None]
    J = I + np.arange(-k, k + 1, dtype=np.int64)[None, :]
    valid = (J >= 0) & (J < n)
    i = I[:, 0].astype(np.float64)
    diag = 1.0 + (i / (n - 1.0)) * (kappa - 1.0) * np.power(rho, (n - 1.0) - i)
    V = np.full(J.shape, off_value)
    V[:, k] = diag
    return _assemble(I, J, V, valid, hi - lo, n)


def irregular_standin(n, mean_len=76, max_len=2000, reach=50_000, seed=0):
    """S4 stand-in for SuiteSparse Queen_4147 (which cannot be fetched here): symmetric
    pattern, log-normal row lengths, random column offsets within +-reach, made SPD by
    diagonal dominance.  Whole matrix only (it is symmetrised)."""
    rng = np.random.default_rng(seed)
    half = np.minimum(np.maximum(rng.lognormal(np.log(mean_len / 2.0) - 0.125, 0.5, n).astype(np.int64), 1),
                      max_len // 2)
    rows = np.repeat(np.arange(n, dtype=np.int64), half)
    offs = rng.integers(1, reach + 1, size=rows.size)
    cols = np.clip(rows + np.where(rng.random(rows.size) < 0.5, -offs, offs), 0, n - 1)
    keep = cols != rows
    rows, cols = rows[keep], cols[keep]
    vals = -rng.random(rows.size)
    U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    S = U + U.T
    S.sum_duplicates()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    A = (S + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


def fem_like_3d(m, dof=3, rows=None):
    """Structured stand-in for 3-D elasticity matrices such as Queen_4147 (n=4.1 M, ~76 nonzeros
    per row): m^3 nodes, `dof` unknowns per node, every node coupled to its 27 neighbours ->
    up to 27*dof = 81 nonzeros per row, columns clustered like an FEM assembly.  Symmetric,
    strictly diagonally dominant (SPD).  m=111, dof=3: n=4,102,893."""
    n = dof * m ** 3
    lo, hi = _rows(n, rows)
    I = np.arange(lo, hi, dtype=np.int64)
    node, d = I // dof, I % dof
    ix, iy, iz = node % m, (node // m) % m, node // (m * m)
    offs = [(a, b, c) for c in (-1, 0, 1) for b in (-1, 0, 1) for a in (-1, 0, 1)]
    cols, valid = [], []
    for (a, b, c) in offs:
        ok = (ix + a >= 0) & (ix + a < m) & (iy + b >= 0) & (iy + b < m) & (iz + c >= 0) & (iz + c < m)
        nb = (ix + a) + m * (iy + b) + m * m * (iz + c)
        for e in range(dof):
            cols.append(nb * dof + e)
            valid.append(ok)
    J = np.stack(cols, axis=1)
    valid = np.stack(valid, axis=1)
    # symmetric values: a function of the unordered pair {row, col}
    # (practically all distinct, as in an assembled FEM matrix: nothing for a value dictionary to find)
    V = -(1.0 + (((I[:, None] + J) * (np.abs(I[:, None] - J) + 1)) % 1000003) / 1000003.0) / (27.0 * dof)
    V[J == I[:, None]] = 2.0
    return _assemble(I[:, None], J, V, valid, hi - lo, n)


def _row_slice(A, rows):
    return A if rows is None else A[rows[0]:rows[1]]


def reference_rhs(A_rows, n):
    """The reference's problem setup (numerical_experiments/figure_gen.py:31-34):
    x_true = 1/sqrt(N), b = A x_true, x0 = 0 -- for the row block ``A_rows``."""
    x_true_full = np.ones(n) / np.sqrt(n)
    b = A_rows @ x_true_full
    m = A_rows.shape[0]
    return b, np.zeros(m), np.full(m, 1.0 / np.sqrt(n))


def model_problem_diag(n, kappa=1e6, rho=0.9):
    """The mpi4py experiment's diagonal model problem as a CSR matrix
    (scaling_experiments_mpi4py/scaling_tests.py:31-36): eigenvalues
    l_i = l_1 + (l_n - l_1) (i/(n-1)) rho^(n-1-i), kappa = l_n/l_1."""
    lambda1 = 1 / kappa
    lambdan = 1
    lam = lambda1 + (lambdan - lambda1) * np.arange(n) / (n - 1) * rho**np.arange(n - 1, -1, -1, dtype='float')
    return sp.diags(lam).tocsr(), lam


WORKLOADS = {
    's1': dict(desc='S1 5-pt Laplacian 1000x1000 (n=1e6, nnz=4,996,000)', n=1000 * 1000,
               make=lambda rows=None: laplace_2d(1000, 1000, rows)),
    's2': dict(desc='S2 7-pt Laplacian 216^3 (n=10,077,696, nnz=70,263,936)', n=216 ** 3,
               make=lambda rows=None: laplace_3d(216, 216, 216, rows)),
    's3': dict(desc='S3 ex2b banded n=1e7, 15 diagonals (nnz=149,999,944), kappa=1e6 rho=0.95 off=1e-4',
               n=10_000_000, make=lambda rows=None: banded_ex2b(10_000_000, 7, rows=rows)),
    's3_8th': dict(desc='one eighth of S3: ex2b banded n=1.25e6, 15 diagonals', n=1_250_000,
                   make=lambda rows=None: banded_ex2b(1_250_000, 7, rows=rows)),
    's2_8th': dict(desc='one eighth of S2: 7-pt Laplacian 216 x 216 x 27 (n=1,259,712)', n=216 * 216 * 27,
                   make=lambda rows=None: laplace_3d(216, 216, 27, rows)),
    's4': dict(desc='S4 stand-in for Queen_4147: irregular symmetric SPD, n=1e6, log-normal row lengths (mean ~76, max 2000), reach 50000, seed 0',
               n=1_000_000, make=lambda rows=None: _row_slice(irregular_standin(1_000_000), rows)),
    's4b': dict(desc='FEM-like stand-in for Queen_4147: 3 dof x 27-point coupling on 80^3 nodes (n=1,536,000, ~81 nnz/row)',
                n=3 * 80 ** 3, make=lambda rows=None: fem_like_3d(80, 3, rows)),
    # reduced sizes for tests / smoke
    's1_small': dict(desc='5-pt Laplacian 64x48', n=64 * 48, make=lambda rows=None: laplace_2d(64, 48, rows)),
    's3_small': dict(desc='ex2b banded n=20000 k=7', n=20000,
                     make=lambda rows=None: banded_ex2b(20000, 7, rows=rows)),
}


def _register_queen():
    """S4 is SuiteSparse Queen_4147 itself when QUEEN_4147_MTX points to the
    MatrixMarket file (it cannot be fetched in the build environment).  Read once per process."""
    import os
    path = os.environ.get('QUEEN_4147_MTX')
    if not path or not os.path.exists(path):
        return
    cache = {}

    def make(rows=None):
        if 'A' not in cache:
            import scipy.io
            cache['A'] = sp.csr_matrix(scipy.io.mmread(path)).astype(np.float64)
            cache['A'].sort_indices()
        return _row_slice(cache['A'], rows)

    import scipy.io
    n = int(scipy.io.mminfo(path)[0])
    WORKLOADS['queen'] = dict(desc=f'SuiteSparse Queen_4147 from {os.path.basename(path)} (n={n})', n=n, make=make)


_register_queen()

