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


def irregular_standin(n, mean_len=76, max_len=2000, reach=50_000, seed=0, shift=1e-4):
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
    # diagonal = sum of the row's |off-diagonals| times (1 + a small varying shift): weakly diagonally dominant, condition
    # number of order 1e4 -- a CG run of a few thousand iterations stays finite (with "+ 1.0" the residual reached an exact
    # zero after a handful of iterations and the benchmark timed a non-finite state)
    ph = np.arange(n) * 0.3819660112501051
    d = np.asarray(abs(S).sum(axis=1)).ravel() * (1.0 + shift * (1.0 + (ph - np.floor(ph)))) + (np.diff(S.indptr) == 0)
    A = (S + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


_FEM_OFFS = [(a, b, c) for c in (-1, 0, 1) for b in (-1, 0, 1) for a in (-1, 0, 1)]      # ascending neighbour id


def _fem_blocks(m, node_lo, node_hi, first_row, dofs, keep_bits, shift):
    """CSR rows of the nodes [node_lo, node_hi) of an m^3 grid whose node i owns the unknowns
    first_row[i] .. first_row[i+1] and is coupled to those of its (kept) 27 neighbours.  Off-diagonal
    values: a symmetric, practically never repeating function of the unordered pair {row, col} in
    (-2/81, -1/81]; diagonal = sum of the row's |off-diagonals| * (1 + shift * (1 + phase(row))): a weighted
    graph Laplacian plus a small varying positive diagonal -- symmetric positive definite with a condition
    number of order 1/shift, and b = A 1 is no eigenvector (a CG run of a few thousand iterations stays finite)."""
    nodes = np.arange(node_lo, node_hi, dtype=np.int64)
    ix, iy, iz = nodes % m, (nodes // m) % m, nodes // (m * m)
    nn = nodes.size
    NB = np.full((nn, 27), -1, dtype=np.int64)
    for o, (a, b, c) in enumerate(_FEM_OFFS):
        ok = (ix + a >= 0) & (ix + a < m) & (iy + b >= 0) & (iy + b < m) & (iz + c >= 0) & (iz + c < m)
        nb = nodes + a + m * b + m * m * c
        if keep_bits is not None and o != 13:
            # the pair {i, j} is kept by the bit of its smaller node (symmetric pattern)
            ok &= keep_bits[nb.clip(0, keep_bits.shape[0] - 1), 26 - o] if o < 13 else keep_bits[nodes, o]
        NB[:, o] = np.where(ok, nb, -1)
    valid = NB >= 0
    nbc = np.where(valid, NB, 0)
    d_nb = np.where(valid, dofs[nbc], 0).astype(np.int64)              # unknowns of each kept neighbour
    L = d_nb.sum(axis=1)                                               # row length of every unknown of the node
    d_own = dofs[nodes].astype(np.int64)
    # one row's columns per node: the kept neighbours' unknowns, ascending
    cnt = d_nb.ravel()
    tot = int(cnt.sum())
    excl = np.cumsum(cnt) - cnt
    pat = np.repeat(first_row[nbc].ravel() - excl, cnt) + np.arange(tot, dtype=np.int64)
    pat_ptr = np.concatenate([[0], np.cumsum(L)])
    # the node's rows: d_own copies of its pattern
    row_len = np.repeat(L, d_own)
    nrows = int(d_own.sum())
    indptr = np.concatenate([[0], np.cumsum(row_len)])
    nnz = int(indptr[-1])
    rexcl = indptr[:-1]
    src = np.repeat(np.repeat(pat_ptr[:-1], d_own) - rexcl, row_len) + np.arange(nnz, dtype=np.int64)
    J = pat[src]
    del src, pat
    r0 = int(first_row[node_lo])
    I = np.repeat(np.arange(r0, r0 + nrows, dtype=np.int64), row_len)
    t = (I + J) * 0.6180339887498949 + np.abs(I - J) * 0.7548776662466927
    V = -(1.0 + (t - np.floor(t))) / 81.0
    del t
    diag = J == I
    V[diag] = 0.0
    rowsum = np.add.reduceat(V, indptr[:-1][row_len > 0]) if nnz else np.zeros(0)
    full = np.zeros(nrows)
    full[row_len > 0] = -rowsum
    rid = np.arange(r0, r0 + nrows, dtype=np.float64)
    ph = rid * 0.3819660112501051
    V[diag] = full * (1.0 + shift * (1.0 + (ph - np.floor(ph))))
    return indptr, J.astype(np.int32), V


def _fem_assemble(m, dofs, keep_bits, shift, rows, chunk_nodes=48 * 1024):
    first_row = np.concatenate([[0], np.cumsum(dofs, dtype=np.int64)])
    n = int(first_row[-1])
    lo, hi = _rows(n, rows)
    na = int(np.searchsorted(first_row, lo, side='right') - 1)
    nb = int(np.searchsorted(first_row, hi, side='left'))
    spans = [(a, min(a + chunk_nodes, nb)) for a in range(na, nb, chunk_nodes)]
    from concurrent.futures import ThreadPoolExecutor
    import os
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2)))) as ex:
        parts = list(ex.map(lambda s: _fem_blocks(m, s[0], s[1], first_row, dofs, keep_bits, shift), spans))
    ptr = [np.zeros(1, dtype=np.int64)]
    base = 0
    for ip, _, _ in parts:
        ptr.append(ip[1:] + base)
        base += int(ip[-1])
    indptr = np.concatenate(ptr)
    indices = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, dtype=np.int32)
    data = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0)
    del parts
    if indptr[-1] < 2**31:
        indptr = indptr.astype(np.int32)
    A = sp.csr_matrix((data, indices, indptr), shape=(int(first_row[nb] - first_row[na]), n))
    a, b = lo - int(first_row[na]), hi - int(first_row[na])
    return A if (a == 0 and b == A.shape[0]) else A[a:b]


def fem_like_3d(m, dof=3, rows=None, shift=1e-4):
    """Structured stand-in for 3-D elasticity matrices such as Queen_4147 (n=4.1 M, ~76 nonzeros
    per row): m^3 nodes, `dof` unknowns per node, every node coupled to its 27 neighbours ->
    up to 27*dof = 81 nonzeros per row, columns clustered like an FEM assembly, values practically all
    distinct (nothing for a value dictionary to find).  Symmetric positive definite (weighted graph
    Laplacian + a small varying diagonal).  m=111, dof=3: n=4,102,893, nnz=326,382,219."""
    return _fem_assemble(m, np.full(m ** 3, dof, dtype=np.int8), None, shift, rows)



def block_band_3dof(nodes, reach, picks=26, seed=3):
    """Rows of aligned runs of three columns with a WIDE footprint: `nodes` nodes of three unknowns, each coupled (full 3 x 3
    blocks) to itself and to `picks` nodes drawn within +-reach.  Not symmetric: a test operator for products (the sliced-row
    planner's window codes need 49..64 granules per slice at reach 120, and more than any window holds beyond ~150)."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for i in range(nodes):
        c = np.union1d(np.clip(i + rng.integers(-reach, reach + 1, size=picks), 0, nodes - 1), [i])
        cc = (c[:, None] * 3 + np.arange(3)[None, :]).ravel()
        for d in range(3):
            rows.append(np.full(cc.size, 3 * i + d)); cols.append(cc)
    r = np.concatenate(rows); c = np.concatenate(cols)
    return sp.csr_matrix((rng.standard_normal(r.size), (r, c)), shape=(3 * nodes, 3 * nodes))

def fem_irregular_3d(m, seed=0, keep=0.8, rows=None, shift=1e-4, dof_choices=(1, 3, 6), dof_probs=(0.3, 0.5, 0.2)):
    """FEM-like locality WITH irregular row lengths (the "CSR-adaptive load-balance stress" of BASELINE config 5,
    labelled stand-in): m^3 nodes carrying 1, 3 or 6 unknowns (seeded draw), 27-point coupling with each node pair
    kept with probability `keep` (symmetric); a row's length is the number of unknowns of its kept neighbours
    (about 20..150, mean ~65).  Symmetric positive definite like fem_like_3d.  m=111: n ~ 4.1 M, ~270 M nonzeros."""
    rng = np.random.default_rng(seed)
    dofs = rng.choice(np.array(dof_choices, dtype=np.int8), size=m ** 3, p=dof_probs)
    keep_bits = rng.random((m ** 3, 27)) < keep
    return _fem_assemble(m, dofs, keep_bits, shift, rows)


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
    # one rank's share of a 2- / 4-GPU run (bench.py: roofline.multi_rank_schedule)
    's3_half': dict(desc='one half of S3: ex2b banded n=5e6, 15 diagonals', n=5_000_000,
                    make=lambda rows=None: banded_ex2b(5_000_000, 7, rows=rows)),
    's3_quarter': dict(desc='one quarter of S3: ex2b banded n=2.5e6, 15 diagonals', n=2_500_000,
                       make=lambda rows=None: banded_ex2b(2_500_000, 7, rows=rows)),
    's2_half': dict(desc='one half of S2: 7-pt Laplacian 216 x 216 x 108 (n=5,038,848)', n=216 * 216 * 108,
                    make=lambda rows=None: laplace_3d(216, 216, 108, rows)),
    's2_quarter': dict(desc='one quarter of S2: 7-pt Laplacian 216 x 216 x 54 (n=2,519,424)', n=216 * 216 * 54,
                       make=lambda rows=None: laplace_3d(216, 216, 54, rows)),
    's4': dict(desc='S4 stand-in for Queen_4147: irregular symmetric SPD, n=1e6, log-normal row lengths (mean ~76, max 2000), reach 50000, seed 0',
               n=1_000_000, make=lambda rows=None: _row_slice(irregular_standin(1_000_000), rows)),
    's4b': dict(desc='FEM-like stand-in for Queen_4147 at its size: 3 dof x 27-point coupling on 111^3 nodes (n=4,102,893, nnz=326,382,219, <= 81 nnz/row)',
                n=3 * 111 ** 3, make=lambda rows=None: fem_like_3d(111, 3, rows)),
    's4b_80': dict(desc='FEM-like stand-in, 80^3 nodes (n=1,536,000, ~81 nnz/row): rounds 2-3 size',
                   n=3 * 80 ** 3, make=lambda rows=None: fem_like_3d(80, 3, rows)),
    's4c': dict(desc='irregular FEM-like stand-in for Queen_4147 at its size: 111^3 nodes of 1 / 3 / 6 unknowns (seed 0), 27-point coupling '
                     'thinned to 80 % (n=4,103,647, nnz=267,748,473, rows of 7..121 nonzeros)',
                n=4_103_647, make=lambda rows=None: fem_irregular_3d(111, rows=rows)),
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

