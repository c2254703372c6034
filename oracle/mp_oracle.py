"""TEST INFRASTRUCTURE ONLY -- not part of the product, never imported by it.

CPU restatement of the reference's mpi4py scaling variants:

* ``pipe_pr_cg(comm, A, b, max_iter)``  MP/cg_variants/pipe_pr_cg.py:7-89
* ``hs_cg(comm, A, b, max_iter)``       MP/cg_variants/hs_cg.py:7-66
* ``cg_cg``, ``gv_cg``, ``pr_cg``        MP/cg_variants/{cg_cg,gv_cg,pr_cg}.py

MP/ = /root/reference/predict_and_recompute/scaling_experiments_mpi4py/.

Semantics kept: x0 = 0, r0 = p0 = b (local slices); the loop is phased
"inner products -> matvec -> ONE reduction -> vector updates" and runs
``max_iter`` times; the return value is ``(x_local, times)`` with
``times = {'tot': seconds}`` on rank 0 and ``None`` elsewhere; the timing fence
is a barrier on both sides of the loop.

Not kept (by design, north_star): the reference's operator is a dense
``n x (n/size)`` *column* block whose product needs an all-reduce of the whole
n-vector.  Here ``A`` is anything with ``matvec_local(v_local) -> y_local`` --
the tests pass a row-block CSR operator with an explicit halo exchange -- so the
only reduction per iteration is the packed scalar one.  ``comm`` needs
``Get_rank()``, ``Get_size()``, ``Barrier()`` and ``allreduce_sum(ndarray) ->
ndarray``.

Pinned by ``tests/golden/make_golden.py``: the reference files are executed there
under a single-rank stand-in for ``mpi4py`` and this restatement must give the
same iterates bit for bit (single rank: no reduction-order freedom).
"""
import time

import numpy as np


class SingleRankComm:
    """Trivial communicator (one rank)."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def Barrier(self):
        pass

    def allreduce_sum(self, a):
        return np.array(a, copy=True)


class DenseColumnBlock:
    """The reference's operator layout (MP/scaling_tests.py:51-54): the rank's
    n x m column block; the product is completed by summing over ranks."""

    def __init__(self, comm, A_block):
        self.comm, self.A = comm, A_block
        self.m = A_block.shape[1]
        self.lo = comm.Get_rank() * self.m

    def matvec_local(self, V):
        full = self.comm.allreduce_sum(np.dot(self.A, V))
        return full[self.lo:self.lo + self.m]


def pipe_pr_cg(comm, A, b, max_iter, dot=np.dot):
    rank = comm.Get_rank()
    times = {'tot': 0.} if rank == 0 else None                 # pipe_pr_cg.py:13-16
    x = np.zeros_like(b)                                       # :23
    # r and s live interleaved in one (m,2) array, as in the reference (:24-28):
    # its inner products therefore run over stride-2 views, which fixes the
    # BLAS summation order this restatement is pinned against.
    RS = np.zeros((len(b), 2))
    RS[:, 0] = b                                               # :25
    r, s = RS[:, 0], RS[:, 1]                                  # :27-28
    p = np.array(b, copy=True)                                 # :26
    s[:] = A.matvec_local(r)                                   # :48-51  (s0 = A r0)
    comm.Barrier()                                             # :54
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for _ in range(max_iter):                              # :58
            part = np.array([dot(p, s), dot(r, s), dot(s, s), dot(r, r)])   # :60-63
            WU = A.matvec_local(RS)                            # :65 (A [r s] in one product)
            mu, dl, gm, nu_ = comm.allreduce_sum(part)         # :67 (the one reduction)
            wp, u = WU[:, 0], WU[:, 1]
            alpha = nu_ / mu                                   # :71
            x += alpha * p                                     # :73
            r -= alpha * s                                     # :74
            w = wp - alpha * u                                 # :75
            nu = nu_ - 2 * alpha * dl + alpha**2 * gm          # :77
            beta = nu / nu_                                    # :78
            p *= beta                                          # :80
            p += r                                             # :81
            s *= beta                                          # :82
            s += w                                             # :83
    comm.Barrier()                                             # :85
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return x, times


def hs_cg(comm, A, b, max_iter, dot=np.dot):
    rank = comm.Get_rank()
    times = {'tot': 0.} if rank == 0 else None                 # hs_cg.py:13-16
    x = np.zeros_like(b)                                       # :24
    r = np.array(b, copy=True)                                 # :25
    p = np.zeros_like(b)                                       # :26
    nu = 1.0                                                   # :21
    comm.Barrier()                                             # :32
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for _ in range(max_iter):                              # :36
            nu_old = nu                                        # :38
            nu = comm.allreduce_sum(np.array([dot(r, r)]))[0]  # :40-42 (reduction 1)
            beta = nu / nu_old                                 # :44
            p *= beta                                          # :46
            p += r                                             # :47
            s = A.matvec_local(p)                              # :49-51
            mu = comm.allreduce_sum(np.array([dot(p, s)]))[0]  # :53-55 (reduction 2)
            alpha = nu / mu                                    # :57
            x += alpha * p                                     # :59
            r -= alpha * s                                     # :60
    comm.Barrier()                                             # :62
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return x, times


def _scalar_allreduce(comm, *vals):
    """The reference keeps its scalars in 1-element arrays (views of the reduction buffers);
    arithmetic on those is what this restatement is pinned against (e.g. ``alpha**2`` on a
    1-element array is an exact square, on a NumPy scalar it is libm pow)."""
    out = comm.allreduce_sum(np.array([float(v) for v in vals]))
    return [out[i:i + 1].copy() for i in range(len(vals))]


def cg_cg(comm, A, b, max_iter, dot=np.dot):
    """Chronopoulos-Gear CG, MP/cg_variants/cg_cg.py:7-71 (two reductions per iteration: the
    product w = A r -- a data reduction in the reference's column-block layout -- and {nu, eta})."""
    rank = comm.Get_rank()
    times = {'tot': 0.} if rank == 0 else None                 # cg_cg.py:13-16
    alpha = np.zeros(1)                                        # :19
    nu = np.ones(1)                                            # :22
    x = np.zeros_like(b)                                       # :33
    r = np.array(b, copy=True)                                 # :34
    p = np.zeros_like(b)                                       # :35
    s = np.zeros_like(b)                                       # :36
    comm.Barrier()                                             # :42
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for k in range(max_iter):                              # :46
            w = A.matvec_local(r)                              # :48-50
            nu_ = nu.copy()                                    # :52
            nu, eta = _scalar_allreduce(comm, dot(r, r), dot(r, w))   # :53-56
            beta = nu / nu_                                    # :58
            p *= beta                                          # :60
            p += r                                             # :61
            s *= beta                                          # :62
            s += w                                             # :63
            mu = eta - (beta / alpha) * nu if k > 0 else eta   # :65
            alpha = nu / mu                                    # :66
            x += alpha * p                                     # :68
            r -= alpha * s                                     # :69
    comm.Barrier()                                             # :71
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return x, times


def gv_cg(comm, A, b, max_iter, dot=np.dot):
    """Ghysels-Vanroose pipelined CG, MP/cg_variants/gv_cg.py:7-88 (ONE reduction per iteration,
    carrying t = A w together with nu and eta)."""
    rank = comm.Get_rank()
    times = {'tot': 0.} if rank == 0 else None                 # gv_cg.py:13-16
    alpha = np.zeros(1)                                        # :19
    nu = np.ones(1)                                            # :29 (the reduction buffer starts as ones)
    x = np.zeros_like(b)                                       # :23
    r = np.array(b, copy=True)                                 # :24
    p = np.zeros_like(b)                                       # :25
    s = np.zeros_like(b)                                       # :26
    u = np.zeros_like(b)                                       # :28
    w = np.array(A.matvec_local(r), copy=True)                 # :43-45  (w = A b, before the clock)
    comm.Barrier()                                             # :48
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for k in range(max_iter):                              # :52
            nu_ = nu.copy()                                    # :54
            part = (dot(r, r), dot(r, w))                      # :56-57
            t = A.matvec_local(w)                              # :58-60 (same reduction as the scalars)
            nu, eta = _scalar_allreduce(comm, *part)
            beta = nu / nu_                                    # :62
            p *= beta                                          # :64
            p += r                                             # :65
            s *= beta                                          # :66
            s += w                                             # :67
            u *= beta                                          # :68
            u += t                                             # :69
            mu = eta - (beta / alpha) * nu if k > 0 else eta   # :71
            alpha = nu / mu                                    # :72
            x += alpha * p                                     # :74
            r -= alpha * s                                     # :75
            w -= alpha * u                                     # :76
    comm.Barrier()                                             # :78
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return x, times


def pr_cg(comm, A, b, max_iter, dot=np.dot):
    """Predict-and-recompute CG (not pipelined), MP/cg_variants/pr_cg.py:7-77."""
    rank = comm.Get_rank()
    times = {'tot': 0.} if rank == 0 else None                 # pr_cg.py:13-16
    x = np.zeros_like(b)                                       # :23
    r = np.array(b, copy=True)                                 # :24
    p = np.array(b, copy=True)                                 # :25
    comm.Barrier()                                             # :46
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for _ in range(max_iter):                              # :50
            s = A.matvec_local(p)                              # :52-53
            mu, delta, gamma, nup = _scalar_allreduce(comm, dot(p, s), dot(r, s), dot(s, s), dot(r, r))   # :55-60
            nu_ = nup.copy()                                   # :62
            alpha = nu_ / mu                                   # :64
            x += alpha * p                                     # :66
            r -= alpha * s                                     # :67
            nu = nu_ - 2 * alpha * delta + alpha**2 * gamma    # :69
            beta = nu / nu_                                    # :70
            p *= beta                                          # :72
            p += r                                             # :73
    comm.Barrier()                                             # :75
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return x, times


def model_problem_eigs(n, kappa=1e6, rho=0.9):
    """Eigenvalues of the reference's diagonal model problem (MP/scaling_tests.py:31-36)."""
    lambda1 = 1 / kappa
    lambdan = 1
    return lambda1 + (lambdan - lambda1) * np.arange(n) / (n - 1) * rho**np.arange(n - 1, -1, -1, dtype='float')
