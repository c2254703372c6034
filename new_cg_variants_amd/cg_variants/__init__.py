"""Serial CG variants with the reference's call shape, computed on one MI355X.

    f(A, b, x0, max_iter, [preconditioner=...], callbacks=[], **kwargs) -> dict

mirrors numerical_experiments/cg_variants/__init__.py:64-74 (same names, argument
meaning and return value): ``A`` a SciPy CSR matrix, ``b``/``x0`` 1-D float64, the
result a dict with ``'name'``, ``'max_iter'`` and one length-``max_iter`` array per
recorder in ``callbacks`` (index 0 = initial state; ``max_iter - 1`` iterations run).
Inputs are not modified.  Breakdown (division by zero) yields inf/nan in the histories,
never an exception -- as in the reference.

Differences, all at the edges:
* the four standard recorders (see ``..callbacks``) are computed on the device; any
  other callable in ``callbacks`` is still honoured, called after every iteration with
  the reference's local names (``x_k``, ``r_k``, ``nu_k`` ...), at the price of a
  device->host copy of the state per iteration;
* ``preconditioner``: a callable that acts as a *diagonal* scaling (the reference's Jacobi
  lambda ``(1/A.diagonal())*x``, figure_gen.py:43, or the identity default) is probed once
  to recover the diagonal and then applied on the device; any other callable is the
  caller's code, as it is in the reference, and is called on the host wherever the reference
  calls ``preconditioner(...)`` (prcg.h: prcg_set_preconditioner) while products, updates
  and inner products stay on the device;
* with an error recorder and no ``x_true`` the reference solves for it with a sparse direct
  solver on the fly (callbacks/error_A_norm.py:36-39); that is done here too up to n = 200,000
  -- beyond that it would never return, so the call asks for ``x_true`` instead.
"""
from collections import OrderedDict

import numpy as np

from .. import _lib as L
from ..callbacks import RECORDER_NAMES
from ..device import DeviceCSR

_OPERATORS = OrderedDict()   # small cache: figure_gen runs nine variants on one matrix
_MAX_CACHED = 2
_MAX_DIRECT_SOLVE = 200_000   # largest system solved on the host for a missing x_true (the reference's on-the-fly spsolve)


def _fingerprint(A):
    """Content hash of the three CSR arrays, so that a matrix modified in place -- even by an edit
    that preserves sums -- is uploaded again instead of meeting a stale device operator."""
    try:
        import xxhash
        h = xxhash.xxh3_64()
        for a in (A.indptr, A.indices, A.data):
            h.update(np.ascontiguousarray(a).data)
        return h.intdigest()
    except ImportError:
        import zlib
        c = 0
        for a in (A.indptr, A.indices, A.data):
            c = zlib.crc32(np.ascontiguousarray(a).data, c)
        return c


def _operator(A, device):
    key = (id(A), A.shape, A.nnz, A.data.__array_interface__['data'][0], device, _fingerprint(A))
    op = _OPERATORS.get(key)
    if op is None:
        op = DeviceCSR(A, device=device)
        _OPERATORS[key] = op
        while len(_OPERATORS) > _MAX_CACHED:
            _, old = _OPERATORS.popitem(last=False)
            old.close()
    else:
        _OPERATORS.move_to_end(key)
    return op


def clear_operator_cache():
    while _OPERATORS:
        _, op = _OPERATORS.popitem()
        op.close()


def _diagonal_of(preconditioner, n):
    """(d, None) with M^-1 v = d * v for a diagonal scaling (d None: the identity), or (None, preconditioner)
    for anything else.  The probe is exact for a diagonal scaling: M^-1 applied to ones IS d, bit for bit.
    A diagonal runs on the device; any other callable is called on the host wherever the reference calls
    `preconditioner(...)` -- it is the caller's code there too (prcg.h: prcg_set_preconditioner)."""
    if preconditioner is None:
        return None, None
    d = getattr(preconditioner, 'inv_diag', None)
    if d is not None:
        return L.f64(d), None
    ones = np.ones(n)
    d = np.asarray(preconditioner(ones), dtype=np.float64)
    if d.shape != (n,):
        raise ValueError('preconditioner must map (n,) to (n,)')
    probe = np.random.default_rng(12345).standard_normal(n)
    if not np.array_equal(np.asarray(preconditioner(probe)), d * probe):
        return None, preconditioner
    if np.all(d == 1.0):
        return None, None
    return np.ascontiguousarray(d), None


class Jacobi:
    """Jacobi preconditioner object: callable like the reference's lambda and carrying
    the reciprocal diagonal so no probing is needed."""

    def __init__(self, A):
        self.inv_diag = 1 / A.diagonal()          # figure_gen.py:43

    def __call__(self, v):
        return self.inv_diag * v


_STATE_NAMES = {   # device vector -> the reference's local name
    'x': 'x_k', 'r': 'r_k', 'p': 'p_k', 's': 's_k', 'w': 'w_k', 'u': 'u_k',
    'rt': 'rt_k', 'st': 'st_k', 'wt': 'wt_k', 'ut': 'ut_k',
}


def _state_vectors(variant, prec):
    if variant == L.HS:
        return ['x', 'r', 'p', 's'] + (['rt'] if prec else [])
    if variant in (L.PR, L.M):
        return ['x', 'r', 'p', 's'] + (['rt', 'st'] if prec else [])
    if variant == L.CG_CG:
        return ['x', 'r', 'p', 's', 'w'] + (['rt'] if prec else [])
    if variant == L.GV:
        return ['x', 'r', 'p', 's', 'w', 'u'] + (['rt', 'wt', 'st'] if prec else [])
    return ['x', 'r', 'p', 's', 'w', 'u'] + (['rt', 'st', 'wt', 'ut'] if prec else [])


def _run(variant, name, A, b, x0, max_iter, preconditioner, callbacks, kwargs, w_replace=None):
    device = int(kwargs.get('device', 0))
    if A.format != 'csr':
        A = A.tocsr()
    n = A.shape[0]
    inv_diag, prec_fn = _diagonal_of(preconditioner, n)

    mask = 0
    foreign = []          # callables we must call ourselves, every iteration
    light = []            # host callbacks that need no vectors
    for cb in callbacks:
        cname = getattr(cb, 'prcg_recorder', None) or getattr(cb, '__name__', '')
        if cname in RECORDER_NAMES:
            mask |= L.HIST_BITS[cname]
        elif getattr(cb, 'prcg_host_light', False) or cname == 'pk':
            light.append(cb)
        else:
            foreign.append(cb)
    x_true = kwargs.get('x_true')
    if (mask & (L.HIST_ERROR_A_NORM | L.HIST_ERROR_2_NORM)) and x_true is None:
        # the reference solves for it on the fly (callbacks/error_A_norm.py:36-39)
        if n > _MAX_DIRECT_SOLVE:
            raise ValueError(f'error_A_norm / error_2_norm need x_true: solving the n = {n} system with a sparse direct solver on the '
                             f'host to obtain it (as the reference would) is not feasible beyond n = {_MAX_DIRECT_SOLVE}; pass x_true=')
        import scipy.sparse.linalg as spla
        x_true = spla.spsolve(A.tocsc().astype(np.double), np.asarray(b, dtype=np.double))
        kwargs['x_true'] = x_true

    op = _operator(A, device)
    output = {'name': name, 'max_iter': max_iter}
    if w_replace is not None:
        # gv_cg.py:9,69-71: the caller's predicate, with the reference's keywords; r_ is the residual of the previous iteration
        flags = {}
        prev = {'r': None}

        def hook(k):
            x, w, r = op.get_vector('x'), op.get_vector('w'), op.get_vector('r')
            if prev['r'] is None:
                prev['r'] = np.asarray(b, dtype=np.float64) - A @ np.asarray(x0, dtype=np.float64)
            fire = w_replace(k=k, A=A, b=b, x=x, w=w, r=r, r_=prev['r'], u=op.get_vector('u'), s=op.get_vector('s'),
                             p=op.get_vector('p'), wk_replace_flags=flags)
            prev['r'] = r
            return fire
        op.set_replace_hook(hook)
    else:
        op.set_replace_hook(None)
    op.begin(variant, b, x0, max_iter, x_true=x_true, inv_diag=inv_diag, hist_mask=mask, preconditioner=prec_fn)

    def call_host(k):
        env = {'output': output, 'k': k, 'max_iter': max_iter, 'A': A, 'b': b, 'x0': x0, 'n': n,
               'kwargs': kwargs, 'callbacks': callbacks}
        for cb in light:
            cb(**env)
        if not foreign:
            return
        op.sync()
        for v in _state_vectors(variant, inv_diag is not None or prec_fn is not None):
            env[_STATE_NAMES[v]] = op.get_vector(v)
        sc = op.get_scalars(k)
        env.update(nu_k=sc[L.S_NU], mu_k=sc[L.S_MU], del_k=sc[L.S_DELTA], gam_k=sc[L.S_GAMMA])
        if variant in (L.CG_CG, L.GV):
            env['eta_k'] = sc[L.S_DELTA]          # slot 1 holds eta = w.r~ for these two
        with np.errstate(all='ignore'):
            env['a_k'] = sc[L.S_NU] / sc[L.S_MU]
        env['b_k'] = op.get_coefficients(k)[1] if k > 0 else 0
        for cb in foreign:
            cb(**env)

    if foreign or light:
        call_host(0)
        for k in range(1, max_iter):
            op.iterate(1)
            call_host(k)
    else:
        op.iterate(max_iter - 1)
    op.sync()
    output.update(op.history())
    return output


def _make(variant, name, preconditioned):
    def take_w_replace(kwargs):
        # gv_cg.py:9 / :93 take a residual-replacement predicate (default: never); the other variants swallow the keyword
        fn = kwargs.pop('w_replace', None)
        return fn if variant == L.GV else None
    if preconditioned:
        def f(A, b, x0, max_iter, preconditioner=None, callbacks=[], **kwargs):
            w_replace = take_w_replace(kwargs)
            return _run(variant, name, A, b, x0, max_iter, preconditioner, callbacks, kwargs, w_replace)
    else:
        def f(A, b, x0, max_iter, callbacks=[], **kwargs):
            w_replace = take_w_replace(kwargs)
            kwargs.pop('preconditioner', None)   # figure_gen.py:59 always passes one
            return _run(variant, name, A, b, x0, max_iter, None, callbacks, kwargs, w_replace)
    f.__name__ = f.__qualname__ = name
    return f


hs_cg = _make(L.HS, 'hs_cg', False)                       # hs_cg.py:9
hs_pcg = _make(L.HS, 'hs_pcg', True)                      # hs_cg.py:70
pr_pcg = _make(L.PR, 'pr_pcg', True)                      # pr_cg.py:166
m_pcg = _make(L.M, 'm_pcg', True)                         # pr_cg.py:172
# the reference's unpreconditioned pr_cg / m_cg raise NameError (pr_cg.py:24,54); here
# they are the identity-preconditioned recurrences, which is what they were meant to be
pr_cg = _make(L.PR, 'pr_cg', False)
m_cg = _make(L.M, 'm_cg', False)
cg_cg = _make(L.CG_CG, 'cg_cg', False)                    # cg_cg.py:9   (Chronopoulos-Gear)
cg_pcg = _make(L.CG_CG, 'cg_pcg', True)                   # cg_cg.py:76
gv_cg = _make(L.GV, 'gv_cg', False)                       # gv_cg.py:9   (Ghysels-Vanroose, pipelined CG)
gv_pcg = _make(L.GV, 'gv_pcg', True)                      # gv_cg.py:93
pipe_p_cg = _make(L.PIPE_P, 'pipe_p_cg', False)           # pipe_pr_cg.py:83
pipe_pr_cg = _make(L.PIPE_PR, 'pipe_pr_cg', False)        # pipe_pr_cg.py:89
pipe_p_m_cg = _make(L.PIPE_P_M, 'pipe_p_m_cg', False)     # pipe_pr_cg.py:95
pipe_pr_m_cg = _make(L.PIPE_PR_M, 'pipe_pr_m_cg', False)  # pipe_pr_cg.py:101
pipe_p_pcg = _make(L.PIPE_P, 'pipe_p_pcg', True)          # pipe_pr_cg.py:195
pipe_pr_pcg = _make(L.PIPE_PR, 'pipe_pr_pcg', True)       # pipe_pr_cg.py:201
pipe_p_m_pcg = _make(L.PIPE_P_M, 'pipe_p_m_pcg', True)    # pipe_pr_cg.py:207
pipe_pr_m_pcg = _make(L.PIPE_PR_M, 'pipe_pr_m_pcg', True) # pipe_pr_cg.py:213

__all__ = ['hs_cg', 'hs_pcg', 'cg_cg', 'cg_pcg', 'gv_cg', 'gv_pcg', 'pr_cg', 'pr_pcg', 'm_cg', 'm_pcg',
           'pipe_p_cg', 'pipe_pr_cg', 'pipe_p_m_cg', 'pipe_pr_m_cg',
           'pipe_p_pcg', 'pipe_pr_pcg', 'pipe_p_m_pcg', 'pipe_pr_m_pcg', 'Jacobi', 'clear_operator_cache']
