"""TEST INFRASTRUCTURE ONLY -- not part of the product, never imported by it.

CPU restatement (NumPy/SciPy, fp64) of the serial CG variants of the reference's
``numerical_experiments`` package, i.e. SURVEY.md section 8 rows a1-a3, a6, a7.

Pinned: ``tests/golden/make_golden.py`` runs the imported reference and this
restatement side by side in the build container and requires *bitwise* equal
histories before it writes the fixtures under ``tests/golden/``.

The reference writes every variant as one long function that re-binds ``_k`` /
``_k1`` names and hands ``locals()`` to its callbacks.  Here every variant is a
pair (``*_start``, ``*_advance``) acting on an explicit ``State`` so that a test
can (a) run free, (b) advance exactly one iteration from a stored state
("teacher forcing") and (c) swap the inner-product routine (``dot=``) for one
that reproduces the device's reduction tree.  The floating-point operations and
their order are those of the reference:

* vector updates are ``a + c*b`` / ``a - c*b`` with the product rounded first
  (NE/cg_variants/pipe_pr_cg.py:61-63,67-68; hs_cg.py:54-58);
* scalar recurrences keep Python's left-to-right evaluation, e.g.
  ``((nu - (2*a)*dl) + (a**2)*gm)`` (NE/cg_variants/pipe_pr_cg.py:64-65);
* ``A @ v`` is SciPy's CSR product, inner products default to ``numpy.dot``.

Paths: NE/ = /root/reference/predict_and_recompute/numerical_experiments/.
"""
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

Vec = np.ndarray


@dataclass
class State:
    """Everything a variant carries from one iteration to the next.

    Vector slots not used by a family stay ``None``.  ``alpha`` is the step that
    the *next* call of ``*_advance`` will apply (the reference's ``a_k``);
    ``beta`` is the one the last call used (``b_k``).
    """
    k: int = 0
    x: Vec = None
    r: Vec = None
    p: Vec = None
    s: Vec = None
    w: Vec = None
    u: Vec = None
    rt: Vec = None   # tilde (preconditioned) companions
    st: Vec = None
    wt: Vec = None
    ut: Vec = None
    nu: float = 0.0
    mu: float = 0.0
    dl: float = 0.0
    gm: float = 0.0
    eta: float = 0.0
    alpha: float = 0.0
    beta: float = 0.0
    nu_pred: float = 0.0  # the predicted nu of the last advance (before recompute)

    def clone(self):
        c = State()
        for f, v in self.__dict__.items():
            setattr(c, f, v.copy() if isinstance(v, np.ndarray) else v)
        return c


def _ident(v):
    return v


def jacobi(A):
    """The reference's Jacobi preconditioner: multiply by the reciprocal diagonal
    (NE/figure_gen.py:43 -- ``(1/A.diagonal())*x``)."""
    inv_d = 1 / A.diagonal()
    return lambda v: inv_d * v


# ---------------------------------------------------------------------------
# Hestenes-Stiefel CG        NE/cg_variants/hs_cg.py:9-67 (hs_cg), 70-131 (hs_pcg)
# ---------------------------------------------------------------------------
def hs_start(A, b, x0, prec=None, dot=np.dot):
    M = prec or _ident
    st = State()
    st.x = np.array(x0, dtype=np.float64, copy=True)          # hs_cg.py:22
    st.r = b - A @ st.x                                        # :23
    st.rt = M(st.r) if prec else None                          # :86
    z = st.rt if prec else st.r
    st.p = z.copy()                                            # :24 / :87
    st.nu = dot(st.r, z)                                       # :25 / :88
    st.s = A @ st.p                                            # :26
    st.mu = dot(st.p, st.s)                                    # :27
    st.alpha = st.nu / st.mu                                   # :28
    return st


def hs_advance(A, st, prec=None, dot=np.dot):
    M = prec or _ident
    a = st.alpha
    nu_old = st.nu
    st.x = st.x + a * st.p                                     # :54
    st.r = st.r - a * st.s                                     # :55
    if prec:
        st.rt = M(st.r)                                        # :118
    z = st.rt if prec else st.r
    st.nu = dot(st.r, z)                                       # :56 / :119
    st.beta = st.nu / nu_old                                   # :57
    st.p = z + st.beta * st.p                                  # :58
    st.s = A @ st.p                                            # :59
    st.mu = dot(st.p, st.s)                                    # :60
    st.alpha = st.nu / st.mu                                   # :61
    st.k += 1
    return st


# ---------------------------------------------------------------------------
# Predict-and-recompute CG (non-pipelined)   NE/cg_variants/pr_cg.py:93-176
#   flavour 'pr' -> pr_pcg, 'm' -> m_pcg (Meurant's nu prediction)
# (the unpreconditioned pr_cg/m_cg of the reference raise NameError, pr_cg.py:24,54;
#  the preconditioned ones with the identity are their oracle -- SURVEY.md 8c)
# ---------------------------------------------------------------------------
def _pow2(a):
    return a**2


def _predict_nu(flavour_m, nu, a, dl, gm, square=_pow2):
    """`a_k1**2` on a NumPy/Python float scalar goes through libm pow(a, 2.0), which is
    NOT always the correctly rounded square (about 1 case in 1000 is one ulp off).  The
    default reproduces the reference; tests that emulate the device (which multiplies)
    pass ``square=lambda a: a*a``."""
    if flavour_m:
        return -nu + square(a) * gm                            # pipe_pr_cg.py:64
    return nu - 2 * a * dl + square(a) * gm                    # pipe_pr_cg.py:65


def pr_start(A, b, x0, prec=None, dot=np.dot):
    M = prec or _ident
    st = State()
    st.x = np.array(x0, dtype=np.float64, copy=True)           # pr_cg.py:106
    st.r = b - A @ st.x                                        # :107
    st.rt = np.array(M(st.r), copy=True)                       # :108
    st.nu = dot(st.rt, st.r)                                   # :109
    st.p = st.rt.copy()                                        # :110
    st.s = A @ st.p                                            # :111
    st.st = np.array(M(st.s), copy=True)                       # :112
    st.mu = dot(st.p, st.s)                                    # :113
    st.alpha = st.nu / st.mu                                   # :114
    st.dl = dot(st.r, st.st)                                   # :115
    st.gm = dot(st.st, st.s)                                   # :116
    return st


def pr_advance(A, st, flavour='pr', prec=None, dot=np.dot, square=_pow2):
    M = prec or _ident
    a = st.alpha
    nu_old = st.nu
    st.x = st.x + a * st.p                                     # :146
    st.r = st.r - a * st.s                                     # :147
    st.rt = st.rt - a * st.st                                  # :148
    st.nu_pred = _predict_nu(flavour == 'm', nu_old, a, st.dl, st.gm, square)   # :149
    st.beta = st.nu_pred / nu_old                              # :150
    st.p = st.rt + st.beta * st.p                              # :151
    st.s = A @ st.p                                            # :152
    st.st = np.array(M(st.s), copy=True)                       # :153
    st.mu = dot(st.p, st.s)                                    # :154
    st.dl = dot(st.r, st.st)                                   # :155
    st.gm = dot(st.st, st.s)                                   # :156
    st.nu = dot(st.rt, st.r)                                   # :157
    st.alpha = st.nu / st.mu                                   # :158
    st.k += 1
    return st


# ---------------------------------------------------------------------------
# Pipelined predict-and-recompute CG
#   NE/cg_variants/pipe_pr_cg.py:9-81 (unpreconditioned), :109-193 (preconditioned)
#   flavour in {'p', 'pr', 'p_m', 'pr_m'}: 'pr*' recomputes w = A r (:70),
#   '*_m' uses Meurant's prediction (:64)
# ---------------------------------------------------------------------------
def pipe_start(A, b, x0, prec=None, dot=np.dot):
    st = State()
    st.x = np.array(x0, dtype=np.float64, copy=True)           # :22
    st.r = b - A @ st.x                                        # :23
    if prec is None:
        st.p = st.r.copy()                                     # :24
        st.nu = dot(st.r, st.r)                                # :25
        st.s = A @ st.p                                        # :26
        st.w = st.s.copy()                                     # :27
        st.u = A @ st.w                                        # :28
        st.mu = dot(st.p, st.s)                                # :29
        st.alpha = st.nu / st.mu                               # :30
        st.dl = dot(st.r, st.s)                                # :35
        st.gm = dot(st.s, st.s)                                # :36
        return st
    st.rt = np.array(prec(st.r), copy=True)                    # :124
    st.p = st.rt.copy()                                        # :125
    st.nu = dot(st.rt, st.r)                                   # :126
    st.s = A @ st.p                                            # :127
    st.st = np.array(prec(st.s), copy=True)                    # :128
    st.w = st.s.copy()                                         # :129
    st.wt = st.st.copy()                                       # :130
    st.u = A @ st.st                                           # :131
    st.ut = np.array(prec(st.u), copy=True)                    # :132
    st.mu = dot(st.p, st.s)                                    # :133
    st.alpha = st.nu / st.mu                                   # :134
    st.dl = dot(st.r, st.st)                                   # :139
    st.gm = dot(st.st, st.s)                                   # :140
    return st


def pipe_advance(A, st, flavour='pr', prec=None, dot=np.dot, square=_pow2):
    recompute = flavour.startswith('pr')
    a = st.alpha
    nu_old = st.nu
    st.x = st.x + a * st.p                                     # :61 / :169
    st.r = st.r - a * st.s                                     # :62 / :170
    if prec is None:
        st.w = st.w - a * st.u                                 # :63
        st.nu_pred = _predict_nu(flavour.endswith('m'), nu_old, a, st.dl, st.gm, square)
        st.beta = st.nu_pred / nu_old                          # :66
        st.p = st.r + st.beta * st.p                           # :67
        st.s = st.w + st.beta * st.s                           # :68
        st.u = A @ st.s                                        # :69
        if recompute:
            st.w = A @ st.r                                    # :70
        st.mu = dot(st.p, st.s)                                # :71
        st.dl = dot(st.r, st.s)                                # :72
        st.gm = dot(st.s, st.s)                                # :73
        st.nu = dot(st.r, st.r)                                # :74
    else:
        st.rt = st.rt - a * st.st                              # :171
        st.w = st.w - a * st.u                                 # :172
        st.wt = st.wt - a * st.ut                              # :173
        st.nu_pred = _predict_nu(flavour.endswith('m'), nu_old, a, st.dl, st.gm, square)   # :174
        st.beta = st.nu_pred / nu_old                          # :175
        st.p = st.rt + st.beta * st.p                          # :176
        st.s = st.w + st.beta * st.s                           # :177
        st.st = st.wt + st.beta * st.st                        # :178
        st.u = A @ st.st                                       # :179
        st.ut = np.array(prec(st.u), copy=True)                # :180
        if recompute:
            st.w = A @ st.rt                                   # :181
            st.wt = np.array(prec(st.w), copy=True)            # :182
        st.mu = dot(st.p, st.s)                                # :183
        st.dl = dot(st.r, st.st)                               # :184
        st.gm = dot(st.st, st.s)                               # :185
        st.nu = dot(st.rt, st.r)                               # :186
    st.alpha = st.nu / st.mu                                   # :75 / :187
    st.k += 1
    return st


# ---------------------------------------------------------------------------
# Competitor baselines
#   Chronopoulos-Gear  NE/cg_variants/cg_cg.py:9-74 (cg_cg), :76-143 (cg_pcg)
#   Ghysels-Vanroose   NE/cg_variants/gv_cg.py:9-91 (gv_cg), :93-181 (gv_pcg); w_replace never fires by default
# ---------------------------------------------------------------------------
def cgcg_start(A, b, x0, prec=None, dot=np.dot):
    M = prec or _ident
    st = State()
    st.x = np.array(x0, dtype=np.float64, copy=True)
    st.r = b - A @ st.x                                        # cg_cg.py:23 / :88
    st.rt = np.array(M(st.r), copy=True) if prec else None     # :89
    z = st.rt if prec else st.r
    st.w = A @ z                                               # :24 / :90
    st.p = z.copy()                                            # :25 / :91
    st.nu = dot(st.r, z)                                       # :26 / :92
    st.eta = dot(st.w, z)                                      # :27 / :93
    st.s = A @ st.p                                            # :28 / :94
    st.u = st.s.copy()                                         # :29 / :95
    st.mu = dot(st.p, st.s)                                    # :30 / :96
    st.alpha = st.nu / st.mu                                   # :31 / :97
    return st


def cgcg_advance(A, st, prec=None, dot=np.dot):
    M = prec or _ident
    a = st.alpha
    nu_old = st.nu
    st.x = st.x + a * st.p                                     # :59 / :127
    st.r = st.r - a * st.s                                     # :60 / :128
    if prec:
        st.rt = np.array(M(st.r), copy=True)                   # :129
    z = st.rt if prec else st.r
    st.w = A @ z                                               # :61 / :130
    st.nu = dot(st.r, z)                                       # :62 / :131
    st.eta = dot(st.w, z)                                      # :63 / :132
    st.beta = st.nu / nu_old                                   # :64 / :133
    st.p = z + st.beta * st.p                                  # :65 / :134
    st.s = st.w + st.beta * st.s                               # :66 / :135
    st.mu = st.eta - (st.beta / a) * st.nu                     # :67 / :136
    st.alpha = st.nu / st.mu                                   # :68 / :137
    st.k += 1
    return st


def gv_start(A, b, x0, prec=None, dot=np.dot):
    st = State()
    st.x = np.array(x0, dtype=np.float64, copy=True)
    st.r = b - A @ st.x                                        # gv_cg.py:26 / :107
    if prec is None:
        st.w = A @ st.r                                        # :27
        st.p = st.r.copy()                                     # :28
        st.s = st.w.copy()                                     # :29
        st.u = A @ st.w                                        # :30
        st.nu = dot(st.r, st.r)                                # :31
        st.eta = dot(st.w, st.r)                               # :32
    else:
        st.rt = np.array(prec(st.r), copy=True)                # :108
        st.w = A @ st.rt                                       # :109
        st.wt = np.array(prec(st.w), copy=True)                # :110
        st.p = st.rt.copy()                                    # :111
        st.s = st.w.copy()                                     # :112
        st.st = st.wt.copy()                                   # :113
        st.u = A @ st.wt                                       # :114
        st.nu = dot(st.r, st.rt)                               # :115
        st.eta = dot(st.w, st.r)                               # :116  (r, not r~ -- as the reference has it)
    st.mu = dot(st.p, st.s)                                    # :33 / :117
    st.alpha = st.nu / st.mu                                   # :34 / :118
    return st


def gv_advance(A, st, prec=None, dot=np.dot, w_replace=None, b=None, flags=None):
    """w_replace: the reference's residual-replacement predicate (gv_cg.py:9,69-71 / :93,156-158), called with the
    reference's keywords after x, r, w are updated; True replaces w by A @ r (also in gv_pcg: r, not r~)."""
    a = st.alpha
    nu_old = st.nu
    r_prev = st.r
    st.x = st.x + a * st.p                                     # :65 / :152
    st.r = st.r - a * st.s                                     # :66 / :153
    if prec is None:
        st.w = st.w - a * st.u                                 # :67
        if w_replace is not None and w_replace(k=st.k + 1, A=A, b=b, x=st.x, w=st.w, r=st.r, r_=r_prev, u=st.u, s=st.s, p=st.p,
                                                wk_replace_flags=flags):          # :69
            st.w = A @ st.r                                    # :71
        t = A @ st.w                                           # :73
        st.nu = dot(st.r, st.r)                                # :74
        st.eta = dot(st.w, st.r)                               # :75
        z, zt = st.r, st.w
    else:
        st.rt = st.rt - a * st.st                              # :154
        st.w = st.w - a * st.u                                 # :155
        if w_replace is not None and w_replace(k=st.k + 1, A=A, b=b, x=st.x, w=st.w, r=st.r, r_=r_prev, u=st.u, s=st.s, p=st.p,
                                                wk_replace_flags=flags):          # :156
            st.w = A @ st.r                                    # :158
        st.wt = np.array(prec(st.w), copy=True)                # :161
        t = A @ st.wt                                          # :162
        st.nu = dot(st.r, st.rt)                               # :163
        st.eta = dot(st.w, st.rt)                              # :164
        z, zt = st.rt, st.wt
    st.beta = st.nu / nu_old                                   # :76 / :165
    st.p = z + st.beta * st.p                                  # :77 / :166
    st.s = st.w + st.beta * st.s                               # :78 / :167
    if prec is not None:
        st.st = zt + st.beta * st.st                           # :168
    st.u = t + st.beta * st.u                                  # :79 / :169
    st.mu = st.eta - (st.beta / a) * st.nu                     # :80 / :170
    st.alpha = st.nu / st.mu                                   # :81 / :171
    st.k += 1
    return st


# ---------------------------------------------------------------------------
# History recorders -- the four callbacks figure_gen uses (NE/figure_gen.py:37)
# ---------------------------------------------------------------------------
def rec_updated_residual_2_norm(A, b, st, x_true):
    return np.linalg.norm(st.r)            # NE/callbacks/updated_residual_2_norm.py:40


def rec_residual_2_norm(A, b, st, x_true):
    return np.linalg.norm(b - A @ st.x)    # NE/callbacks/residual_2_norm.py:41


def rec_error_A_norm(A, b, st, x_true):
    e = st.x - x_true                      # NE/callbacks/error_A_norm.py:47
    with np.errstate(invalid='ignore'):
        return np.sqrt(e.T @ (A @ e))      # :48


def rec_error_2_norm(A, b, st, x_true):
    e = st.x - x_true                      # NE/callbacks/error_2_norm.py:47
    return np.linalg.norm(e)               # :48


RECORDERS = {
    'updated_residual_2_norm': rec_updated_residual_2_norm,
    'residual_2_norm': rec_residual_2_norm,
    'error_A_norm': rec_error_A_norm,
    'error_2_norm': rec_error_2_norm,
}

# family -> (start, advance, takes_flavour)
FAMILIES = {
    'hs': (hs_start, hs_advance, False),
    'pr': (pr_start, pr_advance, True),
    'pipe': (pipe_start, pipe_advance, True),
    'cg_cg': (cgcg_start, cgcg_advance, False),
    'gv': (gv_start, gv_advance, False),
}


def run(family, A, b, x0, max_iter, flavour=None, prec=None, recorders=(),
        x_true=None, dot=np.dot, tap: Optional[Callable] = None, name=None, square=_pow2, dot0=None, w_replace=None):
    """Free-running solve with the reference's loop shape: recorders fire on the
    initial state (index 0) and after each of the ``max_iter - 1`` iterations
    (NE/cg_variants/hs_cg.py:33-36,39,64-65).  ``tap(state)`` sees every state.  ``dot0``: inner-product
    routine of the initial state only (the device sums that one with another kernel's tree)."""
    start, advance, has_flavour = FAMILIES[family]
    out = {'name': name or family, 'max_iter': max_iter}
    for q in recorders:
        out[q] = np.zeros(max_iter)
    st = start(A, b, x0, prec=prec, dot=dot0 or dot)
    flags = {}      # the reference's wk_replace_flags: storage a w_replace predicate may use between iterations

    def record():
        for q in recorders:
            out[q][st.k] = RECORDERS[q](A, b, st, x_true)
        if tap is not None:
            tap(st)

    record()
    with np.errstate(all='ignore'):
        for _ in range(1, max_iter):
            if has_flavour:
                advance(A, st, flavour, prec=prec, dot=dot, square=square)
            elif family == 'gv' and w_replace is not None:
                advance(A, st, prec=prec, dot=dot, w_replace=w_replace, b=b, flags=flags)
            else:
                advance(A, st, prec=prec, dot=dot)
            record()
    out['_final_state'] = st
    return out


# -- entry points with the reference's names and call shape -------------------
def _public(family, flavour, ref_name, preconditioned):
    def f(A, b, x0, max_iter, preconditioner=None, callbacks=(), **kwargs):
        prec = preconditioner if preconditioned else None
        out = run(family, A, b, x0, max_iter, flavour=flavour, prec=prec,
                  recorders=tuple(callbacks), x_true=kwargs.get('x_true'),
                  dot=kwargs.get('dot', np.dot), tap=kwargs.get('tap'), name=ref_name,
                  square=kwargs.get('square', _pow2), dot0=kwargs.get('dot0'), w_replace=kwargs.get('w_replace'))
        return out
    f.__name__ = ref_name
    return f


hs_cg = _public('hs', None, 'hs_cg', False)                    # NE/cg_variants/hs_cg.py:9
hs_pcg = _public('hs', None, 'hs_pcg', True)                   # :70
pr_pcg = _public('pr', 'pr', 'pr_pcg', True)                   # NE/cg_variants/pr_cg.py:166
m_pcg = _public('pr', 'm', 'm_pcg', True)                      # :172
pipe_p_cg = _public('pipe', 'p', 'pipe_p_cg', False)           # NE/cg_variants/pipe_pr_cg.py:83
pipe_pr_cg = _public('pipe', 'pr', 'pipe_pr_cg', False)        # :89
pipe_p_m_cg = _public('pipe', 'p_m', 'pipe_p_m_cg', False)     # :95
pipe_pr_m_cg = _public('pipe', 'pr_m', 'pipe_pr_m_cg', False)  # :101
pipe_p_pcg = _public('pipe', 'p', 'pipe_p_pcg', True)          # :195
pipe_pr_pcg = _public('pipe', 'pr', 'pipe_pr_pcg', True)       # :201
pipe_p_m_pcg = _public('pipe', 'p_m', 'pipe_p_m_pcg', True)    # :207
pipe_pr_m_pcg = _public('pipe', 'pr_m', 'pipe_pr_m_pcg', True) # :213
cg_cg = _public('cg_cg', None, 'cg_cg', False)                 # NE/cg_variants/cg_cg.py:9
cg_pcg = _public('cg_cg', None, 'cg_pcg', True)                # :76
gv_cg = _public('gv', None, 'gv_cg', False)                    # NE/cg_variants/gv_cg.py:9
gv_pcg = _public('gv', None, 'gv_pcg', True)                    # :93


def convergence_summary(error_A_norm_history, tol=1e-5):
    """Iterations to relative A-norm error <= tol and log10 of the best relative
    error, as the reference's table does (NE/figure_gen.py:84-89)."""
    rel = error_A_norm_history / error_A_norm_history[0]
    with np.errstate(all='ignore'):
        return int(np.argmin(rel > tol)), float(np.log10(np.nanmin(rel)))
