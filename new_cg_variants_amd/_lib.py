"""ctypes binding of libprcg.so (the C-ABI declared in include/prcg.h).

The library is built in-tree by ``make`` / ``__graft_entry__.build()``.  There is no
CPU fallback: if the shared object is missing, or no MI355X is visible when a handle is
created, this module raises -- loudly -- instead of computing anything on the host.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PRCG_LIB') or os.path.join(_HERE, 'libprcg.so')   # PRCG_LIB: A/B builds

# constants of include/prcg.h
OK, EINVAL, EHIP, ERCCL, ENOMEM = 0, 1, 2, 3, 4
HS, PIPE_PR, PIPE_P, PIPE_PR_M, PIPE_P_M, PR, M, CG_CG, GV = range(9)
HIST_UPDATED_RESIDUAL_2_NORM, HIST_RESIDUAL_2_NORM, HIST_ERROR_A_NORM, HIST_ERROR_2_NORM = 1, 2, 4, 8
HIST_BITS = {
    'updated_residual_2_norm': HIST_UPDATED_RESIDUAL_2_NORM,
    'residual_2_norm': HIST_RESIDUAL_2_NORM,
    'error_A_norm': HIST_ERROR_A_NORM,
    'error_2_norm': HIST_ERROR_2_NORM,
}
VEC = {'x': 0, 'r': 1, 'p': 2, 's': 3, 'w': 4, 'u': 5, 'rt': 6, 'st': 7, 'wt': 8, 'ut': 9}
S_MU, S_DELTA, S_GAMMA, S_NU, S_RR, S_RES2, S_ERRA2, S_ERR2 = range(8)
NUM_SCALARS = 8


class PrcgError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f'libprcg error {code}: {text}')
        self.code = code


class Timings(C.Structure):
    _fields_ = [('tot_ms', C.c_double), ('iter_ms', C.c_double), ('spmv_ms', C.c_double),
                ('update_ms', C.c_double), ('spmv_samples', C.c_int64), ('iterations', C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_P = C.c_void_p
_dp = C.POINTER(C.c_double)
_SIGNATURES = {
    'prcg_version': (C.c_int, []),
    'prcg_create': (C.c_int, [C.POINTER(_P), C.c_int]),
    'prcg_destroy': (None, [_P]),
    'prcg_last_error': (C.c_char_p, [_P]),
    'prcg_set_option': (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    'prcg_comm_unique_id': (C.c_int, [C.c_char_p, _P]),
    'prcg_comm_init': (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, _P, C.c_int]),
    'prcg_set_csr': (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int, _P, _P]),
    'prcg_set_halo': (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    'prcg_peer_setup': (C.c_int, [_P, C.c_int64, _P, C.POINTER(_P)]),
    'prcg_peer_connect': (C.c_int, [_P, _P, _P, _P]),
    'prcg_world_init': (C.c_int, [_P, C.c_int, C.c_int]),
    'prcg_peer_selftest': (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    'prcg_spmv': (C.c_int, [_P, _P, _P, C.c_int, _dp]),
    'prcg_spmv_ext': (C.c_int, [_P, _P, _P]),
    'prcg_spmm2': (C.c_int, [_P, _P, _P, C.c_int, _dp]),
    'prcg_solve_begin': (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, C.c_uint32]),
    'prcg_iterate': (C.c_int, [_P, C.c_int]),
    'prcg_sync': (C.c_int, [_P]),
    'prcg_iteration': (C.c_int, [_P]),
    'prcg_set_preconditioner': (C.c_int, [_P, _P, _P]),
    'prcg_set_replace_hook': (C.c_int, [_P, _P, _P]),
    'prcg_schedule': (C.c_int, [_P]),
    'prcg_operator_bytes': (C.c_int64, [_P]),
    'prcg_set_iteration': (C.c_int, [_P, C.c_int]),
    'prcg_get_vector': (C.c_int, [_P, C.c_int, _P]),
    'prcg_set_vector': (C.c_int, [_P, C.c_int, _P]),
    'prcg_get_scalars': (C.c_int, [_P, C.c_int, _P]),
    'prcg_set_scalars': (C.c_int, [_P, C.c_int, _P]),
    'prcg_get_coefficients': (C.c_int, [_P, C.c_int, _P]),
    'prcg_get_history': (C.c_int, [_P, _P]),
    'prcg_set_profiling': (C.c_int, [_P, C.c_int]),
    'prcg_get_timings': (C.c_int, [_P, C.POINTER(Timings)]),
    'prcg_solve': (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, C.c_uint32, _P, _P, C.POINTER(Timings)]),
    'prcg_stream_ceiling': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    'prcg_mix_ceiling': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    'prcg_plan_gather': (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int64, _P]),
    'prcg_plan_sell': (C.c_int64, [C.c_int64, _P, _P, _P, _P, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int64, _P,
                                    C.c_int64, _P]),
    'prcg_plan_tiles': (C.c_int64, [C.c_int64, _P, _P, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64)]),
    'prcg_plan_window': (C.c_int64, [C.c_int64, C.c_int64, _P, _P, _P, C.c_int, _P, C.c_int64, _P, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int)]),
    'prcg_plan_window_images': (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, C.c_int, C.c_int, _P]),
    'prcg_plan_sweep': (C.c_int64, [C.c_int64, _P, _P, _P, C.c_int, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P]),
    'prcg_plan_window_patterns': (C.c_int64, [C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P]),
    'prcg_tile_caps': (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'prcg_window_source_ok': (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int64]),
    'prcg_debug_layout': (C.c_int64, [_P, _P, C.c_int64]),
}

REPLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
PREC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))   # prcg_prec_fn
REPLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)                                                 # prcg_replace_fn

_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). '
                'new_cg_variants_amd has no CPU fallback.')
        global _torch_runtime
        _torch_runtime = 'torch' in sys.modules      # which HIP runtime libprcg.so binds to is decided HERE, at load time
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_torch_runtime = None


def default_rccl_path():
    """RCCL must be the copy that matches the HIP runtime libprcg.so itself is bound to: torch wheels bundle their
    own libamdhip64/librccl, so torch's copy iff torch was imported BEFORE libprcg.so was loaded (a torch imported later
    brings a second runtime into the process; streams of one handed to the RCCL of the other fail with "unhandled
    cuda error")."""
    lib()
    if os.environ.get('PRCG_RCCL_LIB'):
        # a named collectives library instead of RCCL (the tests' stand-ins: tests/transport/lib{threads,procs}_ccl.so)
        return os.environ['PRCG_RCCL_LIB']
    if _torch_runtime and 'torch' in sys.modules:
        cand = os.path.join(os.path.dirname(sys.modules['torch'].__file__), 'lib', 'librccl.so')
        if os.path.exists(cand):
            return cand
    for cand in ('/opt/rocm/lib/librccl.so.1', '/opt/rocm/lib/librccl.so'):
        if os.path.exists(cand):
            return cand
    return 'librccl.so.1'


def ptr(a):
    """void* of a C-contiguous ndarray (None -> NULL)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def check(handle, rc):
    if rc != OK:
        msg = lib().prcg_last_error(handle)
        raise PrcgError(rc, msg.decode() if msg else '?')
