import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from new_cg_variants_amd import _lib as L
h = C.c_void_p(); lib = L.lib()
assert lib.prcg_create(C.byref(h), 0) == 0
for n_pairs in (10_000_000, 64_000_000, 200_000_000):
    for mode in (0, 1, 2):
        g = C.c_double()
        rc = lib.prcg_stream_ceiling(h, n_pairs, mode, 10, C.byref(g))
        print(n_pairs, mode, rc, round(g.value, 1), 'GB/s', flush=True)
