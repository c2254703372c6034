"""Build tools/plan_harness.cpp + csrc/prcg_plan.cpp with AddressSanitizer / UBSan (g++, CPU) and run the window
planning pipeline over every golden matrix and the synthetic operators, with and without image sharing.

    python tools/run_plan_asan.py          (re-executes itself with the sanitizer runtimes preloaded)
"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'build_ab', 'libplan_asan.so')


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(['g++', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-std=c++17', '-shared',
                           '-fPIC', '-o', SO, os.path.join(ROOT, 'tools', 'plan_harness.cpp'),
                           os.path.join(ROOT, 'new_cg_variants_amd', 'csrc', 'prcg_plan.cpp'), '-lpthread'])


def main():
    if 'libasan' not in os.environ.get('LD_PRELOAD', ''):
        build()
        libs = [subprocess.check_output(['gcc', f'-print-file-name={n}']).decode().strip() for n in ('libasan.so', 'libubsan.so')]
        env = dict(os.environ, LD_PRELOAD=':'.join(libs), ASAN_OPTIONS='detect_leaks=0')
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
    import numpy as np
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    from new_cg_variants_amd import problems
    lib = ctypes.CDLL(SO)
    P = ctypes.c_void_p
    lib.plan_all.argtypes = [ctypes.c_long, ctypes.c_long, P, P, P, ctypes.c_int, ctypes.c_int]

    def run(A, name):
        A = A.tocsr()
        A.sort_indices()
        ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        d = np.ascontiguousarray(A.data, dtype=np.float64)
        for rows in (64, 128):
            for share in (1, 0):
                rc = lib.plan_all(A.shape[0], A.shape[1], ip.ctypes.data, ix.ctypes.data, d.ctypes.data, rows, share)
                assert rc >= 0, (name, rows, share, rc)
        print(name, A.shape[0], A.nnz, 'ok', flush=True)

    for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'tablemat_*.npz'))):
        z = np.load(f)
        n = int(z['n'])
        run(sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n)), os.path.basename(f))
    run(problems.banded_ex2b(200000, 7), 'band')
    run(problems.laplace_2d(300, 200), 'lap2d')
    run(problems.laplace_3d(40, 40, 40), 'lap3d')
    run(problems.fem_like_3d(12, 3), 'fem')
    run(problems.laplace_3d(72, 8, 64), 'lap3d sweep')
    run(problems.laplace_2d(300, 64), 'lap2d sweep')
    run(problems.banded_ex2b(50000, 7, kappa=1.0), 'band, constant diagonal (pattern operator)')
    print('all clean')


if __name__ == '__main__':
    main()
