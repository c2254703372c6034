import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_matrix(name):
    """CSR matrix + problem data of a golden fixture (tests/golden/matrix_<name>.npz)."""
    z = np.load(os.path.join(GOLDEN, f'matrix_{name}.npz'))
    n = int(z['n'])
    A = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n))
    return A, z


def load_run(matrix, method, prec):
    return np.load(os.path.join(GOLDEN, f'run_{matrix}_{method}_{prec}.npz'))


def golden_state(run, k):
    """dict of the stored reference iterate at k (vectors and scalars)."""
    out = {}
    for f in ('x', 'r', 'p', 's', 'w', 'u', 'rt', 'st', 'wt', 'ut', 'nu', 'mu', 'dl', 'gm', 'eta', 'alpha', 'beta'):
        key = f'state{k}_{f}'
        if key in run.files:
            out[f] = run[key]
    return out


class _LazyMatrices(dict):
    def __missing__(self, name):
        self[name] = load_matrix(name)
        return self[name]

    def __contains__(self, name):
        return dict.__contains__(self, name) or os.path.exists(os.path.join(GOLDEN, f'matrix_{name}.npz'))


@pytest.fixture(scope='session')
def matrices():
    """name -> (CSR matrix, fixture arrays) for every tests/golden/matrix_<name>.npz, loaded on demand."""
    return _LazyMatrices()


def all_runs():
    """(matrix, method, prec) of every golden run file."""
    import glob
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, 'run_*.npz'))):
        tag = os.path.basename(p)[4:-4]
        for m in sorted((os.path.basename(q)[7:-4] for q in glob.glob(os.path.join(GOLDEN, 'matrix_*.npz'))),
                        key=len, reverse=True):
            if tag.startswith(m + '_'):
                method, prec = tag[len(m) + 1:].rsplit('_', 1)
                out.append((m, method, prec))
                break
    return out
