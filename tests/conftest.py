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


@pytest.fixture(scope='session')
def matrices():
    return {m: load_matrix(m) for m in ('bcsstk03', 'nos7')}
