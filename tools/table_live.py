"""Debug helper: the paper's table, row by row, printing as it goes (find the row a fault occurs in)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from new_cg_variants_amd.experiments import figure_run as fr
from new_cg_variants_amd import cg_variants as cgv
from new_cg_variants_amd.callbacks import error_A_norm
G = os.path.join(ROOT, 'tests', 'golden')
rows = json.load(open(os.path.join(G, 'paper_convergence_table.json')))
for row in rows:
    name, prec = row['matrix'], row['preconditioner']
    A = fr.load_matrix(os.path.join(G, f'tablemat_{name}.npz'))
    N = A.shape[0]
    x_true = np.ones(N) / np.sqrt(N); b = A @ x_true; x0 = np.zeros(N)
    P = cgv.Jacobi(A) if prec == 'jacobi' else (lambda v: v)
    for m in row['columns']:
        print(name, prec, m, N, A.nnz, min(row['max_iter'], 4000), flush=True)
        getattr(cgv, m)(A, b, x0, min(row['max_iter'], 4000), callbacks=[error_A_norm], x_true=x_true, preconditioner=P)
print('all rows done')
