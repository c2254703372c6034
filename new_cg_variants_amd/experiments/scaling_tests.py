"""scaling_tests-compatible driver: the reference's mpi4py strong-scaling experiment on MI355X.

    python -m new_cg_variants_amd.experiments.scaling_tests <n> <max_iter> <trial_name>
    python -m torch.distributed.run --nproc-per-node 8 -m new_cg_variants_amd.experiments.scaling_tests 12288 1500 t0

Same arguments, model problem, variant list, printed lines and saved files as
scaling_experiments_mpi4py/scaling_tests.py: the diagonal model problem with kappa = 1e6,
rho = 0.9 (:31-36), b = lambda / sqrt(n) so that the solution is the constant unit vector
(:57), the five variants of :63 each run for exactly <max_iter> iterations, then

    <variant> error: <|1/sqrt(n) - x|>                                   (:81-82)

and ``{"error": ..., "timings": {'tot': seconds, ...}}`` saved to
``./data/<n>/<variant>_<trial_name>.npy`` (:85-86).  One process per GPU (torch.distributed.run
instead of mpiexec); each rank holds its ROW block of the operator as CSR (the reference's dense
n x n/size column block of zeros around a diagonal is the same operator; see scaling/_as_operator).
"""
import os
import sys

import numpy as np


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        sys.exit('usage: scaling_tests <n> <max_iter> <trial_name>')
    n, max_iter, trial_name = int(argv[0]), int(argv[1]), argv[2]
    fd = os.dup(1)              # result lines go to the real stdout; library chatter to stderr
    os.dup2(2, 1)

    def say(line):
        os.write(fd, (line + '\n').encode())

    import scipy.sparse as sp
    from .. import scaling
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')
        comm = scaling.TorchComm()
    else:
        comm = scaling.SelfComm()
    assert n % world == 0, 'n must be a multiple of the number of processes'      # scaling_tests.py:26
    m = n // world
    kappa, rho = 1e6, 0.9                                                          # :30-31
    lambda1, lambdan = 1 / kappa, 1
    Lambda = lambda1 + (lambdan - lambda1) * np.arange(n) / (n - 1) * rho**np.arange(n - 1, -1, -1, dtype='float')
    if rank == 0:
        say(f'trial name: {trial_name}')
        say(f'start distributing to {world} ranks')
    lam = Lambda[rank * m:(rank + 1) * m].copy()
    rows = np.arange(m)
    A_rows = sp.csr_matrix((lam, (rows, rank * m + rows)), shape=(m, n))           # the rank's row block (:51-54)
    b = lam / np.sqrt(n)                                                           # :57
    comm.Barrier()
    if rank == 0:
        say('done distributing')
    op = scaling.RowBlockOperator(comm, A_rows)
    variants = [scaling.hs_cg, scaling.cg_cg, scaling.gv_cg, scaling.pr_cg, scaling.pipe_pr_cg]   # :63
    results = {}
    for variant in variants:
        comm.Barrier()
        sol, t = variant(comm, op, b, max_iter)                                    # :71
        parts = comm.allgather_obj(sol)                                            # :73-76
        if rank == 0:
            sol_raw = np.concatenate(parts)
            error = np.linalg.norm(np.ones(n) / np.sqrt(n) - sol_raw)              # :81
            say(f'{variant.__name__} error: {error}')
            res = {'error': error, 'timings': t}
            os.makedirs(f'./data/{n}', exist_ok=True)
            np.save(f'./data/{n}/{variant.__name__}_{trial_name}', res, allow_pickle=True)   # :85-86
            results[variant.__name__] = res
    op.dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    os.dup2(fd, 1)
    os.close(fd)
    return results


if __name__ == '__main__':
    main()
