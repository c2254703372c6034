"""ex2b-compatible command line: the reference's PETSc strong-scaling driver on MI355X.

    python -m new_cg_variants_amd.experiments.ex2b -n 650000 -k 32 -rho 0.95 -kappa 1e6 \
        -off_value 1e-4 -ksp_type pipeprcg -ksp_max_it 4000 [-num_repeat 1]
    python -m torch.distributed.run --nproc-per-node 8 -m new_cg_variants_amd.experiments.ex2b ...

Same problem, options and final line as scaling_experiments_petsc/ex2b.c: banded SPD model
matrix (ex2b.c:86-96), exact solution u = 1, b = A u (:138-139), zero initial guess, exactly
-ksp_max_it iterations (tolerances 0 and -ksp_norm_type none, strong_scaling_tests.py:66-72),
-num_repeat solves (:184-186), then

    Norm of error <g> iterations <d>                                      (ex2b.c:200)

-ksp_type: cg -> HS-CG, prcg -> PR-CG, pipeprcg -> pipelined PR-CG; with -recompute_q 0
(the reference's "pipeprcg_0" series, strong_scaling_tests.py:60-62) -> the predict-only
pipelined variant.  Defaults of the options are ex2b.c's (:25-26).  Extra: the solve time of
the last repeat is printed to stderr.
"""
import os
import sys
import time

import numpy as np


def parse(argv):
    opt = {'n': 8, 'k': 2, 'num_repeat': 1, 'rho': 0.8, 'kappa': 1e3, 'off_value': 1e-3,
           'ksp_type': 'cg', 'ksp_max_it': 10000, 'recompute_q': 1}
    i = 0
    while i < len(argv):
        key = argv[i].lstrip('-')
        if key in opt and i + 1 < len(argv):
            cast = type(opt[key]) if not isinstance(opt[key], str) else str
            opt[key] = cast(float(argv[i + 1])) if cast in (int, float) else argv[i + 1]
            i += 2
        else:
            # PETSc options that do not change the arithmetic here: -pc_type none,
            # -ksp_norm_type none, -mat_type, -log_view, -ksp_view, -ksp_converged_reason ...
            i += 2 if (i + 1 < len(argv) and not argv[i + 1].startswith('-')) else 1
    return opt


def main(argv=None):
    opt = parse(sys.argv[1:] if argv is None else argv)
    fd = os.dup(1)              # the result line goes to the real stdout; library chatter to stderr
    os.dup2(2, 1)
    from .. import partition, problems, scaling
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')
        comm = scaling.TorchComm()
    else:
        comm = scaling.SelfComm()
    n, k = int(opt['n']), int(opt['k'])
    offsets = partition.even_offsets(n, world)
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    A_rows = problems.banded_ex2b(n, k, kappa=opt['kappa'], rho=opt['rho'], off_value=opt['off_value'], rows=(lo, hi))
    b = A_rows @ np.ones(n)                                     # u = 1, b = A u
    solver = {'cg': scaling.hs_cg, 'prcg': scaling.pr_cg,
              'pipeprcg': scaling.pipe_pr_cg if int(opt['recompute_q']) else scaling.pipe_p_cg}.get(opt['ksp_type'])
    if solver is None:
        sys.exit(f"-ksp_type {opt['ksp_type']} is not available on the device (cg, prcg, pipeprcg)")
    op = scaling.RowBlockOperator(comm, A_rows)
    its = int(opt['ksp_max_it'])
    for _ in range(max(1, int(opt['num_repeat']))):
        t0 = time.perf_counter()
        x, _t = solver(comm, op, b, its)
        dt = time.perf_counter() - t0
    part = float(np.sum((x - 1.0) ** 2))
    total = sum(comm.allgather_obj(part))
    if rank == 0:
        print(f'solve: {dt:.3f} s, {its / dt:.1f} iterations/s on {world} GPU(s)', file=sys.stderr)
        os.write(fd, f'Norm of error {np.sqrt(total):g} iterations {its}\n'.encode())
    op.dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(fd, 1)              # give stdout back (main() may be called from a test or a notebook)
    os.close(fd)


if __name__ == '__main__':
    main()
