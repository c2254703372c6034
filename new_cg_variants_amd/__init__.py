"""new_cg_variants_amd -- the predict-and-recompute CG hot path of
tchen-research/new_cg_variants on AMD MI355X (gfx950).

    from new_cg_variants_amd.cg_variants import hs_cg, pipe_pr_cg, pipe_pr_pcg, ...
    from new_cg_variants_amd.callbacks import error_A_norm, residual_2_norm, ...
    trial = pipe_pr_cg(A, b, x0, max_iter, callbacks=[...], x_true=x_true)   # dict of histories

    from new_cg_variants_amd.scaling import pipe_pr_cg as dist_pipe_pr_cg       # (comm, A, b, max_iter)

Same call shapes as the reference's numerical_experiments/ and
scaling_experiments_mpi4py/ packages; the arithmetic runs in hand-written HIP kernels
behind the C-ABI of include/prcg.h (libprcg.so).  No CPU fallback exists.
"""
__all__ = ['cg_variants', 'callbacks', 'scaling', 'device', 'partition', 'problems']
