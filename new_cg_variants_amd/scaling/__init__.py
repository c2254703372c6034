"""Distributed CG variants with the call shape of the reference's mpi4py experiment:

    sol, times = variant(comm, A, b, max_iter)
    (scaling_experiments_mpi4py/scaling_tests.py:71; cg_variants/{pipe_pr_cg,hs_cg,cg_cg,gv_cg,pr_cg}.py:7)

one process per GPU.  ``comm`` is a communicator-like object (``TorchComm`` wraps
torch.distributed; anything with Get_rank/Get_size/Barrier/allgather_obj/bcast_obj
works); ``A`` is the rank's CSR row block with GLOBAL column ids (shape n_local x n), a
``RowBlockOperator`` already resident on the GPU, or -- exactly what the reference's driver
passes -- the rank's dense n x (n/size) column block of a symmetric operator; ``b`` the rank's slice of the
right-hand side.  As in the reference x0 = 0, r0 = p0 = b, the loop body runs
``max_iter`` times, the return value is the local slice of x and ``{'tot': seconds}`` on
rank 0 (``None`` elsewhere), and the clock starts/stops between barriers
(pipe_pr_cg.py:54-56,85-87).

What differs by design (BASELINE.json north_star): the operator is a sparse row block,
not a dense column block, so the per-iteration traffic is a neighbour halo exchange of
the SpMV input plus ONE 5-double RCCL all-reduce, issued on a side stream and overlapped
with the interior rows of the matrix product (HS-CG: two dependent all-reduces, nothing
to overlap them with -- that contrast is the experiment).
"""
import os
import time

import numpy as np

from .. import _lib as L
from .. import partition
from ..device import DeviceCSR


class TorchComm:
    """torch.distributed as a communicator-like object (control plane only: the
    per-iteration data path is RCCL inside libprcg.so)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self._group = group

    def Get_rank(self):
        return self._dist.get_rank(self._group)

    def Get_size(self):
        return self._dist.get_world_size(self._group)

    def Barrier(self):
        self._dist.barrier(self._group)

    def allgather_obj(self, obj):
        out = [None] * self.Get_size()
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def bcast_obj(self, obj, root=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=root, group=self._group)
        return box[0]


class SelfComm:
    """The one-rank communicator (no torch needed)."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def Barrier(self):
        pass

    def allgather_obj(self, obj):
        return [obj]

    def bcast_obj(self, obj, root=0):
        return obj


class RowBlockOperator:
    """The rank's row block on its GPU, halo plan and RCCL communicator included."""

    def __init__(self, comm, A_rows, device=None, rccl_path=None, knobs=None):
        self.comm = comm
        self.A_rows, self.rccl_path = A_rows, rccl_path
        rank, size = comm.Get_rank(), comm.Get_size()
        n_local, n = A_rows.shape
        sizes = comm.allgather_obj(int(n_local))
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        if self.offsets[-1] != n:
            raise ValueError(f'row blocks cover {self.offsets[-1]} rows of an n={n} operator')
        lo, hi = int(self.offsets[rank]), int(self.offsets[rank + 1])
        self.n_local, self.n = n_local, n
        if device is None:
            # (PRCG_BENCH_DEVICE: every rank on one named device -- rehearsals of an N > 1 launch on fewer GPUs, see bench.py)
            device = int(os.environ.get('PRCG_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        self.device = device
        comm_init = None
        halo = None
        if size > 1:
            A_local, self.ghost_ids = partition.localize(A_rows, lo, hi)
            halo = partition.plan_halo(self.ghost_ids, self.offsets, rank, comm.allgather_obj)
            path = rccl_path or L.default_rccl_path()
            lib = L.lib()
            uid = None
            if rank == 0:
                # One RCCL communicator by default: halo exchange and all-reduce form one chain
                # on the communication stream.  PRCG_DUAL_COMM=1 gives the halo a communicator
                # and a stream of its own so the two run side by side -- faster on paper, but two
                # communicators progressing concurrently is only safe while the GPU can keep both
                # kernels resident, and this build has not been on more than one GPU yet.
                n_ids = 2 if os.environ.get('PRCG_DUAL_COMM', '0') == '1' else 1
                buf = np.zeros((n_ids, 128), dtype=np.uint8)
                from ..device import _stdout_to_stderr
                with _stdout_to_stderr():
                    for i in range(n_ids):
                        L.check(None, lib.prcg_comm_unique_id(path.encode(), L.ptr(buf[i])))
                uid = buf.tobytes()
            uid = comm.bcast_obj(uid, root=0)
            comm_init = (rank, size, uid, path)
        else:
            A_local = A_rows.tocsr()
            self.ghost_ids = np.zeros(0, dtype=np.int64)
        self.dev = DeviceCSR(A_local, device=device, comm_init=comm_init, halo=halo, knobs=knobs)
        # Direct peer exchange over xGMI (include/prcg.h): the per-iteration reduction and halo as stores into the
        # consumers' buffers instead of collectives.  All ranks or none; PRCG_PEER=0 (environment or knob) keeps RCCL.
        self.peer = False
        want = (knobs or {}).get('PRCG_PEER', os.environ.get('PRCG_PEER', '1')) != '0'
        if size > 1 and want:
            self.peer = partition.connect_peer_exchange(self.dev, rank, comm.allgather_obj)


def one_launch_self_check(op, variant, b, x0, inv_diag=None, iters=24):
    """Collective.  Before anything is timed or trusted on a new node: the first iterations of the multi-rank one-launch
    schedule (direct peer exchange: in-kernel waits for stores of OTHER GPUs) against the RCCL two-kernel schedule on the
    same row blocks (stream-ordered collectives only).  The two differ in the summation order of the inner products
    alone (~1e-12); a protocol or visibility problem between GPUs would show as a gross difference, or as a bounded
    wait running out.  Returns None if they agree (or the session does not use the one-launch schedule), else the
    reason -- the same on every rank."""
    comm, dev = op.comm, op.dev
    err, nu_a, one_launch = None, None, False
    try:
        dev.begin(variant, b, x0, iters + 8, inv_diag=inv_diag)
        one_launch = dev.schedule()['fused_comm']
        if one_launch:
            dev.iterate(iters)
            dev.sync()
            nu_a = np.array([dev.get_scalars(k)[L.S_NU] for k in range(0, iters + 1, 3)])
    except L.PrcgError as exc:
        one_launch, err = True, str(exc)
    if not any(comm.allgather_obj(bool(one_launch))):
        return None
    verdict = None
    try:
        ref = RowBlockOperator(comm, op.A_rows, device=op.device, rccl_path=op.rccl_path, knobs={'PRCG_FUSED_COMM': '0', 'PRCG_PEER': '0'})
        ref.dev.begin(variant, b, x0, iters + 8, inv_diag=inv_diag)
        ref.dev.iterate(iters)
        ref.dev.sync()
        nu_b = np.array([ref.dev.get_scalars(k)[L.S_NU] for k in range(0, iters + 1, 3)])
        ref.dev.close()
        if err is not None:
            verdict = err
        elif nu_a is None or not np.all(np.abs(nu_a - nu_b) <= 1e-6 * np.abs(nu_b)):
            verdict = f'one-launch and RCCL schedules disagree: nu = {None if nu_a is None else nu_a.tolist()} vs {nu_b.tolist()}'
    except L.PrcgError as exc:
        verdict = f'reference (RCCL two-kernel) schedule failed: {exc}'
    return next((v for v in comm.allgather_obj(verdict) if v), None)


def _as_operator(comm, A):
    if isinstance(A, RowBlockOperator):
        return A
    if isinstance(A, np.ndarray) and A.ndim == 2:
        # The reference hands each rank the dense n x (n/size) COLUMN block of the operator
        # (scaling_tests.py:51-54) and completes the product with an all-reduce of the n-vector.
        # CG requires a symmetric operator, so column block j is the transpose of row block j:
        # that row block, in CSR, is what the device wants (the model problem is diagonal).
        import scipy.sparse as sp
        m = A.shape[1]
        if A.shape[0] != m * comm.Get_size():
            raise ValueError(f'dense operator block of shape {A.shape}: expected (n, n/size) as in the reference')
        A = sp.csr_matrix(np.ascontiguousarray(A.T))
    return RowBlockOperator(comm, A)


def _timed(comm, op, variant, b, max_iter):
    """One timed solve.  If the one-launch multi-rank schedule cannot be kept fed on this node (its in-kernel waits are
    bounded; a rank that waits too long reports it), the ranks agree on that and repeat the solve on the RCCL
    two-kernel schedule -- the caller gets a result either way."""
    try:
        out, err = _timed_once(comm, op, variant, b, max_iter), None
    except L.PrcgError as exc:
        out, err = None, str(exc)
    errs = [e for e in comm.allgather_obj(err) if e]
    if not errs:
        return out
    if not op.dev.schedule().get('fused_comm'):
        raise L.PrcgError(L.ERCCL, errs[0])
    import sys
    if comm.Get_rank() == 0:
        print(f'[prcg] one-launch multi-rank schedule gave up ({errs[0][:120]}); repeating on the RCCL two-kernel schedule', file=sys.stderr)
    op.dev.close()
    op2 = RowBlockOperator(comm, op.A_rows, device=op.device, rccl_path=op.rccl_path, knobs={'PRCG_FUSED_COMM': '0', 'PRCG_PEER': '0'})
    op.dev, op.peer = op2.dev, False
    return _timed_once(comm, op, variant, b, max_iter)


def _timed_once(comm, op, variant, b, max_iter):
    dev = op.dev
    b = L.f64(b)
    rank = comm.Get_rank()
    times = {'tot': 0., 'c_ip': 0., 'c_mv': 0., 'w_mv': 0., 'w_ip': 0., 'w_vec': 0.} if rank == 0 else None
    # x0 = 0, r0 = p0 = b; one extra history slot because the reference's loop body runs
    # max_iter times (pipe_pr_cg.py:58), not max_iter - 1
    # (HIP loads a kernel's code object on its first launch -- tens of milliseconds per variant: two untimed
    #  iterations of a throw-away session touch every kernel of the loop before the clock starts)
    dev.begin(variant, b, np.zeros_like(b), 4)
    dev.iterate(2)
    dev.sync()
    dev.begin(variant, b, np.zeros_like(b), max_iter + 1)
    comm.Barrier()
    t0 = time.perf_counter()
    dev.iterate(max_iter)
    dev.sync()
    comm.Barrier()
    if rank == 0:
        times['tot'] = time.perf_counter() - t0
    return dev.get_vector('x'), times


def pipe_pr_cg(comm, A, b, max_iter):
    """Pipelined predict-and-recompute CG (scaling_experiments_mpi4py/cg_variants/pipe_pr_cg.py:7)."""
    return _timed(comm, _as_operator(comm, A), L.PIPE_PR, b, max_iter)


def hs_cg(comm, A, b, max_iter):
    """Hestenes-Stiefel CG, the two-reduction baseline (cg_variants/hs_cg.py:7)."""
    return _timed(comm, _as_operator(comm, A), L.HS, b, max_iter)


def pipe_p_cg(comm, A, b, max_iter):
    """Pipelined predict-only CG (no recompute of w = A r): the PETSc series "pipeprcg_0"
    (scaling_experiments_petsc/strong_scaling_tests.py:60-62; cg_impls/pipeprcg.c:33)."""
    return _timed(comm, _as_operator(comm, A), L.PIPE_P, b, max_iter)


def pr_cg(comm, A, b, max_iter):
    """Predict-and-recompute CG, one (blocking) reduction per iteration (cg_variants/pr_cg.py)."""
    return _timed(comm, _as_operator(comm, A), L.PR, b, max_iter)


def cg_cg(comm, A, b, max_iter):
    """Chronopoulos-Gear CG (cg_variants/cg_cg.py:7): one reduction {nu, eta} per iteration, after the
    product it depends on -- nothing to overlap it with."""
    return _timed(comm, _as_operator(comm, A), L.CG_CG, b, max_iter)


def gv_cg(comm, A, b, max_iter):
    """Ghysels-Vanroose pipelined CG (cg_variants/gv_cg.py:7): the reduction {nu, eta} is issued before
    the product t = A w and overlaps it on the communication stream."""
    return _timed(comm, _as_operator(comm, A), L.GV, b, max_iter)


__all__ = ['TorchComm', 'SelfComm', 'RowBlockOperator', 'one_launch_self_check', 'pipe_pr_cg', 'pipe_p_cg', 'hs_cg', 'pr_cg', 'cg_cg', 'gv_cg']
