"""Worker for the multi-rank tests (launched by torch.distributed.run / mp.spawn).

mode 'cpu'  (gloo, no GPU): exercises the HOST side of the N>1 path -- row-block
    partition, halo plan exchange over torch.distributed, and the exchange contract that
    libprcg implements on the device (send_idx order -> ghost slots) -- by running the
    oracle's distributed loop over a NumPy row-block operator that follows the plan
    literally.  The result must equal the single-rank oracle.
mode 'gpu'  (ranks share cuda:0 or own one GPU each): the real thing through libprcg
    (RCCL halo + all-reduce), checked against the single-rank oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class PlanFollowingOperator:
    """NumPy stand-in for the device row block: y_local = A_local [x_local ; ghosts], ghosts
    filled by the SAME plan arrays prcg_set_halo receives (test infrastructure)."""

    def __init__(self, dist, A_local, halo, n_local):
        self.dist, self.A, self.halo, self.n = dist, A_local, halo, n_local

    def exchange(self, V):
        import torch
        V = np.asarray(V)
        nc = 1 if V.ndim == 1 else V.shape[1]
        V2 = V.reshape(self.n, nc)
        peers, sp, rp, sidx = self.halo['peers'], self.halo['send_ptr'], self.halo['recv_ptr'], self.halo['send_idx']
        ghosts = np.zeros((int(rp[-1]), nc))
        reqs, bufs = [], []
        for q, peer in enumerate(peers):
            send = torch.from_numpy(np.ascontiguousarray(V2[sidx[sp[q]:sp[q + 1]]]))
            recv = torch.empty((int(rp[q + 1] - rp[q]), nc), dtype=torch.float64)
            if send.numel():
                reqs.append(self.dist.isend(send, int(peer)))
            if recv.numel():
                reqs.append(self.dist.irecv(recv, int(peer)))
            bufs.append((q, recv, send))
        for r in reqs:
            r.wait()
        for q, recv, _ in bufs:
            ghosts[rp[q]:rp[q + 1]] = recv.numpy()
        return np.concatenate([V2, ghosts], axis=0)

    def matvec_local(self, V):
        ext = self.exchange(V)
        out = np.stack([self.A @ ext[:, c] for c in range(ext.shape[1])], axis=1)
        return out[:, 0] if np.asarray(V).ndim == 1 else out


class OracleComm:
    def __init__(self, tc):
        self.tc = tc

    def Get_rank(self):
        return self.tc.Get_rank()

    def Get_size(self):
        return self.tc.Get_size()

    def Barrier(self):
        self.tc.Barrier()

    def allreduce_sum(self, a):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
        dist.all_reduce(t)
        return t.numpy()


def main(mode, workload, iters, outdir):
    import torch
    import torch.distributed as dist
    from new_cg_variants_amd import partition, problems, scaling
    from oracle import mp_oracle

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tc = scaling.TorchComm()
    wl = problems.WORKLOADS[workload]
    n = wl['n']
    A_full = wl['make']()
    offsets = partition.even_offsets(n, world)
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    A_rows = A_full[lo:hi]
    x_true = np.ones(n) / np.sqrt(n)
    b_full = A_full @ x_true
    b = b_full[lo:hi].copy()

    # single-rank oracle (every rank computes it: small problems)
    class Whole:
        def matvec_local(self, V):
            return A_full @ V
    results = {}
    for name in ('pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'):      # all five files of MP/cg_variants/
        x_ref, _ = getattr(mp_oracle, name)(mp_oracle.SingleRankComm(), Whole(), b_full.copy(), iters)
        if mode == 'cpu':
            A_local, ghost_ids = partition.localize(A_rows, lo, hi)
            halo = partition.plan_halo(ghost_ids, offsets, rank, tc.allgather_obj)
            op = PlanFollowingOperator(dist, A_local, halo, hi - lo)
            # the distributed SpMV alone must reproduce the global product bit for bit
            y = op.matvec_local(x_true[lo:hi] * (1.0 + np.arange(lo, hi)))
            want = (A_full @ (x_true * (1.0 + np.arange(n))))[lo:hi]
            assert np.array_equal(y, want), 'distributed SpMV differs from the global one'
            x, times = getattr(mp_oracle, name)(OracleComm(tc), op, b.copy(), iters)
        else:
            dev = 0 if os.environ.get('PRCG_TEST_SHARE_GPU') == '1' else int(os.environ.get('LOCAL_RANK', '0'))
            rop = scaling.RowBlockOperator(tc, A_rows, device=dev)
            y, _ = rop.dev.matvec(x_true[lo:hi] * (1.0 + np.arange(lo, hi)))
            want = (A_full @ (x_true * (1.0 + np.arange(n))))[lo:hi]
            assert np.array_equal(y, want), 'distributed device SpMV differs from the global one'
            x, times = getattr(scaling, name)(tc, rop, b.copy(), iters)
            rop.dev.close()
        assert (times is not None) == (rank == 0)
        if rank == 0:
            assert 'tot' in times and times['tot'] >= 0
        err = np.linalg.norm(x - x_ref[lo:hi]) / np.linalg.norm(x_ref[lo:hi])
        results[name] = float(err)
    np.save(os.path.join(outdir, f'result_{mode}_{rank}.npy'), results, allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
