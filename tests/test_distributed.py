"""The N>1 path.

CPU (gloo, world_size 2 and 3): partition + halo-plan exchange + the exchange contract of
prcg_set_halo, driven by the oracle's distributed loop (tests/dist_worker.py, mode 'cpu').

GPU: the same worker in mode 'gpu' goes through libprcg (RCCL halo + all-reduce).  A
one-GPU box can only host it if RCCL accepts two ranks on one device; when it refuses
("Duplicate GPU") the test reports that and falls back to the single-rank communicator
test, which still drives the complete two-stream schedule (events, side-stream reduction,
ncclAllReduce) through a 1-rank RCCL communicator.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def rccl_ids(n_ids):
    """n_ids fresh RCCL unique ids as one bytes object (rank-0 side of the bootstrap)."""
    from new_cg_variants_amd import _lib as L
    path = L.default_rccl_path()
    buf = np.zeros((n_ids, 128), dtype=np.uint8)
    for i in range(n_ids):
        L.check(None, L.lib().prcg_comm_unique_id(path.encode(), L.ptr(buf[i])))
    return buf.tobytes(), path


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(mode, world, workload, iters, tmp_path, extra_env=None, timeout=300):
    env = dict(os.environ)
    env['MASTER_ADDR'] = '127.0.0.1'
    env['OMP_NUM_THREADS'] = '1'
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    if extra_env:
        env.update(extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), mode, workload, str(iters), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return p


@pytest.mark.parametrize('world,workload', [(2, 's1_small'), (3, 's3_small')])
def test_row_block_halo_plan_over_gloo(tmp_path, world, workload):
    p = launch('cpu', world, workload, 10, tmp_path)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    for r in range(world):
        res = np.load(tmp_path / f'result_cpu_{r}.npy', allow_pickle=True).item()
        assert set(res) == {'pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'} and all(v <= 1e-11 for v in res.values()), res


def test_halo_plan_is_consistent_between_ranks():
    """send lists of rank a towards b == ghost list of b owned by a, for an irregular matrix."""
    from new_cg_variants_amd import partition, problems
    A = problems.irregular_standin(6000, mean_len=12, max_len=80, reach=900, seed=3)
    offsets, parts = partition.split_serial(A, 4, partition.nnz_balanced_offsets(A.indptr, 4))
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    y = A @ x
    for r, (A_local, ghosts, halo) in enumerate(parts):
        lo, hi = offsets[r], offsets[r + 1]
        ext = np.concatenate([x[lo:hi], x[ghosts]])
        assert np.array_equal(A_local @ ext, y[lo:hi])
        for q, peer in enumerate(halo['peers']):
            # what I send to `peer` must be exactly what `peer` expects from me, in order
            mine = halo['send_idx'][halo['send_ptr'][q]:halo['send_ptr'][q + 1]] + lo
            ph = parts[peer][2]
            qq = list(ph['peers']).index(r)
            theirs = parts[peer][1][ph['recv_ptr'][qq]:ph['recv_ptr'][qq + 1]]
            assert np.array_equal(mine, theirs)
    # balanced: no rank has more than 1.25x the mean nnz
    nnz = [p[0].nnz for p in parts]
    assert max(nnz) <= 1.25 * np.mean(nnz)


@pytest.mark.gpu
def test_single_rank_rccl_communicator_drives_full_schedule(matrices):
    """A 1-rank RCCL communicator switches the engine to its multi-rank schedule (compute
    + communication stream, events, side-stream reduction, ncclAllReduce).  Results must be
    bit-identical to the plain single-GPU path."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd.device import DeviceCSR
    A, z = matrices['nos7']
    n = A.shape[0]
    uid, path = rccl_ids(1)
    plain = DeviceCSR(A, knobs={'PRCG_FUSED': '0'})     # same two-kernel schedule as with a communicator
    comm = DeviceCSR(A, comm_init=(0, 1, uid, path), knobs={'PRCG_FUSED_COMM': '0'})
    for variant in (L.PIPE_PR, L.HS, L.PR, L.CG_CG, L.GV, L.PIPE_P):
        outs = []
        for op in (plain, comm):
            op.begin(variant, z['b'], np.zeros(n), 400, x_true=z['x_true'], hist_mask=15)
            assert not op.schedule()['fused']
            op.iterate(399)
            op.sync()
            outs.append((op.history(), op.get_vector('x')))
        for q in outs[0][0]:
            assert np.array_equal(outs[0][0][q], outs[1][0][q], equal_nan=True), (variant, q)
        assert np.array_equal(outs[0][1], outs[1][1])
    plain.close()
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize('workload,prec', [('s3_small', False), ('s3_small', True), ('s1_small', False)])
def test_one_launch_schedule_with_a_communicator(workload, prec):
    """With a communicator the pipelined variants of a window operator run ONE launch per iteration whose
    waves wait inside the launch for the inner products that the communication stream reduces meanwhile
    (pack -> ncclAllGather -> unpack + publish).  Driven here by a 1-rank communicator, with and without a
    loopback halo (boundary tiles + ghost rows through the all-gather).  Same arithmetic per element as the
    plain one-launch schedule; the inner products are summed in another order (per-launch partials reduced
    by k_gather_pack instead of by the next launch's prologue), so histories agree to rounding on the
    prefix and at convergence level beyond; no wave may have timed out."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    from oracle import ne_oracle as orc
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    inv_diag = (1 / A.diagonal()) if prec else None
    A_loop, halo, moved = loopback_problem(A, 9 if workload == 's3_small' else 70)
    uid, path = rccl_ids(1)
    uid2, _ = rccl_ids(1)
    plain = DeviceCSR(A)
    comm = DeviceCSR(A, comm_init=(0, 1, uid, path), knobs={'PRCG_FUSED_COMM': '1'})
    loop = DeviceCSR(A_loop, comm_init=(0, 1, uid2, path), halo=halo, knobs={'PRCG_FUSED_COMM': '1'})
    for variant in (L.PIPE_PR, L.PIPE_P_M):
        outs = []
        for op in (plain, comm, loop):
            op.begin(variant, b, x0, 300, x_true=x_true, inv_diag=inv_diag, hist_mask=15)
            s = op.schedule()
            assert s['fused'] and s['fused_comm'] == (op is not plain) and (op is plain or s['gather']), s
            op.iterate(299)
            op.sync()                                        # raises if a launch waited beyond its bound
            outs.append((op.history(), op.get_vector('x')))
        for other in outs[1:]:
            for q in outs[0][0]:
                np.testing.assert_allclose(other[0][q][:8], outs[0][0][q][:8], rtol=1e-11, atol=1e-13 * outs[0][0][q][0])
            # beyond the prefix the kappa = 1e6 problems amplify the different summation order like any other
            # (DESIGN.md section 2): the solves must still be the same solve at convergence level
            # (the paper's two statistics, figure_gen.py:86-89; a converged run may end in 0/0 = nan, as in the reference)
            ia, aa = orc.convergence_summary(other[0]['error_A_norm'])
            ib, ab = orc.convergence_summary(outs[0][0]['error_A_norm'])
            assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 2.0, ((ia, aa), (ib, ab))
    for op in (plain, comm, loop):
        op.close()


from new_cg_variants_amd.partition import loopback_problem  # noqa: E402  (one rank's view of a row-block run on one GPU)


@pytest.mark.gpu
@pytest.mark.parametrize('workload,k,n_ids,gather', [
    ('s1_small', 70, 2, '0'), ('s3_small', 9, 2, '0'), ('s3_small', 9, 1, '0'),
    ('s1_small', 70, 1, '1'), ('s3_small', 9, 1, '1'), ('s3_small', 300, 1, '1')])
def test_halo_path_on_one_gpu_through_rccl_loopback(workload, k, n_ids, gather):
    """pack kernel -> ncclSend/ncclRecv (to self) -> ghost slots -> interior/boundary tile
    split -> event choreography, all on one GPU.  Must be bit-identical to the plain path.
    gather='1': the pipelined variants move a small halo inside the one ncclAllGather that
    also carries the partial inner products (k=300 exceeds the slot limit: falls back to
    send/recv + all-reduce by itself)."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    A_loop, halo, moved = loopback_problem(A, k)
    assert moved > 0
    b, x0, x_true = problems.reference_rhs(A, n)
    uid, path = rccl_ids(n_ids)      # 2 ids: halo exchange on its own communicator + stream
    plain = DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    loop = DeviceCSR(A_loop, comm_init=(0, 1, uid, path), halo=halo, knobs={'PRCG_GATHER': gather, 'PRCG_FUSED_COMM': '0'})
    x = np.random.default_rng(2).standard_normal(n)
    y0, _ = plain.matvec(x)
    y1, _ = loop.matvec(x)
    assert np.array_equal(y0, A @ x) and np.array_equal(y1, y0)
    RS = np.stack([x, x[::-1]], axis=1)
    assert np.array_equal(loop.matmat2(RS)[0], plain.matmat2(RS)[0])
    for variant in (L.PIPE_PR, L.HS, L.PR, L.PIPE_P):
        outs = []
        pipelined = variant in (L.PIPE_PR, L.PIPE_P)
        # the variants whose inner products ride on the SpMV see a different tile order in
        # the loopback operator; on the kappa=1e6 problem that rounding-level difference is
        # amplified quickly, so they are compared over a short run
        iters = 120 if pipelined else 25
        for op in (plain, loop):
            op.begin(variant, b, x0, iters, x_true=x_true, hist_mask=15)
            if op is loop:
                sched = op.schedule()
                assert sched['comm'] and not sched['fused']
                assert sched['gather'] == (pipelined and gather == '1' and k < 300), sched
            op.iterate(iters - 1)
            op.sync()
            outs.append((op.history(), op.get_vector('x')))
        for q in outs[0][0]:
            if pipelined and q != 'error_A_norm':
                # every inner product comes from the fused update kernel, whose reduction
                # tree does not depend on the tiling: bit-identical
                assert np.array_equal(outs[0][0][q], outs[1][0][q], equal_nan=True), (variant, q)
            else:
                # inner products fused into the SpMV are summed per tile; the loopback
                # operator has a different (interior / boundary) tile order
                np.testing.assert_allclose(outs[1][0][q], outs[0][0][q], rtol=1e-7 if not pipelined else 1e-11)
        if pipelined:
            assert np.array_equal(outs[0][1], outs[1][1]), variant
        else:
            np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-7)
    plain.close()
    loop.close()


@pytest.mark.gpu
def test_two_ranks_through_rccl_when_the_box_allows_it(tmp_path):
    import torch
    ngpu = torch.cuda.device_count()
    share = ngpu < 2
    p = launch('gpu', 2, 's1_small', 10, tmp_path, {'PRCG_TEST_SHARE_GPU': '1' if share else '0'}, timeout=600)
    if p.returncode != 0 and share:
        tail = (p.stdout + p.stderr)[-3000:]
        pytest.skip('RCCL refused two ranks on one GPU on this box (expected on a 1-GPU box): ' + tail[-400:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    for r in range(2):
        res = np.load(tmp_path / f'result_gpu_{r}.npy', allow_pickle=True).item()
        assert set(res) == {'pipe_pr_cg', 'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg'} and all(v <= 1e-11 for v in res.values()), res


# ---------------------------------------------------------------------------------------
# N > 1 ranks on ONE GPU: ranks in threads, connected by tests/transport/libthreads_ccl.so (a stand-in
# for librccl.so that moves the collectives' bytes with device-to-device copies; RCCL itself refuses two
# ranks on one device).  Real inter-rank data: what rank r reads in its ghost rows was computed by the
# kernels of rank r +- 1.
# ---------------------------------------------------------------------------------------
def run_ranks_in_threads(A, nranks, variant, iters, knobs=None, inv_diag=None, hist_mask=15, offsets=None, peer=False, chunks=None):
    """Solve with `nranks` row blocks, one thread per rank, all on cuda:0.  Returns (x, histories of rank 0, schedules).
    peer: connect the ranks' exchange buffers (direct peer exchange; same process: device addresses instead of IPC handles).
    chunks: lengths of the prcg_iterate calls (default: one call)."""
    import threading
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems
    from new_cg_variants_amd.device import DeviceCSR
    path = os.path.join(ROOT, 'tests', 'transport', 'libthreads_ccl.so')
    assert os.path.exists(path), 'build it with `make` (tests/transport/threads_ccl.hip)'
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    offsets, parts = partition.split_serial(A, nranks, offsets)
    uid = np.zeros(128, dtype=np.uint8)
    L.check(None, L.lib().prcg_comm_unique_id(path.encode(), L.ptr(uid)))
    out = [None] * nranks
    errs = []

    def rank_main(r):
        try:
            A_local, ghost_ids, halo = parts[r]
            lo, hi = int(offsets[r]), int(offsets[r + 1])
            op = DeviceCSR(A_local, comm_init=(r, nranks, uid.tobytes(), path), halo=halo,
                           knobs=knobs[r] if isinstance(knobs, list) else knobs)
            if peer:
                assert partition.connect_peer_exchange(op, r, gather_objects), 'peer exchange did not connect'
            y = op.matvec(x_true[lo:hi] * (1.0 + np.arange(lo, hi)))[0]
            op.begin(variant, b[lo:hi], x0[lo:hi], iters + 1, x_true=x_true[lo:hi],
                     inv_diag=None if inv_diag is None else inv_diag[lo:hi], hist_mask=hist_mask)
            sched = op.schedule()
            for c in (chunks or [iters]):
                op.iterate(c)
            op.sync()
            out[r] = (op.get_vector('x'), op.history(), sched, y)
            op.close()
        except Exception as exc:           # noqa: BLE001 -- reported by the caller
            errs.append((r, repr(exc)))

    # all-gather of Python objects among the rank threads (what torch.distributed.all_gather_object is between processes)
    box = {}
    barrier = threading.Barrier(nranks)
    seq = [0] * nranks

    def gather_objects(obj):
        r = int(threading.current_thread().name.split('-')[-1])
        key = seq[r]
        seq[r] += 1
        box[(key, r)] = obj
        barrier.wait(timeout=120)
        res = [box[(key, q)] for q in range(nranks)]
        barrier.wait(timeout=120)
        return res

    threads = [threading.Thread(target=rank_main, args=(r,), name=f'rank-{r}', daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)          # (daemon threads: a rank that never returns fails the test instead of hanging pytest)
    assert not errs, errs
    assert all(o is not None for o in out), 'a rank did not finish'
    x = np.concatenate([o[0] for o in out])
    y = np.concatenate([o[3] for o in out])
    assert np.array_equal(y, A @ (x_true * (1.0 + np.arange(n)))), 'distributed SpMV differs from the global product'
    return x, out[0][1], [o[2] for o in out]


@pytest.mark.gpu
@pytest.mark.parametrize('workload,nranks,variant', [
    ('s3_small', 2, 'PIPE_PR'), ('s3_small', 4, 'PIPE_PR'), ('s3_small', 3, 'PIPE_P_M'), ('s1_small', 2, 'PIPE_PR'),
    ('s3_small', 2, 'HS'), ('s1_small', 3, 'PR'), ('s3_small', 2, 'CG_CG'), ('s1_small', 2, 'GV')])
def test_ranks_in_threads_two_kernel_schedule(workload, nranks, variant):
    """2-4 ranks with real halos, two-kernel schedule (merged all-gather for the band's 7-row halo, send/recv +
    all-reduce for the stencil's grid-line halo), all variant families.  The distributed SpMV is the global
    one bit for bit; the pipelined variants' inner products come from the update kernel, whose per-rank sums
    are added in rank order, so the whole solve agrees with the single-GPU two-kernel solve to rounding on the
    prefix and at convergence level beyond."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    from oracle import ne_oracle as orc
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    iters = 120
    x, hist, scheds = run_ranks_in_threads(A, nranks, getattr(L, variant), iters, knobs={'PRCG_FUSED_COMM': '0'})
    assert all(s['comm'] and not s['fused'] for s in scheds), scheds
    one = DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    one.begin(getattr(L, variant), b, x0, iters + 1, x_true=x_true, hist_mask=15)
    one.iterate(iters)
    one.sync()
    ref_hist, ref_x = one.history(), one.get_vector('x')
    one.close()
    for q in ref_hist:
        np.testing.assert_allclose(hist[q][:8], ref_hist[q][:8], rtol=1e-11, atol=1e-13 * ref_hist[q][0], err_msg=q)
    ia, aa = orc.convergence_summary(hist['error_A_norm'])
    ib, ab = orc.convergence_summary(ref_hist['error_A_norm'])
    assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 2.0, ((ia, aa), (ib, ab))


@pytest.mark.gpu
@pytest.mark.parametrize('workload,nranks,variant,prec', [
    ('s3_small', 2, 'PIPE_PR', False), ('s3_small', 3, 'PIPE_PR', True), ('s3_small', 2, 'PIPE_P', False)])
def test_ranks_in_threads_one_launch_schedule(workload, nranks, variant, prec):
    """The one-launch schedule with REAL peers: every rank's launch k+1 waits inside the kernel for a publication
    that needs the other ranks' launch k (pack -> all-gather -> unpack), and its boundary tiles read ghost rows the
    neighbours' kernels wrote.  The ranks share one GPU here, so each launches one workgroup per CU
    (PRCG_DEFER_GRID_PER_CU=1: all of them must be resident at once); no wave may time out."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    from oracle import ne_oracle as orc
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    inv_diag = (1 / A.diagonal()) if prec else None
    iters = 200
    x, hist, scheds = run_ranks_in_threads(A, nranks, getattr(L, variant), iters, knobs={'PRCG_DEFER_GRID_PER_CU': '1', 'PRCG_FUSED_COMM': '1'},
                                           inv_diag=inv_diag)
    assert all(s['fused_comm'] and s['gather'] and not s['peer'] for s in scheds), scheds
    one = DeviceCSR(A)
    one.begin(getattr(L, variant), b, x0, iters + 1, x_true=x_true, inv_diag=inv_diag, hist_mask=15)
    one.iterate(iters)
    one.sync()
    ref_hist = one.history()
    one.close()
    for q in ref_hist:
        np.testing.assert_allclose(hist[q][:8], ref_hist[q][:8], rtol=1e-11, atol=1e-13 * ref_hist[q][0], err_msg=q)
    ia, aa = orc.convergence_summary(hist['error_A_norm'])
    ib, ab = orc.convergence_summary(ref_hist['error_A_norm'])
    assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 2.0, ((ia, aa), (ib, ab))


@pytest.mark.gpu
def test_s3_full_size_row_blocks_in_threads():
    """north_star's multi-GPU workload (S3, n = 1e7, ~150 M nonzeros) cut into row blocks whose ranks run in threads
    on ONE GPU: 8 blocks through the two-kernel schedule, 2 blocks through the one-launch schedule (more would not
    all be resident on one GPU).  Residual histories against the single-GPU run."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.WORKLOADS['s3']['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    iters = 30
    one = DeviceCSR(A)
    one.begin(L.PIPE_PR, b, x0, iters + 1, hist_mask=1)
    one.iterate(iters)
    one.sync()
    ref = one.history()['updated_residual_2_norm']
    one.close()
    for nranks, knobs, flag in ((8, {'PRCG_FUSED_COMM': '0'}, False), (2, {'PRCG_DEFER_GRID_PER_CU': '1', 'PRCG_FUSED_COMM': '1'}, True)):
        x, hist, scheds = run_ranks_in_threads(A, nranks, L.PIPE_PR, iters, knobs=knobs, hist_mask=1)
        assert all(s['fused_comm'] == flag and s['gather'] and s['window'] for s in scheds), scheds
        np.testing.assert_allclose(hist['updated_residual_2_norm'][:10], ref[:10], rtol=1e-11)
        np.testing.assert_allclose(hist['updated_residual_2_norm'], ref, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('workload,nranks,variant', [('s2', 4, 'PIPE_PR'), ('s2', 2, 'HS'), ('s4b_80', 3, 'PIPE_PR')])
def test_full_size_stencil_and_fem_row_blocks_in_threads(workload, nranks, variant):
    """The other two multi-GPU configurations of BASELINE.json at full size, cut into row blocks whose ranks run in
    threads on ONE GPU: S2 (7-point Laplacian 216^3; blocks of whole grid planes, halo = one 216^2 plane per side =
    746 KB of (r,s) pairs -> send/recv + all-reduce) and the FEM-like stand-in for Queen_4147 (nnz-balanced split
    points, gather lists per peer).  The distributed SpMV must be the global product bit for bit (checked inside
    run_ranks_in_threads); residual histories against the single-GPU run of the same schedule family."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    iters = 30
    var = getattr(L, variant)
    if workload == 's2':
        planes = 216 // nranks
        offsets = np.arange(nranks + 1, dtype=np.int64) * planes * 216 * 216
    else:
        offsets = partition.nnz_balanced_offsets(A.indptr, nranks)
        nnz = np.diff(A.indptr[offsets])
        assert nnz.max() <= 1.02 * nnz.mean(), nnz
    one = DeviceCSR(A, knobs={'PRCG_FUSED': '0'})
    one.begin(var, b, x0, iters + 1, hist_mask=1)
    one.iterate(iters)
    one.sync()
    ref = one.history()['updated_residual_2_norm']
    one.close()
    x, hist, scheds = run_ranks_in_threads(A, nranks, var, iters, hist_mask=1, offsets=offsets)
    assert all(s['comm'] and not s['fused_comm'] for s in scheds), scheds
    np.testing.assert_allclose(hist['updated_residual_2_norm'][:10], ref[:10], rtol=1e-11)
    np.testing.assert_allclose(hist['updated_residual_2_norm'], ref, rtol=1e-4)


@pytest.mark.gpu
def test_ranks_in_threads_may_run_different_schedules():
    """A rank whose block is no window operator, or whose stream-concurrency probe fails, keeps the two-kernel schedule
    while its peers run one launch per iteration: both issue ONE all-gather per iteration carrying the same things
    (partial inner products of iteration k, rows of (r,s)_k), so a mixed session is the same solve."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.WORKLOADS['s3_small']['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    iters = 150
    knobs = [{'PRCG_DEFER_GRID_PER_CU': '1', 'PRCG_FUSED_COMM': '1'}, {'PRCG_FUSED_COMM': '0'}, {'PRCG_DEFER_GRID_PER_CU': '1', 'PRCG_FUSED_COMM': '1', 'PRCG_WIN': '0'}]
    x, hist, scheds = run_ranks_in_threads(A, 3, L.PIPE_PR, iters, knobs=knobs)
    assert [s['fused_comm'] for s in scheds] == [True, False, False] and all(s['gather'] for s in scheds), scheds
    one = DeviceCSR(A)
    one.begin(L.PIPE_PR, b, x0, iters + 1, x_true=x_true, hist_mask=15)
    one.iterate(iters)
    one.sync()
    ref_hist = one.history()
    one.close()
    for q in ref_hist:
        np.testing.assert_allclose(hist[q][:8], ref_hist[q][:8], rtol=1e-11, atol=1e-13 * ref_hist[q][0], err_msg=q)
    np.testing.assert_allclose(hist['error_A_norm'][-1], ref_hist['error_A_norm'][-1], rtol=0.5)


# ---------------------------------------------------------------------------------------
# Direct peer exchange (include/prcg.h: prcg_peer_setup / prcg_peer_connect): the one-launch schedule whose reduction
# and halo are stores into the consumers' exchange buffers -- no collective inside the loop.
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('workload,prec', [('s3_small', False), ('s3_small', True), ('s1_small', False)])
def test_peer_exchange_with_one_rank_and_a_loopback_halo(workload, prec):
    """One rank whose only peer is itself: the launch's tiles store the "neighbour's" rows into the rank's own exchange
    buffer, its last workgroup the rank's slot, workgroup 0 of the next launch publishes it; boundary tiles read their
    ghost pages from the exchange buffer.  Against the plain one-launch schedule: same arithmetic per element, the
    inner products summed in another order (last workgroup's tree) -- prefix to rounding, convergence level beyond;
    no wave may have timed out; iterate calls of any length."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems
    from new_cg_variants_amd.device import DeviceCSR
    from oracle import ne_oracle as orc
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    inv_diag = (1 / A.diagonal()) if prec else None
    A_loop, halo, moved = partition.loopback_problem(A, 9 if workload == 's3_small' else 70)
    uid, path = rccl_ids(1)
    uid2, _ = rccl_ids(1)
    plain = DeviceCSR(A)
    solo = DeviceCSR(A, comm_init=(0, 1, uid, path))
    loop = DeviceCSR(A_loop, comm_init=(0, 1, uid2, path), halo=halo)
    for op in (solo, loop):
        assert partition.connect_peer_exchange(op, 0, lambda obj: [obj])
    for variant in (L.PIPE_PR, L.PIPE_P_M):
        outs = []
        for op in (plain, solo, loop):
            op.begin(variant, b, x0, 300, x_true=x_true, inv_diag=inv_diag, hist_mask=15)
            s = op.schedule()
            assert s['fused'] and s['fused_comm'] == (op is not plain) and s['peer'] == (op is not plain) and not s['gather'], s
            for c in (1, 2, 50, 246):
                op.iterate(c)
            op.sync()                                        # raises if a launch waited beyond its bound
            outs.append((op.history(), op.get_vector('x')))
        for other in outs[1:]:
            for q in outs[0][0]:
                np.testing.assert_allclose(other[0][q][:8], outs[0][0][q][:8], rtol=1e-11, atol=1e-13 * outs[0][0][q][0])
            ia, aa = orc.convergence_summary(other[0]['error_A_norm'])
            ib, ab = orc.convergence_summary(outs[0][0]['error_A_norm'])
            assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 2.0, ((ia, aa), (ib, ab))
    # teacher forcing reaches the exchange buffers: load the plain engine's state k into the loopback engine, step both
    plain.begin(L.PIPE_PR, b, x0, 40, inv_diag=inv_diag)
    loop.begin(L.PIPE_PR, b, x0, 40, inv_diag=inv_diag)
    plain.iterate(7)
    stored = ['x', 'r', 'p', 's'] + (['rt', 'st'] if prec else [])
    for k in (7, 8, 9):
        for v in stored:
            loop.set_vector(v, plain.get_vector(v))
        loop.set_scalars(k, plain.get_scalars(k))
        loop.set_iteration(k)
        plain.iterate(1)
        loop.iterate(1)
        for v in stored:
            assert np.array_equal(loop.get_vector(v), plain.get_vector(v)), (k, v)
        a, c = loop.get_scalars(k + 1)[:5], plain.get_scalars(k + 1)[:5]
        assert np.max(np.abs(a - c) / np.abs(c)) <= 1e-12
    for op in (plain, solo, loop):
        op.close()


@pytest.mark.gpu
@pytest.mark.parametrize('workload,nranks,variant,prec', [
    ('s3_small', 2, 'PIPE_PR', False), ('s3_small', 3, 'PIPE_PR', True), ('s3_small', 4, 'PIPE_P', False),
    ('s1_small', 2, 'PIPE_PR', False), ('s1_small', 3, 'PIPE_PR_M', True)])
def test_ranks_in_threads_peer_exchange(workload, nranks, variant, prec):
    """2-4 REAL ranks (threads on one GPU) through the direct peer exchange, small halos (band: 7 rows per side) and
    large ones (stencil: a grid line): every rank's launch k+1 waits inside the kernel for the other ranks' launch k,
    reads ghost rows the neighbours' tiles stored into its exchange buffer; no collective in the loop.  The ranks share
    one GPU, so each launches one workgroup per CU (all of them must be resident at once)."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    from oracle import ne_oracle as orc
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    inv_diag = (1 / A.diagonal()) if prec else None
    iters = 200
    x, hist, scheds = run_ranks_in_threads(A, nranks, getattr(L, variant), iters, knobs={'PRCG_DEFER_GRID_PER_CU': '1'},
                                           inv_diag=inv_diag, peer=True, chunks=[1, 3, 96, 100])
    assert all(s['fused_comm'] and s['peer'] and not s['gather'] for s in scheds), scheds
    one = DeviceCSR(A)
    one.begin(getattr(L, variant), b, x0, iters + 1, x_true=x_true, inv_diag=inv_diag, hist_mask=15)
    one.iterate(iters)
    one.sync()
    ref_hist, ref_x = one.history(), one.get_vector('x')
    one.close()
    for q in ref_hist:
        np.testing.assert_allclose(hist[q][:8], ref_hist[q][:8], rtol=1e-11, atol=1e-13 * ref_hist[q][0], err_msg=q)
    ia, aa = orc.convergence_summary(hist['error_A_norm'])
    ib, ab = orc.convergence_summary(ref_hist['error_A_norm'])
    assert abs(ia - ib) <= max(2, 0.05 * ib) and abs(aa - ab) <= 2.0, ((ia, aa), (ib, ab))


@pytest.mark.gpu
@pytest.mark.parametrize('workload,nranks', [('s2', 2), ('s1', 3), ('s3', 2)])
def test_full_size_row_blocks_in_threads_peer_exchange(workload, nranks):
    """BASELINE.json's multi-GPU configurations at full size through the direct peer exchange, ranks in threads on ONE
    GPU: S2 (config 4: 7-point Laplacian 216^3 in 2 blocks of whole planes, halo = one 216^2 plane = 746 KB of pairs
    -- the one-launch schedule serves halos of any size) and S3 (n = 1e7 in 2 blocks).  Two ranks: the persistent
    launches of ALL ranks must be resident on the one GPU at once (250 VGPRs each for S2's geometry)."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    b, x0, x_true = problems.reference_rhs(A, n)
    iters = 30
    offsets = None
    if workload == 's2':
        offsets = np.arange(nranks + 1, dtype=np.int64) * (216 // nranks) * 216 * 216
    if workload == 's1':       # whole grid lines per rank: halo = one line of 1000 rows per side
        offsets = np.array([0, 333, 667, 1000], dtype=np.int64) * 1000
    one = DeviceCSR(A)
    one.begin(L.PIPE_PR, b, x0, iters + 1, hist_mask=1)
    one.iterate(iters)
    one.sync()
    ref = one.history()['updated_residual_2_norm']
    one.close()
    x, hist, scheds = run_ranks_in_threads(A, nranks, L.PIPE_PR, iters, knobs={'PRCG_DEFER_GRID_PER_CU': '1'}, hist_mask=1,
                                           offsets=offsets, peer=True)
    assert all(s['fused_comm'] and s['peer'] and s['window'] for s in scheds), scheds
    np.testing.assert_allclose(hist['updated_residual_2_norm'][:10], ref[:10], rtol=1e-11)
    np.testing.assert_allclose(hist['updated_residual_2_norm'], ref, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('nranks', [2, 3])
def test_peer_exchange_between_processes(nranks):
    """The peer exchange's plumbing between PROCESSES sharing the one GPU (what RCCL refuses): every rank allocates its
    exchange buffer, the 64-byte hipIpc handles travel over the control plane, every rank maps the others' buffers and
    plans its sends from the receivers' halo plans; then six rounds of the exchange primitives -- rows stored straight
    into the neighbours' ghost areas, every rank's slot into every buffer, counters last -- and every rank checks the
    sum of the slots (rank order, bit for bit) and the ghost rows it received against the closed form."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import peer_ipc_worker
    ctx = mp.get_context('spawn')
    pipes = [ctx.Pipe() for _ in range(nranks)]
    procs = [ctx.Process(target=peer_ipc_worker.main, args=(r, nranks, pipes[r][1]), daemon=True) for r in range(nranks)]
    for p in procs:
        p.start()
    conns = [pp[0] for pp in pipes]
    done = {}
    try:
        while len(done) < nranks:
            msgs = {}
            for r, c in enumerate(conns):
                if r in done:
                    continue
                assert c.poll(180), f'rank {r} is silent'
                kind, obj = c.recv()
                if kind == 'done':
                    done[r] = obj
                else:
                    msgs[r] = obj
            if msgs:
                assert len(msgs) + len(done) == nranks and not done, (list(msgs), done)   # lockstep: all gather or none
                everyone = [msgs[r] for r in range(nranks)]
                for c in conns:
                    c.send(everyone)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert len({d['pid'] for d in done.values()}) == nranks
    for r in range(nranks):
        assert done[r]['connected'], done[r]
        assert not done[r]['bad'], done[r]['bad']


class ThreadComm:
    """The communicator-like object of new_cg_variants_amd.scaling for ranks that are threads of this process."""

    def __init__(self, rank, size, shared):
        self.rank, self.size, self.shared = rank, size, shared
        self.seq = 0

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def allgather_obj(self, obj):
        box, barrier = self.shared
        key = self.seq
        self.seq += 1
        box[(key, self.rank)] = obj
        barrier.wait(timeout=180)
        res = [box[(key, q)] for q in range(self.size)]
        barrier.wait(timeout=180)
        return res

    def Barrier(self):
        self.allgather_obj(None)

    def bcast_obj(self, obj, root=0):
        return self.allgather_obj(obj)[root]


@pytest.mark.gpu
def test_row_block_operator_self_check_and_drop_in_call_in_threads():
    """The host side of an N > 1 run exactly as bench.py and the scaling drop-ins drive it -- RowBlockOperator (row block with
    global column ids -> local numbering, halo plan, communicator, peer exchange connected over the control plane),
    scaling.one_launch_self_check (first iterations of the one-launch schedule against the RCCL two-kernel schedule) and
    `sol, t = pipe_pr_cg(comm, A, b, max_iter)` -- with three ranks in threads on the one GPU."""
    import threading
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems, scaling
    from new_cg_variants_amd.device import DeviceCSR
    path = os.path.join(ROOT, 'tests', 'transport', 'libthreads_ccl.so')
    A = problems.WORKLOADS['s3_small']['make']()
    n = A.shape[0]
    nranks = 3
    offsets = partition.even_offsets(n, nranks)
    x_true = np.ones(n) / np.sqrt(n)
    b = A @ x_true
    shared = ({}, threading.Barrier(nranks))
    out, errs = [None] * nranks, []

    def rank_main(r):
        try:
            lo, hi = int(offsets[r]), int(offsets[r + 1])
            comm = ThreadComm(r, nranks, shared)
            op = scaling.RowBlockOperator(comm, A[lo:hi], device=0, rccl_path=path, knobs={'PRCG_DEFER_GRID_PER_CU': '1'})
            assert op.peer
            verdict = scaling.one_launch_self_check(op, L.PIPE_PR, b[lo:hi], np.zeros(hi - lo))
            sol, t = scaling.pipe_pr_cg(comm, op, b[lo:hi], 150)
            out[r] = (verdict, sol, t, op.dev.schedule())
            op.dev.close()
        except Exception as exc:       # noqa: BLE001
            import traceback
            errs.append((r, repr(exc), traceback.format_exc()[-1200:]))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not errs, errs
    assert all(o is not None for o in out), 'a rank did not finish'
    assert all(o[0] is None for o in out), [o[0] for o in out]
    assert all(o[3]['peer'] and o[3]['fused_comm'] for o in out), [o[3] for o in out]
    assert out[0][2]['tot'] > 0 and out[1][2] is None
    x = np.concatenate([o[1] for o in out])
    one = DeviceCSR(A)
    one.begin(L.PIPE_PR, b, np.zeros(n), 152)
    one.iterate(150)
    ref = one.get_vector('x')
    one.close()
    # (150 free-running iterations on the kappa = 1e6 band: the two runs sum their inner products in different orders --
    #  workgroup shapes, rank order -- and the iterates drift apart like the paper's curves do: 0.3-0.7 % seen)
    assert np.linalg.norm(x - ref) <= 2e-2 * np.linalg.norm(ref)
    assert abs(np.linalg.norm(x - x_true) / np.linalg.norm(ref - x_true) - 1.0) <= 0.2


@pytest.mark.gpu
def test_sweep_table_operator_in_a_communicator_session():
    """A sweep-table operator (pattern tiles in plane-sweep order, csrc/prcg_plan.h: plan_sweep_tiles) run by the DEFERRED
    multi-rank kernel: its waves are not the ones the carry bits assume (one wave is the communication wave), so every page is
    loaded -- through the tiles' slot tables all the same.  Teacher-forced steps against the plain schedule on the same
    table: vectors bit for bit."""
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems
    from new_cg_variants_amd.device import DeviceCSR
    A = problems.laplace_3d(128, 64, 64)
    n = A.shape[0]
    b, x0, _ = problems.reference_rhs(A, n)
    knobs = {'PRCG_WIN_SWEEP': '2', 'PRCG_SWEEP_WAVES': '2048'}
    uid, path = rccl_ids(1)
    plain = DeviceCSR(A, knobs=knobs)
    solo = DeviceCSR(A, comm_init=(0, 1, uid, path), knobs=knobs)
    assert partition.connect_peer_exchange(solo, 0, lambda obj: [obj])
    assert plain.layout()['sweep_waves'] > 0 and solo.layout()['sweep_waves'] > 0
    plain.begin(L.PIPE_PR, b, x0, 40)
    solo.begin(L.PIPE_PR, b, x0, 40)
    s = solo.schedule()
    assert s['pattern'] and s['peer'] and s['fused_comm'], s
    plain.iterate(5)
    for k in (5, 6, 7, 8):
        for v in ('x', 'r', 'p', 's'):
            solo.set_vector(v, plain.get_vector(v))
        solo.set_scalars(k, plain.get_scalars(k))
        solo.set_iteration(k)
        plain.iterate(1)
        solo.iterate(1)
        solo.sync()
        for v in ('x', 'r', 'p', 's'):
            assert np.array_equal(solo.get_vector(v), plain.get_vector(v)), (k, v)
        a, c = solo.get_scalars(k + 1)[:5], plain.get_scalars(k + 1)[:5]
        assert np.max(np.abs(a - c) / np.abs(c)) <= 1e-12
    plain.close(); solo.close()


@pytest.mark.gpu
@pytest.mark.parametrize('world,workload', [(2, 's3_8th'), (3, 's2_8th'), (4, 's3')])
def test_bench_launched_as_the_driver_does_with_processes_sharing_the_gpu(world, workload):
    """The driver's N > 1 launch, rehearsed on ONE GPU: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`,
    one PROCESS per rank, gloo control plane, every rank on device 0 (PRCG_BENCH_DEVICE) and the collectives of the set-up
    through tests/transport/libprocs_ccl.so (PRCG_RCCL_LIB; RCCL refuses ranks that share a device).  What runs is the
    product's N > 1 path end to end: row blocks, halo plan over gloo, exchange buffers mapped ACROSS PROCESSES with hipIpc,
    the self-check of the one-launch schedule against the two-kernel schedule, the timed loop in which every rank's
    launches wait in-kernel for the other processes' launches, max-over-ranks timing, the second leg on the RCCL side-stream
    schedule, one JSON line from rank 0."""
    import json
    env = dict(os.environ)
    env.update({'MASTER_ADDR': '127.0.0.1', 'OMP_NUM_THREADS': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'PRCG_BENCH_DEVICE': '0',
                'PRCG_RCCL_LIB': os.path.join(ROOT, 'tests', 'transport', 'libprocs_ccl.so'),
                # ranks that SHARE a GPU: one workgroup per CU each, or a rank's persistent launch keeps the others' out
                'PRCG_DEFER_GRID_PER_CU': '1'})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '60', '--warmup', '10',
           '--workload', workload, '--no-cpu-baseline']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    s = d['config']['schedule']
    print(f"{world} processes on one GPU, {workload}: {d['value']:.0f} it/s, schedule {s}, fallback {d['config']['schedule_fallback']}")
    assert d['n_gpus'] == world and d['steps'] == 60 and d['scaling'] == 'strong'
    assert d['config']['schedule_fallback'] is None, d['config']['schedule_fallback']
    assert s['peer'] and s['fused_comm'] and s['window'] and d['config']['residual_finite'], s
    assert d['value'] > 1        # (processes time-share the GPU while their launches wait for each other: no performance figure)
    # the second, short leg of an N > 1 run: the same loop on the RCCL side-stream schedule (the paper's contrast from one run)
    leg = d['roofline'].get('rccl_schedule')
    assert leg and leg.get('rccl_ranks') == world and not leg.get('error') and leg['value'] > 1 and leg['residual_finite'], leg
    assert d['config']['rccl_ranks'] == world
    print(f"   RCCL side-stream schedule, same run: {leg['value']:.0f} it/s (merged exchange {leg['merged_exchange']})")


@pytest.mark.gpu
def test_scaling_tests_driver_with_processes_sharing_the_gpu(tmp_path):
    """The mpi4py experiment's drop-in driver (experiments/scaling_tests.py; scaling_experiments_mpi4py/scaling_tests.py) launched
    with one PROCESS per rank, two ranks on one GPU through the process stand-in for RCCL: the five variants print their error
    lines and save the reference's result files; the errors agree with a single-process run of the same iteration count to the
    reordering of the inner products."""
    env = dict(os.environ)
    env.update({'MASTER_ADDR': '127.0.0.1', 'OMP_NUM_THREADS': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'PRCG_BENCH_DEVICE': '0',
                'PRCG_RCCL_LIB': os.path.join(ROOT, 'tests', 'transport', 'libprocs_ccl.so'), 'PRCG_DEFER_GRID_PER_CU': '1',
                'PYTHONPATH': ROOT + os.pathsep + env.get('PYTHONPATH', '')})
    out = {}
    for world in (1, 2):
        cwd = tmp_path / f'w{world}'
        cwd.mkdir()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), '-m', 'new_cg_variants_amd.experiments.scaling_tests', '12288', '400', f'procs{world}']
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=str(cwd))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        errs = {ln.split(' error: ')[0]: float(ln.split(' error: ')[1]) for ln in p.stdout.splitlines() if ' error: ' in ln}
        assert set(errs) == {'hs_cg', 'cg_cg', 'gv_cg', 'pr_cg', 'pipe_pr_cg'}, p.stdout[-1500:]
        for v in errs:
            res = np.load(cwd / 'data' / '12288' / f'{v}_procs{world}.npy', allow_pickle=True).item()
            assert res['error'] == errs[v] and res['timings']['tot'] > 0
        out[world] = errs
    print(out)
    for v in out[1]:
        assert np.isfinite(out[2][v]) and 0.2 <= out[2][v] / out[1][v] <= 5.0, (v, out[1][v], out[2][v])
