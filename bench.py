#!/usr/bin/env python3
"""Headline benchmark: pipelined predict-and-recompute CG iterations/second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s3|s2|s1|...] [--variant pipe_pr_cg|hs_cg|...]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU.  torch.distributed (gloo) carries only the control plane (RCCL unique
id, barriers, max-over-ranks of the time); every per-iteration byte -- the halo of the
SpMM input and the single reduction -- moves over xGMI inside libprcg.so.

A "step" is one CG iteration (fused update + two-vector SpMM + reduction) on the
synthetic workload, inputs resident in HBM before the clock starts, no convergence test
(as the reference's PETSc runs: -ksp_norm_type none, strong_scaling_tests.py:70).
The workload is fixed as N grows (strong scaling): S3 = the reference's ex2b banded
model matrix at n = 1e7, 15 diagonals, ~150 M nonzeros, split into row blocks.

Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md section 6).  At N = 1 the
line also carries (all timed in this run, same K and W):
  value_general_csr            the same loop with the value dictionary off: the rate of an operator whose values
                               do not repeat (8 B per nonzero actually move) -- the transferable number
  roofline.plain_values        ... its launch time, bytes and fractions
  roofline.multi_rank_schedule the schedule every rank of an N>1 run executes, on this one GPU: whole S3 with a
                               1-rank communicator, and one rank's share of an 8-GPU run (S3/8 with a loopback halo)
  workloads                    BASELINE.json's other configurations: S1 (config 3), S2 at N=1 (config 4), the two
                               stand-ins for Queen_4147 (config 5), each with its plain-values twin where a dictionary applies
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def spmv_bytes(n, nnz):     # B1: val + col, row_ptr, x read once, y written   
    return 12 * nnz + 4 * (n + 1) + 16 * n


def spmm2_bytes(n, nnz):    # B2: A once, two input and two output vectors
    return 12 * nnz + 4 * (n + 1) + 32 * n


def fused_bytes(n, nnz):    # one-launch iteration: A once, (r,s) read + written, (x,p) read + written
    return 12 * nnz + 4 * (n + 1) + 64 * n


def moved_bytes(algorithmic, n, nnz, operator_bytes):
    """Bytes the launch MUST move: SURVEY 8d's figure with the caller's CSR arrays (12 B per nonzero, 4 B per
    row pointer) replaced by the operator as the device streams it (prcg_operator_bytes: lossless re-encodings --
    1/2-byte window indices, 1-byte value-dictionary indices, 2-byte relative row pointers, dictionaries and tile
    descriptors, stream images that many tiles share counted once)."""
    return algorithmic - 12 * nnz - 4 * (n + 1) + operator_bytes


def kernel_source_sha():
    """Fingerprint of the kernel sources a PMC traffic figure was measured with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'new_cg_variants_amd', 'csrc', '*'))):
        if f.endswith(('.hip', '.hpp', '.h', '.cpp')):
            h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """HBM bytes per launch from the PMC counters (tools/profile_round.sh -> profiles/traffic.json),
    or None when the kernels have changed since they were collected."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        e = t['entries'].get(key) if t.get('kernel_sha') == kernel_source_sha() else None
        return e['bytes_per_launch'] if isinstance(e, dict) else e
    except Exception:
        return None


def cpu_baseline(A, b, x0, family, seconds=15.0):
    """The CPU line: the NumPy/SciPy restatement of the reference's loop (oracle/, pinned
    bitwise against the imported reference in the build container) timed on this host."""
    from oracle import ne_oracle as orc
    start, advance, has_flavour = orc.FAMILIES[family]
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info() if p.get('user_api') == 'blas'] or [1])
    except Exception:
        threads = 1
    st = start(A, b, x0)
    step = (lambda: advance(A, st, 'pr')) if has_flavour else (lambda: advance(A, st))
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    iters = int(max(3, min(200, seconds / max(one, 1e-6))))
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        for _ in range(iters):
            step()
    dt = time.perf_counter() - t0
    return {'value': iters / dt, 'unit': 'iters/s', 'cores': int(threads), 'kind': 'port',
            'sample': f'{iters} iterations of the oracle {family} loop (scipy csr_matvec is single-threaded; '
                      f'BLAS ddot may use up to {threads} threads) on the same matrix, {os.cpu_count()} host cores visible'}


PREWARM = 300      # untimed iterations before any clock starts: the chip's clocks are up, whatever --warmup says
SAMPLE_AFTER = 200  # launches AFTER the timed region with HIP events around every second one: a larger sample of the launch duration
VARIANTS = ['pipe_pr_cg', 'hs_cg', 'pr_cg', 'pipe_pr_pcg', 'pipe_p_cg', 'cg_cg', 'gv_cg']
FAMILY = {'pipe_pr_cg': 'pipe', 'hs_cg': 'hs', 'pr_cg': 'pr', 'pipe_pr_pcg': 'pipe', 'pipe_p_cg': 'pipe', 'cg_cg': 'cg_cg', 'gv_cg': 'gv'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='s3',
                    help='s1 s2 s3 s4 s4b s3_8th s1_small s3_small; queen if QUEEN_4147_MTX names the MatrixMarket file')
    ap.add_argument('--variant', default='pipe_pr_cg', choices=VARIANTS,
                    help='pipe_pr_pcg = the pipelined variant with the Jacobi preconditioner (figure_gen.py:42-44)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-comm', action='store_true',
                    help='N=1 only: the MAIN run uses a 1-rank RCCL communicator (multi-rank schedule)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-plain-values', action='store_true',
                    help='skip the second timed run with the value dictionary off')
    ap.add_argument('--no-multi-rank-leg', action='store_true',
                    help='skip the extra N=1 runs of the multi-rank schedule (1-rank communicator, S3/8 slice)')
    ap.add_argument('--no-rccl-leg', action='store_true',
                    help='N>1 only: skip the second, short timed leg on the RCCL side-stream schedule')
    ap.add_argument('--no-workloads', action='store_true',
                    help="skip the extra N=1 runs of BASELINE.json's other configurations (S1, S2, s4b, s4)")
    args = ap.parse_args()

    # stdout is a protocol here (exactly one JSON line on rank 0): libraries that chat on
    # fd 1 (Gloo's "Rank 0 is connected ...", RCCL's version banner) are sent to stderr
    # for the whole run; the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('PRCG_BENCH_DEVICE'):
        # rehearsal of an N > 1 launch on a box with fewer GPUs than ranks (tests/test_distributed.py, together with
        # PRCG_RCCL_LIB = a collectives stand-in that accepts ranks sharing a device): every rank on the named device
        local_rank = int(os.environ['PRCG_BENCH_DEVICE'])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world

    import torch                                   # first: its HIP runtime is the process's runtime
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)          # torch.cuda.synchronize() below must mean THIS rank's GPU
    from new_cg_variants_amd import _lib as L
    from new_cg_variants_amd import partition, problems, scaling
    from new_cg_variants_amd.device import DeviceCSR

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        comm = scaling.TorchComm()
    else:
        comm = scaling.SelfComm()

    K, W = args.steps, args.warmup
    K_MAIN = K
    # the extra legs (short launches: one eighth of S3 / S2, S1, the multi-rank schedule) are timed over at least 200 steps:
    # 20 steps of a 25 us launch are 0.5 ms, of which the barriers and the event records around them are several per cent
    K_LEG = max(K, 200)
    VAR = {'pipe_pr_cg': L.PIPE_PR, 'hs_cg': L.HS, 'pr_cg': L.PR, 'pipe_pr_pcg': L.PIPE_PR, 'pipe_p_cg': L.PIPE_P,
           'cg_cg': L.CG_CG, 'gv_cg': L.GV}

    def one_rank_comm_device(A, halo=None, knobs=None):
        """a 1-rank communicator session: the multi-rank schedule on this one GPU (its only peer is itself)"""
        uid = np.zeros((1, 128), dtype=np.uint8)
        path = L.default_rccl_path()
        L.check(None, L.lib().prcg_comm_unique_id(path.encode(), L.ptr(uid[0])))
        d = DeviceCSR(A, device=local_rank, comm_init=(0, 1, uid.tobytes(), path), halo=halo, knobs=knobs)
        if (knobs or {}).get('PRCG_PEER', os.environ.get('PRCG_PEER', '1')) != '0':
            partition.connect_peer_exchange(d, 0, lambda obj: [obj])
        return d

    def timed_run(dev, variant, b, x0, inv_diag=None, K=None, sample_after=0):
        """(elapsed, host enqueue time, kernel timings, residual finite, error).  A library error (a one-launch
        iteration of a communicator session that waited longer than its bound for the reduction) is RETURNED, not
        raised: every rank must reach the barriers below, and the decision what to do next is taken collectively.
        The PREWARM iterations bring the chip's clocks up; then the session is begun AFRESH (x0 again), so that the W
        warm-up and the K timed iterations run on the first W + K iterates of the solve -- a finite state whatever the
        operator's conditioning (round 3 timed one stand-in 500 iterations in, after CG had converged to an exact zero).
        sample_after > 0: that many further launches after the timed region with HIP events around every second one
        (timings()['sampled_after_ms']): a larger sample of the launch duration than the timed region allows itself."""
        K = K_MAIN if K is None else K
        err = None
        try:
            dev.begin(variant, b, x0, PREWARM + 1, inv_diag=inv_diag)
            dev.iterate(PREWARM)
            dev.sync()
            dev.begin(variant, b, x0, W + K + sample_after + 1, inv_diag=inv_diag)
            dev.iterate(W)
            dev.sync()
        except L.PrcgError as exc:
            err = str(exc)
        # HIP events bracket every 4th launch at most (two marker packets cost ~7 us of queue time around a 150 us
        # launch: bracketing every launch of a 20-step run made it 7 % slower than a 500-step one)
        dev.set_profiling(max(4, K // 100))
        comm.Barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_enq = 0.0
        if err is None:
            try:
                dev.iterate(K)
                t_enq = time.perf_counter() - t0      # host time to enqueue K iterations (no sync inside)
                dev.sync()
            except L.PrcgError as exc:
                err = str(exc)
        torch.cuda.synchronize()
        comm.Barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            err = next((e for e in comm.allgather_obj(err) if e), None)
        if err is not None:
            return elapsed, t_enq, None, False, err
        finite = bool(np.isfinite(dev.get_scalars(W + K)[L.S_NU]))
        tim = dev.timings()
        if sample_after > 0:
            try:
                dev.set_profiling(2)
                dev.iterate(sample_after)
                dev.sync()
                t2 = dev.timings()
                tim = dict(tim, sampled_after_ms=t2['spmv_ms'], sampled_after=t2['spmv_samples'])
            except L.PrcgError:
                pass
        return elapsed, t_enq, tim, finite, None

    def product_rates(dev, sched, n, nnz):
        """Standalone SpMV / two-vector SpMM of the resident operator (north_star: effective SpMV HBM GB/s)."""
        xin = np.random.default_rng(0).standard_normal(n)
        dev.matvec(xin, reps=30)
        _, ms1 = dev.matvec(xin, reps=50)
        rs = np.stack([xin, xin[::-1]], axis=1)
        dev.matmat2(rs, reps=10)
        _, ms2 = dev.matmat2(rs, reps=50)
        b1, b2 = spmv_bytes(n, nnz), spmm2_bytes(n, nnz)
        opb = dev.operator_bytes()
        m1, m2 = moved_bytes(b1, n, nnz, opb), moved_bytes(b2, n, nnz, opb)
        return {'col_bytes': sched['col_bytes'], 'value_dictionary': sched['value_dict'], 'window_kernels': sched['window'],
                'spmv_ms': ms1, 'spmv_moved_GBps': m1 / ms1 * 1e-6, 'spmv_frac_of_peak': m1 / ms1 * 1e-6 / HBM_PEAK_GBS,
                'spmv_algorithmic_GBps': b1 / ms1 * 1e-6,
                'spmm2_ms': ms2, 'spmm2_moved_GBps': m2 / ms2 * 1e-6, 'spmm2_frac_of_peak': m2 / ms2 * 1e-6 / HBM_PEAK_GBS,
                'spmm2_algorithmic_GBps': b2 / ms2 * 1e-6}

    def launch_bytes(vname, sched, n, nnz):
        """(algorithmic bytes of the dominant launch per SURVEY 8d, its name)"""
        if sched['fused'] and vname.startswith('pipe_'):
            # + (r,s) read and written beside (r~,s~) and the diagonal with Jacobi; + w read and written in the 'p' flavours
            kb = fused_bytes(n, nnz) + (40 * n if vname == 'pipe_pr_pcg' else 0) + (16 * n if vname == 'pipe_p_cg' else 0)
            fam = ('k_win_tiles<2,fused>' + (', pattern tiles' if sched.get('pattern') else '')) if sched['window'] else (
                ('k_sell_win<2,fused>' if sched.get('window_codes') else 'k_sell_tiles<2,fused>') if sched.get('sliced_rows') else 'k_spmv_tiles<2,fused>')
            return kb, 'one-launch pipelined iteration: two-vector SpMM + next vector update + inner products (' + fam + ')'
        if vname.startswith('pipe_'):
            return spmm2_bytes(n, nnz), 'two-vector SpMM, interior launch (' + ('k_win_tiles<2>' if sched['window'] else 'k_spmv_tiles<2>') + ')'
        if sched['fused'] and vname == 'pr_cg':
            return spmv_bytes(n, nnz) + 48 * n, 'one-launch predict-and-recompute iteration (k_win_tiles<1,PROne>)'
        if sched['fused'] and sched['window'] and vname in ('cg_cg', 'gv_cg') and os.environ.get('PRCG_CG_ONE', '1') != '0':
            # window of three old vectors instead of one input vector (+16 n), the row's x, p (r, s) and three new vectors
            extra = {'cg_cg': 16 + 2 * 16 + 3 * 8, 'gv_cg': 16 + 4 * 16 + 3 * 8}[vname]
            return spmv_bytes(n, nnz) - 8 * n + extra * n, ('one launch per ' + ('Chronopoulos-Gear' if vname == 'cg_cg' else 'Ghysels-Vanroose') +
                                                         ' iteration, deferred p, s update (k_win_tiles<1,' + ('CGOne' if vname == 'cg_cg' else 'GVOne') + '>)')
        return spmv_bytes(n, nnz), 'SpMV launch (' + ('k_win_tiles<1>' if sched['window'] else 'k_spmv_tiles<1>') + ')'

    def summarize(vname, dev, elapsed, tim, finite, n, nnz, K=None):
        """value + launch time + must-move bytes + fractions of one timed run on one GPU"""
        K = K_LEG if K is None else K
        sched = dev.schedule()
        kb, kname = launch_bytes(vname, sched, n, nnz)
        opb = dev.operator_bytes()
        mv = moved_bytes(kb, n, nnz, opb)
        ms = tim['spmv_ms']
        return {'value': K / elapsed, 'unit': 'iters/s', 'ms_per_step': elapsed / K * 1e3, 'kernel': kname, 'avg_launch_ms': ms,
                'launches_sampled': tim['spmv_samples'], 'update_kernel_ms': tim['update_ms'],
                'avg_launch_ms_sampled_after': tim.get('sampled_after_ms'), 'launches_sampled_after': tim.get('sampled_after'),
                'bytes_must_move': mv, 'moved_GBps': mv / ms * 1e-6 if ms > 0 else 0.0,
                'frac': mv / ms * 1e-6 / HBM_PEAK_GBS if ms > 0 else 0.0,
                'bytes_algorithmic': kb, 'algorithmic_frac': kb / ms * 1e-6 / HBM_PEAK_GBS if ms > 0 else 0.0,
                'operator_bytes': opb, 'operator_bytes_per_nonzero': opb / max(nnz, 1),
                'window_kernels': sched['window'], 'value_dictionary': sched['value_dict'], 'col_bytes': sched['col_bytes'],
                'pattern_tiles': sched.get('pattern', False), 'sliced_rows': sched.get('sliced_rows', False),
                'sorted_windows': sched.get('sorted_windows', False), 'window_codes': sched.get('window_codes', False), 'nt_loads': sched.get('nt_loads', False), 'stream_stores': sched.get('stream_stores', False),
                'one_launch': sched['fused'], 'residual_finite': finite}

    # ---- the rank's row block of the synthetic operator, right-hand side as the reference --
    wl = problems.WORKLOADS[args.workload]
    n = wl['n']
    if world > 1 and args.workload in ('s4', 'queen'):
        # irregular degrees: split where the nonzeros balance, not the rows (the generator / file gives
        # every rank the whole row pointer anyway)
        offsets = partition.nnz_balanced_offsets(wl['make']().indptr, world)
    else:
        offsets = partition.even_offsets(n, world)
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    A_rows = wl['make'](rows=(lo, hi))
    b, x0, x_true = problems.reference_rhs(A_rows, n)
    nnz_local = int(A_rows.nnz)
    nnz_total = sum(comm.allgather_obj(nnz_local))
    n_local = hi - lo

    t_setup = time.perf_counter()
    if world == 1 and args.force_comm:
        dev = one_rank_comm_device(A_rows.tocsr())
    else:
        op = scaling.RowBlockOperator(comm, A_rows, device=local_rank)
        dev = op.dev
    t_setup = time.perf_counter() - t_setup     # tiling, stream encodings, upload (outside the timed region)
    variant = VAR[args.variant]
    inv_diag = (1.0 / A_rows.tocsr()[:, lo:hi].diagonal()) if args.variant == 'pipe_pr_pcg' else None

    fallback = None
    run_err = None
    if world > 1 and args.variant.startswith('pipe_'):
        # the multi-rank one-launch schedule against the RCCL two-kernel schedule on the first iterations, before anything
        # is timed (scaling.one_launch_self_check): on disagreement every rank runs the timed region on the RCCL schedule
        run_err = scaling.one_launch_self_check(op, variant, b, x0, inv_diag)
    if run_err is None:
        elapsed, t_enq, tim, finite, run_err = timed_run(dev, variant, b, x0, inv_diag, sample_after=SAMPLE_AFTER if world == 1 else 0)
    if run_err is not None:
        # the one-launch schedule of a multi-rank session could not be kept fed on this node (its waits are
        # bounded and reported): every rank rebuilds its operator with the RCCL two-kernel schedule and the run is repeated
        fallback = run_err
        dev.close()
        if world == 1 and args.force_comm:
            dev = one_rank_comm_device(A_rows.tocsr(), knobs={'PRCG_FUSED_COMM': '0', 'PRCG_PEER': '0'})
        else:
            op = scaling.RowBlockOperator(comm, A_rows, device=local_rank, knobs={'PRCG_FUSED_COMM': '0', 'PRCG_PEER': '0'})
            dev = op.dev
        elapsed, t_enq, tim, finite, run_err = timed_run(dev, variant, b, x0, inv_diag)
        if run_err is not None:
            raise RuntimeError(run_err)
    sched = dev.schedule()
    spmv = product_rates(dev, sched, n, nnz_total) if world == 1 else None

    rccl_leg = None

    # An operator whose values do not repeat (an assembled FEM matrix) streams the doubles themselves: time
    # that path too, same matrix, value dictionary off.  This leg IS what SURVEY.md 8d's algorithmic bytes
    # (12 B per nonzero) describe up to the narrower column stream.
    plain = None
    if world == 1 and not args.force_comm and sched['value_dict'] and not args.no_plain_values:
        dev2 = DeviceCSR(A_rows.tocsr(), device=local_rank, knobs={'PRCG_VALDICT': '0'})
        e2, _, tim2, fin2, _err2 = timed_run(dev2, variant, b, x0, inv_diag, sample_after=SAMPLE_AFTER)
        sched2 = dev2.schedule()
        plain = (e2, tim2, fin2, sched2, product_rates(dev2, sched2, n, nnz_total), dev2.operator_bytes())
        dev2.close()

    # What the schedule every rank of an N>1 run executes costs on ONE GPU: (a) the same loop with a 1-rank
    # communicator; (b) one rank's share of an 8-GPU run: one eighth of S3 with a loopback halo (boundary tiles, ghost
    # rows, the whole exchange chain) beside the plain one-launch schedule on the same slice.
    multi = None
    if world == 1 and not args.force_comm and not args.no_multi_rank_leg and args.variant.startswith('pipe_'):
        def comm_leg(A, halo, bb, xx0, dd, knobs=None):
            d3 = one_rank_comm_device(A, halo, knobs)
            e3, q3, tim3, fin3, err3 = timed_run(d3, variant, bb, xx0, dd, K_LEG)
            if err3:
                d3.close()
                raise RuntimeError(err3)
            s3 = d3.schedule()
            out = {'one_launch': s3['fused_comm'], 'merged_exchange': s3['gather'], 'peer_stores': s3.get('peer', False),
                   'value': K_LEG / e3, 'unit': 'iters/s', 'us_per_iteration': e3 / K_LEG * 1e6, 'steps': K_LEG,
                   'launch_us': tim3['spmv_ms'] * 1e3, 'update_us': tim3['update_ms'] * 1e3, 'residual_finite': fin3,
                   'host_enqueue_us_per_step': q3 / K_LEG * 1e6}
            d3.close()
            return out
        try:
            multi = {'what': 'the schedule every rank of an N>1 run executes, timed on this one GPU with a 1-rank communicator '
                             '(at least 200 timed steps)'}
            multi['s3'] = comm_leg(A_rows.tocsr(), None, b, x0, inv_diag)
            # compatibility with earlier rounds' records
            multi.update({k: multi['s3'][k] for k in ('one_launch', 'value', 'unit', 'host_enqueue_us_per_step')})
            multi['ms_per_step'] = multi['s3']['us_per_iteration'] * 1e-3
            if args.workload == 's3':
                def share_leg(name, halo_rows, cut, single_us, single_plain_us, with_rccl):
                    """one rank's share of an N-GPU run on this GPU: plain one-launch schedule, multi-rank schedule with a loopback halo"""
                    wls = problems.WORKLOADS[name]
                    As = wls['make']()
                    bs, xs, _ = problems.reference_rhs(As, wls['n'])
                    ds = (1.0 / As.diagonal()) if inv_diag is not None else None
                    A_loop, halo_s, _ = partition.loopback_problem(As, halo_rows, cut=cut)
                    rec = {'workload': wls['desc'], 'halo_rows_per_side': halo_rows}
                    for key, kn in (('dict', None), ('plain_values', {'PRCG_VALDICT': '0'})):
                        if key == 'plain_values' and plain is None:
                            continue
                        dp = DeviceCSR(As, device=local_rank, knobs=kn)
                        ep, qp, _, _, _ = timed_run(dp, variant, bs, xs, ds, K_LEG)
                        dp.close()
                        leg = comm_leg(A_loop, halo_s, bs, xs, ds, kn)
                        single = single_us if key == 'dict' else single_plain_us
                        r = {'plain_us_per_iteration': ep / K_LEG * 1e6, 'plain_host_enqueue_us_per_step': qp / K_LEG * 1e6,
                             'comm_us_per_iteration': leg['us_per_iteration'], 'comm': leg, 'single_gpu_us_per_iteration': single,
                             'speedup_ceiling': single / leg['us_per_iteration'] if single and leg['us_per_iteration'] > 0 else None}
                        if key == 'dict':
                            rec.update(r)
                            if with_rccl:
                                try:       # the RCCL schedule of the same slice, for comparison
                                    rec['rccl_schedule'] = comm_leg(A_loop, halo_s, bs, xs, ds, {'PRCG_PEER': '0'})
                                except Exception as exc:
                                    rec['rccl_schedule'] = {'error': str(exc)[:200]}
                        else:
                            rec['plain_values'] = r
                    return rec
                s3_us, s3_plain_us = elapsed / K * 1e6, (plain[0] / K * 1e6 if plain is not None else None)
                multi['what_shares'] = ("s3_half / s3_quarter / s3_8th (and s2_*): ONE rank's share of S3 (S2) on 2 / 4 / 8 GPUs, run on this GPU: "
                                        'the plain one-launch schedule, and the multi-rank schedule with a loopback halo (boundary tiles, ghost rows, '
                                        'the whole exchange: rows and partial sums stored into the exchange buffer by the launch itself, next launch '
                                        'waits in-kernel); speedup_ceiling = single-GPU time / that -- a real exchange adds xGMI latency to it')
                multi['s3_half'] = share_leg('s3_half', 7, None, s3_us, s3_plain_us, False)
                multi['s3_quarter'] = share_leg('s3_quarter', 7, None, s3_us, s3_plain_us, False)
                multi['s3_8th'] = share_leg('s3_8th', 7, None, s3_us, s3_plain_us, True)
                multi['s3_8th']['speedup_ceiling_at_8_ranks'] = multi['s3_8th'].get('speedup_ceiling')     # (earlier rounds' key)
                multi['forecast_s3'] = {str(N): multi[k].get('speedup_ceiling') for N, k in ((2, 's3_half'), (4, 's3_quarter'), (8, 's3_8th'))}
                if plain is not None:
                    multi['forecast_s3_plain_values'] = {str(N): (multi[k].get('plain_values') or {}).get('speedup_ceiling')
                                                         for N, k in ((2, 's3_half'), (4, 's3_quarter'), (8, 's3_8th'))}
                # BASELINE config 4: S2's shares (whole grid planes, a halo of one 216 x 216 plane = 746 KB of pairs per side);
                # its single-GPU time comes from the workloads leg below (multi['s2_single_us'] is filled there)
                multi['_s2_pending'] = True
        except Exception as exc:       # RCCL missing on a box: the bench line itself does not depend on it
            multi = dict(multi or {}, error=str(exc)[:300])

    # BASELINE.json's other configurations on this GPU (configs 3, 4 at N=1, 5 through its stand-ins)
    others = None
    if world == 1 and not args.force_comm and not args.no_workloads and args.workload == 's3' and args.variant == 'pipe_pr_cg':
        others = {}
        for name, cfg in (('s1', 'BASELINE config 3'), ('s2', 'BASELINE config 4 at N=1'),
                          ('s4b', "BASELINE config 5, FEM-like stand-in at Queen_4147's size (uniform 3 dof x 27-point coupling)"),
                          ('s4c', "BASELINE config 5, IRREGULAR FEM-like stand-in at Queen_4147's size (nodes of 1 / 3 / 6 unknowns, thinned couplings: "
                                  'rows of 7..121 nonzeros with FEM locality -- the load-balance stress)'),
                          ('s4', 'BASELINE config 5, random-offset irregular stand-in (SURVEY 8d)')):
            try:
                w2 = problems.WORKLOADS[name]
                A2 = w2['make']()
                b2, x2, _ = problems.reference_rhs(A2, w2['n'])
                d2 = DeviceCSR(A2, device=local_rank)
                e, _, tm, fin, er = timed_run(d2, variant, b2, x2, None, K_LEG)
                if er:
                    raise RuntimeError(er)
                rec = dict(summarize(args.variant, d2, e, tm, fin, w2['n'], int(A2.nnz)), config=cfg, workload=w2['desc'],
                           n=w2['n'], nnz=int(A2.nnz))
                rec['traffic'] = measured_traffic(f"{name}:{args.variant}:fused:1:" + ('dict' if d2.schedule()['value_dict'] else 'plain'))
                if rec['traffic']:
                    rec['traffic_over_must_move'] = rec['traffic'] / rec['bytes_must_move']
                    rec['frac_physical'] = rec['traffic'] / (rec['avg_launch_ms'] * 1e-3) * 1e-9 / HBM_PEAK_GBS
                # what the memory system delivers for THIS launch's byte mix with no arithmetic (prcg_mix_ceiling: the operator's
                # stream per 64 rows beside the rows' 64 B of vector traffic): the ceiling of a kernel that moves it
                try:
                    kb64 = int(round(rec['operator_bytes'] / max(w2['n'], 1) * 64 / 1024))
                    mc = d2.mix_ceiling(w2['n'], kb64, reps=6)
                    rec['mix_ceiling_GBps'] = mc
                    rec['mix_ceiling_stream_kb_per_64_rows'] = kb64
                    rec['frac_of_mix_ceiling'] = rec['moved_GBps'] / mc if mc > 0 else None
                except Exception as exc:
                    rec['mix_ceiling_error'] = str(exc)[:200]
                has_dict = d2.schedule()['value_dict']
                d2.close()
                if has_dict:
                    d2 = DeviceCSR(A2, device=local_rank, knobs={'PRCG_VALDICT': '0'})
                    e, _, tm, fin, er = timed_run(d2, variant, b2, x2, None, K_LEG)
                    if not er:
                        rec['plain_values'] = summarize(args.variant, d2, e, tm, fin, w2['n'], int(A2.nnz))
                        rec['plain_values']['traffic'] = measured_traffic(f"{name}:{args.variant}:fused:1:plain")
                    d2.close()
                others[name] = rec
                del A2
            except Exception as exc:
                others[name] = {'error': str(exc)[:300]}
        # S2's shares on 2 / 4 / 8 GPUs (see multi_rank_schedule above), now that its single-GPU time is known
        if multi and multi.pop('_s2_pending', False) and 'value' in others.get('s2', {}):
            try:
                s2_us = others['s2']['ms_per_step'] * 1e3
                s2_plain_us = (others['s2'].get('plain_values') or {}).get('ms_per_step')
                s2_plain_us = s2_plain_us * 1e3 if s2_plain_us else None
                for key, nz in (('s2_half', 108), ('s2_quarter', 54), ('s2_8th', 27)):
                    multi[key] = share_leg(key, 216 * 216, (nz // 2) * 216 * 216, s2_us, s2_plain_us, False)
                multi['forecast_s2'] = {str(N): multi[k].get('speedup_ceiling') for N, k in ((2, 's2_half'), (4, 's2_quarter'), (8, 's2_8th'))}
                if plain is not None:
                    multi['forecast_s2_plain_values'] = {str(N): (multi[k].get('plain_values') or {}).get('speedup_ceiling')
                                                         for N, k in ((2, 's2_half'), (4, 's2_quarter'), (8, 's2_8th'))}
            except Exception as exc:
                multi['s2_shares_error'] = str(exc)[:300]
    if multi:
        multi.pop('_s2_pending', None)

    # BASELINE configs 1 and 2: the paper's small matrices (committed fixtures, data only) at the iteration counts of
    # figure_gen.py:263,289 -- launch-bound systems: the whole pipelined solve of nos7 is ONE launch of ONE workgroup
    if others is not None:
        try:
            import scipy.sparse as sp
            for key, name, vname, iters, cfg in (('bcsstk03_hs_cg', 'bcsstk03', 'hs_cg', 1250, 'BASELINE config 1 (the method the reference runs on its CPU path)'),
                                                 ('nos7_pipe_pr_cg', 'nos7', 'pipe_pr_cg', 7000, 'BASELINE config 2')):
                f = os.path.join(ROOT, 'tests', 'golden', f'matrix_{name}.npz')
                if not os.path.exists(f):
                    continue
                z = np.load(f)
                ns = int(z['n'])
                As = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(ns, ns))
                ds = DeviceCSR(As, device=local_rank)
                best = None
                for _ in range(3):
                    ds.begin(VAR[vname], z['b'], np.zeros(ns), iters, hist_mask=1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ds.iterate(iters - 1)
                    ds.sync()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                hist = ds.history()['updated_residual_2_norm']
                sch = ds.schedule()
                ds.close()
                others[key] = {'config': cfg, 'workload': f'{name} (n={ns}, nnz={As.nnz}), {vname}, {iters} iterations as figure_gen.py', 'value': (iters - 1) / best,
                               'unit': 'iters/s', 'us_per_iteration': best / (iters - 1) * 1e6, 'one_workgroup_solver': sch['small'],
                               'residual_reduction': float(np.nanmin(hist) / hist[0])}
        except Exception as exc:
            others['small_systems_error'] = str(exc)[:300]

    def run_rccl_leg():
        """N > 1: the paper's contrast in one run -- after the default schedule (one launch per iteration, direct peer exchange) the same
        loop on the RCCL schedule (update kernel + two-vector SpMM, halo and the single all-reduce on side streams: the GPU form of
        scaling_experiments_petsc/cg_impls/pipeprcg.c:154-173), a short second leg.  Runs LAST, behind a watchdog on rank 0: if it has
        not finished after 120 s the line is written without it -- a second schedule must never cost the run its first number."""
        if not (world > 1 and args.variant.startswith('pipe_') and fallback is None and sched.get('peer') and not args.no_rccl_leg):
            return None
        try:
            op2 = scaling.RowBlockOperator(comm, A_rows, device=local_rank, knobs={'PRCG_FUSED_COMM': '0', 'PRCG_PEER': '0'})
            k2 = min(K, 100)
            e2, q2, tim2, fin2, err2 = timed_run(op2.dev, variant, b, x0, inv_diag, K=k2)
            s2 = op2.dev.schedule()
            leg = {'what': 'the same loop on the RCCL side-stream schedule (PRCG_PEER=0): update kernel + SpMM, halo send/recv or merged '
                           'all-gather and ONE all-reduce per iteration on communication streams',
                   'rccl_ranks': world, 'value': k2 / e2 if not err2 else None, 'unit': 'iters/s', 'steps': k2,
                   'ms_per_step': e2 / k2 * 1e3, 'residual_finite': fin2, 'error': err2,
                   'merged_exchange': s2.get('gather'), 'one_launch': s2.get('fused_comm'), 'host_enqueue_us_per_step': q2 / k2 * 1e6}
            op2.dev.close()
            return leg
        except Exception as exc:
            return {'rccl_ranks': world, 'error': str(exc)[:300]}

    if rank == 0:
        fused = sched['fused']
        kbytes, kname = launch_bytes(args.variant, sched, n_local, nnz_local)
        ms = tim['spmv_ms']
        opb = dev.operator_bytes()
        moved = moved_bytes(kbytes, n_local, nnz_local, opb)
        achieved = moved / ms * 1e-6 if ms > 0 else 0.0
        tkey = f"{args.workload}:{args.variant}{':fused' if fused else ''}:{world}"
        traffic = measured_traffic(tkey + (':dict' if sched['value_dict'] else ':plain'))
        roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'frac_physical': (traffic / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS) if traffic and ms > 0 else None,
                'frac_8d': kbytes / ms * 1e-6 / HBM_PEAK_GBS if ms > 0 else None,
                'kernel': kname, 'bytes_per_launch': moved, 'avg_launch_ms': ms, 'launches_sampled': tim['spmv_samples'],
                'avg_launch_ms_sampled_after': tim.get('sampled_after_ms'), 'launches_sampled_after': tim.get('sampled_after'),
                'update_kernel_ms': tim['update_ms'],
                'basis': ('achieved / frac = bytes the launch MUST MOVE / mean launch time by HIP events on the compute stream inside the timed region: '
                          'the operator as the device streams it (prcg_operator_bytes: ' f"{sched['col_bytes']} B window / column index + "
                          + ('1 B value-dictionary index' if sched['value_dict'] else '8 B value') + ' per nonzero, relative row '
                          'pointers, tile descriptors; stream images shared by many tiles counted once) + the vector traffic of SURVEY 8d per row.  '
                          'frac_8d = SURVEY.md 8d\'s algorithmic bytes (12 B per nonzero + vectors) / the same time / peak: it EXCEEDS 1 for this '
                          'workload because the operator is compressed (S3 is a two-value band at n = 1e7: 15 MB of descriptors and shared images '
                          'instead of 1.8 GB of CSR) -- it is no roofline fraction.  The 8d ratio that IS one is frac_8d_general_csr: the same matrix '
                          'and loop with every 8-byte value actually streamed (plain_values), with frac_physical_general_csr = PMC traffic / time / '
                          'peak beside it.  frac_physical = PMC traffic of this launch / its time / peak.  stream_ceiling_GBps: what this GPU '
                          'delivered in this run for the launch\'s own byte mix with no arithmetic.'),
                'operator_stream': {'col_bytes': sched['col_bytes'], 'value_dictionary': sched['value_dict'],
                                    'window_kernels': sched['window'], 'bytes': opb, 'bytes_per_nonzero': opb / max(nnz_local, 1),
                                    'csr_bytes': 12 * nnz_local + 4 * (n_local + 1)},
                'effective': {'what': 'SURVEY.md 8d algorithmic CSR bytes (12 B per nonzero + vectors) / the same launch time: '
                                      'north_star\'s "effective" bandwidth; exceeds the peak when the stream is compressed',
                              'bytes_per_launch': kbytes, 'GBps': kbytes / ms * 1e-6 if ms > 0 else 0.0,
                              'frac_of_peak': kbytes / ms * 1e-6 / HBM_PEAK_GBS if ms > 0 else 0.0}}
        if world == 1:
            # "also report a measured stream ceiling" -- in this run, on this GPU: the one-launch iteration's own vector mix
            # (per row 2 x 16 B read, 2 x 16 B written: all the dictionary kernel moves besides 15 MB of operator) over this workload's n
            # rows, with the stores the kernel uses; and a pure 16-byte read over 2 GB (what bounds the plain-value / sliced-row streams)
            try:
                mix = dev.stream_ceiling(n_local, 2 if sched.get('stream_stores', True) else 1)
                mix_plain = dev.stream_ceiling(n_local, 1)
                rd = dev.stream_ceiling(128_000_000, 0, reps=10)
                rd_best = dev.stream_ceiling(128_000_000, 3, reps=10)
                roof['stream_ceiling_GBps'] = mix
                roof['stream_ceiling'] = {'own_mix_GBps': mix, 'own_mix_plain_stores_GBps': mix_plain, 'pure_read_2GB_GBps': rd_best, 'pure_read_2GB_grid_stride_GBps': rd,
                                          'what': f'prcg_stream_ceiling on this GPU in this run: 2 x 16 B read + 2 x 16 B written per row over {n_local} rows '
                                                  '(nontemporal stores as the kernel / plain stores), and a pure 16-byte-per-lane read of 2 GB (contiguous 4 KB chunks per wave, '
                                                  'nontemporal; and the grid-stride form of round 3)'}
                roof['frac_of_stream_ceiling'] = achieved / mix if mix > 0 else None
            except Exception as exc:
                roof['stream_ceiling_GBps'] = None
                roof['stream_ceiling_error'] = str(exc)[:200]
        assert roof['frac'] <= 1.0, 'a physical fraction cannot exceed 1: the byte count is wrong'
        value_general = None
        if plain is not None:
            e2, tim2, fin2, sched2, spmv2, opb2 = plain
            ms2 = tim2['spmv_ms']
            mv2 = moved_bytes(kbytes, n_local, nnz_local, opb2)
            value_general = K / e2
            tr2 = measured_traffic(tkey + ':plain')
            roof['plain_values'] = {
                'what': 'same matrix and loop with the value dictionary off (PRCG_VALDICT=0): the rate of an operator '
                        f"whose values do not repeat; {sched2['col_bytes']} B column stream + 8 B value per nonzero",
                'value': K / e2, 'unit': 'iters/s', 'ms_per_step': e2 / K * 1e3, 'avg_launch_ms': ms2,
                'launches_sampled': tim2['spmv_samples'], 'avg_launch_ms_sampled_after': tim2.get('sampled_after_ms'),
                'launches_sampled_after': tim2.get('sampled_after'),
                'achieved': kbytes / ms2 * 1e-6, 'frac': kbytes / ms2 * 1e-6 / HBM_PEAK_GBS,
                'basis': 'SURVEY.md 8d algorithmic bytes (12 B per nonzero + vectors) / mean launch time',
                'moved_GBps': mv2 / ms2 * 1e-6, 'moved_frac': mv2 / ms2 * 1e-6 / HBM_PEAK_GBS, 'bytes_moved_per_launch': mv2,
                'operator_bytes': opb2,
                'traffic': tr2, 'frac_physical': (tr2 / (ms2 * 1e-3) * 1e-9 / HBM_PEAK_GBS) if tr2 else None,
                'residual_finite': fin2, 'spmv': spmv2}
            # first-class: SURVEY 8d's ratio on the leg it describes, algorithmic and physical
            roof['value_general_csr'] = K / e2
            roof['frac_8d_general_csr'] = roof['plain_values']['frac']
            roof['frac_moved_general_csr'] = roof['plain_values']['moved_frac']
            roof['frac_physical_general_csr'] = roof['plain_values']['frac_physical']
            if roof.get('stream_ceiling'):
                roof['plain_values']['frac_of_pure_read_ceiling'] = (mv2 / ms2 * 1e-6) / roof['stream_ceiling']['pure_read_2GB_GBps']
            try:
                kb64 = int(round(opb2 / max(n_local, 1) * 64 / 1024))
                mc2 = dev.mix_ceiling(n_local, kb64, reps=6)
                roof['plain_values']['mix_ceiling_GBps'] = mc2
                roof['plain_values']['frac_of_mix_ceiling'] = (mv2 / ms2 * 1e-6) / mc2 if mc2 > 0 else None
                roof['frac_of_mix_ceiling_general_csr'] = roof['plain_values']['frac_of_mix_ceiling']
            except Exception as exc:
                roof['plain_values']['mix_ceiling_error'] = str(exc)[:200]
            assert roof['plain_values']['frac'] <= 1.0
        if spmv:
            roof['spmv'] = spmv
        if multi:
            roof['multi_rank_schedule'] = multi
        out = {
            'metric': f'{args.variant} iterations/sec (synthetic banded CSR, fp64)',
            'value': K / elapsed, 'unit': 'iters/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'value_general_csr': value_general,
            'value_general_csr_note': ('iterations/s of the SAME loop on the SAME matrix with the value dictionary off: every nonzero moves its '
                                       '8-byte value, as for an assembled matrix whose values do not repeat.  `value` rides on S3\'s repeated '
                                       'coefficients (one constant off the diagonal: 15 MB of operator instead of 1.2 GB); this is the number '
                                       'that transfers to other matrices') if value_general is not None else None,
            'config': {'workload': wl['desc'], 'n': n, 'nnz': nnz_total, 'variant': args.variant,
                       'partition': f'row blocks x{world}' + (' (nnz-balanced)' if world > 1 and args.workload in ('s4', 'queen') else ''), 'rhs': 'x_true=1/sqrt(n), b=A x_true, x0=0',
                       'residual_finite': finite, 'fresh_session_before_timed_steps': True, 'host_enqueue_us_per_step': t_enq / K * 1e6,
                       'operator_setup_s': t_setup, 'prewarm_steps': PREWARM,
                       'schedule_fallback': fallback, 'rccl_ranks': (world if (rccl_leg and not rccl_leg.get('error')) or fallback else 0),
                       'schedule': {k: v for k, v in sched.items()}},
            'roofline': roof,
        }
        if others:
            out['workloads'] = others
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(A_rows.tocsr(), b, x0, FAMILY[args.variant], args.cpu_seconds)
        else:
            out['cpu_baseline'] = None
        line_written = []

        def write_line():
            if not line_written:
                line_written.append(1)
                sys.stdout.flush()
                os.write(json_fd, (json.dumps(out) + '\n').encode())

        if world > 1:
            import threading

            def give_up():
                out['roofline']['rccl_schedule'] = {'rccl_ranks': world, 'error': 'the RCCL second leg did not finish within 120 s: line written without it'}
                write_line()
                os._exit(0)
            dog = threading.Timer(120.0, give_up)
            dog.daemon = True
            dog.start()
            leg = run_rccl_leg()
            dog.cancel()
            if leg:
                out['roofline']['rccl_schedule'] = leg
                out['config']['rccl_ranks'] = world if not leg.get('error') else out['config']['rccl_ranks']
        write_line()
    elif world > 1:
        run_rccl_leg()

    dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
