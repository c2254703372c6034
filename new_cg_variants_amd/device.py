"""Host-side owner of one libprcg handle: one GPU, one row block, one rank.

This is the thin layer between SciPy CSR arrays and the C-ABI (include/prcg.h); every
numerical operation happens in the HIP kernels behind it.
"""
import contextlib
import ctypes as C
import os
import sys

import numpy as np

from . import _lib as L


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on fd 1 when a communicator is created; a program
    whose stdout is a protocol (bench.py prints exactly one JSON line) must not see it."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class DeviceCSR:
    """A CSR row block resident on one MI355X.

    ``A`` is what the reference passes around as ``A``: a ``scipy.sparse`` CSR matrix
    (numerical_experiments/figure_gen.py:350).  For a multi-rank run ``A`` is the rank's
    row block with LOCAL column numbering and ``halo`` the plan produced by
    ``partition.plan_halo`` (columns >= n_rows are ghosts).
    """

    def __init__(self, A, device=0, comm_init=None, halo=None, knobs=None, world=None):
        """knobs: dict of PRCG_* experiment switches for THIS handle (prcg_set_option), e.g.
        {'PRCG_FUSED': '0'} keeps the two-kernel schedule on one GPU.  The process environment
        is not touched."""
        self._h = C.c_void_p()
        self._lib = L.lib()
        rc = self._lib.prcg_create(C.byref(self._h), int(device))
        if rc != L.OK:
            msg = self._lib.prcg_last_error(None)
            self._h = C.c_void_p()
            raise L.PrcgError(rc, msg.decode() if msg else '?')
        for k, v in (knobs or {}).items():
            self._check(self._lib.prcg_set_option(self._h, str(k).encode(), str(v).encode()))
        self.rank, self.nranks = 0, 1
        if comm_init is not None:
            rank, nranks, uid, path = comm_init      # uid: 128 or 256 bytes (one or two RCCL ids)
            ids = np.frombuffer(uid, dtype=np.uint8).copy()
            assert ids.size in (128, 256)
            with _stdout_to_stderr():
                self._check(self._lib.prcg_comm_init(self._h, path.encode() if path else None, rank, nranks,
                                                     L.ptr(ids), ids.size // 128))
            self.rank, self.nranks = rank, nranks
        elif world is not None:        # rank / world size without a communicator (prcg.h: prcg_world_init; peer-exchange plumbing only)
            self._check(self._lib.prcg_world_init(self._h, int(world[0]), int(world[1])))
            self.rank, self.nranks = int(world[0]), int(world[1])
        self._set_matrix(A, halo)

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        L.check(self._h, rc)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.prcg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _set_matrix(self, A, halo):
        if not hasattr(A, 'indptr'):
            raise TypeError('A must be a scipy.sparse CSR matrix/array')
        if A.format != 'csr':
            A = A.tocsr()
        if A.dtype != np.float64:
            raise TypeError(f'fp64 only (got {A.dtype}); the reference works in double precision')
        n_rows, n_cols = A.shape
        n_ghost = n_cols - n_rows
        if n_ghost < 0:
            raise ValueError('row block must have at least n_rows columns (local numbering)')
        indptr = np.ascontiguousarray(A.indptr)
        is64 = indptr.dtype == np.int64
        if not is64:
            indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        data = np.ascontiguousarray(A.data, dtype=np.float64)
        self.n, self.n_ghost, self.nnz = int(n_rows), int(n_ghost), int(A.nnz)
        self._check(self._lib.prcg_set_csr(self._h, self.n, self.n_ghost, self.nnz, L.ptr(indptr), int(is64),
                                           L.ptr(indices), L.ptr(data)))
        self.halo = halo
        if halo is not None:
            peers = np.ascontiguousarray(halo['peers'], dtype=np.int32)
            send_ptr = np.ascontiguousarray(halo['send_ptr'], dtype=np.int64)
            send_idx = np.ascontiguousarray(halo['send_idx'], dtype=np.int32)
            recv_ptr = np.ascontiguousarray(halo['recv_ptr'], dtype=np.int64)
            self._check(self._lib.prcg_set_halo(self._h, len(peers), L.ptr(peers), L.ptr(send_ptr),
                                                L.ptr(send_idx), L.ptr(recv_ptr)))

    # -- direct peer exchange (prcg.h: prcg_peer_setup / prcg_peer_connect) ------------------------------
    def peer_setup(self, max_ghost_any_rank):
        """Allocate this rank's exchange buffer; returns (64-byte IPC handle, device address)."""
        handle = np.zeros(64, dtype=np.uint8)
        ptr = C.c_void_p()
        self._check(self._lib.prcg_peer_setup(self._h, int(max_ghost_any_rank), L.ptr(handle), C.byref(ptr)))
        return handle.tobytes(), int(ptr.value or 0)

    def peer_connect(self, handles, same_process_ptrs, send_dst_off):
        """handles: list of 64-byte IPC handles in rank order (or None); same_process_ptrs: list of device addresses
        (0: use the handle) or None; send_dst_off[q]: where this rank's rows begin in peer q's ghost area."""
        hb = None if handles is None else np.frombuffer(b''.join(handles), dtype=np.uint8).copy()
        pp = None if same_process_ptrs is None else (C.c_void_p * len(same_process_ptrs))(*[C.c_void_p(int(v) or None) for v in same_process_ptrs])
        off = np.ascontiguousarray(send_dst_off, dtype=np.int64)
        if off.size == 0:
            off = np.zeros(1, dtype=np.int64)
        self._check(self._lib.prcg_peer_connect(self._h, L.ptr(hb), pp, L.ptr(off)))

    def peer_selftest(self, k, rows, slot):
        """One round of the exchange primitives (prcg.h: prcg_peer_selftest): returns (sum of all ranks' slots, ghost area)."""
        rows, slot = L.f64(rows), L.f64(slot)
        assert rows.shape == (self.n, 2) and slot.shape == (5,)
        sums, ghost = np.zeros(5), np.zeros((max(self.n_ghost, 1), 2))
        self._check(self._lib.prcg_peer_selftest(self._h, int(k), L.ptr(rows), L.ptr(slot), L.ptr(sums), L.ptr(ghost)))
        return sums, ghost[:self.n_ghost]

    # -- products (tests / bench) ---------------------------------------------------------
    def matvec(self, x, reps=1):
        """y = A x on the device; returns (y, mean ms per launch)."""
        x = L.f64(x)
        assert x.shape == (self.n,)
        y = np.empty(self.n)
        ms = C.c_double(0.0)
        self._check(self._lib.prcg_spmv(self._h, L.ptr(x), L.ptr(y), int(reps), C.byref(ms)))
        return y, ms.value

    def matvec_ext(self, x_ext):
        """y = A_local [x_own ; x_ghost] with the ghost entries supplied by the caller (no halo
        exchange, no communicator): one rank's share of a row-block product."""
        x_ext = L.f64(x_ext)
        assert x_ext.shape == (self.n + self.n_ghost,)
        y = np.empty(self.n)
        self._check(self._lib.prcg_spmv_ext(self._h, L.ptr(x_ext), L.ptr(y)))
        return y

    def matmat2(self, RS, reps=1):
        """[w u] = A [r s] for an (n,2) array; returns ((n,2) array, mean ms per launch)."""
        RS = L.f64(RS)
        assert RS.shape == (self.n, 2)
        WU = np.empty((self.n, 2))
        ms = C.c_double(0.0)
        self._check(self._lib.prcg_spmm2(self._h, L.ptr(RS), L.ptr(WU), int(reps), C.byref(ms)))
        return WU, ms.value

    # -- solver session ------------------------------------------------------------------------
    def begin(self, variant, b, x0, max_iter, x_true=None, inv_diag=None, hist_mask=0, preconditioner=None):
        """inv_diag: Jacobi on the device.  preconditioner: any callable v -> M^-1 v (what the reference's *_pcg
        functions take); it runs on the host wherever the reference calls it (prcg.h: prcg_set_preconditioner)."""
        b, x0 = L.f64(b), L.f64(x0)
        assert b.shape == (self.n,) and x0.shape == (self.n,)
        assert inv_diag is None or preconditioner is None
        xt = None if x_true is None else L.f64(x_true)
        dv = None if inv_diag is None else L.f64(inv_diag)
        if preconditioner is not None:
            n = self.n

            def call(_ctx, count, v, out):
                try:
                    res = np.asarray(preconditioner(np.ctypeslib.as_array(v, shape=(count,)).copy()), dtype=np.float64)
                    if res.shape != (count,):
                        return 2
                    np.ctypeslib.as_array(out, shape=(count,))[:] = res
                    return 0
                except Exception:          # noqa: BLE001 -- reported through the C return code
                    import traceback
                    traceback.print_exc()
                    return 1
            self._prec_fn = L.PREC_FN(call)            # keep the trampoline alive for the whole session
            self._check(self._lib.prcg_set_preconditioner(self._h, C.cast(self._prec_fn, C.c_void_p), None))
        else:
            self._prec_fn = None
            self._check(self._lib.prcg_set_preconditioner(self._h, None, None))
        self._check(self._lib.prcg_solve_begin(self._h, int(variant), L.ptr(b), L.ptr(x0), int(max_iter),
                                               L.ptr(xt), L.ptr(dv), int(hist_mask)))
        self.max_iter, self.hist_mask = int(max_iter), int(hist_mask)

    def set_replace_hook(self, fn):
        """Ghysels-Vanroose residual replacement (prcg.h: prcg_set_replace_hook): fn(k) -> truthy replaces w by A r in
        iteration k; None removes the hook.  Set before begin()."""
        if fn is None:
            self._replace_fn = None
            self._check(self._lib.prcg_set_replace_hook(self._h, None, None))
            return

        def call(_ctx, k):
            try:
                return 1 if fn(int(k)) else 0
            except Exception:          # noqa: BLE001 -- a failing predicate does not replace; the traceback is shown
                import traceback
                traceback.print_exc()
                return 0
        self._replace_fn = L.REPLACE_FN(call)          # keep the trampoline alive
        self._check(self._lib.prcg_set_replace_hook(self._h, C.cast(self._replace_fn, C.c_void_p), None))

    def iterate(self, iters):
        self._check(self._lib.prcg_iterate(self._h, int(iters)))

    def sync(self):
        self._check(self._lib.prcg_sync(self._h))

    @property
    def k(self):
        return self._lib.prcg_iteration(self._h)

    def schedule(self):
        """Flags of the schedule the current session runs (prcg.h PRCG_SCHED_*)."""
        s = self._lib.prcg_schedule(self._h)
        return {'fused': bool(s & 1), 'small': bool(s & 2), 'comm': bool(s & 4), 'gather': bool(s & 8),
                'dual_comm': bool(s & 16), 'value_dict': bool(s & 32),
                'col_bytes': 0 if s & 65536 else (1 if s & 64 else (2 if s & 128 else 4)), 'tile_steps': (s >> 8) & 15,
                'pattern': bool(s & 65536), 'window': bool(s & 4096), 'fused_comm': bool(s & 8192), 'peer': bool(s & 16384), 'sliced_rows': bool(s & 32768),
                'stream_stores': bool(s & 131072), 'sorted_windows': bool(s & 262144), 'nt_loads': bool(s & 524288), 'window_codes': bool(s & 2097152),
                'medium': bool(s & 1048576)}

    def layout(self):
        """Diagnostic (prcg.h: prcg_debug_layout): what the summation order of the one-launch iteration's inner
        products depends on -- tile rows in table order, workgroups and waves per workgroup of the last launch."""
        need = -int(self._lib.prcg_debug_layout(self._h, L.ptr(np.zeros(1, dtype=np.int64)), 0))
        out = np.zeros(max(need, 8), dtype=np.int64)
        got = int(self._lib.prcg_debug_layout(self._h, L.ptr(out), out.size))
        if got < 8:
            raise RuntimeError('prcg_debug_layout failed')
        return {'window': bool(out[0]), 'geometry': int(out[1]), 'rows_per_tile': int(out[2]), 'tiles': out[8:got].reshape(-1, 2).copy(),
                'grid': int(out[4]), 'waves_per_block': int(out[5]), 'interior_tiles': int(out[6]), 'xcd_chunked': bool(int(out[7]) & 1), 'sweep_waves': int(out[7]) >> 8}

    def operator_bytes(self):
        """Bytes of the operator as the device streams it (prcg.h: prcg_operator_bytes)."""
        return int(self._lib.prcg_operator_bytes(self._h))

    def set_iteration(self, k):
        self._check(self._lib.prcg_set_iteration(self._h, int(k)))

    def get_vector(self, name):
        out = np.empty(self.n)
        self._check(self._lib.prcg_get_vector(self._h, L.VEC[name], L.ptr(out)))
        return out

    def set_vector(self, name, v):
        v = L.f64(v)
        assert v.shape == (self.n,)
        self._check(self._lib.prcg_set_vector(self._h, L.VEC[name], L.ptr(v)))

    def get_scalars(self, k):
        out = np.empty(L.NUM_SCALARS)
        self._check(self._lib.prcg_get_scalars(self._h, int(k), L.ptr(out)))
        return out

    def set_scalars(self, k, values):
        v = L.f64(values)
        assert v.shape == (L.NUM_SCALARS,)
        self._check(self._lib.prcg_set_scalars(self._h, int(k), L.ptr(v)))

    def get_coefficients(self, k):
        out = np.empty(3)
        self._check(self._lib.prcg_get_coefficients(self._h, int(k), L.ptr(out)))
        return out

    def history(self):
        """dict recorder-name -> array(max_iter) for the recorders of this session."""
        names = [q for q, bit in sorted(L.HIST_BITS.items(), key=lambda kv: kv[1]) if self.hist_mask & bit]
        if not names:
            return {}
        buf = np.zeros((len(names), self.max_iter))
        self._check(self._lib.prcg_get_history(self._h, L.ptr(buf)))
        return {q: buf[i].copy() for i, q in enumerate(names)}

    def set_profiling(self, stride):
        self._check(self._lib.prcg_set_profiling(self._h, int(stride)))

    def stream_ceiling(self, n_pairs, mode, reps=20):
        """GB/s the memory system delivers for a byte mix over arrays of n_pairs 16-byte entries (prcg_test.h:
        prcg_stream_ceiling): mode 0 pure read, 1 = 2 x 16 B in + 2 x 16 B out per row (the one-launch iteration's vector
        traffic) with plain stores, 2 with nontemporal stores, 3 pure read in contiguous 4 KB chunks per wave, nontemporal."""
        g = C.c_double()
        self._check(self._lib.prcg_stream_ceiling(self._h, int(n_pairs), int(mode), int(reps), C.byref(g)))
        return float(g.value)

    def mix_ceiling(self, n_rows, stream_kb_per_64_rows, reps=10):
        """GB/s the memory system delivers for the byte mix of a one-launch iteration that streams that many KB of operator per 64
        rows beside the rows' 64 B of vector traffic (prcg_test.h: prcg_mix_ceiling)."""
        g = C.c_double()
        self._check(self._lib.prcg_mix_ceiling(self._h, int(n_rows), int(stream_kb_per_64_rows), int(reps), C.byref(g)))
        return float(g.value)

    def timings(self):
        t = L.Timings()
        self._check(self._lib.prcg_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    def solve(self, variant, b, x0, max_iter, x_true=None, inv_diag=None, hist_mask=0):
        """One C call: begin + (max_iter-1) iterations + histories + x (prcg_solve)."""
        b, x0 = L.f64(b), L.f64(x0)
        xt = None if x_true is None else L.f64(x_true)
        dv = None if inv_diag is None else L.f64(inv_diag)
        nh = bin(hist_mask).count('1')
        hist = np.zeros((max(nh, 1), max_iter))
        x = np.empty(self.n)
        t = L.Timings()
        self._check(self._lib.prcg_solve(self._h, int(variant), L.ptr(b), L.ptr(x0), int(max_iter), L.ptr(xt),
                                         L.ptr(dv), int(hist_mask), L.ptr(hist) if nh else None, L.ptr(x),
                                         C.byref(t)))
        self.max_iter, self.hist_mask = int(max_iter), int(hist_mask)
        names = [q for q, bit in sorted(L.HIST_BITS.items(), key=lambda kv: kv[1]) if hist_mask & bit]
        return x, {q: hist[i].copy() for i, q in enumerate(names)}, t.as_dict()


def plan_tiles(indptr, row_class=None, cap_nnz=None, cap_rows=None):
    """Host-only view of the CSR-adaptive tiling (no GPU needed): returns
    (tiles[(row_begin,row_end)], n_class0)."""
    lib = L.lib()
    if cap_nnz is None or cap_rows is None:
        a, b = C.c_int(0), C.c_int(0)
        lib.prcg_tile_caps(C.byref(a), C.byref(b))
        cap_nnz = cap_nnz or a.value
        cap_rows = cap_rows or b.value
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    n = len(indptr) - 1
    rc = None if row_class is None else np.ascontiguousarray(row_class, dtype=np.uint8)
    cap = n + 1
    out = np.zeros((cap, 2), dtype=np.int32)
    n0 = C.c_int64(0)
    got = lib.prcg_plan_tiles(n, L.ptr(indptr), L.ptr(rc), int(cap_nnz), int(cap_rows), L.ptr(out), cap,
                              C.byref(n0))
    if got < 0:
        raise RuntimeError('prcg_plan_tiles failed')
    return out[:got].copy(), int(n0.value)
