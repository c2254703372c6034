/*
 * prcg.h -- C-ABI of libprcg.so: predict-and-recompute CG on AMD MI355X (gfx950).
 *
 * The reference (tchen-research/new_cg_variants) has no FFI layer: its hot path is
 * reached through two Python call signatures,
 *
 *   trial  = method(A, b, x0, max_iter, callbacks=..., x_true=..., preconditioner=...)
 *            numerical_experiments/figure_gen.py:59   (solvers: cg_variants/hs_cg.py:9,
 *            cg_variants/pipe_pr_cg.py:9-105,109-216, cg_variants/pr_cg.py:93-176)
 *   sol, t = variant(comm, A, b, max_iter)
 *            scaling_experiments_mpi4py/scaling_tests.py:71 (cg_variants/pipe_pr_cg.py:7,
 *            cg_variants/hs_cg.py:7)
 *
 * and everything below them is SciPy/NumPy/BLAS/MPI.  This header is the boundary a
 * maintainer binds instead (ctypes stub: INTEGRATION.md): plain pointers and sizes,
 * no Python or torch types.  One handle = one GPU = one rank (one process per GPU);
 * the caller owns every host buffer before and after each call, the library owns
 * all device memory, streams and the RCCL communicator.
 *
 * Return value: 0 on success, otherwise one of PRCG_E*; text via prcg_last_error().
 * Numerical breakdown is NOT an error: inf/nan land in the histories exactly as in
 * the reference (which never raises; figure_gen.py:89 uses nanmin).
 * A handle is not re-entrant; distinct handles may be used from distinct threads.
 */
#ifndef PRCG_H
#define PRCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prcg_handle prcg_t;

/* ---- error classes --------------------------------------------------------- */
#define PRCG_OK       0
#define PRCG_EINVAL   1   /* bad argument / call order                     */
#define PRCG_EHIP     2   /* HIP runtime error (no GPU, OOM, launch fault) */
#define PRCG_ERCCL    3   /* RCCL could not be loaded or returned an error */
#define PRCG_ENOMEM   4   /* host allocation failed                        */

/* ---- variants: which recurrence prcg_solve_begin sets up --------------------
 * names = the reference's function names, numerical_experiments/cg_variants/__init__.py:64-74 */
#define PRCG_HS          0  /* hs_cg / hs_pcg            hs_cg.py:9,70               */
#define PRCG_PIPE_PR     1  /* pipe_pr_cg / pipe_pr_pcg  pipe_pr_cg.py:89,201        */
#define PRCG_PIPE_P      2  /* pipe_p_cg / pipe_p_pcg    pipe_pr_cg.py:83,195        */
#define PRCG_PIPE_PR_M   3  /* pipe_pr_m_cg / _pcg       pipe_pr_cg.py:101,213       */
#define PRCG_PIPE_P_M    4  /* pipe_p_m_cg / _pcg        pipe_pr_cg.py:95,207        */
#define PRCG_PR          5  /* pr_pcg                    pr_cg.py:166                */
#define PRCG_M           6  /* m_pcg                     pr_cg.py:172                */
#define PRCG_CG_CG       7  /* cg_cg / cg_pcg  (Chronopoulos-Gear)   cg_cg.py:9,76      */
#define PRCG_GV          8  /* gv_cg / gv_pcg  (Ghysels-Vanroose)    gv_cg.py:9,93      */
#define PRCG_NUM_VARIANTS 9

/* ---- history recorders (bit mask), = the four callbacks of figure_gen.py:37 ---- */
#define PRCG_HIST_UPDATED_RESIDUAL_2_NORM  1u  /* callbacks/updated_residual_2_norm.py:40 */
#define PRCG_HIST_RESIDUAL_2_NORM          2u  /* callbacks/residual_2_norm.py:41         */
#define PRCG_HIST_ERROR_A_NORM             4u  /* callbacks/error_A_norm.py:47-48         */
#define PRCG_HIST_ERROR_2_NORM             8u  /* callbacks/error_2_norm.py:47-48         */
#define PRCG_HIST_ALL                     15u

/* ---- state vectors addressable by prcg_get_vector / prcg_set_vector ---------
 * (the reference's x_k, r_k, p_k, s_k, w_k, u_k and the tilde companions) */
#define PRCG_VEC_X   0
#define PRCG_VEC_R   1
#define PRCG_VEC_P   2
#define PRCG_VEC_S   3
#define PRCG_VEC_W   4
#define PRCG_VEC_U   5
#define PRCG_VEC_RT  6
#define PRCG_VEC_ST  7
#define PRCG_VEC_WT  8
#define PRCG_VEC_UT  9
#define PRCG_NUM_VECS 10

/* ---- per-iteration scalars (one row of prcg_get_scalars) ---------------------- */
#define PRCG_S_MU     0   /* p.s                                   */
#define PRCG_S_DELTA  1   /* r.s~   (cg_cg / gv: eta = w.r~)        */
#define PRCG_S_GAMMA  2   /* s~.s                                  */
#define PRCG_S_NU     3   /* r~.r                                  */
#define PRCG_S_RR     4   /* r.r   (= nu when unpreconditioned)    */
#define PRCG_S_RES2   5   /* |b - A x|^2        if recorded        */
#define PRCG_S_ERRA2  6   /* e'Ae, e = x-x_true if recorded        */
#define PRCG_S_ERR2   7   /* |e|^2              if recorded        */
#define PRCG_NUM_SCALARS 8

typedef struct prcg_timings {
    double tot_ms;          /* wall time of the last prcg_solve loop (host clock, synced) */
    double iter_ms;         /* tot_ms / iterations                                        */
    double spmv_ms;         /* mean device time of the SpMV/SpMM launch (HIP events)      */
    double update_ms;       /* mean device time of the fused vector-update launch         */
    int64_t spmv_samples;   /* number of launches spmv_ms averages over                   */
    int64_t iterations;     /* iterations timed                                           */
} prcg_timings;

/* ---- life cycle -------------------------------------------------------------- */
int  prcg_create(prcg_t** h, int device_id);
void prcg_destroy(prcg_t* h);
/* h may be NULL: text of the last error of a failed prcg_create / prcg_comm_unique_id */
const char* prcg_last_error(const prcg_t* h);
/* ABI version of this header */
int  prcg_version(void);

/* Experiment switches (the PRCG_* names of INTEGRATION.md, e.g. "PRCG_FUSED" = "0"): prcg_create
 * takes their defaults from the environment, this call sets one for ONE handle -- before
 * prcg_set_csr, which fixes the operator's encodings.  No counterpart in the reference. */
int prcg_set_option(prcg_t* h, const char* key, const char* value);

/* ---- multi-GPU: one RCCL communicator per handle --------------------------------
 * replaces comm = MPI.COMM_WORLD (scaling_tests.py:21) for the data path.
 * rccl_path: path of the librccl.so to dlopen (NULL -> "librccl.so.1").  Rank 0
 * calls prcg_comm_unique_id (once per id) and ships the 128-byte ids to the other
 * ranks by any means (torch.distributed in bench.py); then every rank calls
 * prcg_comm_init with n_ids = 1 or 2 consecutive ids.  With 1 id the neighbour exchange
 * and the all-reduce of an iteration form one chain on the communication stream; with 2
 * ids the halo exchange gets a communicator and a stream of its own and the two run
 * side by side. */
int prcg_comm_unique_id(const char* rccl_path, void* id128);
int prcg_comm_init(prcg_t* h, const char* rccl_path, int rank, int nranks, const void* ids, int n_ids);

/* Ghysels-Vanroose residual replacement (gv_cg.py:9,69-71; gv_pcg :93,156-158): `w_replace(k=..., x=..., w=..., r=..., ...)`
 * is the CALLER's predicate.  fn(ctx, k) is called inside prcg_iterate of a PRCG_GV session after x, r, (r~), w of
 * iteration k are updated and before t = A w~; it may read the state with prcg_get_vector (x, r, w are the new ones, p, s, u
 * the old ones -- what the reference passes); a non-zero return replaces w by A r (r, not r~, also with a preconditioner, as
 * the reference does) before the iteration goes on.  Sessions with a hook run the unfused schedule on one stream. */
typedef int (*prcg_replace_fn)(void* ctx, int k);
int prcg_set_replace_hook(prcg_t* h, prcg_replace_fn fn, void* ctx);

/* ---- direct peer exchange over xGMI ----------------------------------------------------
 * The reduction and the halo of the pipelined loop WITHOUT a collective (replaces, inside the loop, the
 * `comm.Allreduce` of scaling_experiments_mpi4py/cg_variants/pipe_pr_cg.py:67 and the neighbour exchange its dense
 * column blocks make implicit): every rank owns one exchange buffer in fine-grained device memory, mapped into every
 * other rank's process (hipIpc); the iteration launch stores the rows its neighbours need and its five partial inner
 * products straight into their buffers, and waits inside the kernel for theirs.  One host call
 * per iteration, no communication stream.  Call after prcg_set_csr / prcg_set_halo, on EVERY rank or on none:
 *   prcg_peer_setup    allocates this rank's buffer for `max_ghost_any_rank` ghost rows (the largest ghost count of any
 *                      rank: every buffer has the same layout), returns its 64-byte hipIpcMemHandle and, for ranks that
 *                      share the process (tests: ranks in threads), its device address;
 *   prcg_peer_connect  ipc_handles: nranks x 64 bytes, rank order (own entry ignored), or same_process_ptrs[q] non-null
 *                      for a rank of this process; send_dst_off[q]: where in peer q's ghost area (prcg_set_halo's
 *                      receive order of THAT rank) this rank's rows for it begin.
 * Window operators only (others keep the RCCL schedule); PRCG_PEER=0 turns it off.  Waits are bounded: a rank that
 * waits ~10 s for another sets an error that prcg_sync and the next prcg_iterate report. */
int prcg_peer_setup(prcg_t* h, int64_t max_ghost_any_rank, void* ipc_handle64, void** local_ptr);
int prcg_peer_connect(prcg_t* h, const void* ipc_handles, void* const* same_process_ptrs, const int64_t* send_dst_off);
/* ---- operator -------------------------------------------------------------------
 * The rank's row block in CSR (what `A` is in figure_gen.py:350 / scaling_tests.py:51),
 * column indices already LOCAL: [0,n_rows) = owned entries, [n_rows, n_rows+n_ghost)
 * = ghost entries received from peers, in the order fixed by prcg_set_halo.
 * indptr: int32 or int64 (indptr_is64), n_rows+1 entries; nnz < 2^31.  Indices need
 * not be sorted, but the sequential per-row summation order of SciPy's csr_matvec is
 * reproduced only in the order given.  Copies to the device and builds the
 * CSR-adaptive tile tables (interior rows / rows that touch ghosts). */
int prcg_set_csr(prcg_t* h, int64_t n_rows, int64_t n_ghost, int64_t nnz,
                 const void* indptr, int indptr_is64,
                 const int32_t* indices, const double* data);

/* Halo plan (needed iff n_ghost > 0).  Peer q = peer_rank[q]:
 *   send_idx[send_ptr[q] .. send_ptr[q+1])  local rows whose entries peer q needs
 *   ghost slots [recv_ptr[q], recv_ptr[q+1]) receive from peer q, in the peer's send order. */
int prcg_set_halo(prcg_t* h, int n_peers, const int32_t* peer_rank,
                  const int64_t* send_ptr, const int32_t* send_idx,
                  const int64_t* recv_ptr);

/* y = A x on the device, `reps` times (x: n_rows host doubles; ghosts exchanged when
 * n_ghost > 0).  ms_avg (nullable) = mean device time per launch by HIP events.
 * Replaces `A @ v` -> scipy _sparsetools.csr_matvec (hs_cg.py:23,26,59). */
int prcg_spmv(prcg_t* h, const double* x, double* y, int reps, double* ms_avg);
/* y = A_local [x_own ; x_ghost]: the row block's product with the ghost entries supplied by the
 * caller (n_rows + n_ghost host doubles) instead of by a halo exchange -- no communicator needed.
 * Runs the interior launch and the boundary launch of the overlapped schedule.  What one rank of
 * `A @ v` computes once its halo has arrived (scaling_experiments_mpi4py/cg_variants/hs_cg.py:49-51). */
int prcg_spmv_ext(prcg_t* h, const double* x_ext, double* y);
/* [w u] = A [r s], the fused two-vector product of the pipelined loop
 * (pipe_pr_cg.py:69-70; mpi4py pipe_pr_cg.py:65).  rs, wu: n_rows x 2 row-major. */
int prcg_spmm2(prcg_t* h, const double* rs, double* wu, int reps, double* ms_avg);

/* ---- solver session ----------------------------------------------------------------
 * begin: upload b, x0 (+ optional x_true, inv_diag), run the variant's initialisation
 *        (hs_cg.py:22-28 / pipe_pr_cg.py:22-36,122-140 / pr_cg.py:106-116) and record
 *        history index 0.  max_iter as in the reference: histories have max_iter
 *        entries, index 0 = initial state, at most max_iter-1 iterations follow.
 *        inv_diag != NULL selects the Jacobi-preconditioned recurrences with
 *        z = inv_diag * r (figure_gen.py:43); NULL = the unpreconditioned ones.
 * iterate: enqueue `iters` iterations on the device; returns without waiting.
 * sync: wait for everything enqueued. */
int prcg_solve_begin(prcg_t* h, int variant, const double* b, const double* x0,
                     int max_iter, const double* x_true, const double* inv_diag,
                     uint32_t hist_mask);
int prcg_iterate(prcg_t* h, int iters);
int prcg_sync(prcg_t* h);
/* current iteration index k (0 after begin) */
int prcg_iteration(const prcg_t* h);
/* which schedule the current session runs (valid after prcg_solve_begin): bit 0 one launch per
 * iteration (fused SpMM + update), bit 1 one-workgroup solver, bit 2 communicator present,
 * bit 3 merged exchange (halo rows ride on the one all-gather that carries the partial inner
 * products), bit 4 second halo communicator, bits 5..7 the operator's stream encodings (valid after
 * prcg_set_csr); bits 8..11 tile size in 256-slot steps.
 * No counterpart in the reference (diagnostics for tests and benchmarks). */
#define PRCG_SCHED_FUSED 1
#define PRCG_SCHED_SMALL 2
#define PRCG_SCHED_COMM 4
#define PRCG_SCHED_GATHER 8
#define PRCG_SCHED_DUAL_COMM 16
#define PRCG_SCHED_VALDICT 32   /* interior tiles stream 1-byte value-dictionary indices (lossless) */
#define PRCG_SCHED_COL8 64      /* ... and 1-byte tile-relative column offsets */
#define PRCG_SCHED_COL16 128    /* ... 2-byte */
#define PRCG_SCHED_FUSED_COMM 8192 /* one launch per iteration WITH a communicator: the interior launch waits in-kernel
                                      for the reduced inner products of the previous iteration */
#define PRCG_SCHED_SELL 32768  /* lane-per-row kernels over 64-row slices (medium-length rows: assembled FEM matrices) */
#define PRCG_SCHED_PEER 16384  /* ... through the direct peer exchange (prcg_peer_setup / prcg_peer_connect): no collective in the loop */
#define PRCG_SCHED_PATTERN 65536 /* ... window kernels over PATTERN tiles (constant-coefficient stencils): no per-nonzero stream at all --
                                    per tile one pattern record (slot offsets + values) and the rows' 16-bit slot masks */
#define PRCG_SCHED_STREAM_STORES 131072 /* the one-launch iteration writes its row results with nontemporal stores (vectors far larger than the caches) */
#define PRCG_SCHED_SELL_SORTED 262144  /* sliced rows with a sorting window wider than a slice (SELL-C-sigma: row lengths vary) */
#define PRCG_SCHED_NT_LOADS 524288     /* sliced rows: value / column-code streams read with nontemporal loads (operator far larger than the Infinity Cache) */
#define PRCG_SCHED_SELL_WINDOW 2097152 /* sliced rows with WINDOW codes: a slice's input entries staged in LDS, per nonzero an LDS read instead of a gather */
#define PRCG_SCHED_MEDIUM 1048576      /* mid-size system: the whole pipelined solve in one launch of a few co-operating workgroups */
#define PRCG_SCHED_WINDOW 4096  /* row-per-lane window kernels (bands, stencils): the column stream holds indices into the tile's
                                   LDS-staged window of the input vector */
/* A preconditioner that is not a diagonal scaling: `fn(ctx, n, v, out)` must write M^-1 v to out (host buffers,
 * n doubles each; return 0).  It stands for the `preconditioner` callable of the reference's *_pcg functions
 * (numerical_experiments/cg_variants/hs_cg.py:70, pr_cg.py:93, pipe_pr_cg.py:109, cg_cg.py:74, gv_cg.py:87), which
 * is the caller's code there too.  Sessions begun with inv_diag == NULL while a function is set call it wherever
 * the reference calls `preconditioner(...)`: the vector goes to the host, the result comes back -- two PCIe copies
 * and a stream synchronisation per application (1 per iteration for hs / cg_cg / gv / pr / m and the 'p' pipelined
 * flavours, 2 for the 'pr' pipelined flavours), on the schedules in which every tilde vector is a stored vector.
 * Single GPU.  fn == NULL removes it.  Jacobi stays on the device: pass inv_diag to prcg_solve_begin instead. */
typedef int (*prcg_prec_fn)(void* ctx, int64_t n, const double* v, double* out);
int prcg_set_preconditioner(prcg_t* h, prcg_prec_fn fn, void* ctx);
int prcg_schedule(const prcg_t* h);
/* Bytes of the operator AS THE DEVICE STREAMS IT (the lossless re-encodings of the caller's CSR built at
 * prcg_set_csr: narrow column / window indices, value-dictionary indices, relative row pointers, tile
 * descriptors; stream images that many tiles share are counted once): what one matrix product must read
 * from memory at least once.  bench.py's roofline is computed from this figure plus the vector traffic
 * (SURVEY.md section 8d gives the algorithmic figure for the caller's CSR: 12 B per nonzero + 4 B per row).
 * -1 without an operator.  No counterpart in the reference (scipy streams the CSR arrays as they are). */
int64_t prcg_operator_bytes(const prcg_t* h);
/* teacher forcing: declare that the state now loaded (prcg_set_vector / prcg_set_scalars
 * for iteration k) IS iteration k; the next prcg_iterate(h,1) produces k+1 */
int prcg_set_iteration(prcg_t* h, int k);

/* state access (synchronises first); vectors are n_rows host doubles */
int prcg_get_vector(prcg_t* h, int which, double* out);
int prcg_set_vector(prcg_t* h, int which, const double* in);
/* scalars of iteration k: PRCG_NUM_SCALARS doubles */
int prcg_get_scalars(prcg_t* h, int k, double* out);
int prcg_set_scalars(prcg_t* h, int k, const double* in);
/* coefficients used BY iteration k (k>=1): out[0]=alpha (a_k1), out[1]=beta (b_k),
 * out[2]=predicted nu */
int prcg_get_coefficients(prcg_t* h, int k, double* out);

/* histories: for each bit set in hist_mask (ascending bit order) max_iter doubles;
 * entries beyond the current iteration are 0, as numpy.zeros(max_iter) leaves them. */
int prcg_get_history(prcg_t* h, double* hist);
/* HIP-event sampling of the SpMV/SpMM and update launches inside prcg_iterate:
 * every `stride`-th iteration is bracketed (0 = off).  Read back with prcg_get_timings. */
int prcg_set_profiling(prcg_t* h, int stride);
int prcg_get_timings(prcg_t* h, prcg_timings* t);

/* one call = begin + (max_iter-1) iterations + histories + x; the drop-in for
 * `method(A,b,x0,max_iter,...)`.  hist: [popcount(hist_mask)][max_iter], nullable. */
int prcg_solve(prcg_t* h, int variant, const double* b, const double* x0, int max_iter,
               const double* x_true, const double* inv_diag, uint32_t hist_mask,
               double* hist, double* x_out, prcg_timings* t);

#ifdef __cplusplus
}
#endif
#endif /* PRCG_H */
