// libprcg.so engine: handle, operator upload, solver sessions, multi-GPU schedule.
// Implements include/prcg.h.  Host code only; kernels live in prcg_kernels.hip.
//
// Per-iteration schedule of the pipelined variants (the GPU form of the PETSc plug-in's
// VecDotBegin / PetscCommSplitReductionBegin / KSP_MatMult / VecDotEnd sequence,
// scaling_experiments_petsc/cg_impls/pipeprcg.c:154-173):
//
//   compute stream : [update k: x,r,p,s + dot partials] -E1-> [SpMM interior rows]
//                    -wait Eh-> [SpMM rows touching ghosts] -wait Er-> [update k+1]
//   comm stream    : wait E1 -> [pack halo] [ncclSend/Recv] -Eh-> [reduce partials]
//                    [ncclAllReduce, 5 doubles] -Er->
//
// i.e. ONE reduction per iteration, hidden behind the matrix product.  The host only
// enqueues; it never synchronises inside prcg_iterate.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <new>
#include <utility>
#include <string>
#include <array>
#include <vector>

#include "../../include/prcg.h"
#include "../../include/prcg_test.h"
#include "prcg_kernels.h"
#include "prcg_plan.h"
#include "prcg_rccl.h"

using namespace prcg;

namespace {

std::string g_last_error;   // errors with no handle to hang them on

constexpr int kNS = PRCG_NUM_SCALARS;
static_assert(kPartialStride == PRCG_NUM_SCALARS, "the small-system kernel indexes the scalar history with kPartialStride");
constexpr int kCoefStride = 4;
constexpr int kMaxProfSamples = 512;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;      // size asked for
    size_t cap = 0;        // size of the allocation
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; cap = 0; } }
    hipError_t alloc(size_t nbytes, bool zero = true) {
        release();
        if (nbytes == 0) nbytes = 16;
        hipError_t e = hipMalloc(&p, nbytes);
        if (e != hipSuccess) { p = nullptr; return e; }
        bytes = cap = nbytes;
        if (zero) {
            // the null stream does not order with our non-blocking streams: wait here
            e = hipMemset(p, 0, nbytes);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        }
        return e;
    }
    // Session state: keep the allocation of the previous solve when it is large enough (a sweep of
    // solves on one operator -- figure_gen.py:343-363 runs nine variants per matrix -- then allocates
    // once) and clear it on the stream the session works on: no hipMalloc, no host synchronisation.
    hipError_t ensure(size_t nbytes, hipStream_t st) {
        if (nbytes == 0) nbytes = 16;
        if (!p || cap < nbytes || cap > 2 * nbytes + (1u << 20)) {
            release();
            hipError_t e = hipMalloc(&p, nbytes);
            if (e != hipSuccess) { p = nullptr; return e; }
            cap = nbytes;
        }
        bytes = nbytes;
        return hipMemsetAsync(p, 0, nbytes, st);
    }
    double* d() const { return static_cast<double*>(p); }
    int* i() const { return static_cast<int*>(p); }
};

struct EventPair { hipEvent_t a = nullptr, b = nullptr; };

}  // namespace

struct prcg_handle {
    int dev = 0;
    std::string err;
    hipStream_t sc = nullptr;   // compute
    hipStream_t sm = nullptr;   // reductions + all-reduce
    hipStream_t sh = nullptr;   // halo exchange (own stream + own communicator: runs beside the all-reduce)
    hipEvent_t e_upd = nullptr, e_halo = nullptr, e_red = nullptr;
    hipEvent_t e_kdone = nullptr;   // completion signal of a launch itself (hipExtLaunchKernel), see FusedState::done
    hipEvent_t e_rdone = nullptr;   // ... of the unpack launch that ends the communication chain of an iteration

    // ---- communicator ----
    Rccl* rccl = nullptr;
    ncclComm_t comm = nullptr;    // all-reduce (and, if comm_halo is null, the halo too)
    ncclComm_t comm_halo = nullptr;
    int rank = 0, nranks = 1;

    // ---- operator ----
    bool have_csr = false;
    int64_t n = 0, g = 0, nnz = 0;
    DevBuf indptr, col, val, tiles;
    DevBuf col16, col8, tile_base;       // 16- / 8-bit tile-relative column encodings (see CsrDev)
    bool c16_int = false, c16_bnd = false;   // ... usable for all interior / all boundary tiles
    bool c8_int = false, c8_bnd = false;
    bool want_c16 = true;                // PRCG_COL16=0 turns it off
    bool want_c8 = true;                 // PRCG_COL8=0: never narrower than 16 bit
    DevBuf vidx8, vdict, vdesc;          // value dictionary (see CsrDev): 1-byte indices, entries, {first,count} per tile
    bool vd_int = false, vd_bnd = false;
    bool want_vdict = true;              // PRCG_VALDICT=0 turns it off
    int nt_int = 0, nt_bnd = 0;          // interior tiles first, then boundary tiles
    int steps = kDefaultTileSteps;       // tile size the table was planned for
    TileKnobs kn;                        // PRCG_GRID_PER_CU, PRCG_TILE_ORDER
    int steps_override = 0;              // PRCG_TILE_STEPS
    // ---- window tiles (row-per-lane kernels, prcg_win.hip): all tiles of the operator or none ----
    bool want_win = true;                // PRCG_WIN=0 turns them off
    int win_per_cu = 0;                  // PRCG_WIN_GRID_PER_CU
    int win_max_mean = 24;               // PRCG_WIN_MAX_MEAN: longest mean row the window form is tried for
    int win_rows_override = 0;           // PRCG_WIN_ROWS = 64 | 128: rows per window tile (default: by mean row length)
    bool win = false;
    int win_geom = 0, win_rows = 0;
    bool win_vd = false;
    bool win_pat = false;        // pattern tiles (geometry 5): no index streams, the rows' slot masks in wrel, records in wpat
    bool want_pat = true;        // PRCG_WIN_PAT=0: constant-coefficient stencils keep the stream geometries
    int want_sweep = 1;          // PRCG_WIN_SWEEP=0: pattern tiles in row order (no page carried from tile to tile); 2: sweep tables
                                 // for operators of any size (default: 5e6 rows and more -- below, the row order with big workgroups wins)
    int sweep_waves = 0, sweep_tiles = 0;
    int sweep_max_waves = 6144;  // PRCG_SWEEP_WAVES: most waves a sweep table may ask for
    DevBuf wpat;
    int nwt_int = 0, nwt_bnd = 0;
    DevBuf wtiles, wcw, wvidx, wvdict, wrel;
    // ---- sliced rows (lane-per-row kernels for medium-length rows, prcg_sell.hip): all rows of the operator or none ----
    bool want_sell = true;               // PRCG_SELL=0 turns them off
    int sell_per_cu = 0;                 // PRCG_SELL_GRID_PER_CU
    bool sell = false;
    int nst_int = 0, nst_bnd = 0;        // interior slices first, then slices touching ghost columns
    int64_t sell_bytes = 0;              // bytes of the re-laid operator a product reads
    DevBuf sval, scol, sslices, srows, sgran;
    int sell_window = 0;                 // > 0: WINDOW codes -- the kernels stage a slice's input entries in LDS (prcg_plan.h); the most granules of a slice
    int sell_window_opt = 64;            // PRCG_SELL_WINDOW=0 turns them off (delta codes, gathers from memory); 1..64: granules a slice may use
    double sell_overhead_opt = 0.0;      // PRCG_SELL_MAX_OVERHEAD_PCT: most padded nonzeros per nonzero, in percent (experiments)
    int sell_sigma_opt = 0;              // PRCG_SELL_SIGMA: sorting window of the sliced layout in rows (0: chosen by the planner)
    int sell_planes_opt = 0;             // PRCG_SELL_PLANES: grid planes interleaved in the slice table (<= 1: row order, the default:
                                         // interleaving 8 planes cost s4b at 80^3 nodes 9 % -- profiles/r04_sweeps.md)
    int sell_nt = 0;                     // the value / code streams are read with nontemporal loads: chosen per operator in prcg_set_csr
    int sell_nt_opt = -1;                // PRCG_SELL_NT=0|1 overrides
    int place_k = 8;                     // PRCG_PLACE=k: the pipelined session's vectors are placed k times and the fastest placement is kept (place_session_vectors); 0 / 1: off
    void* placed_xp = nullptr;           // ... the (x,p) allocation that has been through it
    int sell_gb = 0, sell_defer = 0;     // PRCG_SELL_GB=0|4|8, PRCG_SELL_DEFER=0|1: request orders inside the sliced-row kernels (prcg_sell.hip)
    int sell_sigma = 0, sell_planes = 0, sell_run = 1; // what the planner chose
    bool sell_runs_opt = true;           // PRCG_SELL_RUNS=0: a column code per nonzero even where the rows are runs of three
    int64_t sell_stride = 0;
    bool want_big = true;                // PRCG_WIN_BIG=0: short launches keep the small workgroups too
    int win_order = 0;                   // 1: XCD-chunked tile order of the window launches (opt-in: PRCG_WIN_ORDER=1)
    int win_order_override = -1;
    int win_period = 0;                  // tiles t and t + win_period read the same stream images (0: no such period found)
    bool want_share = true;              // PRCG_WIN_SHARE=0: every window tile keeps its own stream images
    int64_t win_stream_bytes = 0;        // bytes of the encoded operator a product must read at least once (window form)
    bool side_stream = false;            // one GPU: reduce the partials beside the SpMM (PRCG_SIDE_STREAM=1);
                                         // measured slower than in-order (cross-stream event waits ~15 us/iter)
    DevBuf tmp_ext;                      // 2*(n+g) doubles: SpMV input scratch with ghost room
    DevBuf t1;                           // 2*n doubles: SpMV output scratch
    DevBuf partA, partB;                 // block partials: update kernels / SpMV epilogues
    DevBuf ticket;                       // arrival counter of the fused final reduction
    bool fused_final = false;            // PRCG_FUSED_FINAL=1: last block of the update kernel reduces the
                                         // partials (correct, but its per-block release fence writes back every
                                         // XCD L2: update 140 -> 250 us at S3; kept as an experiment)

    // ---- halo plan ----
    int n_peers = 0;
    std::vector<int> peer_rank;
    std::vector<int64_t> send_ptr, recv_ptr;
    DevBuf send_idx, send_buf;
    bool have_halo = false;

    // ---- merged exchange: with a small halo, ONE all-gather per iteration carries the rank's five
    // partial inner products and the rows its neighbours need (pipelined variants) ----
    bool want_gather = true;             // PRCG_GATHER=0 turns it off (must agree on all ranks)
    int64_t gather_max_bytes = 8192;     // PRCG_GATHER_MAX_BYTES: largest per-rank slot that still rides along
    bool gather_planned = false, gather_ok = false;
    bool gather = false;                 // this session uses it
    int g_slot = 0;                      // doubles per rank slot: 8 + 2 * (largest send list of any rank)
    DevBuf gbuf, ghost_src, gtab;

    // ---- session ----
    bool in_session = false;
    int variant = -1;
    bool prec = false;
    int max_iter = 0;
    int k = 0;
    uint32_t hist_mask = 0;
    bool have_xtrue = false;
    DevBuf r2, s2, rt2, st2;     // one-launch predict-and-recompute: the second copies of r, s (r~, s~ with Jacobi)
    DevBuf x, xp, p, p2, rs, rs2, rst, rst2, wu, wt, wv, r, s, rt, st, b, xt, dinv, e_ext;
    DevBuf w, u, tvec;           // cg_cg / gv: w (ghost room), u, t = A w~
    bool fused = false;          // this session runs the one-launch-per-iteration pipelined kernel
    bool fused_comm = false;     // ... with a communicator: the interior launch waits in-kernel for the reduction
    bool want_fused_comm = true; // PRCG_FUSED_COMM=0: communicator sessions keep the two-kernel schedule
    bool ext_signal = true;      // PRCG_EXT_SIGNAL=0: separate hipEventRecord instead of the launch's own completion signal
    int defer_per_cu = 0;        // PRCG_DEFER_GRID_PER_CU: workgroups per CU of the deferred launch (several ranks sharing one
                                 // GPU in the tests must all be resident at once: 1)
    DevBuf pub, pub_err;         // publication record of the reduced inner products / timeout flag
    bool red_pending = false;    // the communication chain of the previous iteration is outstanding ...
    hipEvent_t red_event = nullptr;   // ... and this event marks its end
    bool want_fused = true;      // PRCG_FUSED=0 turns it off
    bool medium = false;         // this session runs the few-workgroup solver (prcg_medium.hip): mid-size systems
    bool want_medium = false;    // PRCG_MEDIUM=1 turns it on (opt-in: correct, but at ~5 us per grid-wide hand-off -- two per iteration -- it does not
                                 // yet beat one launch per iteration, 8-12 us: profiles/r04_sweeps.md)
    bool medium_ok = false;      // the operator has a medium plan (prcg_set_csr)
    int med_groups = 0, med_window = 0;
    unsigned long long med_seq = 0;
    DevBuf m_val, m_col, m_slices, m_rows, m_wave_first, m_window, m_own, m_exch, m_slots, m_err;
    bool small = false;          // this session runs the one-workgroup solver (n <= 4096)
    int small_mode = 0;          // 0: matrix in LDS, 1: matrix in registers
    bool small_hs = false;       // Hestenes-Stiefel session of a small system: the whole solve in one launch of one workgroup
    int max_row_len = 0;
    bool want_small = true;      // PRCG_SMALL=0 turns it off
    double* rs_cur = nullptr;    // fused: the current SpMM input pairs: rs / rs2 ((r,s)), with Jacobi rst / rst2 ((r~,s~))
    DevBuf partC;                // fused: second partials buffer (ping-pong with partB)
    // host-callback preconditioner (prcg_set_preconditioner): M^-1 v is computed by the caller's function on host
    // copies of v; sessions that use it run the schedules in which every tilde vector is a stored vector
    prcg_prec_fn cb = nullptr; void* cb_ctx = nullptr; bool cb_session = false;
    prcg_replace_fn replace_fn = nullptr; void* replace_ctx = nullptr;       // gv_cg's w_replace predicate (prcg_set_replace_hook)
    std::vector<double> cb_in, cb_out;
    DevBuf cb_stage, ut;         // staging for strided operands; u~ = M^-1 u of the pipelined variants
    bool pr_packed = false;      // ... unpreconditioned, no per-iteration vector recorder: inside a prcg_iterate call the state lives PACKED,
                                 // one 32-byte entry (z, zs, p, x) per row in q / q2 (kEpiPROneQ: 16-byte accesses only); packed at the
                                 // call's first iteration, unpacked into r, s, p, x at its end
    bool pr_q_valid = false;     // ... the packed copy q_cur is the current state
    int want_pr_pack = -1;       // PRCG_PR_PACK=0|1; default: pattern-tile operators only (S2: +7.6 %; S3 +-0; S2 with plain values -12 %)
    double* q_cur = nullptr;
    DevBuf q, q2;
    bool pr_fused = false;       // non-pipelined predict-and-recompute (pr, m) on a window operator: ONE launch per iteration
                                 // (window formed as (z - a zs) + b p_old); z, zs, p double-buffered:
    double* cur_r = nullptr; double* cur_s = nullptr; double* cur_rt = nullptr; double* cur_st = nullptr;
    double* cur_w = nullptr;     // ... Ghysels-Vanroose: w double-buffered the same way (w / w2)
    DevBuf w2;
    bool cg_fused = false;       // Chronopoulos-Gear / Ghysels-Vanroose on a window operator: two launches (product with the window formed as
                                 // r - a s; p, s update that sums the product's partials itself); r double-buffered (cur_r)
    bool cg_one = false;         // ... in ONE launch per iteration (launch_win_cg_one): the p, s (u) update of an iteration is deferred into the next
                                 // launch's window formation; s (cg) / t, u (gv) double-buffered as well
    bool cg_lag = false;         // ... and the update of iteration pend_k is pending (its partials: pend_buf / pend_parts)
    bool want_cg_one = true;     // PRCG_CG_ONE=0: the two-launch schedule
    double* cur_u = nullptr; double* cur_t = nullptr;      // (cur_s: above)
    DevBuf u2, t2;
    bool hs_fused = false;       // Hestenes-Stiefel without reduction launches: 2 launches per iteration on window
                                 // operators (update; product with the direction formed in the staged window), else 3
    double* p_cur = nullptr;     // ... the current direction: p / p2 (the product launch writes the other one)
    int hs_pend_mu = 0;          // ... mu of iteration pend_k exists only as this many block partials in partB
    int last_grid = 0;           // workgroups of the last one-launch iteration (prcg_debug_layout)
    int pend_parts = 0;          // fused: dots[pend_k] exist only as this many block partials ...
    int pend_k = -1;             // ... of iteration pend_k, in pend_buf
    double* pend_buf = nullptr;
    DevBuf dots, coef;

    // ---- profiling ----
    int prof_stride = 0;
    std::vector<EventPair> ev_spmv, ev_upd;
    int n_ev_spmv = 0, n_ev_upd = 0;
    double last_tot_ms = 0.0;
    int64_t last_iters = 0;

    // operator view for a launch over tiles [first, ...): interior launches pass first = 0,
    // boundary launches first = nt_int; a launch over ALL tiles needs both classes to qualify
    CsrDev csr(int first = 0, bool all = true) const {
        const bool boundary = first >= nt_int && nt_bnd > 0;
        const bool ok = all ? (c16_int && (nt_bnd == 0 || c16_bnd)) : (boundary ? c16_bnd : c16_int);
        const bool ok8 = all ? (c8_int && (nt_bnd == 0 || c8_bnd)) : (boundary ? c8_bnd : c8_int);
        const bool okv = all ? (vd_int && (nt_bnd == 0 || vd_bnd)) : (boundary ? vd_bnd : vd_int);
        return CsrDev{indptr.i(), col.i(), val.d(), ok ? static_cast<const unsigned short*>(col16.p) : nullptr,
                      ok8 ? static_cast<const unsigned char*>(col8.p) : nullptr,
                      (ok || ok8) ? static_cast<const int*>(tile_base.p) + first : nullptr,
                      okv ? static_cast<const unsigned char*>(vidx8.p) : nullptr,
                      okv ? static_cast<const double*>(vdict.p) : nullptr,
                      okv ? static_cast<const int2*>(vdesc.p) + first : nullptr};
    }
    const Tile* tile_ptr(int first = 0) const { return static_cast<const Tile*>(tiles.p) + first; }
    WinDev wdev() const {
        const bool b16 = win_geom >= 2;
        return WinDev{indptr.i(), val.d(), b16 ? nullptr : static_cast<const unsigned char*>(wcw.p),
                      b16 ? static_cast<const unsigned short*>(wcw.p) : nullptr,
                      win_vd ? static_cast<const unsigned char*>(wvidx.p) : nullptr,
                      win_vd ? static_cast<const double*>(wvdict.p) : nullptr,
                      static_cast<const unsigned short*>(wrel.p), static_cast<const PatRec*>(wpat.p), sweep_waves, sweep_tiles, want_big ? 1 : 0, win_order,
                      win_period};
    }
    const WTile* wtile_ptr(int first = 0) const { return static_cast<const WTile*>(wtiles.p) + first; }
    SellDev sdev() const { return SellDev{indptr.i(), val_sell(), static_cast<const unsigned short*>(scol.p), static_cast<const int*>(srows.p), sell_nt, sell_run, sell_gb, sell_defer, sell_window > 0 ? static_cast<const int*>(sgran.p) : nullptr, sell_window}; }
    const double* val_sell() const { return static_cast<const double*>(sval.p); }
    const void* sslice_ptr(int first = 0) const { return static_cast<const char*>(sslices.p) + (size_t)first * 32; }
    // any communicator -- even a 1-rank one -- selects the two-stream schedule
    bool multi() const { return comm != nullptr; }

    // ---- direct peer exchange over xGMI (PeerDev, prcg_kernels.h): the multi-rank one-launch schedule without a collective ----
    int stream_stores = 0;               // the one-launch iteration writes its row results with streaming stores: chosen per operator in
                                         // prcg_set_csr (vectors far larger than the 256 MB Infinity Cache), PRCG_STREAM_STORES=0|1 overrides
    int stream_override = -1;
    int want_fused_comm_rccl = 0;        // PRCG_FUSED_COMM=1: one launch per iteration with the RCCL all-gather chain on the communication
                                         // stream (in-kernel wait for kernels of another stream: validated with one rank only -- opt-in)
    bool want_peer = true;               // PRCG_PEER=0: never use the peer exchange even when connected
    void* xbuf = nullptr;                // this rank's exchange buffer
    size_t xbuf_bytes = 0;
    bool xbuf_fine = false;              // allocated fine-grained (what other GPUs' stores need); else plain device memory (one GPU)
    int ghost_cap = 0;                   // ghost rows per parity in EVERY rank's buffer (largest ghost count of any rank)
    std::vector<void*> peer_opened;      // other ranks' buffers opened with hipIpcOpenMemHandle
    PeerDev peer_host{};
    DevBuf peer_dev, peer_ents, peer_tile_send;
    bool peer_ok = false;                // connected: every rank's buffer is mapped, send entries planned
    bool peer = false;                   // this session uses it
    uint64_t peer_epoch = 0;
    unsigned* err_host = nullptr;        // pinned host word a timed-out wave also writes: prcg_iterate sees it without a sync
    std::vector<int32_t> send_idx_host;  // prcg_set_halo's send lists (host copy: the peer plan is built from them)
    std::vector<int32_t> wt_rb, wt_re;   // rows of the window tiles in table order

    bool debug_short_sources = false;    // PRCG_DEBUG_SHORT_SOURCES=1 (TESTS ONLY): Hestenes-Stiefel sessions allocate r without
                                         // the spare entries a window source needs -- the launch must be refused, not fault
    // every vector a window launch may stage (the pointer handed to the launch lies inside one of them)
    const DevBuf* owner(const void* ptr) const {
        const DevBuf* all[] = {&tmp_ext, &t1, &x, &xp, &p, &p2, &rs, &rs2, &rst, &rst2, &wu, &wt, &wv, &r, &r2, &s, &s2, &rt, &rt2,
                               &st, &st2, &b, &xt, &dinv, &e_ext, &w, &w2, &u, &u2, &tvec, &t2, &ut, &cb_stage, &q, &q2};
        const char* c = static_cast<const char*>(ptr);
        for (const DevBuf* d : all)
            if (d->p && c >= static_cast<const char*>(d->p) && c < static_cast<const char*>(d->p) + d->bytes) return d;
        return nullptr;
    }
};

namespace {

int fail(prcg_t* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_last_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                      \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(h, PRCG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

#define NCCLCHK(h, expr)                                                                      \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess)                                                               \
            return fail(h, PRCG_ERCCL, "%s failed: %s (%s:%d)", #expr,                        \
                        (h)->rccl->GetErrorString(r__), __FILE__, __LINE__);                  \
    } while (0)

#define LAUNCHCHK(h, grid)                                                                    \
    do {                                                                                      \
        const int g__ = (grid);                                                               \
        if (g__ == -2) return PRCG_EINVAL;        /* refused by check_sources: message set */  \
        if (g__ < 0) return fail(h, PRCG_EHIP, "kernel launch failed (%s:%d)", __FILE__, __LINE__); \
    } while (0)

#define CHECK(h, cond, ...)                                                                   \
    do { if (!(cond)) return fail(h, PRCG_EINVAL, __VA_ARGS__); } while (0)

// Tile size (256-nnz steps per wave tile).  Short rows (stencils: 5-7 nonzeros) do better
// with 256-slot tiles -- more rows per lane would otherwise serialise the reduce phase
// (measured: S1 34.3 k vs 31.3 k it/s, S2 3236 vs 3013) -- longer rows with 512-slot tiles
// (S3, 15 per row: 2115 vs 1885 it/s).  PRCG_TILE_STEPS = 1 | 2 | 4 overrides.
int pick_tile_steps(int override_, int64_t n = 0, int64_t nnz = 0) {
    if (override_ == 1 || override_ == 2 || override_ == 4) return override_;
    if (n > 0 && nnz < 10 * n) return 1;
    // medium rows (FEM-like, ~50+ nonzeros): the lane-per-row sums are a serial chain per row, so
    // bigger tiles (more rows summed side by side per wave) win (s4b: 2983 vs 2740 it/s)
    if (n > 0 && nnz >= 48 * n) return 4;
    return kDefaultTileSteps;
}

// A window launch stages whole 64-column pages of its source vectors and the narrow column encodings of the
// CSR-adaptive kernels decode a few out-of-tile bytes per tile: every such source must hold its n + g entries AND
// kGatherPad spare ones behind them (prcg_window_source_ok).  Checked at every launch from the size of the
// allocation the pointer lies in -- a short source is an error code, not a memory fault.
int check_sources(prcg_t* h, std::initializer_list<std::pair<const void*, int>> srcs) {
    for (const auto& sc : srcs) {
        if (!sc.first) continue;
        const DevBuf* d = h->owner(sc.first);
        const int64_t avail = d ? (int64_t)(static_cast<const char*>(d->p) + d->bytes - static_cast<const char*>(sc.first)) : -1;
        if (!d || !prcg_window_source_ok(h->n, h->g, sc.second, avail))
            return fail(h, PRCG_EINVAL, "a matrix-product source vector holds %lld bytes, fewer than the %lld a launch over %lld rows + "
                        "%lld ghosts may touch (%d components, %d spare entries): launch refused",
                        (long long)avail, (long long)((h->n + h->g + kGatherPad) * (int64_t)sc.second * 8), (long long)h->n,
                        (long long)h->g, sc.second, kGatherPad);
    }
    return PRCG_OK;
}
#define SRCCHK(h, ...) do { if (check_sources(h, {__VA_ARGS__})) return -2; } while (0)              // inside a launch helper (returns a grid)
#define SRCCHK2(h, ...) do { if (check_sources(h, {__VA_ARGS__})) return PRCG_EINVAL; } while (0)   // inside an iterate_* function

// The matrix products of the engine: window kernels when the operator qualified, else the
// CSR-adaptive tile kernels.  which: 0 = every tile, 1 = interior tiles, 2 = tiles touching ghosts.
int eng_spmv(prcg_t* h, hipStream_t st, int which, const double* x, double* y, SpmvEpilogue epi, const double* ep_r,
             const double* ep_d, double* ep_st, double* partials) {
    SRCCHK(h, {x, 1});
    if (h->win) {
        const int first = which == 2 ? h->nwt_int : 0;
        const int nt = which == 0 ? h->nwt_int + h->nwt_bnd : (which == 1 ? h->nwt_int : h->nwt_bnd);
        return launch_win_spmv(st, h->wdev(), h->wtile_ptr(first), nt, h->win_geom, x, y, epi, ep_r, ep_d, ep_st, partials,
                               h->win_per_cu);
    }
    if (h->sell) {
        const int first = which == 2 ? h->nst_int : 0;
        const int nt = which == 0 ? h->nst_int + h->nst_bnd : (which == 1 ? h->nst_int : h->nst_bnd);
        return launch_sell_spmv(st, h->sdev(), h->sslice_ptr(first), nt, x, y, epi, ep_r, ep_d, ep_st, partials, h->sell_per_cu);
    }
    const int first = which == 2 ? h->nt_int : 0;
    const int nt = which == 0 ? h->nt_int + h->nt_bnd : (which == 1 ? h->nt_int : h->nt_bnd);
    const CsrDev A = which == 0 ? h->csr() : (which == 1 ? h->csr(0, h->nt_bnd == 0) : h->csr(h->nt_int, false));
    return launch_spmv(st, A, h->tile_ptr(first), nt, h->steps, x, y, epi, ep_r, ep_d, ep_st, partials, h->kn);
}
int eng_spmm2(prcg_t* h, hipStream_t st, int which, const double* rs, double* wu, int mask) {
    SRCCHK(h, {rs, 2});
    if (h->win) {
        const int first = which == 2 ? h->nwt_int : 0;
        const int nt = which == 0 ? h->nwt_int + h->nwt_bnd : (which == 1 ? h->nwt_int : h->nwt_bnd);
        return launch_win_spmm2(st, h->wdev(), h->wtile_ptr(first), nt, h->win_geom, rs, wu, mask, h->win_per_cu);
    }
    if (h->sell) {
        const int first = which == 2 ? h->nst_int : 0;
        const int nt = which == 0 ? h->nst_int + h->nst_bnd : (which == 1 ? h->nst_int : h->nst_bnd);
        return launch_sell_spmm2(st, h->sdev(), h->sslice_ptr(first), nt, rs, wu, mask, h->sell_per_cu);
    }
    const int first = which == 2 ? h->nt_int : 0;
    const int nt = which == 0 ? h->nt_int + h->nt_bnd : (which == 1 ? h->nt_int : h->nt_bnd);
    const CsrDev A = which == 0 ? h->csr() : (which == 1 ? h->csr(0, h->nt_bnd == 0) : h->csr(h->nt_int, false));
    return launch_spmm2(st, A, h->tile_ptr(first), nt, h->steps, rs, wu, mask, h->kn);
}
int eng_fused(prcg_t* h, hipStream_t st, const FusedState& f, int which = 0) {
    SRCCHK(h, {f.in_old, 2});
    if (h->win) {
        const int first = which == 2 ? h->nwt_int : 0;
        const int nt = which == 0 ? h->nwt_int + h->nwt_bnd : (which == 1 ? h->nwt_int : h->nwt_bnd);
        return launch_win_pipe_fused(st, h->wdev(), h->wtile_ptr(first), nt, h->win_geom, f,
                                     f.deferred ? h->defer_per_cu : h->win_per_cu);
    }
    if (h->sell) return launch_sell_pipe_fused(st, h->sdev(), h->sslice_ptr(0), h->nst_int + h->nst_bnd, f, h->sell_per_cu);
    return launch_pipe_fused(st, h->csr(), h->tile_ptr(), h->nt_int + h->nt_bnd, h->steps, f, h->kn);
}

bool is_pipe(int v) { return v == PRCG_PIPE_PR || v == PRCG_PIPE_P || v == PRCG_PIPE_PR_M || v == PRCG_PIPE_P_M; }
bool is_pr(int v) { return v == PRCG_PR || v == PRCG_M; }
bool is_cg_family(int v) { return v == PRCG_CG_CG || v == PRCG_GV; }
bool pipe_recompute(int v) { return v == PRCG_PIPE_PR || v == PRCG_PIPE_PR_M; }
bool meurant(int v) { return v == PRCG_PIPE_PR_M || v == PRCG_PIPE_P_M || v == PRCG_M; }

// ---- halo exchange of an nc-component extended vector, all on `st` ----------------------
int exchange(prcg_t* h, double* vec_ext, int nc, hipStream_t st) {
    // a rank takes part in the exchange if it has ghosts OR if a peer needs its rows (with a
    // pattern-symmetric operator the two coincide, but the plan decides, not the pattern)
    if (!h->multi()) return PRCG_OK;
    CHECK(h, h->g == 0 || h->have_halo, "matrix has ghost columns but prcg_set_halo was not called");
    if (!h->have_halo || h->n_peers == 0) return PRCG_OK;
    // the halo communicator is used on its own stream only
    ncclComm_t cm = (h->comm_halo && st == h->sh) ? h->comm_halo : h->comm;
    const int64_t nsend = h->send_ptr[h->n_peers];
    launch_pack(st, h->send_buf.d(), vec_ext, h->send_idx.i(), nsend, nc);
    NCCLCHK(h, h->rccl->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int q = 0; q < h->n_peers && bad == ncclSuccess; ++q) {
        const int64_t ns = h->send_ptr[q + 1] - h->send_ptr[q];
        const int64_t nr = h->recv_ptr[q + 1] - h->recv_ptr[q];
        if (ns > 0)
            bad = h->rccl->Send(h->send_buf.d() + h->send_ptr[q] * nc, (size_t)(ns * nc), ncclDouble, h->peer_rank[q], cm, st);
        if (nr > 0 && bad == ncclSuccess)
            bad = h->rccl->Recv(vec_ext + (h->n + h->recv_ptr[q]) * nc, (size_t)(nr * nc), ncclDouble, h->peer_rank[q], cm, st);
    }
    // never leave the group open: a failed send/recv still gets its GroupEnd before the error is reported
    const ncclResult_t fin = h->rccl->GroupEnd();
    if (bad != ncclSuccess) return fail(h, PRCG_ERCCL, "ncclSend/ncclRecv failed: %s", h->rccl->GetErrorString(bad));
    NCCLCHK(h, fin);
    return PRCG_OK;
}

int allreduce(prcg_t* h, double* buf, int count, hipStream_t st) {
    if (!h->multi()) return PRCG_OK;
    NCCLCHK(h, h->rccl->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, h->comm, st));
    return PRCG_OK;
}

// Collective (every rank calls it, from prcg_solve_begin): is the halo small enough to ride
// on the reduction?  If so, learn where in each neighbour's slot this rank's ghost rows lie:
// the ranks all-gather their (peer, offset, count) send tables once.
int plan_gather(prcg_t* h) {
    h->gather = false;
    if (!h->multi() || !h->want_gather || h->fused_final) return PRCG_OK;
    if (h->gather_planned) { h->gather = h->gather_ok; return PRCG_OK; }
    h->gather_planned = true;
    h->gather_ok = false;
    const int np = h->have_halo ? h->n_peers : 0;
    const int64_t nsend = np > 0 ? h->send_ptr[np] : 0;
    const int R = h->nranks;
    // (a) largest send list / peer count of any rank
    HIPCHK(h, h->gtab.alloc(2 * sizeof(double)));
    double mine[2] = {(double)nsend, (double)np}, mx[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(h->gtab.p, mine, sizeof mine, hipMemcpyHostToDevice, h->sc));
    NCCLCHK(h, h->rccl->AllReduce(h->gtab.p, h->gtab.p, 2, ncclDouble, ncclMax, h->comm, h->sc));
    HIPCHK(h, hipMemcpyAsync(mx, h->gtab.p, sizeof mx, hipMemcpyDeviceToHost, h->sc));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    const int64_t max_send = (int64_t)mx[0];
    const int max_peers = (int)mx[1];
    const int64_t slot = 8 + 2 * max_send;
    if (slot * (int64_t)sizeof(double) > h->gather_max_bytes) return PRCG_OK;      // same verdict on every rank
    // (b) everybody's send table: [n_peers, (peer, first row of the list, rows) ...]
    const int T = 1 + 3 * max_peers;
    std::vector<double> tab((size_t)R * T, 0.0);
    double* my = tab.data() + (size_t)h->rank * T;
    my[0] = np;
    for (int q = 0; q < np; ++q) {
        my[1 + 3 * q] = h->peer_rank[q];
        my[2 + 3 * q] = (double)h->send_ptr[q];
        my[3 + 3 * q] = (double)(h->send_ptr[q + 1] - h->send_ptr[q]);
    }
    HIPCHK(h, h->gtab.alloc((size_t)R * T * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(h->gtab.d() + (size_t)h->rank * T, my, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->sc));
    NCCLCHK(h, h->rccl->AllGather(h->gtab.d() + (size_t)h->rank * T, h->gtab.p, (size_t)T, ncclDouble, h->comm, h->sc));
    HIPCHK(h, hipMemcpyAsync(tab.data(), h->gtab.p, tab.size() * sizeof(double), hipMemcpyDeviceToHost, h->sc));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    // (c) ghost j  <-  pair index into the gathered buffer
    std::vector<int32_t> src((size_t)h->g + 1, 0);
    const int bad = plan_gather_sources(h->rank, T, tab.data(), np, h->peer_rank.data(), h->recv_ptr.data(), slot, src.data());
    CHECK(h, bad == 0, "halo plans disagree: rank %d sends no list of the expected length to rank %d (or index overflow)",
          bad > 0 ? h->peer_rank[bad - 1] : -1, h->rank);
    HIPCHK(h, h->ghost_src.alloc(src.size() * sizeof(int32_t)));
    HIPCHK(h, hipMemcpy(h->ghost_src.p, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(h, h->gbuf.alloc((size_t)R * slot * sizeof(double)));
    HIPCHK(h, hipMemset(h->gbuf.p, 0, (size_t)R * slot * sizeof(double)));
    h->g_slot = (int)slot;
    h->gather_ok = true;
    h->gather = true;
    return PRCG_OK;
}

// y = A x (x extended, ghosts exchanged), everything on the compute stream.  Used by the
// initialisation, the recorders and prcg_spmv -- not by the timed loop.
int dist_spmv(prcg_t* h, double* x_ext, double* y, SpmvEpilogue epi, const double* ep_r,
              const double* ep_d, double* ep_st, int* grid_out) {
    int rc = exchange(h, x_ext, 1, h->sc);
    if (rc) return rc;
    const int grid = eng_spmv(h, h->sc, 0, x_ext, y, epi, ep_r, ep_d, ep_st, h->partB.d());
    LAUNCHCHK(h, grid);
    if (grid_out) *grid_out = grid;
    return PRCG_OK;
}

// the iterate x: second-class citizen of the pair array XP in the pipelined variants
const double* x_ptr(prcg_t* h) { return is_pipe(h->variant) ? h->xp.d() : h->x.d(); }
int x_stride(prcg_t* h) { return is_pipe(h->variant) ? 2 : 1; }

double* dots_at(prcg_t* h, int k) { return h->dots.d() + (size_t)k * kNS; }
double* coef_at(prcg_t* h, int k) { return h->coef.d() + (size_t)k * kCoefStride; }

void fused_flush(prcg_t* h);
void pr_unpack(prcg_t* h);
void hs_flush(prcg_t* h);
void cg_flush(prcg_t* h);
int apply_prec(prcg_t* h, const double* src, int sstride, double* dst, int dstride);

// ---- history recorders for the state of iteration k (compute stream) -------------------
int record(prcg_t* h, int k) {
    const uint32_t m = h->hist_mask;
    if (!(m & (PRCG_HIST_RESIDUAL_2_NORM | PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM))) return PRCG_OK;
    if (h->fused && !h->fused_comm) fused_flush(h);    // the recorders reuse the partials buffers
    if (h->hs_fused) hs_flush(h);
    if (h->pr_fused) { fused_flush(h); pr_unpack(h); }
    if (h->cg_one) cg_flush(h);
    if (h->fused_comm && h->red_pending) HIPCHK(h, hipStreamWaitEvent(h->sc, h->red_event, 0));
    if (h->peer && h->pend_parts > 0) {
        // (peer exchange: the last launch's partials are still to be sent by the next launch's communication wave -- the
        //  recorders reuse the buffer, so the slot goes out now)
        launch_peer_collect(h->sc, static_cast<const PeerDev*>(h->peer_dev.p), h->pend_k, h->pend_buf, h->pend_parts, dots_at(h, h->pend_k),
                            h->pub.d(), static_cast<unsigned*>(h->pub_err.p));
        h->pend_parts = 0;
    }
    const int64_t n = h->n;
    int rc;
    if (m & PRCG_HIST_RESIDUAL_2_NORM) {
        // |b - A x|   callbacks/residual_2_norm.py:41
        launch_copy(h->sc, h->tmp_ext.d(), 1, x_ptr(h), x_stride(h), n);
        if ((rc = dist_spmv(h, h->tmp_ext.d(), h->t1.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;
        const int grid = launch_diff_sq(h->sc, h->b.d(), 1, h->t1.d(), n, h->partA.d(), PRCG_S_RES2);
        LAUNCHCHK(h, grid);
        launch_reduce_final(h->sc, h->partA.d(), grid, dots_at(h, k), PRCG_S_RES2, PRCG_S_RES2, 1);
    }
    if (m & PRCG_HIST_ERROR_2_NORM) {
        // |x - x_true|   callbacks/error_2_norm.py:47-48
        const int grid = launch_diff_sq(h->sc, x_ptr(h), x_stride(h), h->xt.d(), n, h->partA.d(), PRCG_S_ERR2);
        LAUNCHCHK(h, grid);
        launch_reduce_final(h->sc, h->partA.d(), grid, dots_at(h, k), PRCG_S_ERR2, PRCG_S_ERR2, 1);
    }
    if (m & PRCG_HIST_ERROR_A_NORM) {
        // e = x - x_true; e'(A e)   callbacks/error_A_norm.py:47-48
        launch_sub(h->sc, h->e_ext.d(), 1, x_ptr(h), x_stride(h), h->xt.d(), 1, n);
        int grid = 0;
        if ((rc = dist_spmv(h, h->e_ext.d(), h->t1.d(), kEpiDotXY, nullptr, nullptr, nullptr, &grid))) return rc;
        launch_reduce_final(h->sc, h->partB.d(), grid, dots_at(h, k), 0, PRCG_S_ERRA2, 1);
    }
    return allreduce(h, dots_at(h, k) + PRCG_S_RES2, 3, h->sc);
}

void prof_begin(prcg_t* h, std::vector<EventPair>& evs, int& count, int k, bool& on) {
    on = h->prof_stride > 0 && (k % h->prof_stride) == 0 && count < kMaxProfSamples;
    if (!on) return;
    if ((int)evs.size() <= count) {
        EventPair ep;
        if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) { on = false; return; }
        evs.push_back(ep);
    }
    (void)hipEventRecord(evs[count].a, h->sc);
}
void prof_end(prcg_t* h, std::vector<EventPair>& evs, int& count, bool on) {
    if (!on) return;
    (void)hipEventRecord(evs[count].b, h->sc);
    ++count;
}

// SpMM of the pipelined loop.  The fixed-order reduction of the update kernel's block
// partials (and, with a communicator, the halo exchange and the one all-reduce) run on
// the communication stream, hidden behind the matrix product on the compute stream.
// `grid_upd` = number of partial blocks the update kernel just wrote to partA.
int pipe_spmm_and_reduce(prcg_t* h, int k, int grid_upd, bool profile) {
    double* in_ext = h->prec ? h->rst.d() : h->rs.d();
    const int mask = pipe_recompute(h->variant) ? 3 : 2;
    bool on = false;
    int rc;
    if (!h->side_stream && !h->multi()) {
        // everything in order on one stream
        if (!h->fused_final) launch_reduce_final(h->sc, h->partA.d(), grid_upd, dots_at(h, k), 0, 0, 5);
        if (profile) prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
        LAUNCHCHK(h, eng_spmm2(h, h->sc, 0, in_ext, h->wu.d(), mask));
        if (profile) prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        if (h->cb_session) {
            // u~ = preconditioner(u); w~ = preconditioner(w) in the flavours that recompute w (pipe_pr_cg.py:178-182)
            if ((rc = apply_prec(h, h->wu.d() + 1, 2, h->ut.d(), 1))) return rc;
            if (pipe_recompute(h->variant) && (rc = apply_prec(h, h->wu.d(), 2, h->wt.d(), 1))) return rc;
        }
        return PRCG_OK;
    }
    if (h->gather) {
        // small halo: pack (fixed-order reduction of the partials + the neighbours' rows) ->
        // ONE all-gather -> unpack (sums in rank order, ghost rows into place), all beside the
        // interior product
        const int np = h->have_halo ? h->n_peers : 0;
        double* slot = h->gbuf.d() + (size_t)h->rank * h->g_slot;
        HIPCHK(h, hipEventRecord(h->e_upd, h->sc));
        HIPCHK(h, hipStreamWaitEvent(h->sm, h->e_upd, 0));
        launch_gather_pack(h->sm, h->partA.d(), grid_upd, slot, in_ext, h->send_idx.i(), np > 0 ? (int)h->send_ptr[np] : 0);
        NCCLCHK(h, h->rccl->AllGather(slot, h->gbuf.p, (size_t)h->g_slot, ncclDouble, h->comm, h->sm));
        launch_gather_unpack(h->sm, h->gbuf.d(), h->g_slot, h->nranks, dots_at(h, k), in_ext + 2 * h->n,
                             h->ghost_src.i(), (int)h->g);
        HIPCHK(h, hipEventRecord(h->e_red, h->sm));
        if (profile) prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
        LAUNCHCHK(h, eng_spmm2(h, h->sc, 1, in_ext, h->wu.d(), mask));
        if (profile) prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        HIPCHK(h, hipStreamWaitEvent(h->sc, h->e_red, 0));
        if (h->nt_bnd > 0) LAUNCHCHK(h, eng_spmm2(h, h->sc, 2, in_ext, h->wu.d(), mask));
        return PRCG_OK;
    }
    const bool halo = h->multi() && h->have_halo && h->n_peers > 0;
    HIPCHK(h, hipEventRecord(h->e_upd, h->sc));
    // With a halo communicator of its own the neighbour exchange runs on its own stream,
    // beside the reduction chain; with ONE communicator everything is one chain on the
    // reduction stream (halo first: the boundary rows need it before the next update needs
    // the inner products).
    hipStream_t hs = h->comm_halo ? h->sh : h->sm;
    HIPCHK(h, hipStreamWaitEvent(h->sm, h->e_upd, 0));
    if (halo) {
        if (hs != h->sm) HIPCHK(h, hipStreamWaitEvent(hs, h->e_upd, 0));
        if ((rc = exchange(h, in_ext, 2, hs))) return rc;
        HIPCHK(h, hipEventRecord(h->e_halo, hs));
    }
    // block partials -> 5 doubles -> the one all-reduce
    if (!h->fused_final) launch_reduce_final(h->sm, h->partA.d(), grid_upd, dots_at(h, k), 0, 0, 5);
    if ((rc = allreduce(h, dots_at(h, k), 5, h->sm))) return rc;
    HIPCHK(h, hipEventRecord(h->e_red, h->sm));

    if (profile) prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    LAUNCHCHK(h, eng_spmm2(h, h->sc, 1, in_ext, h->wu.d(), mask));
    if (profile) prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    if (halo) {
        HIPCHK(h, hipStreamWaitEvent(h->sc, h->e_halo, 0));
        LAUNCHCHK(h, eng_spmm2(h, h->sc, 2, in_ext, h->wu.d(), mask));
    }
    HIPCHK(h, hipStreamWaitEvent(h->sc, h->e_red, 0));
    return PRCG_OK;
}

// single-vector SpMV with an epilogue, interior rows overlapped with the halo of x_ext.
// partials of the two launches are laid end to end in partB; *nparts = their count.
int overlapped_spmv(prcg_t* h, int k, double* x_ext, double* y, SpmvEpilogue epi, const double* ep_r,
                    const double* ep_d, double* ep_st, int* nparts) {
    bool on = false;
    if (!h->multi()) {
        prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
        const int grid = eng_spmv(h, h->sc, 0, x_ext, y, epi, ep_r, ep_d, ep_st, h->partB.d());
        LAUNCHCHK(h, grid);
        prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        *nparts = grid;
        return PRCG_OK;
    }
    hipStream_t hs = h->comm_halo ? h->sh : h->sm;
    HIPCHK(h, hipEventRecord(h->e_upd, h->sc));
    HIPCHK(h, hipStreamWaitEvent(hs, h->e_upd, 0));
    int rc = exchange(h, x_ext, 1, hs);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->e_halo, hs));
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int g1 = eng_spmv(h, h->sc, 1, x_ext, y, epi, ep_r, ep_d, ep_st, h->partB.d());
    LAUNCHCHK(h, g1);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    HIPCHK(h, hipStreamWaitEvent(h->sc, h->e_halo, 0));
    const int g2 = eng_spmv(h, h->sc, 2, x_ext, y, epi, ep_r, ep_d, ep_st, h->partB.d() + (size_t)g1 * kPartialStride);
    LAUNCHCHK(h, g2);
    *nparts = g1 + g2;
    return PRCG_OK;
}

PipeUpdateArgs pipe_args(prcg_t* h, int k) {
    PipeUpdateArgs a{};
    a.n = h->n;
    a.xp = h->xp.d();
    a.rs = h->rs.d(); a.rst = h->prec ? h->rst.d() : nullptr;
    a.wu = h->wu.d(); a.wt = h->wt.d();
    a.d = (h->prec && !h->cb_session) ? h->dinv.d() : nullptr;
    a.ut = h->cb_session ? h->ut.d() : nullptr;
    a.dots_prev = k > 0 ? dots_at(h, k - 1) : dots_at(h, 0);
    a.coef_out = coef_at(h, k);
    a.partials = h->partA.d();
    a.final_out = h->fused_final ? dots_at(h, k) : nullptr;
    a.ticket = static_cast<unsigned*>(h->ticket.p);
    a.meurant = meurant(h->variant);
    a.recompute_w = pipe_recompute(h->variant);
    return a;
}

// One launch per iteration (single GPU, all four pipelined flavours, with or without Jacobi): state k-1 =
// {XP, the SpMM input pairs in rs_cur, dots[k-1]} (+ (r,s) with Jacobi, + w (w~) for the 'p' flavours); the
// kernel computes A [in.x in.y] row by row and applies update k to the row at once.  The reduction of the inner products is NOT
// overlapped with anything here (it is needed at the head of the next launch) -- which is
// why ranks with a communicator keep the two-kernel schedule below.
// make dots[pend_k] real if the last fused launch left it as block partials
void fused_flush(prcg_t* h) {
    if (h->pend_parts > 0) {
        launch_reduce_final(h->sc, h->pend_buf, h->pend_parts, dots_at(h, h->pend_k), 0, 0, 5);
        h->pend_parts = 0;
    }
}

int iterate_pipe_fused(prcg_t* h, int k) {
    double* const bufA = h->prec ? h->rst.d() : h->rs.d();
    double* const bufB = h->prec ? h->rst2.d() : h->rs2.d();
    double* rs_old = h->rs_cur;
    double* rs_new = (h->rs_cur == bufA) ? bufB : bufA;
    // partials of iteration k-1 (if still pending) are consumed by this launch's prologue
    FusedPrev prev{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (h->pend_parts > 0 && h->pend_k == k - 1) {
        prev = FusedPrev{h->pend_buf, h->pend_parts, dots_at(h, k - 1), nullptr, nullptr, nullptr};
    } else {
        fused_flush(h);
    }
    double* part_out = (h->pend_buf == h->partB.d()) ? h->partC.d() : h->partB.d();
    bool on = false;
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const bool rec = pipe_recompute(h->variant);
    FusedState f{};
    f.in_old = rs_old; f.in_new = rs_new; f.xp = h->xp.d();
    f.rs = h->prec ? h->rs.d() : nullptr;
    f.dinv = h->prec ? h->dinv.d() : nullptr;
    f.w = rec ? nullptr : h->wv.d();
    f.wt = (!rec && h->prec) ? h->wt.d() : nullptr;
    f.dots_prev = dots_at(h, k - 1); f.coef_out = coef_at(h, k); f.partials = part_out;
    f.meurant = meurant(h->variant); f.recompute_w = rec;
    f.stream_stores = h->stream_stores;
    f.prev = prev;
    const int grid = eng_fused(h, h->sc, f);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    h->pend_parts = grid; h->pend_k = k; h->pend_buf = part_out; h->last_grid = grid;
    h->rs_cur = rs_new;
    return PRCG_OK;
}

// One launch per iteration WITH a communicator (window operators): the GPU form of
// VecDotBegin ... KSP_MatMult ... VecDotEnd (scaling_experiments_petsc/cg_impls/pipeprcg.c:154-173).
//
//   compute stream : [launch k: every wave: products of its first interior tiles | wait pub >= k-1 | their updates |
//                     remaining tiles fused, the tiles that touch ghost rows last] [launch k+1 ...          (back to back)
//   comm stream    : wait "launch k done" -> [pack partials + rows] [ncclAllGather] [unpack: dots[k], ghosts of
//                     (r,s)_k into place, release, publish k]
//
// The reduction AND the halo of iteration k travel while launch k+1 computes A [r s] on interior rows -- the overlap
// the two-kernel schedule has, without storing (w,u), without a separate update launch, without a second launch for
// the boundary rows and without any wait of the compute stream on the communication stream: the only
// synchronisation the compute stream sees is inside the kernel.
int iterate_pipe_fused_comm(prcg_t* h, int k) {
    double* const bufA = h->prec ? h->rst.d() : h->rs.d();
    double* const bufB = h->prec ? h->rst2.d() : h->rs2.d();
    double* in_old = h->rs_cur;
    double* in_new = (h->rs_cur == bufA) ? bufB : bufA;
    double* part_out = (k & 1) ? h->partB.d() : h->partC.d();
    const bool rec = pipe_recompute(h->variant);
    FusedState f{};
    f.in_old = in_old; f.in_new = in_new; f.xp = h->xp.d();
    f.rs = h->prec ? h->rs.d() : nullptr;
    f.dinv = h->prec ? h->dinv.d() : nullptr;
    f.w = rec ? nullptr : h->wv.d();
    f.wt = (!rec && h->prec) ? h->wt.d() : nullptr;
    f.dots_prev = dots_at(h, k - 1); f.coef_out = coef_at(h, k); f.partials = part_out;
    f.meurant = meurant(h->variant); f.recompute_w = rec;
    // (the deferred form gains nothing from streaming stores where the plain schedule gains 20 % -- S3 on one rank 143.7 against
    //  142.7 us -- and loses 13 % where the vectors just exceed the Infinity Cache -- one half of S3: 78.1 against 68.0 us;
    //  profiles/r04_sweeps.md: plain stores unless PRCG_STREAM_STORES=1 asks)
    f.stream_stores = h->stream_override > 0 ? 1 : 0;
    f.prev = FusedPrev{};
    f.prev.pub = h->pub.d(); f.prev.want = (unsigned)(k - 1); f.prev.err = static_cast<unsigned*>(h->pub_err.p);
    f.deferred = 1;
    f.prev.nt_int = h->nwt_int;
    // ONE launch over all tiles: the boundary tiles come last in the table and are touched only after the wave has
    // seen the publication (which the ghost rows precede).  The launch signals the communication stream with its own
    // completion (no marker packet on the compute stream: the next iteration's launch follows directly).
    const bool ext_signal = h->ext_signal && !h->peer;
    f.done = ext_signal ? h->e_kdone : nullptr;
    if (h->peer) {
        // direct peer exchange: this launch is the whole iteration -- its tiles send the neighbours' rows, its last workgroup
        // this rank's partial sums, and its workgroup 0 turns the ranks' slots of iteration k-1 into the publication
        f.prev.px = static_cast<const PeerDev*>(h->peer_dev.p);
        f.prev.dots_prev_out = dots_at(h, k - 1);
        f.prev.err_host = h->err_host;
        // the partial sums launch k-1 left are summed and SENT by this launch's communication wave (none pending: the
        // slot of iteration k-1 is already out -- session start, teacher forcing, or the end of the last prcg_iterate call)
        if (h->pend_parts > 0 && h->pend_k == k - 1) { f.prev.prev_partials = h->pend_buf; f.prev.nprev = h->pend_parts; }
    }
    bool on = false;
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int g1 = eng_fused(h, h->sc, f, 0);
    LAUNCHCHK(h, g1);
    h->last_grid = g1;
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    if (h->peer) {
        h->pend_parts = g1; h->pend_k = k; h->pend_buf = part_out;
        h->rs_cur = in_new;
        return PRCG_OK;
    }
    const int g2 = 0;
    if (ext_signal) {
        HIPCHK(h, hipStreamWaitEvent(h->sm, h->e_kdone, 0));
    } else {
        HIPCHK(h, hipEventRecord(h->e_upd, h->sc));
        HIPCHK(h, hipStreamWaitEvent(h->sm, h->e_upd, 0));
    }
    const int nparts = g1 + g2;
    int rc;
    if (h->gather) {
        const int np = h->have_halo ? h->n_peers : 0;
        double* slot = h->gbuf.d() + (size_t)h->rank * h->g_slot;
        launch_gather_pack(h->sm, part_out, nparts, slot, in_new, h->send_idx.i(), np > 0 ? (int)h->send_ptr[np] : 0);
        NCCLCHK(h, h->rccl->AllGather(slot, h->gbuf.p, (size_t)h->g_slot, ncclDouble, h->comm, h->sm));
        launch_gather_unpack(h->sm, h->gbuf.d(), h->g_slot, h->nranks, dots_at(h, k), in_new + 2 * h->n, h->ghost_src.i(),
                             (int)h->g, h->pub.d(), (unsigned)k, ext_signal ? h->e_rdone : nullptr);
        if (!ext_signal) HIPCHK(h, hipEventRecord(h->e_red, h->sm));
        h->red_event = ext_signal ? h->e_rdone : h->e_red;
    } else {
        if (h->have_halo && h->n_peers > 0 && (rc = exchange(h, in_new, 2, h->sm))) return rc;
        launch_reduce_final(h->sm, part_out, nparts, dots_at(h, k), 0, 0, 5);
        if ((rc = allreduce(h, dots_at(h, k), 5, h->sm))) return rc;
        launch_publish(h->sm, dots_at(h, k), h->pub.d(), (unsigned)k);
        HIPCHK(h, hipEventRecord(h->e_red, h->sm));
        h->red_event = h->e_red;
    }
    h->red_pending = true;
    h->rs_cur = in_new;
    return PRCG_OK;
}

int iterate_pipe(prcg_t* h, int k) {
    // pipe_pr_cg.py:61-75 / :169-187
    if (h->fused_comm) return iterate_pipe_fused_comm(h, k);
    if (h->fused) return iterate_pipe_fused(h, k);
    PipeUpdateArgs a = pipe_args(h, k);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    const int grid = launch_pipe_update(h->sc, a);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    return pipe_spmm_and_reduce(h, k, grid, true);
}

HsArgs hs_args(prcg_t* h, int k) {
    HsArgs a{};
    a.n = h->n;
    a.x = h->x.d(); a.r = h->r.d(); a.rt = h->prec ? h->rt.d() : nullptr;
    a.p = h->p_cur; a.s = h->s.d(); a.d = (h->prec && !h->cb_session) ? h->dinv.d() : nullptr;
    a.dots_prev = k > 0 ? dots_at(h, k - 1) : dots_at(h, 0);
    a.dots_cur = dots_at(h, k);
    a.coef_out = coef_at(h, k);
    a.partials = h->partA.d();
    return a;
}

int iterate_hs(prcg_t* h, int k) {
    // hs_cg.py:54-61: two dependent reductions per iteration -- nothing to overlap them with
    HsArgs a = hs_args(h, k);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    const int g1 = launch_hs_update_xr(h->sc, a);
    LAUNCHCHK(h, g1);
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    launch_reduce_final(h->sc, h->partA.d(), g1, dots_at(h, k), PRCG_S_NU, PRCG_S_NU, 2);
    int rc;
    if (h->cb_session) {
        // r~ = preconditioner(r) on the host (hs_pcg :118), then nu = r~.r as an inner product of its own
        if ((rc = apply_prec(h, h->r.d(), 1, h->rt.d(), 1))) return rc;
        const int g2 = launch_dot(h->sc, h->rt.d(), h->r.d(), h->n, h->partB.d(), 0);
        LAUNCHCHK(h, g2);
        launch_reduce_final(h->sc, h->partB.d(), g2, dots_at(h, k), 0, PRCG_S_NU, 1);
    }
    rc = allreduce(h, dots_at(h, k) + PRCG_S_NU, 2, h->sc);              // reduction 1: nu
    if (rc) return rc;
    LAUNCHCHK(h, launch_hs_update_p(h->sc, a));
    int nparts = 0;
    if ((rc = overlapped_spmv(h, k, h->p_cur, h->s.d(), kEpiDotXY, nullptr, nullptr, nullptr, &nparts))) return rc;
    launch_reduce_final(h->sc, h->partB.d(), nparts, dots_at(h, k), 0, PRCG_S_MU, 1);
    return allreduce(h, dots_at(h, k) + PRCG_S_MU, 1, h->sc);           // reduction 2: mu
}

// make mu of iteration pend_k real if the last product launch left it as block partials
void hs_flush(prcg_t* h) {
    if (h->hs_pend_mu > 0) {
        launch_reduce_final(h->sc, h->partB.d(), h->hs_pend_mu, dots_at(h, h->pend_k), 0, PRCG_S_MU, 1);
        h->hs_pend_mu = 0;
    }
}

// hs_cg.py:54-62 (hs_pcg :116-125) on one GPU without reduction launches.  The two inner products still separate
// the iteration into two phases (that IS Hestenes-Stiefel), but each phase's block partials are summed by every
// workgroup of the NEXT launch (same fixed order everywhere):
//   launch 1  [mu_k1 from the product's partials; a = nu_k1 / mu_k1]  x += a p;  r -= a s;  (r~ = d r);  nu_k partials
//   launch 2  [nu_k from launch 1's partials; b = nu_k / nu_k1]  window operators: p = z + b p_old formed while the
//             input window is staged, s = A p, mu_k partials, p_k written to the other direction buffer;
//             CSR-adaptive tiles: the p update (with that prologue) and the product are two launches.
int iterate_hs_fused(prcg_t* h, int k) {
    HsArgs a = hs_args(h, k);
    const bool have_mu = h->hs_pend_mu > 0 && h->pend_k == k - 1;
    if (!have_mu) hs_flush(h);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    const int g1 = launch_hs_update_xr(h->sc, a, have_mu ? h->partB.d() : nullptr, have_mu ? h->hs_pend_mu : 0, dots_at(h, k - 1));
    LAUNCHCHK(h, g1);
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    h->hs_pend_mu = 0;
    int grid = 0;
    on = false;
    if (h->win) {
        FusedPrev hs{};
        hs.prev_partials = h->partA.d(); hs.nprev = g1;
        hs.dots_prev_out = dots_at(h, k); hs.dots_old = dots_at(h, k - 1);
        double* p_new = (h->p_cur == h->p.d()) ? h->p2.d() : h->p.d();
        SRCCHK2(h, {h->prec ? h->rt.d() : h->r.d(), 1}, {h->p_cur, 1});
        prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
        grid = launch_win_hs(h->sc, h->wdev(), h->wtile_ptr(0), h->nwt_int + h->nwt_bnd, h->win_geom,
                             h->prec ? h->rt.d() : h->r.d(), h->p_cur, p_new, h->s.d(), h->partB.d(), coef_at(h, k), hs,
                             h->win_per_cu);
        LAUNCHCHK(h, grid);
        prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        h->p_cur = p_new;
    } else {
        LAUNCHCHK(h, launch_hs_update_p(h->sc, a, h->partA.d(), g1, dots_at(h, k)));
        prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
        grid = eng_spmv(h, h->sc, 0, h->p_cur, h->s.d(), kEpiDotXY, nullptr, nullptr, nullptr, h->partB.d());
        LAUNCHCHK(h, grid);
        prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    }
    h->hs_pend_mu = grid; h->pend_k = k;
    return PRCG_OK;
}

// pr_cg.py:146-158 (pr_pcg, m_pcg) in ONE launch per iteration on a window operator (launch_win_pr_one): the
// partials of launch k-1 are summed by every workgroup of launch k (as in iterate_pipe_fused), a and b follow,
// the staged window of the new direction is formed from the old r~ (r), s~ (s), p, and the row's own vectors and
// the five partials are written by the lane that summed the row.  r~ / s~ / p (r / s / p without Jacobi) are
// double-buffered: other tiles still stage the old values.
int iterate_pr_fused(prcg_t* h, int k) {
    const bool jac = h->prec;
    double*& z_cur = jac ? h->cur_rt : h->cur_r;
    double*& zs_cur = jac ? h->cur_st : h->cur_s;
    double* z_a = jac ? h->rt.d() : h->r.d();   double* z_b = jac ? h->rt2.d() : h->r2.d();
    double* zs_a = jac ? h->st.d() : h->s.d();  double* zs_b = jac ? h->st2.d() : h->s2.d();
    double* z_new = (z_cur == z_a) ? z_b : z_a;
    double* zs_new = (zs_cur == zs_a) ? zs_b : zs_a;
    double* p_new = (h->p_cur == h->p.d()) ? h->p2.d() : h->p.d();
    FusedPrev f{};
    if (h->pend_parts > 0 && h->pend_k == k - 1) {
        f.prev_partials = h->pend_buf; f.nprev = h->pend_parts; f.dots_prev_out = dots_at(h, k - 1);
    } else {
        fused_flush(h);
    }
    f.dots_old = dots_at(h, k - 1);
    f.pr.z_old = z_cur; f.pr.zs_old = zs_cur; f.pr.p_old = h->p_cur;
    f.pr.z_new = z_new; f.pr.zs_new = zs_new; f.pr.p_new = p_new;
    f.pr.x = h->x.d();
    f.pr.r = jac ? h->r.d() : nullptr; f.pr.s = jac ? h->s.d() : nullptr; f.pr.d = jac ? h->dinv.d() : nullptr;
    if (h->pr_packed) {
        // the state as 16-byte pairs (z, zs) and (p, x): packed once per prcg_iterate call (pr_unpack at its end); each of q / q2
        // holds the (z, zs) pairs with ghost room and spare entries, then the (p, x) pairs
        const size_t half = (size_t)2 * (h->n + h->g + kGatherPad);
        if (!h->pr_q_valid) {
            h->q_cur = h->q.d();
            launch_copy(h->sc, h->q_cur + 0, 2, z_cur, 1, h->n);
            launch_copy(h->sc, h->q_cur + 1, 2, zs_cur, 1, h->n);
            launch_copy(h->sc, h->q_cur + half + 0, 2, h->p_cur, 1, h->n);
            launch_copy(h->sc, h->q_cur + half + 1, 2, h->x.d(), 1, h->n);
            h->pr_q_valid = true;
        }
        double* q_other = (h->q_cur == h->q.d()) ? h->q2.d() : h->q.d();
        f.pr.q_old = h->q_cur; f.pr.px_old = h->q_cur + half;
        f.pr.q_new = q_other; f.pr.px_new = q_other + half;
    }
    double* part_out = (h->pend_buf == h->partB.d()) ? h->partC.d() : h->partB.d();
    bool on = false;
    if (h->pr_packed) SRCCHK2(h, {f.pr.q_old, 2}, {f.pr.px_old, 2});
    else SRCCHK2(h, {f.pr.z_old, 1}, {f.pr.zs_old, 1}, {f.pr.p_old, 1});
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int grid = launch_win_pr_one(h->sc, h->wdev(), h->wtile_ptr(0), h->nwt_int + h->nwt_bnd, h->win_geom, f,
                                       (meurant(h->variant) ? 1 : 0) | (h->stream_stores ? 2 : 0), part_out, coef_at(h, k), h->win_per_cu);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    h->pend_parts = grid; h->pend_k = k; h->pend_buf = part_out; h->last_grid = grid;
    if (h->pr_packed) h->q_cur = f.pr.q_new;
    else { z_cur = z_new; zs_cur = zs_new; h->p_cur = p_new; }
    return PRCG_OK;
}

// end of a prcg_iterate call of a packed predict-and-recompute session: the packed entries back into r, s, p, x
void pr_unpack(prcg_t* h) {
    if (!h->pr_packed || !h->pr_q_valid) return;
    const size_t half = (size_t)2 * (h->n + h->g + kGatherPad);
    launch_copy(h->sc, h->cur_r, 1, h->q_cur + 0, 2, h->n);
    launch_copy(h->sc, h->cur_s, 1, h->q_cur + 1, 2, h->n);
    launch_copy(h->sc, h->p_cur, 1, h->q_cur + half + 0, 2, h->n);
    launch_copy(h->sc, h->x.d(), 1, h->q_cur + half + 1, 2, h->n);
    h->pr_q_valid = false;
}

PrArgs pr_args(prcg_t* h, int k) {
    PrArgs a{};
    a.n = h->n;
    a.x = h->x.d(); a.r = h->r.d(); a.rt = h->prec ? h->rt.d() : nullptr;
    a.p = h->p.d(); a.s = h->s.d(); a.st_ = h->prec ? h->st.d() : nullptr;
    a.dots_prev = k > 0 ? dots_at(h, k - 1) : dots_at(h, 0);
    a.coef_out = coef_at(h, k);
    a.partials = h->partA.d();
    a.meurant = meurant(h->variant);
    a.precond = h->prec;
    return a;
}

int pr_spmv_and_reduce(prcg_t* h, int k, int grid_upd) {
    // s = A p; s~ = M^-1 s; mu, delta, gamma ride on the SpMV (pr_cg.py:152-157)
    int nparts = 0;
    int rc;
    if (h->cb_session) {
        // s = A p; s~ = preconditioner(s) on the host; mu, delta, gamma as inner products of their own
        if ((rc = overlapped_spmv(h, k, h->p.d(), h->s.d(), kEpiNone, nullptr, nullptr, nullptr, &nparts))) return rc;
        if ((rc = apply_prec(h, h->s.d(), 1, h->st.d(), 1))) return rc;
        int g = launch_dot(h->sc, h->p.d(), h->s.d(), h->n, h->partB.d(), 0);
        LAUNCHCHK(h, g);
        LAUNCHCHK(h, launch_dot(h->sc, h->r.d(), h->st.d(), h->n, h->partB.d(), 1));
        LAUNCHCHK(h, launch_dot(h->sc, h->st.d(), h->s.d(), h->n, h->partB.d(), 2));
        launch_reduce_final(h->sc, h->partA.d(), grid_upd, dots_at(h, k), PRCG_S_NU, PRCG_S_NU, 2);
        launch_reduce_final(h->sc, h->partB.d(), g, dots_at(h, k), 0, 0, 3);
        return allreduce(h, dots_at(h, k), 5, h->sc);
    }
    rc = overlapped_spmv(h, k, h->p.d(), h->s.d(), kEpiPR, h->r.d(), h->prec ? h->dinv.d() : nullptr,
                         h->prec ? h->st.d() : nullptr, &nparts);
    if (rc) return rc;
    launch_reduce_final(h->sc, h->partA.d(), grid_upd, dots_at(h, k), PRCG_S_NU, PRCG_S_NU, 2);
    launch_reduce_final(h->sc, h->partB.d(), nparts, dots_at(h, k), 0, 0, 3);
    return allreduce(h, dots_at(h, k), 5, h->sc);                       // the one reduction
}

int iterate_pr(prcg_t* h, int k) {
    PrArgs a = pr_args(h, k);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    const int grid = launch_pr_update(h->sc, a);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    return pr_spmv_and_reduce(h, k, grid);
}

CgArgs cg_args(prcg_t* h, int k) {
    CgArgs a{};
    a.n = h->n;
    a.x = h->x.d(); a.r = h->cur_r; a.rt = h->prec ? h->rt.d() : nullptr;
    a.w = h->cur_w; a.wt = h->prec ? h->wt.d() : nullptr;
    a.p = h->p.d(); a.s = h->cur_s;
    a.st_ = (h->prec && h->variant == PRCG_GV) ? h->st.d() : nullptr;
    a.u = h->variant == PRCG_GV ? h->cur_u : nullptr;
    a.t = h->cur_t;
    a.z = h->prec ? h->rt.d() : h->cur_r;
    a.d = (h->prec && !h->cb_session) ? h->dinv.d() : nullptr;
    a.dots_prev = k > 0 ? dots_at(h, k - 1) : dots_at(h, 0);
    a.dots_cur = dots_at(h, k);
    a.dots_cur_w = dots_at(h, k);
    a.coef_out = coef_at(h, k);
    a.partials = h->partA.d();
    return a;
}

// Chronopoulos-Gear on a window operator in TWO launches (was four): the product launch forms its window as the new
// residual r - a s (times d), writes x, r, r~, w and the partials of eta, nu; the p, s update sums them in its prologue
// (same tree as k_reduce_final), derives b and mu and leaves the complete scalars of iteration k behind.
int iterate_cgcg_fused(prcg_t* h, int k) {
    double* r_new = (h->cur_r == h->r.d()) ? h->r2.d() : h->r.d();
    FusedPrev f{};
    f.dots_old = dots_at(h, k - 1);
    f.pr.z_old = h->cur_r; f.pr.zs_old = h->s.d(); f.pr.p_old = h->p.d();
    f.pr.z_new = r_new; f.pr.zs_new = h->prec ? h->rt.d() : nullptr;
    f.pr.x = h->x.d(); f.pr.d = h->prec ? h->dinv.d() : nullptr;
    bool on = false;
    SRCCHK2(h, {f.pr.z_old, 1}, {f.pr.zs_old, 1}, {f.pr.d, 1});
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int grid = launch_win_cg_w(h->sc, h->wdev(), h->wtile_ptr(0), h->nwt_int + h->nwt_bnd, h->win_geom, f, h->w.d(),
                                     h->partB.d(), coef_at(h, k), h->win_per_cu);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    h->cur_r = r_new;
    on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    LAUNCHCHK(h, launch_cg_update_ps(h->sc, cg_args(h, k), h->partB.d(), grid));
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    return PRCG_OK;
}

// Ghysels-Vanroose in two launches, the same way: the window is the new w formed as w - a u (times d).
int iterate_gv_fused(prcg_t* h, int k) {
    double* w_new = (h->cur_w == h->w.d()) ? h->w2.d() : h->w.d();
    FusedPrev f{};
    f.dots_old = dots_at(h, k - 1);
    f.pr.z_old = h->cur_w; f.pr.zs_old = h->u.d(); f.pr.p_old = h->p.d();
    f.pr.z_new = w_new; f.pr.zs_new = h->prec ? h->wt.d() : nullptr;
    f.pr.x = h->x.d(); f.pr.r = h->r.d(); f.pr.s = h->s.d();
    f.pr.d = h->prec ? h->dinv.d() : nullptr;
    f.pr.rt = h->prec ? h->rt.d() : nullptr; f.pr.st = h->prec ? h->st.d() : nullptr;
    bool on = false;
    SRCCHK2(h, {f.pr.z_old, 1}, {f.pr.zs_old, 1}, {f.pr.d, 1});
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int grid = launch_win_gv_w(h->sc, h->wdev(), h->wtile_ptr(0), h->nwt_int + h->nwt_bnd, h->win_geom, f, h->tvec.d(),
                                     h->partB.d(), coef_at(h, k), h->win_per_cu);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    h->cur_w = w_new;
    on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    LAUNCHCHK(h, launch_cg_update_ps(h->sc, cg_args(h, k), h->partB.d(), grid));
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    return PRCG_OK;
}

// close the iteration whose p, s (u) update is still pending (one-launch Chronopoulos-Gear / Ghysels-Vanroose): what the
// next launch would do while forming its window, as a launch of its own -- at the end of a prcg_iterate call, before a recorder
void cg_flush(prcg_t* h) {
    if (!h->cg_lag) return;
    if (h->prec) launch_mul(h->sc, h->rt.d(), 1, h->dinv.d(), 1, h->cur_r, 1, h->n);        // r~ = M^-1 r (the launches keep it in registers)
    (void)launch_cg_update_ps(h->sc, cg_args(h, h->pend_k), h->pend_buf, h->pend_parts);
    h->cg_lag = false;
    h->pend_parts = 0;
}

// ONE launch per iteration of Chronopoulos-Gear / Ghysels-Vanroose on a window operator (launch_win_cg_one): the launch of
// iteration k closes iteration k-1 in its prologue (b, mu, a from that launch's partials) and applies its p, s (u) update
// while forming the window -- the new residual (new w) expressed in old vectors.
int iterate_cg_one(prcg_t* h, int k) {
    const bool gv = h->variant == PRCG_GV;
    if (h->cg_lag && h->pend_k != k - 1) cg_flush(h);
    FusedPrev f{};
    if (h->cg_lag) {
        f.prev_partials = h->pend_buf; f.nprev = h->pend_parts;
        f.dots_prev_out = dots_at(h, k - 1); f.dots_old = dots_at(h, k - 2); f.lag.coef_prev = coef_at(h, k - 1);
    } else {
        f.dots_old = dots_at(h, k - 1);
    }
    auto other = [](double* cur, DevBuf& a, DevBuf& b) { return cur == a.d() ? b.d() : a.d(); };
    double *z0n, *z1n, *z2n;
    if (gv) {
        f.lag.z0 = h->cur_w; f.lag.z1 = h->cur_t; f.lag.z2 = h->cur_u;
        z0n = other(h->cur_w, h->w, h->w2); z1n = other(h->cur_t, h->tvec, h->t2); z2n = other(h->cur_u, h->u, h->u2);
        f.lag.r = h->r.d(); f.lag.s = h->s.d();
    } else {
        f.lag.z0 = h->cur_r; f.lag.z1 = h->cur_w; f.lag.z2 = h->cur_s;
        z0n = other(h->cur_r, h->r, h->r2); z1n = other(h->cur_w, h->w, h->w2); z2n = other(h->cur_s, h->s, h->s2);
        f.lag.d = h->prec ? h->dinv.d() : nullptr;
    }
    f.lag.z0n = z0n; f.lag.z1n = z1n; f.lag.z2n = z2n;
    f.lag.x = h->x.d(); f.lag.p = h->p.d();
    double* part_out = (h->pend_buf == h->partB.d()) ? h->partC.d() : h->partB.d();
    SRCCHK2(h, {f.lag.z0, 1}, {f.lag.z1, 1}, {f.lag.z2, 1}, {f.lag.d, 1});
    bool on = false;
    prof_begin(h, h->ev_spmv, h->n_ev_spmv, k, on);
    const int grid = launch_win_cg_one(h->sc, h->wdev(), h->wtile_ptr(0), h->nwt_int + h->nwt_bnd, h->win_geom, f, gv ? 1 : 0, part_out,
                                       coef_at(h, k), h->win_per_cu);
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
    if (gv) { h->cur_w = z0n; h->cur_t = z1n; h->cur_u = z2n; }
    else { h->cur_r = z0n; h->cur_w = z1n; h->cur_s = z2n; }
    h->pend_parts = grid; h->pend_k = k; h->pend_buf = part_out; h->cg_lag = true;
    return PRCG_OK;
}

// Chronopoulos-Gear (cg_cg.py:59-68): ONE reduction per iteration, after the SpMV it depends on
int iterate_cgcg(prcg_t* h, int k) {
    HsArgs a = hs_args(h, k);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    LAUNCHCHK(h, launch_hs_update_xr(h->sc, a));                              // x, r, (r~)
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    double* z = h->prec ? h->rt.d() : h->r.d();
    int nparts = 0;
    int rc;
    if (h->cb_session && (rc = apply_prec(h, h->r.d(), 1, h->rt.d(), 1))) return rc;   // r~ = preconditioner(r)  cg_cg.py:118
    rc = overlapped_spmv(h, k, z, h->w.d(), kEpiCG, h->r.d(), nullptr, nullptr, &nparts);   // w = A r~; nu, eta
    if (rc) return rc;
    launch_reduce_final(h->sc, h->partB.d(), nparts, dots_at(h, k), 0, 0, 5);
    if ((rc = allreduce(h, dots_at(h, k), 5, h->sc))) return rc;
    LAUNCHCHK(h, launch_cg_update_ps(h->sc, cg_args(h, k)));                  // p, s; mu by recurrence
    return PRCG_OK;
}

// Ghysels-Vanroose (gv_cg.py:65-81): the inner products come BEFORE the SpMV they overlap with
int iterate_gv(prcg_t* h, int k) {
    CgArgs a = cg_args(h, k);
    bool on = false;
    prof_begin(h, h->ev_upd, h->n_ev_upd, k, on);
    const int grid = launch_gv_update1(h->sc, a, false);                      // x, r, r~, w, w~; nu, eta
    LAUNCHCHK(h, grid);
    prof_end(h, h->ev_upd, h->n_ev_upd, on);
    int rc;
    bool replaced = false;
    if (h->replace_fn) {
        // gv_cg.py:69-71: the caller's predicate sees the state as the reference passes it (x, r, w new; p, s, u old)
        HIPCHK(h, hipStreamSynchronize(h->sc));
        const int kk = h->k;
        h->k = k;                                                        // (prcg_get_* inside the hook address iteration k)
        const int fire = h->replace_fn(h->replace_ctx, k);
        h->k = kk;
        if (fire) {
            if ((rc = dist_spmv(h, h->r.d(), h->cur_w, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;   // w = A r
            if (h->prec && !h->cb_session) launch_mul(h->sc, h->wt.d(), 1, h->dinv.d(), 1, h->cur_w, 1, h->n);      // w~ = M^-1 w
            replaced = true;
        }
    }
    if (h->cb_session && (rc = apply_prec(h, h->w.d(), 1, h->wt.d(), 1))) return rc;  // w~ = preconditioner(w)  gv_cg.py:161
    const bool side = h->multi() && !h->replace_fn;
    if (side) {
        HIPCHK(h, hipEventRecord(h->e_upd, h->sc));
        HIPCHK(h, hipStreamWaitEvent(h->sm, h->e_upd, 0));
        launch_reduce_final(h->sm, h->partA.d(), grid, dots_at(h, k), 0, 0, 5);
        if ((rc = allreduce(h, dots_at(h, k), 5, h->sm))) return rc;
        HIPCHK(h, hipEventRecord(h->e_red, h->sm));
    } else {
        launch_reduce_final(h->sc, h->partA.d(), grid, dots_at(h, k), 0, 0, 5);
        if (replaced) {     // eta = w.r~ with the replaced w (gv_cg.py:75 / :164)
            const int g2 = launch_dot(h->sc, h->cur_w, h->prec ? h->rt.d() : h->r.d(), h->n, h->partB.d(), 0);
            LAUNCHCHK(h, g2);
            launch_reduce_final(h->sc, h->partB.d(), g2, dots_at(h, k), 0, PRCG_S_DELTA, 1);
        }
        if (h->multi() && (rc = allreduce(h, dots_at(h, k), 5, h->sc))) return rc;
    }
    double* zt = h->prec ? h->wt.d() : h->w.d();
    int nparts = 0;
    if ((rc = overlapped_spmv(h, k, zt, h->tvec.d(), kEpiNone, nullptr, nullptr, nullptr, &nparts))) return rc;   // t = A w~
    if (side) HIPCHK(h, hipStreamWaitEvent(h->sc, h->e_red, 0));
    LAUNCHCHK(h, launch_cg_update_ps(h->sc, a));                              // p, s, s~, u; mu by recurrence
    return PRCG_OK;
}

// One place for every PRCG_* switch: prcg_create reads the environment through it, prcg_set_option
// sets them per handle.  Returns false for an unknown key.
bool apply_option(prcg_t* h, const char* key, const char* val) {
    if (!key || !val) return false;
    const std::string k(key);
    const long v = atol(val);
    if (k == "PRCG_SIDE_STREAM") h->side_stream = v != 0;
    else if (k == "PRCG_FUSED_FINAL") h->fused_final = v != 0;
    else if (k == "PRCG_FUSED") h->want_fused = v != 0;
    else if (k == "PRCG_WIN_SHARE") h->want_share = v != 0;
    else if (k == "PRCG_SMALL") h->want_small = v != 0;
    else if (k == "PRCG_MEDIUM") h->want_medium = v != 0;
    else if (k == "PRCG_COL16") h->want_c16 = v != 0;
    else if (k == "PRCG_COL8") h->want_c8 = v != 0;
    else if (k == "PRCG_VALDICT") h->want_vdict = v != 0;
    else if (k == "PRCG_GATHER") h->want_gather = v != 0;
    else if (k == "PRCG_GATHER_MAX_BYTES") { if (v >= 64) h->gather_max_bytes = v; }
    else if (k == "PRCG_GRID_PER_CU") h->kn.per_cu = (v >= 1 && v <= 16) ? (int)v : 0;
    else if (k == "PRCG_TILE_ORDER") h->kn.chunked = (val[0] == 'c') ? 1 : 0;
    else if (k == "PRCG_TILE_STEPS") h->steps_override = (v == 1 || v == 2 || v == 4) ? (int)v : 0;
    else if (k == "PRCG_WIN") h->want_win = v != 0;
    else if (k == "PRCG_FUSED_COMM") { h->want_fused_comm = v != 0; h->want_fused_comm_rccl = v != 0; }
    else if (k == "PRCG_PEER") h->want_peer = v != 0;
    else if (k == "PRCG_SELL") h->want_sell = v != 0;
    else if (k == "PRCG_CG_ONE") h->want_cg_one = v != 0;
    else if (k == "PRCG_WIN_ORDER") h->win_order_override = v != 0;
    else if (k == "PRCG_WIN_BIG") h->want_big = v != 0;
    else if (k == "PRCG_WIN_PAT") h->want_pat = v != 0;
    else if (k == "PRCG_WIN_SWEEP") h->want_sweep = (v >= 0 && v <= 2) ? (int)v : 1;
    else if (k == "PRCG_SWEEP_WAVES") h->sweep_max_waves = (v >= 64 && v <= 16384) ? (int)v : 6144;
    else if (k == "PRCG_SELL_GRID_PER_CU") h->sell_per_cu = (v >= 1 && v <= 8) ? (int)v : 0;
    else if (k == "PRCG_SELL_SIGMA") h->sell_sigma_opt = (v >= 64 && v <= (1 << 20)) ? (int)v : 0;
    else if (k == "PRCG_SELL_PLANES") h->sell_planes_opt = (v >= 0 && v <= 64) ? (int)v : 0;
    else if (k == "PRCG_SELL_RUNS") h->sell_runs_opt = v != 0;
    else if (k == "PRCG_SELL_WINDOW") h->sell_window_opt = v == 1 ? 64 : (v >= 0 && v <= 64) ? (int)v : 64;
    else if (k == "PRCG_SELL_MAX_OVERHEAD_PCT") h->sell_overhead_opt = (v >= 100 && v <= 800) ? (double)v / 100.0 : 0.0;
    else if (k == "PRCG_PR_PACK") h->want_pr_pack = v != 0 ? 1 : 0;
    else if (k == "PRCG_PLACE") h->place_k = (v >= 0 && v <= 8) ? (int)v : 0;
    else if (k == "PRCG_SELL_GB") h->sell_gb = (v == 4 || v == 8) ? (int)v : 0;
    else if (k == "PRCG_SELL_DEFER") h->sell_defer = v != 0;
    else if (k == "PRCG_SELL_NT") { h->sell_nt_opt = v != 0; h->sell_nt = v != 0; }
    else if (k == "PRCG_STREAM_STORES") h->stream_override = v != 0;
    else if (k == "PRCG_EXT_SIGNAL") h->ext_signal = v != 0;
    else if (k == "PRCG_DEFER_GRID_PER_CU") h->defer_per_cu = (v >= 1 && v <= 4) ? (int)v : 0;
    else if (k == "PRCG_WIN_GRID_PER_CU") h->win_per_cu = (v >= 1 && v <= 32) ? (int)v : 0;
    else if (k == "PRCG_WIN_MAX_MEAN") { if (v >= 1) h->win_max_mean = (int)v; }
    else if (k == "PRCG_WIN_ROWS") h->win_rows_override = (v == 64 || v == 128) ? (int)v : 0;
    else if (k == "PRCG_DEBUG_SHORT_SOURCES") h->debug_short_sources = v != 0;
    else return false;
    return true;
}
const char* const kOptionKeys[] = {"PRCG_SIDE_STREAM", "PRCG_FUSED_FINAL", "PRCG_FUSED", "PRCG_SMALL", "PRCG_MEDIUM", "PRCG_COL16", "PRCG_COL8",
                                   "PRCG_VALDICT", "PRCG_GATHER", "PRCG_GATHER_MAX_BYTES", "PRCG_GRID_PER_CU", "PRCG_TILE_ORDER",
                                   "PRCG_TILE_STEPS", "PRCG_WIN", "PRCG_WIN_GRID_PER_CU", "PRCG_WIN_MAX_MEAN", "PRCG_FUSED_COMM", "PRCG_WIN_ROWS", "PRCG_EXT_SIGNAL", "PRCG_DEFER_GRID_PER_CU",
                                   "PRCG_WIN_SHARE", "PRCG_DEBUG_SHORT_SOURCES", "PRCG_PEER", "PRCG_STREAM_STORES", "PRCG_SELL", "PRCG_SELL_GRID_PER_CU", "PRCG_SELL_SIGMA", "PRCG_SELL_PLANES", "PRCG_PLACE", "PRCG_SELL_NT", "PRCG_SELL_GB", "PRCG_SELL_DEFER", "PRCG_SELL_RUNS", "PRCG_SELL_WINDOW", "PRCG_SELL_MAX_OVERHEAD_PCT", "PRCG_PR_PACK", "PRCG_CG_ONE", "PRCG_WIN_ORDER", "PRCG_WIN_BIG", "PRCG_WIN_PAT", "PRCG_WIN_SWEEP", "PRCG_SWEEP_WAVES"};

int h2d(prcg_t* h, double* dst, const double* src, int64_t count) {
    HIPCHK(h, hipMemcpyAsync(dst, src, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->sc));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    return PRCG_OK;
}
int d2h(prcg_t* h, double* dst, const double* src, int64_t count) {
    HIPCHK(h, hipMemcpyAsync(dst, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->sc));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    return PRCG_OK;
}

// dst = M^-1 src: the inverse diagonal on the device, or the caller's function on the host (the reference's
// `preconditioner(v)`, figure_gen.py:42-44 -- there, too, it is the caller's Python callable).
int apply_prec(prcg_t* h, const double* src, int sstride, double* dst, int dstride) {
    const int64_t n = h->n;
    if (!h->cb_session) {
        launch_mul(h->sc, dst, dstride, h->dinv.d(), 1, src, sstride, n);
        return PRCG_OK;
    }
    h->cb_in.resize((size_t)n); h->cb_out.resize((size_t)n);
    const double* dsrc = src;
    if (sstride != 1) { launch_copy(h->sc, h->cb_stage.d(), 1, src, sstride, n); dsrc = h->cb_stage.d(); }
    int rc = d2h(h, h->cb_in.data(), dsrc, n);
    if (rc) return rc;
    if (h->cb(h->cb_ctx, n, h->cb_in.data(), h->cb_out.data()) != 0)
        return fail(h, PRCG_EINVAL, "the preconditioner callback reported a failure");
    if (dstride == 1) return h2d(h, dst, h->cb_out.data(), n);
    if ((rc = h2d(h, h->cb_stage.d(), h->cb_out.data(), n))) return rc;
    launch_copy(h->sc, dst, dstride, h->cb_stage.d(), 1, n);
    return PRCG_OK;
}

// where does state vector `which` live?  base pointer + stride (in doubles)
bool locate(prcg_t* h, int which, double** base, int* stride) {
    *stride = 1;
    *base = nullptr;
    const int v = h->variant;
    if (is_pipe(v)) {
        switch (which) {
        case PRCG_VEC_X: *base = h->xp.d(); *stride = 2; return true;
        case PRCG_VEC_P: *base = h->xp.d() + 1; *stride = 2; return true;
        case PRCG_VEC_R: *base = ((h->fused && !h->prec) ? h->rs_cur : h->rs.d()); *stride = 2; return true;
        case PRCG_VEC_S: *base = ((h->fused && !h->prec) ? h->rs_cur : h->rs.d()) + 1; *stride = 2; return true;
        case PRCG_VEC_W:
            if (h->fused) { if (pipe_recompute(v)) return false; *base = h->wv.d(); return true; }   // stored recurrence
            *base = h->wu.d(); *stride = 2; return true;
        case PRCG_VEC_U: if (h->fused) return false; *base = h->wu.d() + 1; *stride = 2; return true;
        case PRCG_VEC_RT: if (!h->prec) return false; *base = (h->fused ? h->rs_cur : h->rst.d()); *stride = 2; return true;
        case PRCG_VEC_ST: if (!h->prec) return false; *base = (h->fused ? h->rs_cur : h->rst.d()) + 1; *stride = 2; return true;
        // host-callback preconditioner sessions: w~ and u~ are stored state (what the callback returned), in every flavour
        case PRCG_VEC_WT: if (!h->prec || (pipe_recompute(v) && !h->cb_session)) return false; *base = h->wt.d(); return true;
        case PRCG_VEC_UT: if (!h->cb_session) return false; *base = h->ut.d(); return true;
        default: return false;
        }
    }
    if (is_cg_family(v)) {
        switch (which) {
        case PRCG_VEC_X: *base = h->x.d(); return true;
        case PRCG_VEC_P: *base = h->p.d(); return true;
        case PRCG_VEC_R: *base = h->cur_r; return true;
        case PRCG_VEC_S: *base = h->cur_s; return true;
        case PRCG_VEC_W: *base = h->cur_w; return true;
        case PRCG_VEC_U: if (v != PRCG_GV) return false; *base = h->cur_u; return true;
        case PRCG_VEC_RT: if (!h->prec) return false; *base = h->rt.d(); return true;
        case PRCG_VEC_WT: if (!h->prec || v != PRCG_GV) return false; *base = h->wt.d(); return true;
        case PRCG_VEC_ST: if (!h->prec || v != PRCG_GV) return false; *base = h->st.d(); return true;
        default: return false;
        }
    }
    switch (which) {
    case PRCG_VEC_X: *base = h->x.d(); return true;
    case PRCG_VEC_P: *base = h->p_cur; return true;
    case PRCG_VEC_R: *base = h->cur_r; return true;
    case PRCG_VEC_S: *base = h->cur_s; return true;
    case PRCG_VEC_RT: if (!h->prec) return false; *base = h->cur_rt; return true;
    case PRCG_VEC_ST: if (!h->prec || !is_pr(v)) return false; *base = h->cur_st; return true;
    default: return false;
    }
}

void destroy_events(std::vector<EventPair>& evs) {
    for (auto& e : evs) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    evs.clear();
}

}  // namespace

// =========================================================================================
extern "C" {

int prcg_version(void) { return 1; }

const char* prcg_last_error(const prcg_t* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int prcg_create(prcg_t** out, int device_id) {
    if (!out) return fail(nullptr, PRCG_EINVAL, "prcg_create: null output pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PRCG_EHIP, "prcg_create: no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, PRCG_EINVAL, "prcg_create: device %d out of range [0,%d)", device_id, ndev);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, PRCG_EHIP, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    prcg_t* h = new (std::nothrow) prcg_handle();
    if (!h) return fail(nullptr, PRCG_ENOMEM, "out of host memory");
    h->dev = device_id;
    // every switch lives in the handle from here on (no lazily read environment anywhere else)
    for (const char* key : kOptionKeys)
        if (const char* e = getenv(key)) (void)apply_option(h, key, e);
    // the communication stream outranks the compute stream: its small kernels (halo pack,
    // partial reduction, RCCL) must get CU slots while the matrix product floods the chip
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&h->sc, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&h->sm, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&h->sh, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->e_upd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->e_halo, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->e_red, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->e_kdone) != hipSuccess || hipEventCreate(&h->e_rdone) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->err_host), 64, hipHostMallocDefault) != hipSuccess) {
        prcg_destroy(h);
        return fail(nullptr, PRCG_EHIP, "prcg_create: stream/event creation failed");
    }
    *h->err_host = 0u;
    *out = h;
    return PRCG_OK;
}

void prcg_destroy(prcg_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->dev);
    (void)hipDeviceSynchronize();
    for (void* q : h->peer_opened) (void)hipIpcCloseMemHandle(q);
    if (h->xbuf) (void)hipFree(h->xbuf);
    if (h->err_host) (void)hipHostFree(h->err_host);
    if (h->comm_halo && h->rccl) (void)h->rccl->CommDestroy(h->comm_halo);
    if (h->comm && h->rccl) (void)h->rccl->CommDestroy(h->comm);
    destroy_events(h->ev_spmv);
    destroy_events(h->ev_upd);
    if (h->e_upd) (void)hipEventDestroy(h->e_upd);
    if (h->e_halo) (void)hipEventDestroy(h->e_halo);
    if (h->e_red) (void)hipEventDestroy(h->e_red);
    if (h->e_kdone) (void)hipEventDestroy(h->e_kdone);
    if (h->e_rdone) (void)hipEventDestroy(h->e_rdone);
    if (h->sc) (void)hipStreamDestroy(h->sc);
    if (h->sm) (void)hipStreamDestroy(h->sm);
    if (h->sh) (void)hipStreamDestroy(h->sh);
    delete h;
}

int prcg_set_preconditioner(prcg_t* h, prcg_prec_fn fn, void* ctx) {
    if (!h) return PRCG_EINVAL;
    h->cb = fn;
    h->cb_ctx = ctx;
    return PRCG_OK;
}

int prcg_set_replace_hook(prcg_t* h, prcg_replace_fn fn, void* ctx) {
    if (!h) return PRCG_EINVAL;
    h->replace_fn = fn;
    h->replace_ctx = ctx;
    return PRCG_OK;
}

int prcg_set_option(prcg_t* h, const char* key, const char* value) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, key && value, "prcg_set_option: null key or value");
    CHECK(h, !h->have_csr, "prcg_set_option: options are fixed once the operator is set (call it before prcg_set_csr)");
    CHECK(h, apply_option(h, key, value), "prcg_set_option: unknown option '%s'", key);
    return PRCG_OK;
}

int prcg_comm_unique_id(const char* rccl_path, void* id128) {
    if (!id128) return fail(nullptr, PRCG_EINVAL, "prcg_comm_unique_id: null buffer");
    std::string err;
    Rccl* r = Rccl::get(rccl_path, err);
    if (!r) return fail(nullptr, PRCG_ERCCL, "%s", err.c_str());
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(nullptr, PRCG_ERCCL, "ncclGetUniqueId: %s", r->GetErrorString(rc));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return PRCG_OK;
}

int prcg_comm_init(prcg_t* h, const char* rccl_path, int rank, int nranks, const void* id128, int n_ids) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, nranks >= 1 && rank >= 0 && rank < nranks, "prcg_comm_init: bad rank %d of %d", rank, nranks);
    CHECK(h, id128 != nullptr, "prcg_comm_init: null unique id");
    CHECK(h, n_ids == 1 || n_ids == 2, "prcg_comm_init: n_ids must be 1 or 2");
    CHECK(h, h->comm == nullptr, "prcg_comm_init: communicator already initialised");
    std::string err;
    h->rccl = Rccl::get(rccl_path, err);
    if (!h->rccl) return fail(h, PRCG_ERCCL, "%s", err.c_str());
    HIPCHK(h, hipSetDevice(h->dev));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(h, h->rccl->CommInitRank(&h->comm, nranks, id, rank));
    if (n_ids == 2) {
        memcpy(&id, static_cast<const char*>(id128) + sizeof id, sizeof id);
        NCCLCHK(h, h->rccl->CommInitRank(&h->comm_halo, nranks, id, rank));
    }
    h->rank = rank;
    h->nranks = nranks;
    return PRCG_OK;
}

int prcg_set_csr(prcg_t* h, int64_t n_rows, int64_t n_ghost, int64_t nnz, const void* indptr, int indptr_is64,
                 const int32_t* indices, const double* data) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, n_rows >= 0 && n_ghost >= 0 && nnz >= 0, "prcg_set_csr: negative size");
    CHECK(h, nnz < (int64_t)2147483000, "prcg_set_csr: nnz=%lld does not fit int32 row pointers", (long long)nnz);
    CHECK(h, n_rows + n_ghost < (int64_t)2147483000, "prcg_set_csr: too many rows for int32 column indices");
    CHECK(h, indptr && (nnz == 0 || (indices && data)), "prcg_set_csr: null array");
    h->in_session = false;   // a new operator invalidates any open session
    HIPCHK(h, hipSetDevice(h->dev));

    // --- validate on the host before anything reaches a kernel ---
    std::vector<int32_t> ip((size_t)n_rows + 1);
    for (int64_t i = 0; i <= n_rows; ++i) {
        const int64_t v = indptr_is64 ? static_cast<const int64_t*>(indptr)[i] : static_cast<const int32_t*>(indptr)[i];
        CHECK(h, v >= 0 && v <= nnz, "prcg_set_csr: indptr[%lld]=%lld out of [0,nnz]", (long long)i, (long long)v);
        CHECK(h, i == 0 || v >= ip[i - 1], "prcg_set_csr: indptr not monotone at row %lld", (long long)i);
        ip[i] = (int32_t)v;
    }
    CHECK(h, ip[0] == 0 && ip[n_rows] == nnz, "prcg_set_csr: indptr must run from 0 to nnz");
    const int64_t ncols = n_rows + n_ghost;
    std::vector<uint8_t> cls((size_t)n_rows, 0);
    int max_len = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        uint8_t c = 0;
        for (int32_t q = ip[i]; q < ip[i + 1]; ++q) {
            const int32_t j = indices[q];
            CHECK(h, j >= 0 && j < ncols, "prcg_set_csr: column index %d out of [0,%lld) in row %lld", j,
                  (long long)ncols, (long long)i);
            if (j >= n_rows) c = 1;
        }
        cls[i] = c;
        if (ip[i + 1] - ip[i] > max_len) max_len = ip[i + 1] - ip[i];
    }
    h->max_row_len = max_len;
    std::vector<Tile> t0, t1;
    h->steps = pick_tile_steps(h->steps_override, n_rows, nnz);
    plan_tiles(n_rows, ip.data(), n_ghost > 0 ? cls.data() : nullptr, tile_cap_nnz(h->steps), kTileCapRows, t0, t1);
    std::vector<Tile> all(t0);
    all.insert(all.end(), t1.begin(), t1.end());

    // --- window tiles (row-per-lane kernels): tried first; when every tile of the operator qualifies the
    // narrow encodings of the CSR-adaptive kernels below are not built at all ---
    h->win = false; h->win_vd = false; h->nwt_int = h->nwt_bnd = 0;
    WinPlan wp;
    std::vector<WTile> wall;
    std::vector<uint8_t> wvidx;
    std::vector<double> wvdict;
    // pattern tiles first (constant-coefficient stencils: 64-row tiles, at most kWinPatPages pages, every tile one pattern --
    // prcg_plan.h: plan_window_patterns): no per-nonzero stream at all
    h->win_pat = false;
    std::vector<PatRec> pats;
    std::vector<uint16_t> pmasks;
    h->sweep_waves = h->sweep_tiles = 0;
    if (h->want_win && h->want_pat && (h->want_sweep == 2 || (h->want_sweep == 1 && n_rows >= 5000000)) && h->want_vdict &&
        !h->win_rows_override && n_ghost == 0 && nnz > 0 && nnz <= (int64_t)kPatSlots * n_rows) {
        // a stencil on a regular grid, long launches: sweep order (a wave's consecutive tiles = the same rows of consecutive grid
        // planes; the pages they share stay in LDS -- prcg_plan.h: plan_sweep_tiles), then the pattern check as for any tiling
        SweepPlan sw;
        if (plan_sweep_tiles(n_rows, ncols, ip.data(), indices, kWinPatPages, h->sweep_max_waves, sw) &&
            plan_window_patterns(sw.tiles, ip.data(), sw.cw.data(), data, pats, pmasks)) {
            h->win = true; h->win_pat = true; h->win_vd = true; h->win_geom = kWinPatGeom; h->win_rows = 64;
            h->sweep_waves = sw.waves; h->sweep_tiles = (int)sw.tiles.size();
            wall.swap(sw.tiles);
            wp.t0 = wall; wp.t1.clear();
        }
    }
    if (!h->win_pat && h->want_win && h->want_pat && h->want_vdict && !h->win_rows_override && n_rows >= 64 && nnz > 0 && nnz <= (int64_t)kPatSlots * n_rows) {
        WinPlan wq;
        plan_window_tiles(n_rows, ncols, ip.data(), indices, n_ghost > 0 ? cls.data() : nullptr, 64, kWinCapNnz, kWinPatPages, wq);
        if (wq.ok0 && wq.ok1 && wq.t0.size() + wq.t1.size() < (size_t)(1 << 26)) {
            std::vector<WTile> wa(wq.t0);
            wa.insert(wa.end(), wq.t1.begin(), wq.t1.end());
            if (plan_window_patterns(wa, ip.data(), wq.cw.data(), data, pats, pmasks)) {
                h->win = true; h->win_pat = true; h->win_vd = true; h->win_geom = kWinPatGeom; h->win_rows = 64;
                wall.swap(wa);
                wp.t0.swap(wq.t0); wp.t1.swap(wq.t1);
            }
        }
    }
    if (!h->win_pat && h->want_win && n_rows >= 64 && nnz > 0 && nnz <= (int64_t)h->win_max_mean * n_rows) {
        int rows = h->win_rows_override ? h->win_rows_override : (nnz < 10 * n_rows ? 128 : 64);
        plan_window_tiles(n_rows, ncols, ip.data(), indices, n_ghost > 0 ? cls.data() : nullptr, rows, kWinCapNnz,
                          win_max_pages(rows), wp);
        int most = wp.pages0 > wp.pages1 ? wp.pages0 : wp.pages1;
        if (!h->win_rows_override && rows == 128 && wp.ok0 && wp.ok1 && most > 8) {
            // short rows whose 128-row tiles need more than eight pages (a 3-D stencil: its plane neighbours): 64-row tiles of
            // at most eight pages stream better (S2: +6 % with the dictionary, +4.5 % plain; r03_sweeps.md J), where they qualify
            WinPlan w64;
            plan_window_tiles(n_rows, ncols, ip.data(), indices, n_ghost > 0 ? cls.data() : nullptr, 64, kWinCapNnz, win_max_pages(64), w64);
            const int m64 = w64.pages0 > w64.pages1 ? w64.pages0 : w64.pages1;
            if (w64.ok0 && w64.ok1 && win_geometry(64, m64) >= 0) {
                wp.t0.swap(w64.t0); wp.t1.swap(w64.t1); wp.cw.swap(w64.cw);
                wp.pages0 = w64.pages0; wp.pages1 = w64.pages1;
                rows = 64; most = m64;
            }
        }
        const int geom = win_geometry(rows, most);
        if (wp.ok0 && wp.ok1 && geom >= 0 && wp.t0.size() + wp.t1.size() < (size_t)(1 << 26)) {
            h->win = true; h->win_geom = geom; h->win_rows = rows;
            wall = wp.t0;
            wall.insert(wall.end(), wp.t1.begin(), wp.t1.end());
            if (h->want_vdict) {
                wvidx.assign((size_t)nnz + 32, 0);
                h->win_vd = plan_window_dict(wall, data, kWinDictMax, wvidx, wvdict);
                if (!h->win_vd) { for (auto& t : wall) t.vd_first = t.vd_count = 0; }
            }
        }
    }
    // --- sliced rows (lane-per-row kernels): operators that are no window operators but whose rows are long enough for
    // a lane each -- assembled FEM matrices -- when the padding to the slices' longest rows stays below 25 % ---
    h->sell = false; h->nst_int = h->nst_bnd = 0;
    SellPlan sp;
    // (rows of 24 nonzeros and more: shorter rows that are no window operator keep the CSR-adaptive kernels with their
    //  narrow column / value encodings -- a lane per row pays once a row is a sizeable share of a tile)
    if (!h->win && h->want_sell && n_rows >= 64 && nnz >= 24 * n_rows) {
        SellOptions so;
        so.sigma = h->sell_sigma_opt;
        so.planes = h->sell_planes_opt;
        so.allow_runs = h->sell_runs_opt;
        so.window_granules = h->sell_window_opt;
        if (h->sell_overhead_opt > 0.0) so.max_overhead = h->sell_overhead_opt;
        h->sell = plan_sell(n_rows, ip.data(), indices, data, n_ghost > 0 ? cls.data() : nullptr, so, sp);
    }
    const bool classic_enc = !h->win && !h->sell;      // column / value re-encodings of the CSR-adaptive kernels

    // --- 16-bit tile-relative column encoding (host, once) ---
    std::vector<int32_t> tbase(all.size() + 1, 0);
    std::vector<uint16_t> c16;
    std::vector<uint8_t> c8;
    bool fit_int = classic_enc && h->want_c16 && !all.empty(), fit_bnd = classic_enc && h->want_c16;
    bool fit8_int = fit_int && h->want_c8, fit8_bnd = fit_bnd && h->want_c8;
    if (classic_enc && h->want_c16) {
        const int cap = tile_cap_nnz(h->steps);
        for (size_t ti = 0; ti < all.size(); ++ti) {
            const Tile& tl = all[ti];
            if (tl.nnz_end - tl.nnz_begin > cap || tl.nnz_end == tl.nnz_begin) continue;   // long row / empty: not streamed
            int32_t lo_c = indices[tl.nnz_begin], hi_c = lo_c;
            for (int32_t q = tl.nnz_begin; q < tl.nnz_end; ++q) {
                lo_c = indices[q] < lo_c ? indices[q] : lo_c;
                hi_c = indices[q] > hi_c ? indices[q] : hi_c;
            }
            tbase[ti] = lo_c;
            if (hi_c - lo_c >= 65536) { if (ti < t0.size()) fit_int = false; else fit_bnd = false; }
            if (hi_c - lo_c >= 256) { if (ti < t0.size()) fit8_int = false; else fit8_bnd = false; }
        }
        if (fit_int || (fit_bnd && !t1.empty())) {
            c16.assign((size_t)nnz + 8, 0);
            for (size_t ti = 0; ti < all.size(); ++ti) {
                const bool ok = ti < t0.size() ? fit_int : fit_bnd;
                const Tile& tl = all[ti];
                if (!ok || tl.nnz_end - tl.nnz_begin > cap) continue;
                for (int32_t q = tl.nnz_begin; q < tl.nnz_end; ++q) c16[q] = (uint16_t)(indices[q] - tbase[ti]);
            }
        }
    }
    h->c16_int = fit_int && !c16.empty();
    h->c16_bnd = fit_bnd && !c16.empty() && !t1.empty();
    fit8_int = fit8_int && h->c16_int;
    fit8_bnd = fit8_bnd && h->c16_bnd;
    if (fit8_int || fit8_bnd) {
        c8.assign((size_t)nnz + 8, 0);
        for (size_t ti = 0; ti < all.size(); ++ti) {
            const bool ok = ti < t0.size() ? fit8_int : fit8_bnd;
            const Tile& tl = all[ti];
            if (!ok || tl.nnz_end - tl.nnz_begin > tile_cap_nnz(h->steps)) continue;
            for (int32_t q = tl.nnz_begin; q < tl.nnz_end; ++q) c8[q] = (uint8_t)(indices[q] - tbase[ti]);
        }
    }
    h->c8_int = fit8_int;
    h->c8_bnd = fit8_bnd;

    // --- value dictionary (host, once): per streamed tile the distinct bit patterns of its values;
    // a class of tiles qualifies if none of its tiles needs more than kDictMax entries.
    std::vector<uint8_t> vidx;
    std::vector<double> vdict;
    std::vector<int32_t> vdesc;     // {first entry, count} per tile
    bool vd_int = classic_enc && h->want_vdict && !t0.empty(), vd_bnd = classic_enc && h->want_vdict && !t1.empty();
    if (vd_int || vd_bnd) {
        const int cap = tile_cap_nnz(h->steps);
        vidx.assign((size_t)nnz + 8, 0);
        vdesc.assign(2 * (all.size() + 1), 0);
        vdict.reserve(all.size() * 4);
        constexpr int kHash = 256;          // open addressing, <= kDictMax live keys
        uint64_t keys[kHash];
        int16_t slot_of[kHash];
        for (size_t ti = 0; ti < all.size(); ++ti) {
            const bool interior = ti < t0.size();
            if (!(interior ? vd_int : vd_bnd)) continue;
            const Tile& tl = all[ti];
            if (tl.nnz_end - tl.nnz_begin > cap || tl.nnz_end == tl.nnz_begin) continue;   // long row / empty: not streamed
            for (int i = 0; i < kHash; ++i) slot_of[i] = -1;
            const size_t first = vdict.size();
            int count = 0;
            bool ok = true;
            for (int32_t q = tl.nnz_begin; q < tl.nnz_end; ++q) {
                uint64_t bits;
                memcpy(&bits, &data[q], sizeof bits);
                uint32_t hsh = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 56);   // 8 bits
                while (slot_of[hsh] >= 0 && keys[hsh] != bits) hsh = (hsh + 1) & (kHash - 1);
                if (slot_of[hsh] < 0) {
                    if (count == kDictMax) { ok = false; break; }
                    keys[hsh] = bits;
                    slot_of[hsh] = (int16_t)count++;
                    vdict.push_back(data[q]);
                }
                vidx[q] = (uint8_t)slot_of[hsh];
            }
            if (!ok) {
                vdict.resize(first);
                if (interior) vd_int = false; else vd_bnd = false;
                continue;
            }
            vdesc[2 * ti] = (int32_t)first;
            vdesc[2 * ti + 1] = count;
        }
        if (vdict.size() >= (size_t)INT32_MAX) vd_int = vd_bnd = false;
    }
    h->vd_int = vd_int;
    h->vd_bnd = vd_bnd && !t1.empty();

    // --- upload (arrays padded so the 16-byte stream loads never leave the allocation) ---
    const size_t pad = 32;
    HIPCHK(h, h->indptr.alloc(((size_t)n_rows + 1 + pad) * sizeof(int32_t)));
    HIPCHK(h, h->col.alloc(((size_t)nnz + pad) * sizeof(int32_t)));
    HIPCHK(h, h->val.alloc(((size_t)nnz + pad) * sizeof(double)));
    HIPCHK(h, h->tiles.alloc((all.size() + 1) * sizeof(Tile)));
    HIPCHK(h, hipMemcpy(h->indptr.p, ip.data(), ((size_t)n_rows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (nnz > 0) {
        HIPCHK(h, hipMemcpy(h->col.p, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->val.p, data, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    }
    if (!all.empty())
        HIPCHK(h, hipMemcpy(h->tiles.p, all.data(), all.size() * sizeof(Tile), hipMemcpyHostToDevice));
    HIPCHK(h, h->tile_base.alloc(tbase.size() * sizeof(int32_t)));
    HIPCHK(h, hipMemcpy(h->tile_base.p, tbase.data(), tbase.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(h, h->col16.alloc(c16.empty() ? 16 : c16.size() * sizeof(uint16_t)));
    if (!c16.empty())
        HIPCHK(h, hipMemcpy(h->col16.p, c16.data(), c16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(h, h->col8.alloc(c8.empty() ? 16 : c8.size()));
    if (!c8.empty()) HIPCHK(h, hipMemcpy(h->col8.p, c8.data(), c8.size(), hipMemcpyHostToDevice));
    const bool any_vd = h->vd_int || h->vd_bnd;
    HIPCHK(h, h->vidx8.alloc(any_vd ? vidx.size() : 16));
    HIPCHK(h, h->vdict.alloc(any_vd ? (vdict.size() + kDictMax) * sizeof(double) : 16));
    HIPCHK(h, h->vdesc.alloc(any_vd ? vdesc.size() * sizeof(int32_t) : 16));
    if (any_vd) {
        HIPCHK(h, hipMemcpy(h->vidx8.p, vidx.data(), vidx.size(), hipMemcpyHostToDevice));
        if (!vdict.empty())
            HIPCHK(h, hipMemcpy(h->vdict.p, vdict.data(), vdict.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->vdesc.p, vdesc.data(), vdesc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (h->sell) {
        h->nst_int = (int)sp.s0.size(); h->nst_bnd = (int)sp.s1.size();
        std::vector<SellSlice> sall(sp.s0);
        sall.insert(sall.end(), sp.s1.begin(), sp.s1.end());
        HIPCHK(h, h->sval.alloc(sp.val.size() * sizeof(double), false));
        HIPCHK(h, hipMemcpy(h->sval.p, sp.val.data(), sp.val.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, h->scol.alloc(sp.col.size() * sizeof(uint16_t), false));
        HIPCHK(h, hipMemcpy(h->scol.p, sp.col.data(), sp.col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIPCHK(h, h->sslices.alloc((sall.size() + 1) * sizeof(SellSlice)));
        if (!sall.empty()) HIPCHK(h, hipMemcpy(h->sslices.p, sall.data(), sall.size() * sizeof(SellSlice), hipMemcpyHostToDevice));
        HIPCHK(h, h->srows.alloc((sp.rows.size() + 64) * sizeof(int32_t)));
        if (!sp.rows.empty()) HIPCHK(h, hipMemcpy(h->srows.p, sp.rows.data(), sp.rows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // what a product reads of the operator: 8 B per (padded) value, 2 B per (padded) column code, the slice descriptors, and
        // the row pointers (slices of consecutive rows) or the slices' (row, stored length) pairs
        HIPCHK(h, h->sgran.alloc((sp.gran.size() + 64) * sizeof(int32_t)));
        if (!sp.gran.empty()) HIPCHK(h, hipMemcpy(h->sgran.p, sp.gran.data(), sp.gran.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        h->sell_window = sp.window;
        // ... and with WINDOW codes the granule starts (the window pages themselves are vector traffic: every entry a slice touches,
        // read once per slice instead of once per nonzero)
        h->sell_bytes = sp.padded_nnz * 8 + sp.col_entries * 2 + (int64_t)sall.size() * 32 + 4 * (n_rows + 1) + (int64_t)sp.rows.size() * 4 +
                        (int64_t)sp.gran.size() * 4;
        h->sell_sigma = sp.sigma; h->sell_planes = sp.planes; h->sell_stride = sp.stride_rows; h->sell_run = sp.run;
        sp = SellPlan{};
    }
    // mid-size systems (no ghosts, at most 131,072 rows): the plan of the few-workgroup solver, used by pipelined sessions
    // that record nothing but the recurrence residual (prcg_solve_begin decides)
    h->medium_ok = false;
    if (h->want_medium && n_ghost == 0 && n_rows >= 256 && n_rows <= (int64_t)kMedMaxGroups * 16 * kMedSlices * 64 && nnz <= (int64_t)1 << 23) {
        MediumPlan mp;
        if (plan_medium(n_rows, ip.data(), indices, data, kMedMaxGroups, kMedSlices, kMedMaxWindow, mp)) {
            HIPCHK(h, h->m_val.alloc(mp.sell.val.size() * sizeof(double), false));
            HIPCHK(h, hipMemcpy(h->m_val.p, mp.sell.val.data(), mp.sell.val.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_col.alloc(mp.sell.col.size() * sizeof(uint16_t), false));
            HIPCHK(h, hipMemcpy(h->m_col.p, mp.sell.col.data(), mp.sell.col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_slices.alloc((mp.sell.s0.size() + 1) * sizeof(SellSlice)));
            HIPCHK(h, hipMemcpy(h->m_slices.p, mp.sell.s0.data(), mp.sell.s0.size() * sizeof(SellSlice), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_rows.alloc((mp.sell.rows.size() + 128) * sizeof(int32_t)));
            if (!mp.sell.rows.empty()) HIPCHK(h, hipMemcpy(h->m_rows.p, mp.sell.rows.data(), mp.sell.rows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_wave_first.alloc(mp.wave_first.size() * sizeof(int32_t)));
            HIPCHK(h, hipMemcpy(h->m_wave_first.p, mp.wave_first.data(), mp.wave_first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_window.alloc(mp.window.size() * sizeof(int32_t)));
            HIPCHK(h, hipMemcpy(h->m_window.p, mp.window.data(), mp.window.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_own.alloc(mp.own.size() * sizeof(int32_t)));
            HIPCHK(h, hipMemcpy(h->m_own.p, mp.own.data(), mp.own.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIPCHK(h, h->m_exch.alloc((size_t)4 * n_rows * sizeof(double) + 64));
            HIPCHK(h, h->m_slots.alloc((size_t)3 * kMedMaxGroups * 8 * sizeof(double)));      // sums of even / odd iterations, flags
            HIPCHK(h, h->m_err.alloc(64));
            h->med_groups = mp.groups; h->med_window = mp.window_pairs;
            h->medium_ok = true;
        }
    }
    h->peer_ok = false;
    h->wt_rb.clear(); h->wt_re.clear();
    if (h->win) {
        h->nwt_int = (int)wp.t0.size(); h->nwt_bnd = (int)wp.t1.size();
        for (const auto& t : wall) { h->wt_rb.push_back(t.rb); h->wt_re.push_back(t.re); }
        // the tiles' stream images: byte-identical ones are stored once (prcg_plan.h: share_window_streams)
        std::vector<uint8_t> vstore;
        std::vector<uint16_t> rstore;
        size_t cw_bytes = 0;
        if (h->win_pat) {
            // pattern tiles: the rows' slot masks take the place of the row pointers; no window-index / value-index images
            rstore.swap(pmasks);
            rstore.resize(rstore.size() + 64, 0);
            HIPCHK(h, h->wcw.alloc(64));
            HIPCHK(h, h->wpat.alloc((pats.size() + 1) * sizeof(PatRec)));
            HIPCHK(h, hipMemcpy(h->wpat.p, pats.data(), pats.size() * sizeof(PatRec), hipMemcpyHostToDevice));
            cw_bytes = pats.size() * sizeof(PatRec);
            vstore.assign(64, 0);
        } else if (h->win_geom >= 2) {
            std::vector<uint16_t> cstore;
            share_window_streams<uint16_t>(wall, ip.data(), wp.cw.data(), h->win_vd ? wvidx.data() : nullptr, h->want_share,
                                           cstore, vstore, rstore);
            cw_bytes = cstore.size() * sizeof(uint16_t);
            HIPCHK(h, h->wcw.alloc(cw_bytes));
            HIPCHK(h, hipMemcpy(h->wcw.p, cstore.data(), cw_bytes, hipMemcpyHostToDevice));
        } else {
            std::vector<uint8_t> c8w(wp.cw.size()), cstore;
            for (size_t q = 0; q < wp.cw.size(); ++q) c8w[q] = (uint8_t)wp.cw[q];
            share_window_streams<uint8_t>(wall, ip.data(), c8w.data(), h->win_vd ? wvidx.data() : nullptr, h->want_share,
                                          cstore, vstore, rstore);
            cw_bytes = cstore.size();
            HIPCHK(h, h->wcw.alloc(cw_bytes));
            HIPCHK(h, hipMemcpy(h->wcw.p, cstore.data(), cw_bytes, hipMemcpyHostToDevice));
        }
        // period of the images over the interior tiles (a stencil on a regular grid: a grid line, a grid plane): the
        // smallest P with image(t + P) == image(t) for every t of a long stretch in the middle of the table
        h->win_period = 0;
        if (h->want_share && h->win_vd && !h->win_pat && wp.t0.size() > 4096) {
            const size_t nt0 = wp.t0.size(), t0 = nt0 / 3;
            for (size_t P = 2; P <= 4096 && t0 + 3 * P < nt0; ++P) {
                if (wall[t0 + P].spare != wall[t0].spare || wall[t0].spare == 0) continue;
                bool ok = true;
                for (size_t j = 0; j < 2 * P && ok; ++j) ok = wall[t0 + j + P].spare == wall[t0 + j].spare;
                if (ok) { h->win_period = (int)P; break; }
            }
            if (wall[t0 + 1].spare == wall[t0].spare && wall[t0 + 2].spare == wall[t0].spare) h->win_period = 0;   // (period 1: nothing to align)
        }
        // tile order: PRCG_WIN_ORDER=1 lets every XCD sweep one contiguous eighth of the table (a 3-D stencil's plane neighbours
        // then meet in one XCD's L2: S2 reads 0.39 instead of 0.67 GB per launch through the fabric) -- measured 2 % SLOWER at S2
        // (profiles/r03_sweeps.md H: the 128-row stencil kernels are not bound by bytes), so the chip-wide front stays the default
        h->win_order = h->win_order_override > 0 ? 1 : 0;
        HIPCHK(h, h->wrel.alloc(rstore.size() * sizeof(uint16_t)));
        HIPCHK(h, hipMemcpy(h->wrel.p, rstore.data(), rstore.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIPCHK(h, h->wtiles.alloc((wall.size() + 1) * sizeof(WTile)));
        HIPCHK(h, hipMemcpy(h->wtiles.p, wall.data(), wall.size() * sizeof(WTile), hipMemcpyHostToDevice));
        h->win_stream_bytes = (int64_t)(wall.size() * sizeof(WTile) + cw_bytes + rstore.size() * sizeof(uint16_t)) +
                              (h->win_pat ? 0 : (h->win_vd ? (int64_t)(vstore.size() + wvdict.size() * sizeof(double)) : (int64_t)nnz * 8));
        if (h->win_vd) {
            HIPCHK(h, h->wvidx.alloc(vstore.size()));
            HIPCHK(h, hipMemcpy(h->wvidx.p, vstore.data(), vstore.size(), hipMemcpyHostToDevice));
            HIPCHK(h, h->wvdict.alloc((wvdict.size() + kWinDictMax) * sizeof(double)));
            if (!wvdict.empty())
                HIPCHK(h, hipMemcpy(h->wvdict.p, wvdict.data(), wvdict.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    h->n = n_rows; h->g = n_ghost; h->nnz = nnz;
    h->nt_int = (int)t0.size(); h->nt_bnd = (int)t1.size();
    // what one launch moves -- the operator as streamed plus 64 bytes of vectors per row: beyond the Infinity Cache
    // (256 MB) the next launch finds none of its row results cached anyway (S3 +23 %, s4b +4 %, S2 +2 %; S1 and one
    // eighth of S3 fit and lose 4-8 % with streaming stores)
    HIPCHK(h, h->tmp_ext.alloc((size_t)2 * (n_rows + n_ghost + kGatherPad) * sizeof(double)));
    HIPCHK(h, h->t1.alloc((size_t)2 * n_rows * sizeof(double)));
    HIPCHK(h, h->partA.alloc((size_t)8192 * kPartialStride * sizeof(double)));
    HIPCHK(h, h->partB.alloc((size_t)8192 * kPartialStride * sizeof(double)));
    HIPCHK(h, h->ticket.alloc(64));
    h->have_csr = true;
    h->have_halo = false;
    h->gather_planned = false;
    h->stream_stores = h->stream_override >= 0 ? h->stream_override
                                               : ((int64_t)64 * n_rows + prcg_operator_bytes(h) > (int64_t)256 << 20);
    // sliced rows with a sorting window: a slice's rows lie anywhere in the window, its 16-byte row results are PARTS of cache
    // lines that the other slices of the window complete -- plain stores let the L2 merge them (nontemporal ones wrote 1.54 x the
    // bytes: s4c 706 -> 675 us, profiles/r04_sweeps.md)
    if (h->sell && h->sell_sigma > 64 && h->stream_override < 0) h->stream_stores = 0;
    // ... and the value / code streams of an operator far larger than the Infinity Cache are read with nontemporal loads
    // (s4b at 3.4 GB: 638 -> 610 us, s4c +4.6 %; at 1.3 GB -1 %)
    if (h->sell && h->sell_nt_opt < 0) h->sell_nt = h->sell_bytes >= (int64_t)2000 << 20;
    return PRCG_OK;
}

int prcg_set_halo(prcg_t* h, int n_peers, const int32_t* peer_rank, const int64_t* send_ptr, const int32_t* send_idx,
                  const int64_t* recv_ptr) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->have_csr, "prcg_set_halo: call prcg_set_csr first");
    CHECK(h, n_peers >= 0 && (n_peers == 0 || (peer_rank && send_ptr && recv_ptr)), "prcg_set_halo: null array");
    HIPCHK(h, hipSetDevice(h->dev));
    h->n_peers = n_peers;
    h->peer_rank.assign(peer_rank, peer_rank + n_peers);
    h->send_ptr.assign(1, 0);
    h->recv_ptr.assign(1, 0);
    if (n_peers > 0) {
        h->send_ptr.assign(send_ptr, send_ptr + n_peers + 1);
        h->recv_ptr.assign(recv_ptr, recv_ptr + n_peers + 1);
    }
    CHECK(h, h->send_ptr[0] == 0 && h->recv_ptr[0] == 0, "prcg_set_halo: pointers must start at 0");
    for (int q = 0; q < n_peers; ++q) {
        // peer == own rank is allowed: RCCL does self send/recv inside a group, which is how
        // the one-GPU tests drive the whole halo path (loopback ghosts)
        CHECK(h, h->peer_rank[q] >= 0 && h->peer_rank[q] < h->nranks, "prcg_set_halo: bad peer rank %d",
              h->peer_rank[q]);
        CHECK(h, h->send_ptr[q + 1] >= h->send_ptr[q] && h->recv_ptr[q + 1] >= h->recv_ptr[q],
              "prcg_set_halo: pointers not monotone");
    }
    CHECK(h, h->recv_ptr[n_peers] == h->g, "prcg_set_halo: receive counts (%lld) != n_ghost (%lld)",
          (long long)h->recv_ptr[n_peers], (long long)h->g);
    const int64_t nsend = h->send_ptr[n_peers];
    for (int64_t j = 0; j < nsend; ++j)
        CHECK(h, send_idx[j] >= 0 && send_idx[j] < h->n, "prcg_set_halo: send index %d out of range", send_idx[j]);
    HIPCHK(h, h->send_idx.alloc((size_t)(nsend + 1) * sizeof(int32_t)));
    HIPCHK(h, h->send_buf.alloc((size_t)(nsend + 1) * 2 * sizeof(double)));
    if (nsend > 0)
        HIPCHK(h, hipMemcpy(h->send_idx.p, send_idx, (size_t)nsend * sizeof(int32_t), hipMemcpyHostToDevice));
    h->send_idx_host.assign(send_idx, send_idx + nsend);
    h->peer_ok = false;
    h->have_halo = true;
    h->gather_planned = false;
    return PRCG_OK;
}

int prcg_peer_setup(prcg_t* h, int64_t max_ghost_any_rank, void* ipc_handle64, void** local_ptr) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->have_csr, "prcg_peer_setup: call prcg_set_csr (and prcg_set_halo) first");
    CHECK(h, h->nranks >= 1 && h->nranks <= kMaxPeerRanks, "prcg_peer_setup: 1..%d ranks", kMaxPeerRanks);
    CHECK(h, max_ghost_any_rank >= h->g && max_ghost_any_rank < (int64_t)1 << 28, "prcg_peer_setup: ghost capacity %lld < this rank's %lld ghosts",
          (long long)max_ghost_any_rank, (long long)h->g);
    HIPCHK(h, hipSetDevice(h->dev));
    h->peer_ok = false;
    const size_t bytes = peer_buffer_doubles(h->nranks, (int)max_ghost_any_rank) * sizeof(double) + 4096;
    if (!h->xbuf || h->xbuf_bytes < bytes || h->ghost_cap != (int)max_ghost_any_rank) {
        CHECK(h, h->peer_opened.empty() && !h->xbuf, "prcg_peer_setup: the exchange buffer of a connected handle cannot change size");
        // fine-grained device memory: stores of OTHER GPUs become visible to this one's loads without a kernel boundary
        h->xbuf_fine = hipExtMallocWithFlags(&h->xbuf, bytes, hipDeviceMallocFinegrained) == hipSuccess;
        if (!h->xbuf_fine) {
            (void)hipGetLastError();
            HIPCHK(h, hipMalloc(&h->xbuf, bytes));
        }
        h->xbuf_bytes = bytes;
        h->ghost_cap = (int)max_ghost_any_rank;
        HIPCHK(h, hipMemset(h->xbuf, 0, bytes));
        HIPCHK(h, hipDeviceSynchronize());
    }
    if (ipc_handle64) {
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI hands IPC handles around as 64 bytes");
        hipIpcMemHandle_t hd;
        memset(&hd, 0, sizeof hd);
        if (hipIpcGetMemHandle(&hd, h->xbuf) != hipSuccess) { (void)hipGetLastError(); memset(&hd, 0, sizeof hd); }   // same-process peers still work
        memcpy(ipc_handle64, &hd, sizeof hd);
    }
    if (local_ptr) *local_ptr = h->xbuf;
    return PRCG_OK;
}

int prcg_peer_connect(prcg_t* h, const void* ipc_handles, void* const* same_process_ptrs, const int64_t* send_dst_off) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->xbuf != nullptr, "prcg_peer_connect: call prcg_peer_setup first");
    CHECK(h, h->g == 0 || h->have_halo, "prcg_peer_connect: ghost columns but no halo plan");
    const int np = h->have_halo ? h->n_peers : 0;
    CHECK(h, np == 0 || send_dst_off != nullptr, "prcg_peer_connect: null destination offsets");
    HIPCHK(h, hipSetDevice(h->dev));
    h->peer_ok = false;
    const int R = h->nranks;
    PeerDev& P = h->peer_host;
    if (h->peer_opened.empty()) {
        for (int q = 0; q < kMaxPeerRanks; ++q) P.peer[q] = nullptr;
        for (int q = 0; q < R; ++q) {
            if (q == h->rank) { P.peer[q] = static_cast<double*>(h->xbuf); continue; }
            if (same_process_ptrs && same_process_ptrs[q]) { P.peer[q] = static_cast<double*>(same_process_ptrs[q]); continue; }
            CHECK(h, ipc_handles != nullptr, "prcg_peer_connect: no handle for rank %d", q);
            hipIpcMemHandle_t hd;
            memcpy(&hd, static_cast<const char*>(ipc_handles) + (size_t)q * sizeof hd, sizeof hd);
            void* mapped = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&mapped, hd, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
                return fail(h, PRCG_EHIP, "hipIpcOpenMemHandle for rank %d's exchange buffer: %s", q, hipGetErrorString(e));
            h->peer_opened.push_back(mapped);
            P.peer[q] = static_cast<double*>(mapped);
        }
    }
    P.mine = static_cast<double*>(h->xbuf);
    P.rank = h->rank; P.nranks = R; P.n_own = (int)h->n; P.ghost_cap = h->ghost_cap; P.epoch = 0;
    // send entries {row, destination rank, index in its ghost area}, grouped by the tile that owns the row
    struct Ent { int32_t row, peer, dst, tile; };
    std::vector<Ent> ents;
    if (h->win && np > 0) {
        const size_t nt = h->wt_rb.size();
        std::vector<int32_t> order(nt);
        for (size_t i = 0; i < nt; ++i) order[i] = (int32_t)i;
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return h->wt_rb[a] < h->wt_rb[b]; });
        for (int q = 0; q < np; ++q) {
            const int64_t cnt = h->send_ptr[q + 1] - h->send_ptr[q];
            CHECK(h, send_dst_off[q] >= 0 && send_dst_off[q] + cnt <= h->ghost_cap,
                  "prcg_peer_connect: rows for rank %d would land outside its ghost area", h->peer_rank[q]);
            for (int64_t j = 0; j < cnt; ++j) {
                const int32_t row = h->send_idx_host[(size_t)(h->send_ptr[q] + j)];
                // the tile whose rows hold `row`: last tile (by first row) that starts at or before it
                size_t lo = 0, hi = nt;
                while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (h->wt_rb[order[mid]] <= row) lo = mid; else hi = mid; }
                const int32_t t = order[lo];
                CHECK(h, row >= h->wt_rb[t] && row < h->wt_re[t], "prcg_peer_connect: row %d lies in no tile", row);
                ents.push_back(Ent{row, h->peer_rank[q], (int32_t)(send_dst_off[q] + j), t});
            }
        }
        std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.tile != b.tile ? a.tile < b.tile : a.row < b.row; });
    }
    const size_t nt = h->wt_rb.size();
    std::vector<int32_t> tsend(2 * (nt + 1), 0), flat(4 * (ents.size() + 1), 0);
    {
        size_t e = 0;
        for (size_t t = 0; t < nt; ++t) {
            tsend[2 * t] = (int32_t)e;
            while (e < ents.size() && ents[e].tile == (int32_t)t) ++e;
            tsend[2 * t + 1] = (int32_t)e;
        }
        for (size_t i = 0; i < ents.size(); ++i) { flat[4 * i] = ents[i].row; flat[4 * i + 1] = ents[i].peer; flat[4 * i + 2] = ents[i].dst; }
    }
    HIPCHK(h, h->peer_tile_send.alloc(tsend.size() * sizeof(int32_t)));
    HIPCHK(h, hipMemcpy(h->peer_tile_send.p, tsend.data(), tsend.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(h, h->peer_ents.alloc(flat.size() * sizeof(int32_t)));
    HIPCHK(h, hipMemcpy(h->peer_ents.p, flat.data(), flat.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(h, h->peer_dev.alloc(sizeof(PeerDev)));
    P.tile_send = static_cast<const int2*>(h->peer_tile_send.p);
    P.send_ent = static_cast<const int4*>(h->peer_ents.p);
    P.n_send = (int)ents.size();
    HIPCHK(h, hipMemcpy(h->peer_dev.p, &P, sizeof P, hipMemcpyHostToDevice));
    if (h->win) {
        launch_flag_send_tiles(h->sc, h->wtiles.p, h->peer_tile_send.p, (int)nt);
        HIPCHK(h, hipStreamSynchronize(h->sc));
    }
    // (a non-window operator keeps the two-kernel schedule: the exchange is the iteration launch's own, and only the
    //  window kernels have it)
    h->peer_ok = h->win;
    return PRCG_OK;
}

int prcg_world_init(prcg_t* h, int rank, int nranks) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, nranks >= 1 && rank >= 0 && rank < nranks, "prcg_world_init: bad rank %d of %d", rank, nranks);
    CHECK(h, h->comm == nullptr, "prcg_world_init: the handle already has a communicator");
    h->rank = rank;
    h->nranks = nranks;
    return PRCG_OK;
}

int prcg_peer_selftest(prcg_t* h, int k, const double* rows2n, const double* slot5, double* sums5, double* ghost2g) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->peer_ok, "prcg_peer_selftest: call prcg_peer_setup / prcg_peer_connect first");
    CHECK(h, k >= 0 && rows2n && slot5 && sums5, "prcg_peer_selftest: bad argument");
    HIPCHK(h, hipSetDevice(h->dev));
    const PeerDev* px = static_cast<const PeerDev*>(h->peer_dev.p);
    DevBuf rows, slot, out, err;
    HIPCHK(h, rows.alloc((size_t)2 * (h->n + 1) * sizeof(double)));
    HIPCHK(h, slot.alloc(64));
    HIPCHK(h, out.alloc(64));
    HIPCHK(h, err.alloc(64));
    HIPCHK(h, hipMemcpy(rows.p, rows2n, (size_t)2 * h->n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(slot.p, slot5, 5 * sizeof(double), hipMemcpyHostToDevice));
    if (h->peer_host.epoch == 0) {
        h->peer_host.epoch = (unsigned long long)(++h->peer_epoch) << 32;
        HIPCHK(h, hipMemcpy(h->peer_dev.p, &h->peer_host, sizeof(PeerDev), hipMemcpyHostToDevice));
    }
    launch_peer_push(h->sc, px, rows.d(), slot.d(), k, 1);
    launch_peer_collect(h->sc, px, k, nullptr, 0, out.d(), nullptr, static_cast<unsigned*>(err.p));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    unsigned e = 0;
    HIPCHK(h, hipMemcpy(&e, err.p, sizeof e, hipMemcpyDeviceToHost));
    if (e) return fail(h, PRCG_ERCCL, "prcg_peer_selftest: the other ranks' slots of round %d did not arrive", k);
    HIPCHK(h, hipMemcpy(sums5, out.p, 5 * sizeof(double), hipMemcpyDeviceToHost));
    if (ghost2g && h->g > 0)
        HIPCHK(h, hipMemcpy(ghost2g, static_cast<const double*>(h->xbuf) + peer_ghost_off(h->nranks, h->ghost_cap, k & 1),
                            (size_t)2 * h->g * sizeof(double), hipMemcpyDeviceToHost));
    return PRCG_OK;
}

static int timed_product(prcg_t* h, int nc, const double* in, double* out, int reps, double* ms_avg) {
    CHECK(h, h->have_csr, "no matrix: call prcg_set_csr first");
    CHECK(h, in && out && reps >= 1, "bad argument");
    CHECK(h, h->g == 0 || (h->multi() && h->have_halo), "ghost columns need a communicator and a halo plan");
    HIPCHK(h, hipSetDevice(h->dev));
    int rc = h2d(h, h->tmp_ext.d(), in, h->n * nc);
    if (rc) return rc;
    if ((rc = exchange(h, h->tmp_ext.d(), nc, h->sc))) return rc;
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a));
    HIPCHK(h, hipEventCreate(&b));
    double total = 0.0;
    for (int i = 0; i < reps; ++i) {
        HIPCHK(h, hipEventRecord(a, h->sc));
        if (nc == 1)
            LAUNCHCHK(h, eng_spmv(h, h->sc, 0, h->tmp_ext.d(), h->t1.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr));
        else
            LAUNCHCHK(h, eng_spmm2(h, h->sc, 0, h->tmp_ext.d(), h->t1.d(), 3));
        HIPCHK(h, hipEventRecord(b, h->sc));
        HIPCHK(h, hipEventSynchronize(b));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, a, b));
        total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (ms_avg) *ms_avg = total / reps;
    return d2h(h, out, h->t1.d(), h->n * nc);
}

int prcg_spmv(prcg_t* h, const double* x, double* y, int reps, double* ms_avg) {
    if (!h) return PRCG_EINVAL;
    return timed_product(h, 1, x, y, reps, ms_avg);
}

int prcg_stream_ceiling(prcg_t* h, int64_t n_pairs, int mode, int reps, double* gbytes_per_s) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, n_pairs >= 1024 && mode >= 0 && mode <= 3 && reps >= 1 && gbytes_per_s, "prcg_stream_ceiling: bad argument");
    HIPCHK(h, hipSetDevice(h->dev));
    DevBuf a, b, c;
    const size_t bytes = (size_t)n_pairs * 16;
    HIPCHK(h, a.alloc(bytes));
    const bool mix = mode == 1 || mode == 2;
    HIPCHK(h, b.alloc(mix ? bytes : 64));
    HIPCHK(h, c.alloc(mix ? bytes : 64));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_stream_probe(h->sc, mode, a.d(), b.d(), c.d(), (size_t)n_pairs);
    HIPCHK(h, hipEventRecord(e0, h->sc));
    for (int i = 0; i < reps; ++i) launch_stream_probe(h->sc, mode, a.d(), b.d(), c.d(), (size_t)n_pairs);
    HIPCHK(h, hipEventRecord(e1, h->sc));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double moved = (double)bytes * (mix ? 4.0 : 1.0) * reps;
    *gbytes_per_s = ms > 0.f ? moved / (ms * 1e-3) * 1e-9 : 0.0;
    return PRCG_OK;
}

int prcg_mix_ceiling(prcg_t* h, int64_t n_rows, int stream_kb_per_64_rows, int reps, double* gbytes_per_s) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, n_rows >= 4096 && stream_kb_per_64_rows >= 0 && stream_kb_per_64_rows <= 256 && reps >= 1 && gbytes_per_s, "prcg_mix_ceiling: bad argument");
    HIPCHK(h, hipSetDevice(h->dev));
    DevBuf v, x, r, rn;
    const size_t pieces = (size_t)n_rows / 64;
    const size_t vbytes = (pieces + 1) * (size_t)stream_kb_per_64_rows * 1024 + 4096, pbytes = ((size_t)n_rows + 64) * 16;
    HIPCHK(h, v.alloc(vbytes));
    HIPCHK(h, x.alloc(pbytes));
    HIPCHK(h, r.alloc(pbytes));
    HIPCHK(h, rn.alloc(pbytes));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_stream_mix(h->sc, v.d(), x.d(), r.d(), rn.d(), (size_t)n_rows, stream_kb_per_64_rows);
    HIPCHK(h, hipEventRecord(e0, h->sc));
    for (int i = 0; i < reps; ++i) launch_stream_mix(h->sc, v.d(), x.d(), r.d(), rn.d(), (size_t)n_rows, stream_kb_per_64_rows);
    HIPCHK(h, hipEventRecord(e1, h->sc));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double moved = ((double)pieces * stream_kb_per_64_rows * 1024.0 + 64.0 * (double)pieces * 64.0) * reps;
    *gbytes_per_s = ms > 0.f ? moved / (ms * 1e-3) * 1e-9 : 0.0;
    return PRCG_OK;
}

int prcg_spmv_ext(prcg_t* h, const double* x_ext, double* y) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->have_csr, "no matrix: call prcg_set_csr first");
    CHECK(h, x_ext && y, "bad argument");
    HIPCHK(h, hipSetDevice(h->dev));
    int rc = h2d(h, h->tmp_ext.d(), x_ext, h->n + h->g);
    if (rc) return rc;
    // interior tiles, then the tiles that touch ghost columns: the two launches of the overlapped schedule
    LAUNCHCHK(h, eng_spmv(h, h->sc, 1, h->tmp_ext.d(), h->t1.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr));
    LAUNCHCHK(h, eng_spmv(h, h->sc, 2, h->tmp_ext.d(), h->t1.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr));
    return d2h(h, y, h->t1.d(), h->n);
}

int prcg_spmm2(prcg_t* h, const double* rs, double* wu, int reps, double* ms_avg) {
    if (!h) return PRCG_EINVAL;
    return timed_product(h, 2, rs, wu, reps, ms_avg);
}

// What the row results of a one-launch iteration cost beside the operator's read stream depends on WHERE the written arrays lie:
// the same byte mix runs at two speeds from process to process (tools/mixbench.hip: s4b's mix 467-477 or 530-560 us, the read
// stream alone 443 either way; r04_sweeps.md D, K) and from allocation to allocation inside a process.  The session's three
// large arrays -- (x,p), the (r,s) pairs and their second copy -- are therefore allocated k times, every placement is timed with
// the mix probe against the operator's own stream (k_stream_mix: reads of the stream, reads of two pair arrays, nontemporal
// writes of two), and the fastest is kept.  A few milliseconds per placement, once per handle (the allocations are kept for
// later sessions).  Buffers are zeroed again afterwards; nothing has been written into them yet.
int place_session_vectors(prcg_t* h, size_t xp_bytes, size_t rs_bytes) {
    const int64_t n = h->n;
    const double* stream = nullptr;
    size_t stream_bytes = 0;
    if (h->sell) { stream = h->val_sell(); stream_bytes = h->sval.bytes; }
    else if (h->val.p && h->val.bytes >= (size_t)h->nnz * 8) { stream = h->val.d(); stream_bytes = (size_t)h->nnz * 8; }
    const size_t pieces = (size_t)n / 64;
    if (pieces < 4096 || rs_bytes < (size_t)n * 16 || xp_bytes < (size_t)n * 16) return PRCG_OK;      // (262,144 rows and more: smaller sessions are launch-bound)
    // KB of stream per 64 rows, so that the probe stays inside the stream's allocation (and a dictionary operator: none)
    int kb = 0;
    if (stream && !(h->win && (h->win_vd || h->win_pat))) {
        kb = (int)std::min<size_t>(stream_bytes / (pieces + 1) / 1024, 200);
    }
    if (kb == 0) stream = h->xp.d();               // (the probe's first loads read one line of it)
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    auto probe = [&](double* xp, double* rs, double* rs2, float* ms) -> int {
        launch_stream_mix(h->sc, stream, xp, rs, rs2, (size_t)n, kb);
        HIPCHK(h, hipEventRecord(e0, h->sc));
        for (int i = 0; i < 3; ++i) launch_stream_mix(h->sc, stream, xp, rs, rs2, (size_t)n, kb);
        HIPCHK(h, hipEventRecord(e1, h->sc));
        HIPCHK(h, hipEventSynchronize(e1));
        HIPCHK(h, hipEventElapsedTime(ms, e0, e1));
        return PRCG_OK;
    };
    float best = 0.f;
    int rc = probe(h->xp.d(), h->rs.d(), h->rs2.d(), &best);
    if (rc) return rc;
    if (getenv("PRCG_PLAN_DEBUG")) fprintf(stderr, "place_session_vectors: placement 0: %.1f us (stream %d KB per 64 rows)\n", best / 3 * 1e3, kb);
    // (every placement stays allocated until the choice is made: a freed one would be handed out again for the next)
    std::vector<std::array<DevBuf, 3>> cand((size_t)h->place_k - 1);
    int best_c = -1;
    for (int c = 0; c + 1 < h->place_k; ++c) {
        DevBuf& cx = cand[(size_t)c][0]; DevBuf& cr = cand[(size_t)c][1]; DevBuf& cr2 = cand[(size_t)c][2];
        if (cx.alloc(xp_bytes, false) != hipSuccess || cr.alloc(rs_bytes, false) != hipSuccess || cr2.alloc(rs_bytes, false) != hipSuccess) {
            (void)hipGetLastError();                 // (no room for another placement: the ones so far compete; the error is not the session's)
            break;
        }
        HIPCHK(h, hipMemsetAsync(cx.p, 0, xp_bytes, h->sc));
        HIPCHK(h, hipMemsetAsync(cr.p, 0, rs_bytes, h->sc));
        HIPCHK(h, hipMemsetAsync(cr2.p, 0, rs_bytes, h->sc));
        float ms = 0.f;
        if ((rc = probe(cx.d(), cr.d(), cr2.d(), &ms))) return rc;
        if (getenv("PRCG_PLAN_DEBUG")) fprintf(stderr, "place_session_vectors: placement %d: %.1f us\n", c + 1, ms / 3 * 1e3);
        if (ms < best) { best = ms; best_c = c; }
    }
    if (best_c >= 0) {
        DevBuf& cx = cand[(size_t)best_c][0]; DevBuf& cr = cand[(size_t)best_c][1]; DevBuf& cr2 = cand[(size_t)best_c][2];
        std::swap(h->xp.p, cx.p); std::swap(h->xp.cap, cx.cap); std::swap(h->xp.bytes, cx.bytes);
        std::swap(h->rs.p, cr.p); std::swap(h->rs.cap, cr.cap); std::swap(h->rs.bytes, cr.bytes);
        std::swap(h->rs2.p, cr2.p); std::swap(h->rs2.cap, cr2.cap); std::swap(h->rs2.bytes, cr2.bytes);
    }
    cand.clear();                                   // (the others are freed)
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(h, hipMemsetAsync(h->xp.p, 0, h->xp.bytes, h->sc));
    HIPCHK(h, hipMemsetAsync(h->rs.p, 0, h->rs.bytes, h->sc));
    HIPCHK(h, hipMemsetAsync(h->rs2.p, 0, h->rs2.bytes, h->sc));
    h->placed_xp = h->xp.p;
    return PRCG_OK;
}

int prcg_solve_begin(prcg_t* h, int variant, const double* b, const double* x0, int max_iter, const double* x_true,
                     const double* inv_diag, uint32_t hist_mask) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->have_csr, "prcg_solve_begin: call prcg_set_csr first");
    CHECK(h, variant >= 0 && variant < PRCG_NUM_VARIANTS, "prcg_solve_begin: unknown variant %d", variant);
    CHECK(h, b && x0, "prcg_solve_begin: null b or x0");
    CHECK(h, max_iter >= 1, "prcg_solve_begin: max_iter must be >= 1");
    CHECK(h, (hist_mask & ~PRCG_HIST_ALL) == 0, "prcg_solve_begin: unknown history bits");
    CHECK(h, !(hist_mask & (PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM)) || x_true,
          "prcg_solve_begin: error histories need x_true");
    CHECK(h, h->g == 0 || (h->multi() && h->have_halo), "ghost columns need a communicator and a halo plan");
    HIPCHK(h, hipSetDevice(h->dev));
    // every vector that feeds a matrix product has ghost room AND kGatherPad spare entries: the
    // narrow column encodings decode a few out-of-tile bytes per tile against the tile's own base
    // (products nobody reads); the pad keeps those gathers inside the allocation without a test
    const int64_t n = h->n, ne = h->n + h->g + kGatherPad;
    const size_t D = sizeof(double);
    h->in_session = false;
    h->variant = variant;
    h->fused = false;
    h->hs_fused = false;
    h->cg_fused = false;
    h->pr_fused = false;
    h->pr_packed = false;
    h->hs_pend_mu = 0;
    h->small = false; h->small_hs = false;
    h->medium = false;
    h->gather = false;
    h->cb_session = h->cb != nullptr && inv_diag == nullptr;
    h->prec = inv_diag != nullptr || h->cb_session;
    CHECK(h, !(h->cb_session && h->multi()), "a host-callback preconditioner runs on one GPU only");
    if (h->cb_session) HIPCHK(h, h->cb_stage.ensure((size_t)h->n * sizeof(double), h->sc));
    h->max_iter = max_iter;
    h->hist_mask = hist_mask;
    h->have_xtrue = x_true != nullptr;
    h->k = 0;
    h->n_ev_spmv = h->n_ev_upd = 0;

    HIPCHK(h, h->x.ensure((size_t)n * D, h->sc));
    HIPCHK(h, h->b.ensure((size_t)n * D, h->sc));
    HIPCHK(h, h->xt.ensure((size_t)n * D, h->sc));
    HIPCHK(h, h->e_ext.ensure((size_t)ne * D, h->sc));
    HIPCHK(h, h->dinv.ensure((size_t)ne * D, h->sc));     // (a window source of the Chronopoulos-Gear product launch)
    HIPCHK(h, h->dots.ensure((size_t)(max_iter + 1) * kNS * D, h->sc));
    HIPCHK(h, h->coef.ensure((size_t)(max_iter + 1) * kCoefStride * D, h->sc));
    int rc;
    if ((rc = h2d(h, h->x.d(), x0, n))) return rc;
    if ((rc = h2d(h, h->b.d(), b, n))) return rc;
    if (x_true && (rc = h2d(h, h->xt.d(), x_true, n))) return rc;
    if (inv_diag && (rc = h2d(h, h->dinv.d(), inv_diag, n))) return rc;
    hipStream_t sc = h->sc;
    double* tmp = h->tmp_ext.d();
    double* t1 = h->t1.d();

    // r0 = b - A x0   (hs_cg.py:23, pipe_pr_cg.py:23)
    launch_copy(sc, tmp, 1, h->x.d(), 1, n);
    if ((rc = dist_spmv(h, tmp, t1, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;

    if (is_pipe(variant)) {
        h->fused = h->want_fused && !h->multi() && h->g == 0 && !h->cb_session;
        // with a communicator: the same kernel in its deferred form (window operators only)
        h->fused_comm = false;
        h->red_pending = false;
        // direct peer exchange (every rank connected, window operator): the one-launch schedule without a collective.
        // Whether it is connected is the same on every rank (the host side connects all ranks or none).
        h->peer = h->peer_ok && h->want_peer && h->want_fused && h->multi() && h->win && !h->fused_final && !h->cb_session;
        if (!h->peer && (rc = plan_gather(h))) return rc;
        // with a communicator: the same kernel in its deferred form -- window operators whose halo rides on the
        // one all-gather per iteration (bands; the merged exchange).  Larger halos (send/recv + all-reduce chain)
        // keep the two-kernel schedule: RCCL's point-to-point path was seen to stall for ~1 s on its first use
        // from the communication stream while a launch waited for it (profiles/r02_sweeps.md).
        // (the one-launch schedule over the RCCL all-gather chain is opt-in, PRCG_FUSED_COMM=1: its launches wait inside the
        //  kernel for kernels of ANOTHER stream to become resident, which has only ever been validated with one rank)
        h->fused_comm = h->peer || (h->want_fused && h->want_fused_comm_rccl && h->multi() && h->win && !h->fused_final && h->gather);
        if (h->fused_comm) h->fused = true;        // state layout, derived vectors: as the one-launch schedule
        HIPCHK(h, h->xp.ensure((size_t)2 * n * D, h->sc));
        HIPCHK(h, h->rs.ensure((size_t)2 * (h->prec ? n : ne) * D, h->sc));
        HIPCHK(h, h->rs2.ensure((h->fused && !h->prec) ? (size_t)2 * ne * D : 16, h->sc));
        if (h->place_k > 1 && h->fused && !h->prec && h->placed_xp != h->xp.p && h->rs.bytes == h->rs2.bytes) {       // (every rank for itself: its vectors are its own)
            int prc = place_session_vectors(h, h->xp.bytes, h->rs.bytes);
            if (prc) return prc;
        }
        HIPCHK(h, h->rst2.ensure((h->fused && h->prec) ? (size_t)2 * ne * D : 16, h->sc));
        HIPCHK(h, h->wv.ensure((h->fused && !pipe_recompute(variant)) ? (size_t)n * D : 16, h->sc));
        HIPCHK(h, h->partC.ensure(h->fused ? (size_t)8192 * kPartialStride * sizeof(double) : 16, h->sc));
        HIPCHK(h, h->pub.ensure(h->fused_comm ? kPubDoubles * sizeof(double) : 16, h->sc));
        HIPCHK(h, h->pub_err.ensure(64, h->sc));
        h->pend_parts = 0; h->pend_k = -1; h->pend_buf = nullptr;
        h->rs_cur = h->rs.d();
        // one-workgroup solver: only when nothing but the recurrence residual is recorded
        h->small = h->fused && !h->fused_comm && h->want_small && !h->prec && pipe_recompute(variant) &&
                   !(hist_mask & (PRCG_HIST_RESIDUAL_2_NORM | PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM)) &&
                   small_fits(h->n, h->nnz, h->max_row_len, &h->small_mode);
        // few-workgroup solver for the systems the one-workgroup solver cannot hold: same conditions, up to 131,072 rows
        h->medium = !h->small && h->medium_ok && h->want_medium && h->fused && !h->fused_comm && !h->prec && pipe_recompute(variant) &&
                    !(hist_mask & (PRCG_HIST_RESIDUAL_2_NORM | PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM)) && !h->multi();
        HIPCHK(h, h->rst.ensure(h->prec ? (size_t)2 * ne * D : 16, h->sc));
        if (h->prec) h->rs_cur = h->rst.d();
        HIPCHK(h, h->wu.ensure((size_t)2 * n * D, h->sc));
        HIPCHK(h, h->wt.ensure(h->prec ? (size_t)n * D : 16, h->sc));
        HIPCHK(h, h->ut.ensure(h->cb_session ? (size_t)n * D : 16, h->sc));
        double* RS = h->rs.d();
        double* WU = h->wu.d();
        double* XP = h->xp.d();
        launch_copy(sc, XP, 2, h->x.d(), 1, n);                             // x = x0         :22
        launch_sub(sc, RS, 2, h->b.d(), 1, t1, 1, n);                       // r = b - A x
        if (!h->prec) {
            launch_copy(sc, XP + 1, 2, RS, 2, n);                           // p = r          :24
            launch_copy(sc, tmp, 1, RS, 2, n);
            if ((rc = dist_spmv(h, tmp, t1, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;
            launch_copy(sc, RS + 1, 2, t1, 1, n);                           // s = A p        :26
            launch_copy(sc, WU, 2, t1, 1, n);                               // w = s          :27
            launch_copy(sc, tmp, 1, t1, 1, n);
            if ((rc = dist_spmv(h, tmp, t1, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;
            launch_copy(sc, WU + 1, 2, t1, 1, n);                           // u = A w        :28
        } else {
            double* RST = h->rst.d();
            if ((rc = apply_prec(h, RS, 2, RST, 2))) return rc;            // r~ = M^-1 r    :124
            launch_copy(sc, XP + 1, 2, RST, 2, n);                          // p = r~         :125
            launch_copy(sc, tmp, 1, RST, 2, n);
            if ((rc = dist_spmv(h, tmp, t1, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;
            launch_copy(sc, RS + 1, 2, t1, 1, n);                           // s = A p        :127
            if ((rc = apply_prec(h, t1, 1, RST + 1, 2))) return rc;        // s~ = M^-1 s    :128
            launch_copy(sc, WU, 2, t1, 1, n);                               // w = s          :129
            launch_copy(sc, h->wt.d(), 1, RST + 1, 2, n);                   // w~ = s~        :130
            launch_copy(sc, tmp, 1, RST + 1, 2, n);
            if ((rc = dist_spmv(h, tmp, t1, kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;
            launch_copy(sc, WU + 1, 2, t1, 1, n);                           // u = A s~       :131
            if (h->cb_session && (rc = apply_prec(h, WU + 1, 2, h->ut.d(), 1))) return rc;   // u~ = M^-1 u  :132
        }
        if (h->fused && !pipe_recompute(variant)) launch_copy(sc, h->wv.d(), 1, WU, 2, n);   // the stored w of the 'p' flavours
        PipeUpdateArgs a = pipe_args(h, 0);
        const int grid = launch_pipe_dots(sc, a);                           // nu, mu, delta, gamma
        LAUNCHCHK(h, grid);
        if (!h->fused_final) launch_reduce_final(sc, h->partA.d(), grid, dots_at(h, 0), 0, 0, 5);
        if ((rc = allreduce(h, dots_at(h, 0), 5, sc))) return rc;
        if (h->gather) {
            // COLLECTIVE part, the same on every rank of a merged-exchange session whatever schedule the rank itself
            // ends up with (a rank's block may be no window operator, or its probe below may fail: the ranks then
            // still issue the same collectives in the same order -- one all-gather per iteration either way):
            // the first use of the communicator from the communication stream happens HERE, not inside a launch that
            // waits for it, and the ghosts of the initial input pairs are in place for the first boundary tiles.
            HIPCHK(h, hipStreamSynchronize(sc));
            double* slot = h->gbuf.d() + (size_t)h->rank * h->g_slot;
            NCCLCHK(h, h->rccl->AllGather(slot, h->gbuf.p, (size_t)h->g_slot, ncclDouble, h->comm, h->sm));
            HIPCHK(h, hipStreamSynchronize(h->sm));
            if ((rc = exchange(h, h->rs_cur, 2, sc))) return rc;
        }
        if (h->peer) {
            // The initial state into the exchange buffers: this rank's boundary rows into the neighbours' ghost areas and its
            // slot of "iteration 0" (rank 0 carries the reduced inner products).  Ordered behind the all-reduce above on this
            // stream, i.e. behind everything the neighbours still had in flight from an earlier session on theirs.
            h->peer_host.epoch = (unsigned long long)(++h->peer_epoch) << 32;
            h->peer_host.n_own = (int)h->n;
            HIPCHK(h, hipMemcpyAsync(h->peer_dev.p, &h->peer_host, sizeof(PeerDev), hipMemcpyHostToDevice, sc));
            HIPCHK(h, hipStreamSynchronize(sc));                       // (peer_host must not change under the copy)
            *h->err_host = 0u;
            launch_peer_push(sc, static_cast<const PeerDev*>(h->peer_dev.p), h->rs_cur, dots_at(h, 0), 0, h->rank == 0);
            launch_peer_collect(sc, static_cast<const PeerDev*>(h->peer_dev.p), 0, nullptr, 0, nullptr, h->pub.d(), static_cast<unsigned*>(h->pub_err.p));
        } else if (h->fused_comm) {
            // LOCAL part.  Can a kernel of the communication stream run while a kernel of the compute stream waits for
            // it?  (HIP may have mapped both streams to one hardware queue -- then the deferred form would only ever
            // time out.)  Probe once per session: a one-wave kernel on sc waits ~2 ms at most for a record that a
            // kernel on sm publishes.
            HIPCHK(h, hipStreamSynchronize(sc));
            launch_probe_wait(sc, h->pub.d(), 0x7f000000u, static_cast<unsigned*>(h->pub_err.p));
            launch_publish(h->sm, dots_at(h, 0), h->pub.d(), 0x7f000000u);
            HIPCHK(h, hipStreamSynchronize(h->sm));
            HIPCHK(h, hipStreamSynchronize(sc));
            unsigned perr = 0;
            HIPCHK(h, hipMemcpy(&perr, h->pub_err.p, sizeof perr, hipMemcpyDeviceToHost));
            HIPCHK(h, hipMemset(h->pub_err.p, 0, sizeof perr));
            if (perr) { h->fused_comm = false; h->fused = false; }       // streams are serialised here: two-kernel schedule
        }
        if (h->fused_comm && !h->peer) launch_publish(sc, dots_at(h, 0), h->pub.d(), 0u);   // iteration 1 waits for "0": the initial inner products
    } else if (is_cg_family(variant)) {
        // x, r, r~, w, w~ (all with ghost room: whichever feeds the SpMV), p, s, s~, u, t
        HIPCHK(h, h->p.ensure((size_t)ne * D, h->sc));
        h->p_cur = h->p.d();
        HIPCHK(h, h->r.ensure((size_t)ne * D, h->sc));
        h->cur_r = h->r.d();
        h->cg_fused = h->want_fused && !h->multi() && h->g == 0 && h->win && !h->cb_session &&
                      !(variant == PRCG_GV && h->replace_fn);      // (the predicate is called between the update and the product)
        // one launch per iteration: both variants unpreconditioned, Chronopoulos-Gear with Jacobi too
        h->cg_one = h->cg_fused && h->want_cg_one && (variant == PRCG_CG_CG || !h->prec);
        h->cg_lag = false;
        HIPCHK(h, h->r2.ensure((h->cg_fused && variant == PRCG_CG_CG) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->w2.ensure((h->cg_fused && (variant == PRCG_GV || h->cg_one)) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->s2.ensure((h->cg_one && variant == PRCG_CG_CG) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->u2.ensure((h->cg_one && variant == PRCG_GV) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->t2.ensure((h->cg_one && variant == PRCG_GV) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->partC.ensure(h->cg_one ? (size_t)8192 * kPartialStride * sizeof(double) : 16, h->sc));
        h->pend_parts = 0; h->pend_k = -1; h->pend_buf = nullptr;
        HIPCHK(h, h->rt.ensure(h->prec ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->w.ensure((size_t)ne * D, h->sc));
        h->cur_w = h->w.d();
        HIPCHK(h, h->wt.ensure(h->prec ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->s.ensure((size_t)ne * D, h->sc));
        HIPCHK(h, h->st.ensure(h->prec ? (size_t)n * D : 16, h->sc));
        HIPCHK(h, h->u.ensure((size_t)ne * D, h->sc));      // (a window source of the Ghysels-Vanroose product launch)
        HIPCHK(h, h->tvec.ensure((size_t)ne * D, h->sc));   // (... of its one-launch form)
        h->cur_s = h->s.d(); h->cur_u = h->u.d(); h->cur_t = h->tvec.d();
        launch_sub(sc, h->r.d(), 1, h->b.d(), 1, t1, 1, n);                 // r = b - A x      cg_cg.py:23
        if (h->prec && (rc = apply_prec(h, h->r.d(), 1, h->rt.d(), 1))) return rc;   // r~ = M^-1 r   :89
        double* z = h->prec ? h->rt.d() : h->r.d();
        launch_copy(sc, h->p.d(), 1, z, 1, n);                              // p = r~           :25 / :91
        int grid = 0;
        if (variant == PRCG_CG_CG) {
            // w = A r~ with nu = r.r~, eta = w.r~ (:24,26,27); s = A p with mu = p.s (:28,30)
            if ((rc = dist_spmv(h, z, h->w.d(), kEpiCG, h->r.d(), nullptr, nullptr, &grid))) return rc;
            launch_reduce_final(sc, h->partB.d(), grid, dots_at(h, 0), 0, 0, 5);
            if ((rc = dist_spmv(h, h->p.d(), h->s.d(), kEpiDotXY, nullptr, nullptr, nullptr, &grid))) return rc;
            launch_reduce_final(sc, h->partB.d(), grid, dots_at(h, 0), 0, PRCG_S_MU, 1);
            if ((rc = allreduce(h, dots_at(h, 0), 5, sc))) return rc;
        } else {
            // gv_cg.py:26-33 / gv_pcg :96-109
            if ((rc = dist_spmv(h, z, h->w.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;   // w = A r~
            if (h->prec && (rc = apply_prec(h, h->w.d(), 1, h->wt.d(), 1))) return rc;                         // w~
            launch_copy(sc, h->s.d(), 1, h->w.d(), 1, n);                                                      // s = w
            if (h->prec) launch_copy(sc, h->st.d(), 1, h->wt.d(), 1, n);                                       // s~ = w~
            double* zt = h->prec ? h->wt.d() : h->w.d();
            if ((rc = dist_spmv(h, zt, h->u.d(), kEpiNone, nullptr, nullptr, nullptr, nullptr))) return rc;  // u = A w~
            CgArgs a = cg_args(h, 0);
            const int g1 = launch_gv_update1(sc, a, true);                                                     // nu, eta
            LAUNCHCHK(h, g1);
            launch_reduce_final(sc, h->partA.d(), g1, dots_at(h, 0), 0, 0, 5);
            // mu = p.s as a true inner product at the start (gv_cg.py:33)
            const int g2 = launch_dot(sc, h->p.d(), h->s.d(), n, h->partB.d(), 0);
            LAUNCHCHK(h, g2);
            launch_reduce_final(sc, h->partB.d(), g2, dots_at(h, 0), 0, PRCG_S_MU, 1);
            if ((rc = allreduce(h, dots_at(h, 0), 5, sc))) return rc;
        }
    } else {
        // HS and non-pipelined PR share the layout x, r, (r~), p(+ghosts), s, (s~)
        HIPCHK(h, h->p.ensure((size_t)ne * D, h->sc));
        h->p_cur = h->p.d();
        h->hs_fused = variant == PRCG_HS && h->want_fused && !h->multi() && h->g == 0 && !h->cb_session;
        // one-workgroup solver (k_small_hs): only when nothing but the recurrence residual is recorded
        h->small_hs = h->hs_fused && h->want_small && !h->prec &&
                      !(hist_mask & (PRCG_HIST_RESIDUAL_2_NORM | PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM)) &&
                      small_fits(h->n, h->nnz, h->max_row_len, &h->small_mode);
        // r (r~) is the staged-window source of the Hestenes-Stiefel product launch: like every vector that feeds a
        // product it has the spare entries behind its end (a window page of the last tile may reach past row n)
        HIPCHK(h, h->r.ensure((size_t)(h->debug_short_sources ? n : ne) * D, h->sc));
        HIPCHK(h, h->s.ensure((size_t)ne * D, h->sc));
        HIPCHK(h, h->rt.ensure(h->prec ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->st.ensure(h->prec ? (size_t)ne * D : 16, h->sc));
        // one launch per iteration for pr / m on a window operator: second copies of what the window is formed from
        h->pr_fused = is_pr(variant) && h->want_fused && !h->multi() && h->g == 0 && h->win && !h->cb_session;
        HIPCHK(h, h->p2.ensure(((h->hs_fused && h->win) || h->pr_fused) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->r2.ensure((h->pr_fused && !h->prec) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->s2.ensure((h->pr_fused && !h->prec) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->rt2.ensure((h->pr_fused && h->prec) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->st2.ensure((h->pr_fused && h->prec) ? (size_t)ne * D : 16, h->sc));
        HIPCHK(h, h->partC.ensure(h->pr_fused ? (size_t)8192 * kPartialStride * sizeof(double) : 16, h->sc));
        h->pr_packed = h->pr_fused && (h->want_pr_pack < 0 ? h->win_pat : h->want_pr_pack != 0) && !h->prec &&
                       !(hist_mask & (PRCG_HIST_RESIDUAL_2_NORM | PRCG_HIST_ERROR_A_NORM | PRCG_HIST_ERROR_2_NORM));
        h->pr_q_valid = false;
        HIPCHK(h, h->q.ensure(h->pr_packed ? (size_t)4 * ne * D : 16, h->sc));
        HIPCHK(h, h->q2.ensure(h->pr_packed ? (size_t)4 * ne * D : 16, h->sc));
        h->cur_r = h->r.d(); h->cur_s = h->s.d(); h->cur_rt = h->rt.d(); h->cur_st = h->st.d();
        h->pend_parts = 0; h->pend_k = -1; h->pend_buf = nullptr;
        launch_sub(sc, h->r.d(), 1, h->b.d(), 1, t1, 1, n);                 // r = b - A x
        if (h->prec) {
            if ((rc = apply_prec(h, h->r.d(), 1, h->rt.d(), 1))) return rc;   // r~ = M^-1 r
            launch_copy(sc, h->p.d(), 1, h->rt.d(), 1, n);                  // p = r~
        } else {
            launch_copy(sc, h->p.d(), 1, h->r.d(), 1, n);                   // p = r
        }
        if (variant == PRCG_HS) {
            HsArgs a = hs_args(h, 0);
            const int g1 = launch_hs_init_dots(sc, a);                      // nu = r.r~      hs_cg.py:25
            LAUNCHCHK(h, g1);
            launch_reduce_final(sc, h->partA.d(), g1, dots_at(h, 0), PRCG_S_NU, PRCG_S_NU, 2);
            if (h->cb_session) {        // nu = r~.r with the r~ the callback returned
                const int g2 = launch_dot(sc, h->rt.d(), h->r.d(), n, h->partB.d(), 0);
                LAUNCHCHK(h, g2);
                launch_reduce_final(sc, h->partB.d(), g2, dots_at(h, 0), 0, PRCG_S_NU, 1);
            }
            if ((rc = allreduce(h, dots_at(h, 0) + PRCG_S_NU, 2, sc))) return rc;
            int grid = 0;
            if ((rc = dist_spmv(h, h->p.d(), h->s.d(), kEpiDotXY, nullptr, nullptr, nullptr, &grid))) return rc;
            launch_reduce_final(sc, h->partB.d(), grid, dots_at(h, 0), 0, PRCG_S_MU, 1);   // mu = p.s  :27
            if ((rc = allreduce(h, dots_at(h, 0) + PRCG_S_MU, 1, sc))) return rc;
        } else {
            PrArgs a = pr_args(h, 0);
            const int g1 = launch_pr_init_dots(sc, a);                      // nu = r~.r      pr_cg.py:109
            LAUNCHCHK(h, g1);
            int grid = 0;
            if (h->cb_session) {
                if ((rc = pr_spmv_and_reduce(h, 0, g1))) return rc;                       // (s~ from the callback)
            } else {
                if ((rc = dist_spmv(h, h->p.d(), h->s.d(), kEpiPR, h->r.d(), h->prec ? h->dinv.d() : nullptr,
                                    h->prec ? h->st.d() : nullptr, &grid))) return rc;    // s, s~, mu, dl, gm
                launch_reduce_final(sc, h->partA.d(), g1, dots_at(h, 0), PRCG_S_NU, PRCG_S_NU, 2);
                launch_reduce_final(sc, h->partB.d(), grid, dots_at(h, 0), 0, 0, 3);
                if ((rc = allreduce(h, dots_at(h, 0), 5, sc))) return rc;
            }
        }
    }
    if ((rc = record(h, 0))) return rc;
    HIPCHK(h, hipStreamSynchronize(sc));
    h->in_session = true;
    return PRCG_OK;
}

int prcg_iterate(prcg_t* h, int iters) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session, "prcg_iterate: no open session");
    CHECK(h, iters >= 0, "prcg_iterate: negative count");
    CHECK(h, h->k + iters <= h->max_iter, "prcg_iterate: %d more iterations exceed max_iter=%d (k=%d)", iters,
          h->max_iter, h->k);
    HIPCHK(h, hipSetDevice(h->dev));
    if (h->fused_comm && *h->err_host != 0u)
        return fail(h, PRCG_ERCCL, "a one-launch iteration waited more than its bound for the other ranks (a peer stalled or died); "
                                   "the session's results are invalid -- PRCG_PEER=0 / PRCG_FUSED_COMM=0 select the two-kernel schedule");
    if (h->small_hs && iters > 0) {
        // Hestenes-Stiefel: all `iters` iterations inside one launch of one workgroup
        hs_flush(h);
        SmallArgs sa{};
        sa.n = (int)h->n; sa.nnz = (int)h->nnz;
        sa.indptr = h->indptr.i(); sa.col = h->col.i(); sa.val = h->val.d();
        sa.xp = h->x.d(); sa.rs = h->r.d(); sa.hs_p = h->p_cur; sa.hs_s = h->s.d();
        sa.dots = h->dots.d(); sa.coef = h->coef.d();
        sa.k0 = h->k; sa.iters = iters; sa.meurant = 0;
        bool on = false;
        prof_begin(h, h->ev_spmv, h->n_ev_spmv, 0, on);
        LAUNCHCHK(h, launch_small_hs(h->sc, sa, h->small_mode));
        prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        h->k += iters;
        return PRCG_OK;
    }
    if (h->small && iters > 0) {
        // all `iters` iterations inside one launch of one workgroup
        SmallArgs sa{};
        sa.n = (int)h->n; sa.nnz = (int)h->nnz;
        sa.indptr = h->indptr.i(); sa.col = h->col.i(); sa.val = h->val.d();
        sa.xp = h->xp.d(); sa.rs = h->rs_cur;
        sa.dots = h->dots.d(); sa.coef = h->coef.d();
        sa.k0 = h->k; sa.iters = iters; sa.meurant = meurant(h->variant);
        bool on = false;
        prof_begin(h, h->ev_spmv, h->n_ev_spmv, 0, on);
        LAUNCHCHK(h, launch_small_pipe_pr(h->sc, sa, h->small_mode));
        prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
        h->k += iters;
        return PRCG_OK;
    }
    if (h->medium && iters > 0) {
        // all `iters` iterations inside one launch of a few co-operating workgroups (prcg_medium.hip)
        for (int left = iters; left > 0;) {
            const int now = left < (1 << 20) ? left : (1 << 20);
            MediumArgs ma{};
            ma.n = (int)h->n; ma.G = h->med_groups;
            ma.slices = static_cast<const int4*>(h->m_slices.p);
            ma.val = h->m_val.d(); ma.col16 = static_cast<const unsigned short*>(h->m_col.p);
            ma.rows = static_cast<const int*>(h->m_rows.p); ma.indptr = h->indptr.i();
            ma.wave_first = static_cast<const int*>(h->m_wave_first.p);
            ma.wg_window = static_cast<const int2*>(h->m_window.p);
            ma.wg_own = static_cast<const int2*>(h->m_own.p);
            ma.xp = h->xp.d(); ma.rs = h->rs_cur; ma.exch = h->m_exch.d(); ma.slots = h->m_slots.d();
            ma.dots = h->dots.d(); ma.coef = h->coef.d();
            ma.k0 = h->k; ma.iters = now; ma.meurant = meurant(h->variant);
            ma.seq = ++h->med_seq; ma.err = static_cast<unsigned*>(h->m_err.p);
            bool on = false;
            prof_begin(h, h->ev_spmv, h->n_ev_spmv, 0, on);
            LAUNCHCHK(h, launch_medium_pipe_pr(h->sc, ma, h->med_window));
            prof_end(h, h->ev_spmv, h->n_ev_spmv, on);
            h->k += now;
            left -= now;
        }
        return PRCG_OK;
    }
    for (int i = 0; i < iters; ++i) {
        const int k = h->k + 1;
        int rc;
        if (is_pipe(h->variant)) rc = iterate_pipe(h, k);
        else if (h->variant == PRCG_HS) rc = h->hs_fused ? iterate_hs_fused(h, k) : iterate_hs(h, k);
        else if (is_cg_family(h->variant) && h->cg_one) rc = iterate_cg_one(h, k);
        else if (h->variant == PRCG_CG_CG) rc = h->cg_fused ? iterate_cgcg_fused(h, k) : iterate_cgcg(h, k);
        else if (h->variant == PRCG_GV) rc = h->cg_fused ? iterate_gv_fused(h, k) : iterate_gv(h, k);
        else rc = h->pr_fused ? iterate_pr_fused(h, k) : iterate_pr(h, k);
        if (rc) return rc;
        if ((rc = record(h, k))) return rc;
        h->k = k;
    }
    if (h->fused && !h->fused_comm) fused_flush(h);    // dots of the last iteration: one reduction per call, not per iteration
    if (h->hs_fused) hs_flush(h);
    if (h->pr_fused) { fused_flush(h); pr_unpack(h); }
    if (h->cg_one) cg_flush(h);
    if (h->fused_comm && h->red_pending) {
        // the caller may read or rewrite state next (recorders, teacher forcing): finish the exchange of the last iteration
        HIPCHK(h, hipStreamWaitEvent(h->sc, h->red_event, 0));
    }
    if (h->peer && h->pend_parts > 0 && h->pend_k == h->k) {
        // the inner products of the last iteration: this rank's slot goes out now (no next launch to send it), then every
        // rank's slot is added in rank order
        launch_peer_collect(h->sc, static_cast<const PeerDev*>(h->peer_dev.p), h->k, h->pend_buf, h->pend_parts, dots_at(h, h->k),
                            h->pub.d(), static_cast<unsigned*>(h->pub_err.p));
        h->pend_parts = 0;
    }
    return PRCG_OK;
}

int prcg_sync(prcg_t* h) {
    if (!h) return PRCG_EINVAL;
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->sh));
    HIPCHK(h, hipStreamSynchronize(h->sm));
    HIPCHK(h, hipStreamSynchronize(h->sc));
    if (h->in_session && h->fused_comm && h->pub_err.p) {
        unsigned err = 0;
        HIPCHK(h, hipMemcpy(&err, h->pub_err.p, sizeof err, hipMemcpyDeviceToHost));
        if (err) *h->err_host = 1u;
        if (err) return fail(h, PRCG_ERCCL, "a one-launch iteration waited more than its bound for the reduced inner products "
                                            "(communication stream starved or a peer stalled); results are invalid. "
                                            "PRCG_FUSED_COMM=0 selects the two-kernel schedule");
    }
    if (h->in_session && h->medium && h->m_err.p) {
        unsigned err = 0;
        HIPCHK(h, hipMemcpy(&err, h->m_err.p, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(h, PRCG_EHIP, "the few-workgroup solver waited more than its bound for one of its workgroups (not all of them "
                                           "resident?); results are invalid.  PRCG_MEDIUM=0 selects one launch per iteration");
    }
    return PRCG_OK;
}

int prcg_iteration(const prcg_t* h) { return h ? h->k : -1; }

int64_t prcg_operator_bytes(const prcg_t* h) {
    if (!h || !h->have_csr) return -1;
    if (h->win) return h->win_stream_bytes;
    if (h->sell) return h->sell_bytes;
    // CSR-adaptive tiles: row pointers, tile table, column stream as encoded, values or dictionary indices
    const int64_t colb = h->c8_int ? 1 : (h->c16_int ? 2 : 4);
    const int64_t nt = (int64_t)h->nt_int + h->nt_bnd;
    return 4 * (h->n + 1) + nt * (int64_t)sizeof(Tile) + h->nnz * colb + (h->vd_int ? h->nnz + nt * 16 : h->nnz * 8);
}

int prcg_schedule(const prcg_t* h) {
    if (!h) return -1;
    return ((h->fused || h->hs_fused || h->pr_fused || h->cg_fused) ? PRCG_SCHED_FUSED : 0) | (h->fused_comm ? PRCG_SCHED_FUSED_COMM : 0) |
           (h->peer ? PRCG_SCHED_PEER : 0) | (h->sell ? PRCG_SCHED_SELL | PRCG_SCHED_COL16 : 0) | ((h->small || h->small_hs) ? PRCG_SCHED_SMALL : 0) | (h->medium ? PRCG_SCHED_MEDIUM : 0) | (h->comm ? PRCG_SCHED_COMM : 0) |
           (h->gather ? PRCG_SCHED_GATHER : 0) | (h->comm_halo ? PRCG_SCHED_DUAL_COMM : 0) | ((h->steps & 15) << 8) |
           (h->stream_stores ? PRCG_SCHED_STREAM_STORES : 0) | ((h->sell && h->sell_sigma > 64) ? PRCG_SCHED_SELL_SORTED : 0) |
           ((h->sell && h->sell_nt) ? PRCG_SCHED_NT_LOADS : 0) | ((h->sell && h->sell_window > 0) ? PRCG_SCHED_SELL_WINDOW : 0) |
           ((h->win ? h->win_vd : h->vd_int) ? PRCG_SCHED_VALDICT : 0) |
           (h->win ? (h->win_pat ? PRCG_SCHED_PATTERN : (h->win_geom < 2 ? PRCG_SCHED_COL8 : PRCG_SCHED_COL16)) | PRCG_SCHED_WINDOW
                   : (h->c8_int ? PRCG_SCHED_COL8 : (h->c16_int ? PRCG_SCHED_COL16 : 0)));
}

int prcg_set_iteration(prcg_t* h, int k) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session, "prcg_set_iteration: no open session");
    CHECK(h, k >= 0 && k < h->max_iter, "prcg_set_iteration: k out of range");
    h->k = k;
    if (h->peer && is_pipe(h->variant)) {
        // teacher forcing: the exchange buffers must describe the state just loaded -- rows and slot of "iteration k" again
        int rc = prcg_sync(h);
        if (rc) return rc;
        if ((rc = allreduce(h, h->t1.d(), 1, h->sc))) return rc;      // (collective: nobody is still reading what the pushes overwrite)
        launch_peer_push(h->sc, static_cast<const PeerDev*>(h->peer_dev.p), h->rs_cur, dots_at(h, k), k, h->rank == 0);
        launch_peer_collect(h->sc, static_cast<const PeerDev*>(h->peer_dev.p), k, nullptr, 0, nullptr, h->pub.d(), static_cast<unsigned*>(h->pub_err.p));
        HIPCHK(h, hipStreamSynchronize(h->sc));
        h->pend_parts = 0;
        return PRCG_OK;
    }
    if (h->gather && is_pipe(h->variant)) {
        // teacher forcing: what the deferred launch of iteration k+1 waits for and what its boundary tiles read
        // must describe the state just loaded (the exchange is collective over the session's ranks)
        int rc = prcg_sync(h);
        if (rc) return rc;
        if (h->fused_comm) launch_publish(h->sc, dots_at(h, k), h->pub.d(), (unsigned)k);
        if ((rc = exchange(h, h->fused ? h->rs_cur : (h->prec ? h->rst.d() : h->rs.d()), 2, h->sc))) return rc;
        HIPCHK(h, hipStreamSynchronize(h->sc));
        h->red_pending = false;
    }
    return PRCG_OK;
}

int prcg_get_vector(prcg_t* h, int which, double* out) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && out, "prcg_get_vector: no session or null buffer");
    int rc = prcg_sync(h);
    if (rc) return rc;
    const int64_t n = h->n;
    double* base; int stride;
    if (!locate(h, which, &base, &stride)) {
        // fused schedule: u = A s (A s~) and, in the 'pr' flavours, w = A r (A r~) are never stored --
        // recompute on request from the current SpMM input pairs
        const bool tilde = is_pipe(h->variant) && h->prec && (which == PRCG_VEC_UT || which == PRCG_VEC_WT);
        if (h->fused && (which == PRCG_VEC_W || which == PRCG_VEC_U || tilde))
            LAUNCHCHK(h, eng_spmm2(h, h->sc, 0, h->rs_cur, h->wu.d(), 3));
        if (h->fused && (which == PRCG_VEC_W || which == PRCG_VEC_U)) {
            launch_copy(h->sc, h->t1.d(), 1, h->wu.d() + (which == PRCG_VEC_U ? 1 : 0), 2, n);
            return d2h(h, out, h->t1.d(), n);
        }
        // derived tilde vectors of the Jacobi 'pr' flavours: w~ = M^-1 w, u~ = M^-1 u
        if (tilde) {
            launch_mul(h->sc, h->t1.d(), 1, h->dinv.d(), 1, h->wu.d() + (which == PRCG_VEC_UT ? 1 : 0), 2, n);
            return d2h(h, out, h->t1.d(), n);
        }
        return fail(h, PRCG_EINVAL, "prcg_get_vector: vector %d is not part of variant %d", which, h->variant);
    }
    if (stride == 1) return d2h(h, out, base, n);
    launch_copy(h->sc, h->t1.d(), 1, base, stride, n);
    return d2h(h, out, h->t1.d(), n);
}

int prcg_set_vector(prcg_t* h, int which, const double* in) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && in, "prcg_set_vector: no session or null buffer");
    int rc = prcg_sync(h);
    if (rc) return rc;
    const int64_t n = h->n;
    double* base; int stride;
    if (!locate(h, which, &base, &stride)) {
        if (h->fused && (which == PRCG_VEC_W || which == PRCG_VEC_U)) return PRCG_OK;                          // derived
        if (is_pipe(h->variant) && h->prec && (which == PRCG_VEC_UT || which == PRCG_VEC_WT)) return PRCG_OK;  // derived
        return fail(h, PRCG_EINVAL, "prcg_set_vector: vector %d is not part of variant %d", which, h->variant);
    }
    if (stride == 1) return h2d(h, base, in, n);
    if ((rc = h2d(h, h->t1.d(), in, n))) return rc;
    launch_copy(h->sc, base, stride, h->t1.d(), 1, n);
    HIPCHK(h, hipStreamSynchronize(h->sc));
    return PRCG_OK;
}

int prcg_get_scalars(prcg_t* h, int k, double* out) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && out && k >= 0 && k <= h->max_iter, "prcg_get_scalars: bad argument");
    int rc = prcg_sync(h);
    if (rc) return rc;
    return d2h(h, out, dots_at(h, k), kNS);
}

int prcg_set_scalars(prcg_t* h, int k, const double* in) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && in && k >= 0 && k <= h->max_iter, "prcg_set_scalars: bad argument");
    int rc = prcg_sync(h);
    if (rc) return rc;
    if ((rc = h2d(h, dots_at(h, k), in, kNS))) return rc;
    if (h->peer) return PRCG_OK;      // (prcg_set_iteration re-sends rows and slot once the whole state is loaded)
    if (h->fused_comm) {      // the deferred launch of iteration k+1 reads the published copy
        launch_publish(h->sc, dots_at(h, k), h->pub.d(), (unsigned)k);
        HIPCHK(h, hipStreamSynchronize(h->sc));
    }
    return PRCG_OK;
}

int prcg_get_coefficients(prcg_t* h, int k, double* out) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && out && k >= 1 && k <= h->max_iter, "prcg_get_coefficients: bad argument");
    int rc = prcg_sync(h);
    if (rc) return rc;
    return d2h(h, out, coef_at(h, k), 3);
}

int prcg_get_history(prcg_t* h, double* hist) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, h->in_session && hist, "prcg_get_history: no session or null buffer");
    int rc = prcg_sync(h);
    if (rc) return rc;
    const int m = h->max_iter;
    std::vector<double> all((size_t)(m + 1) * kNS);
    if ((rc = d2h(h, all.data(), h->dots.d(), (int64_t)(m + 1) * kNS))) return rc;
    static const int slot_of_bit[4] = {PRCG_S_RR, PRCG_S_RES2, PRCG_S_ERRA2, PRCG_S_ERR2};
    int row = 0;
    for (int bit = 0; bit < 4; ++bit) {
        if (!(h->hist_mask & (1u << bit))) continue;
        double* dst = hist + (size_t)row * m;
        for (int k = 0; k < m; ++k)
            dst[k] = k <= h->k ? std::sqrt(all[(size_t)k * kNS + slot_of_bit[bit]]) : 0.0;
        ++row;
    }
    return PRCG_OK;
}

int prcg_set_profiling(prcg_t* h, int stride) {
    if (!h) return PRCG_EINVAL;
    CHECK(h, stride >= 0, "prcg_set_profiling: negative stride");
    h->prof_stride = stride;
    h->n_ev_spmv = h->n_ev_upd = 0;
    return PRCG_OK;
}

int prcg_get_timings(prcg_t* h, prcg_timings* t) {
    if (!h || !t) return PRCG_EINVAL;
    int rc = prcg_sync(h);
    if (rc) return rc;
    memset(t, 0, sizeof *t);
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < h->n_ev_spmv; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_spmv[i].a, h->ev_spmv[i].b) == hipSuccess) s1 += ms;
    }
    for (int i = 0; i < h->n_ev_upd; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_upd[i].a, h->ev_upd[i].b) == hipSuccess) s2 += ms;
    }
    t->spmv_samples = h->n_ev_spmv;
    t->spmv_ms = h->n_ev_spmv ? s1 / h->n_ev_spmv : 0.0;
    t->update_ms = h->n_ev_upd ? s2 / h->n_ev_upd : 0.0;
    t->tot_ms = h->last_tot_ms;
    t->iterations = h->last_iters;
    t->iter_ms = h->last_iters ? h->last_tot_ms / h->last_iters : 0.0;
    return PRCG_OK;
}

int prcg_solve(prcg_t* h, int variant, const double* b, const double* x0, int max_iter, const double* x_true,
               const double* inv_diag, uint32_t hist_mask, double* hist, double* x_out, prcg_timings* t) {
    if (!h) return PRCG_EINVAL;
    int rc = prcg_solve_begin(h, variant, b, x0, max_iter, x_true, inv_diag, hist_mask);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = prcg_iterate(h, max_iter - 1))) return rc;
    if ((rc = prcg_sync(h))) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    h->last_tot_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    h->last_iters = max_iter - 1;
    if (hist && hist_mask && (rc = prcg_get_history(h, hist))) return rc;
    if (x_out && (rc = prcg_get_vector(h, PRCG_VEC_X, x_out))) return rc;
    if (t) rc = prcg_get_timings(h, t);
    return rc;
}

int64_t prcg_plan_tiles(int64_t n, const int32_t* indptr, const uint8_t* row_class, int cap_nnz, int cap_rows,
                        int32_t* tiles_out, int64_t capacity, int64_t* n_class0) {
    if (n < 0 || !indptr || cap_nnz < 1 || cap_rows < 1 || (!tiles_out && capacity > 0)) return -1;
    std::vector<Tile> t0, t1;
    plan_tiles(n, indptr, row_class, cap_nnz, cap_rows, t0, t1);
    const int64_t total = (int64_t)t0.size() + (int64_t)t1.size();
    if (n_class0) *n_class0 = (int64_t)t0.size();
    if (total > capacity) return -total;   // tell the caller how much room is needed
    int64_t o = 0;
    for (const auto& t : t0) { tiles_out[2 * o] = t.row_begin; tiles_out[2 * o + 1] = t.row_end; ++o; }
    for (const auto& t : t1) { tiles_out[2 * o] = t.row_begin; tiles_out[2 * o + 1] = t.row_end; ++o; }
    return total;
}

int64_t prcg_plan_window(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, const uint8_t* row_class,
                         int rows_per_tile, int32_t* tiles_out, int64_t capacity, uint16_t* cw_out, int64_t* n_class0,
                         int* most_pages) {
    if (n < 0 || n_cols < n || !indptr || (indptr[n] > 0 && !indices) || (rows_per_tile != 64 && rows_per_tile != 128) ||
        (!tiles_out && capacity > 0))
        return -1;
    WinPlan wp;
    plan_window_tiles(n, n_cols, indptr, indices, row_class, rows_per_tile, kWinCapNnz, win_max_pages(rows_per_tile), wp);
    if (!wp.ok0 || !wp.ok1) return 0;                       // not a window operator
    const int64_t total = (int64_t)wp.t0.size() + (int64_t)wp.t1.size();
    if (n_class0) *n_class0 = (int64_t)wp.t0.size();
    if (most_pages) *most_pages = wp.pages0 > wp.pages1 ? wp.pages0 : wp.pages1;
    if (total > capacity) return -total;
    int64_t o = 0;
    for (const auto* v : {&wp.t0, &wp.t1})
        for (const auto& t : *v) { memcpy(tiles_out + 20 * o, &t, 20 * sizeof(int32_t)); ++o; }
    if (cw_out) memcpy(cw_out, wp.cw.data(), (size_t)indptr[n] * sizeof(uint16_t));
    return total;
}

int64_t prcg_plan_window_patterns(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, const double* data,
                                  const uint8_t* row_class, int32_t* tiles_out, int64_t tile_capacity, void* pat_out,
                                  int64_t pat_capacity, uint16_t* masks_out, int64_t mask_capacity, int64_t* counts_out) {
    if (n < 0 || n_cols < n || !indptr || (indptr[n] > 0 && (!indices || !data)) || !counts_out) return -1;
    WinPlan wp;
    plan_window_tiles(n, n_cols, indptr, indices, row_class, 64, kWinCapNnz, kWinPatPages, wp);
    if (!wp.ok0 || !wp.ok1) return 0;
    std::vector<WTile> all(wp.t0);
    all.insert(all.end(), wp.t1.begin(), wp.t1.end());
    std::vector<PatRec> pats;
    std::vector<uint16_t> masks;
    if (!plan_window_patterns(all, indptr, wp.cw.data(), data, pats, masks)) return 0;
    counts_out[0] = (int64_t)all.size(); counts_out[1] = (int64_t)pats.size(); counts_out[2] = (int64_t)masks.size();
    if ((int64_t)all.size() > tile_capacity || (int64_t)pats.size() > pat_capacity || (int64_t)masks.size() > mask_capacity ||
        !tiles_out || !pat_out || !masks_out)
        return -(int64_t)all.size();
    static_assert(sizeof(WTile) == 24 * sizeof(int32_t), "24 int32 per tile");
    memcpy(tiles_out, all.data(), all.size() * sizeof(WTile));
    memcpy(pat_out, pats.data(), pats.size() * sizeof(PatRec));
    memcpy(masks_out, masks.data(), masks.size() * sizeof(uint16_t));
    return 1;
}

int64_t prcg_plan_sweep(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, int max_waves,
                        int32_t* tiles_out, int64_t tile_capacity, void* pat_out, int64_t pat_capacity, uint16_t* masks_out,
                        int64_t mask_capacity, int64_t* counts_out) {
    if (n < 0 || !indptr || (indptr[n] > 0 && (!indices || !data)) || !counts_out) return -1;
    SweepPlan sw;
    std::vector<PatRec> pats;
    std::vector<uint16_t> masks;
    if (!plan_sweep_tiles(n, n, indptr, indices, kWinPatPages, max_waves, sw)) return 0;
    if (!plan_window_patterns(sw.tiles, indptr, sw.cw.data(), data, pats, masks)) return 0;
    counts_out[0] = (int64_t)sw.tiles.size(); counts_out[1] = (int64_t)pats.size(); counts_out[2] = (int64_t)masks.size();
    counts_out[3] = sw.waves; counts_out[4] = sw.plane; counts_out[5] = sw.rows_per_tile;
    if ((int64_t)sw.tiles.size() > tile_capacity || (int64_t)pats.size() > pat_capacity || (int64_t)masks.size() > mask_capacity ||
        !tiles_out || !pat_out || !masks_out)
        return -(int64_t)sw.tiles.size();
    memcpy(tiles_out, sw.tiles.data(), sw.tiles.size() * sizeof(WTile));
    memcpy(pat_out, pats.data(), pats.size() * sizeof(PatRec));
    memcpy(masks_out, masks.data(), masks.size() * sizeof(uint16_t));
    return 1;
}

int prcg_plan_window_images(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, const uint8_t* row_class,
                            int rows_per_tile, int share, int64_t* out) {
    if (n < 0 || n_cols < n || !indptr || (indptr[n] > 0 && !indices) || (rows_per_tile != 64 && rows_per_tile != 128) || !out)
        return -1;
    WinPlan wp;
    plan_window_tiles(n, n_cols, indptr, indices, row_class, rows_per_tile, kWinCapNnz, win_max_pages(rows_per_tile), wp);
    if (!wp.ok0 || !wp.ok1) return 0;
    std::vector<WTile> all(wp.t0);
    all.insert(all.end(), wp.t1.begin(), wp.t1.end());
    std::vector<uint16_t> cstore, rstore;
    std::vector<uint8_t> vstore;
    const StreamStats st = share_window_streams<uint16_t>(all, indptr, wp.cw.data(), nullptr, share != 0, cstore, vstore, rstore);
    // what every tile will read must be what its own image holds
    bool same = true;
    for (const auto& t : all) {
        const int pad = t.lo & 15;
        same = same && (t.src_c & 15) == 0 && (size_t)t.src_c + pad + (t.hi - t.lo) <= cstore.size() &&
               memcmp(cstore.data() + t.src_c + pad, wp.cw.data() + t.lo, (size_t)(t.hi - t.lo) * sizeof(uint16_t)) == 0;
        for (int r = t.rb; r <= t.re && same; ++r) same = rstore[(size_t)t.src_r + (r - t.rb)] == (uint16_t)(indptr[r] - t.lo);
    }
    out[0] = (int64_t)all.size(); out[1] = st.cw_images; out[2] = st.rel_images;
    out[3] = (int64_t)cstore.size(); out[4] = (int64_t)rstore.size(); out[5] = same ? 1 : 0;
    return 1;
}

int64_t prcg_plan_sell(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, const uint8_t* row_class,
                       double max_overhead, int sigma, int planes, int allow_runs, int window_granules, int32_t* slices_out, int64_t capacity,
                       double* val_out, uint16_t* col_out, int64_t array_capacity, int32_t* rows_out, int64_t rows_capacity,
                       int32_t* gran_out, int64_t gran_capacity, int64_t* stats) {
    if (n < 0 || !indptr || (indptr[n] > 0 && (!indices || !data)) || (!slices_out && capacity > 0)) return -1;
    SellPlan sp;
    SellOptions so;
    so.max_overhead = max_overhead; so.sigma = sigma; so.planes = planes; so.allow_runs = allow_runs != 0;
    so.window_granules = window_granules;
    if (!plan_sell(n, indptr, indices, data, row_class, so, sp)) return 0;
    const int64_t total = (int64_t)sp.s0.size() + (int64_t)sp.s1.size();
    if (stats) {
        stats[0] = (int64_t)sp.s0.size(); stats[1] = (int64_t)sp.val.size(); stats[2] = (int64_t)sp.col.size(); stats[3] = sp.padded_nnz;
        stats[4] = sp.sigma; stats[5] = sp.stride_rows; stats[6] = sp.planes; stats[7] = (int64_t)sp.rows.size();
        stats[8] = sp.col_entries; stats[9] = sp.run; stats[10] = sp.window; stats[11] = (int64_t)sp.gran.size();
    }
    if (total > capacity || (val_out && (int64_t)sp.val.size() > array_capacity) || (col_out && (int64_t)sp.col.size() > array_capacity) ||
        (rows_out && (int64_t)sp.rows.size() > rows_capacity) || (gran_out && (int64_t)sp.gran.size() > gran_capacity)) return -total;
    int64_t o = 0;
    for (const auto* v : {&sp.s0, &sp.s1})
        for (const auto& t : *v) { memcpy(slices_out + 8 * o, &t, 8 * sizeof(int32_t)); ++o; }
    if (val_out) memcpy(val_out, sp.val.data(), sp.val.size() * sizeof(double));
    if (col_out) memcpy(col_out, sp.col.data(), sp.col.size() * sizeof(uint16_t));
    if (rows_out && !sp.rows.empty()) memcpy(rows_out, sp.rows.data(), sp.rows.size() * sizeof(int32_t));
    if (gran_out && !sp.gran.empty()) memcpy(gran_out, sp.gran.data(), sp.gran.size() * sizeof(int32_t));
    return total;
}

int prcg_plan_gather(int rank, int doubles_per_table, const double* tables, int n_peers, const int32_t* peer_rank,
                     const int64_t* recv_ptr, int64_t slot_doubles, int32_t* ghost_src) {
    if (rank < 0 || doubles_per_table < 1 || !tables || n_peers < 0 || slot_doubles < 8 || (slot_doubles & 1)) return -1;
    if (n_peers > 0 && (!peer_rank || !recv_ptr || !ghost_src)) return -1;
    return plan_gather_sources(rank, doubles_per_table, tables, n_peers, peer_rank, recv_ptr, slot_doubles, ghost_src);
}

int64_t prcg_debug_layout(const prcg_t* h, int64_t* out, int64_t capacity) {
    if (!h || !h->have_csr || !out) return -1;
    const int64_t nt = h->win ? (int64_t)h->nwt_int + h->nwt_bnd : (int64_t)h->nt_int + h->nt_bnd;
    const int64_t need = 8 + 2 * nt;
    if (capacity < need) return -need;
    out[0] = h->win ? 1 : 0;
    out[1] = h->win ? h->win_geom : -1;
    out[2] = h->win ? h->win_rows : 0;
    out[3] = nt;
    out[4] = h->last_grid;                                                        // workgroups of the last one-launch iteration
    out[5] = h->win ? win_fused_waves_per_block(h->win_geom, h->win_vd, h->fused_comm, h->nwt_int + h->nwt_bnd, h->want_big, h->sweep_waves) : 4;
    out[6] = h->win ? h->nwt_int : h->nt_int;
    out[7] = h->win ? (int64_t)h->win_order | ((int64_t)h->sweep_waves << 8) : 0;    // bit 0: XCD-chunked tile order; >> 8: waves of a sweep table
    std::vector<int32_t> rows((size_t)nt * 2);
    if (h->win) {
        std::vector<WTile> t((size_t)nt);
        if (nt && hipMemcpy(t.data(), h->wtiles.p, (size_t)nt * sizeof(WTile), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (int64_t i = 0; i < nt; ++i) { out[8 + 2 * i] = t[(size_t)i].rb; out[9 + 2 * i] = t[(size_t)i].re; }
    } else {
        std::vector<Tile> t((size_t)nt);
        if (nt && hipMemcpy(t.data(), h->tiles.p, (size_t)nt * sizeof(Tile), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (int64_t i = 0; i < nt; ++i) { out[8 + 2 * i] = t[(size_t)i].row_begin; out[9 + 2 * i] = t[(size_t)i].row_end; }
    }
    return need;
}

int prcg_window_source_ok(int64_t n_rows, int64_t n_ghost, int components, int64_t bytes_available) {
    if (n_rows < 0 || n_ghost < 0 || components < 1) return 0;
    return bytes_available >= (n_rows + n_ghost + (int64_t)kGatherPad) * components * (int64_t)sizeof(double) ? 1 : 0;
}

void prcg_tile_caps(int* cap_nnz, int* cap_rows) {
    const char* e = getenv("PRCG_TILE_STEPS");
    if (cap_nnz) *cap_nnz = tile_cap_nnz(pick_tile_steps(e ? atoi(e) : 0));
    if (cap_rows) *cap_rows = kTileCapRows;
}

}  // extern "C"
