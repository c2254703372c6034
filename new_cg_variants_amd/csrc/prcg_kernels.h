// Launch wrappers of the gfx950 kernels (definitions: prcg_kernels.hip).
// Internal to libprcg.so -- the public boundary is include/prcg.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace prcg {

// ---- tile geometry (must agree with the host planner) -----------------------------
// A tile is a run of consecutive rows handled by ONE wavefront: its nonzeros are
// streamed with 16-byte loads, the products staged in that wave's LDS slice, and each
// row reduced sequentially (left to right, as scipy's csr_matvec does) by one lane.
constexpr int kDefaultTileSteps = 2;                // 256-nnz steps per tile (1, 2 or 4)
inline int tile_cap_nnz(int steps) { return 256 * steps - 3; }   // -3: the stream starts 16-B aligned
constexpr int kTileCapRows = 256;
constexpr int kMaxGridBlocks = 2048;                // 8 blocks x 256 CUs
constexpr int kPartialStride = 8;                   // doubles per block in a partials array

#ifndef PRCG_TILE_DEFINED
#define PRCG_TILE_DEFINED
struct alignas(16) Tile { int row_begin, row_end, nnz_begin, nnz_end; };
#endif

// ---- direct peer exchange over xGMI (multi-rank one-launch schedule) ------------------------
// Every rank owns one EXCHANGE BUFFER (fine-grained device memory, mapped into every other rank's process with
// hipIpc): per parity (iteration & 1) one 64-byte SLOT per rank -- that rank's five partial inner products of the
// iteration and a 64-bit counter -- and a GHOST AREA that the neighbours' launches fill with the rows this rank needs.
// Launch k of a rank stores with system scope straight into the consumers' buffers: the rows of iteration k its tiles
// update, as soon as a tile is done.  Its COMMUNICATION WAVE (wave 0 of workgroup 0; takes no tiles) first adds the
// partial sums launch k-1 left (kernel boundary: complete, and launch k-1's row stores have arrived), sends them as
// the rank's slot of iteration k-1 into EVERY rank's buffer, waits until all R slots of its own buffer carry that
// counter, adds them in rank order -- the same bits on every rank -- and publishes the sums to the other waves, which
// meanwhile compute the part of the iteration that needs neither (products of their first tiles).  The tiles that
// read ghost rows come last and read them from the rank's own buffer.  No collective, no communication stream, one
// host call per iteration.
//   layout (doubles):  slot(par, q) at (par * R + q) * 8 : [5 sums, pad, pad, counter]
//                      ghost(par)   at 16 * R + par * 2 * (ghost_cap + 64) : ghost_cap pairs (+ 64 spare: a window page)
//   counter = epoch | (k + 1): epoch = session number << 32 (stale slots of an earlier session are smaller)
constexpr int kMaxPeerRanks = 16;
struct PeerDev {
    double* mine;                       // this rank's exchange buffer
    double* peer[kMaxPeerRanks];        // every rank's exchange buffer as mapped here (peer[rank] == mine)
    const int2* tile_send;              // per tile of the table in use: [first, end) of its entries in send_ent
    const int4* send_ent;               // {local row, destination rank, index in the destination's ghost area, 0}, by tile
    int rank, nranks, n_own, ghost_cap;
    unsigned long long epoch;
    int n_send;                         // entries in send_ent
};
__host__ __device__ inline int peer_slot_off(int nranks, int par, int q) { return (par * nranks + q) * 8; }
__host__ __device__ inline long long peer_ghost_off(int nranks, int ghost_cap, int par) {
    return 16LL * nranks + (long long)par * 2 * (ghost_cap + 64);
}
inline size_t peer_buffer_doubles(int nranks, int ghost_cap) { return (size_t)peer_ghost_off(nranks, ghost_cap, 2); }

// partial inner products of the previous one-launch iteration (see launch_pipe_fused)
// (+ the in-place operands of the Jacobi / 'p' flavours, see FusedState)
struct FusedPrev {
    const double* prev_partials; int nprev; double* dots_prev_out; double* rs; double* w; double* wt;
    // multi-rank one-launch schedule: the inner products of the previous iteration arrive WHILE this launch
    // runs (reduced across ranks on the communication stream, published in `pub`: kPubCopies records of 64
    // bytes, each 5 doubles + a 32-bit iteration counter at byte 48, all stored write-through).  Each wave first computes the products of
    // its first tiles, then waits for pub's counter to reach `want`, then applies the deferred updates.
    const double* pub; unsigned want; unsigned* err;
    unsigned* err_host;  // pinned host word a timed-out wave also sets (prcg_iterate reads it without a synchronisation), or null
    const PeerDev* px;   // non-null: direct peer exchange (above) -- the launch's last workgroup sends this rank's partial sums,
                         // its tiles send the rows the neighbours need, workgroup 0 turns the ranks' slots into `pub`
    int nt_int;       // tiles [nt_int, ntiles) touch ghost columns: they are neither computed nor even requested before
                      // the publication has arrived (the ghost rows travel with it)
    const double* dots_old;   // Hestenes-Stiefel product launch: the scalars of iteration k-1 (nu_k1 for b_k = nu_k / nu_k1)
                              // (one-launch predict-and-recompute: the reduced scalars of k-1 when nprev == 0)
    // one-launch predict-and-recompute iteration (launch_win_pr_one): the three vectors the window is formed from
    // (z = r~ or r, zs = s~ or s, the direction p) are read from the *_old arrays and written to the *_new ones
    // (other tiles still stage the old values); x -- and with Jacobi the plain r, s -- are updated in place
    struct PrOne {
        const double* z_old; const double* zs_old; const double* p_old;
        double* z_new; double* zs_new; double* p_new;
        double* x; double* r; double* s; const double* d;
        double* rt; const double* st;      // Ghysels-Vanroose with Jacobi: r~ (updated in place) and s~ of the row
        // PACKED form of the unpreconditioned iteration (kEpiPROneQ): the four values of a row as two 16-byte pairs, (z, zs) in
        // q_old / q_new and (p, x) in px_old / px_new -- window pages: two 16-byte loads per lane instead of three 8-byte ones
        // (the row's own x rides along), row results: two 16-byte stores, each a contiguous kilobyte per wave, instead of
        // four 8-byte ones; the separate arrays above are not touched
        const double* q_old; double* q_new; const double* px_old; double* px_new;
    } pr;
    // ONE launch per iteration of Chronopoulos-Gear / Ghysels-Vanroose (launch_win_cg_one): the p, s (u) update of the
    // previous iteration is deferred INTO this launch -- see the comment there.  z0, z1, z2: the three old vectors the
    // window is formed from (cg: r, w, s; gv: w, t, u), z?n their new buffers (other tiles still stage the old ones);
    // x, p (and r, s for gv) in place; d: inverse diagonal or null; coef_prev: where a, b of the previous iteration go.
    struct Lag {
        const double* z0; const double* z1; const double* z2;
        double* z0n; double* z1n; double* z2n;
        double* x; double* p; double* r; double* s; const double* d; double* coef_prev;
    } lag;
};
// State of the one-launch pipelined iteration (pipe_pr_cg.py:61-75 unpreconditioned, :169-187 Jacobi):
// the two-vector product of the SpMM input pair array `in_old` with the NEXT vector update applied row
// by row.  in = (r,s) unpreconditioned, (r~,s~) with Jacobi; in_new = the same array of the next
// iteration (other rows still gather the old one).  rs: the plain (r,s) pairs with Jacobi (updated in
// place: only the row itself reads them).  w / wt: the stored w, w~ of the 'p' flavours (in place).
struct FusedState {
    const double* in_old; double* in_new; double* xp;
    double* rs; const double* dinv; double* w; double* wt;
    const double* dots_prev; double* coef_out; double* partials;
    int meurant, recompute_w;
    FusedPrev prev;
    int stream_stores; // 1: the row results go out with streaming (nontemporal) stores: vectors far larger than the caches
    int deferred;     // 1: the launch waits in-kernel for prev.pub (communicator sessions, interior tiles)
    hipEvent_t done;  // non-null: the launch's own completion signal is this event (hipExtLaunchKernel): the
                      // communication stream waits for it, and NO marker packet sits between two launches
                      // on the compute stream (a separate hipEventRecord cost ~15 us of idle queue per iteration)
};

// epilogues fused into the single-vector SpMV
enum SpmvEpilogue {
    kEpiNone = 0,
    kEpiDotXY = 1,   // partial[0] += x_i * y_i                       (HS: mu = p.s; e'Ae)
    kEpiPR = 2,      // st = d*y (or y); partials mu=p.s, dl=r.st, gm=st.s   (pr_pcg)
    kEpiPipeFused = 3,   // two-vector only: the next pipelined vector update, fused row by row
    kEpiCG = 4,      // partials nu = r.x, eta = y.x, rr = r.r                 (cg_cg: x = r~, y = w)
    kEpiPipeFusedP = 5,  // ... 'p' flavours: w is the stored recurrence, only u = A s is used from the product
    kEpiPipeFusedJ = 6,  // ... Jacobi, 'pr' flavours: input (r~,s~); w~ = d w, u~ = d u in registers
    kEpiPipeFusedPJ = 7, // ... Jacobi, 'p' flavours
    kEpiHS = 8,      // window kernels only: the Hestenes-Stiefel product launch (launch_win_hs)
    kEpiPROne = 9,   // window kernels only: one-launch predict-and-recompute iteration (launch_win_pr_one)
    kEpiPROneJ = 10, // ... with Jacobi
    kEpiCGW = 11,    // window kernels only: Chronopoulos-Gear product launch, window formed as r - a s (launch_win_cg_w)
    kEpiCGWJ = 12,   // ... with Jacobi: d (r - a s)
    kEpiGVW = 13,    // window kernels only: Ghysels-Vanroose product launch, window formed as w - a u (launch_win_gv_w)
    kEpiGVWJ = 14,   // ... with Jacobi: d (w - a u)
    kEpiCGOne = 15,  // window kernels only: ONE launch per Chronopoulos-Gear iteration (launch_win_cg_one)
    kEpiCGOneJ = 16, // ... with Jacobi
    kEpiGVOne = 17,  // window kernels only: ONE launch per Ghysels-Vanroose iteration (unpreconditioned)
    kEpiPROneQ = 18, // window kernels only: one-launch predict-and-recompute iteration on the PACKED state, pairs (z, zs) and (p, x) (PrOne::q_old, px_old)
};
constexpr bool epi_pr_one(int e) { return e == kEpiPROne || e == kEpiPROneJ || e == kEpiPROneQ; }
constexpr bool epi_cg_w(int e) { return e == kEpiCGW || e == kEpiCGWJ; }
constexpr bool epi_gv_w(int e) { return e == kEpiGVW || e == kEpiGVWJ; }
// the launches that form their window from several old vectors and update the row's own vectors (FusedPrev::PrOne)
constexpr bool epi_rowset(int e) { return epi_pr_one(e) || epi_cg_w(e) || epi_gv_w(e); }
constexpr bool epi_lag(int e) { return e == kEpiCGOne || e == kEpiCGOneJ || e == kEpiGVOne; }
constexpr bool epi_fused(int e) { return e == kEpiPipeFused || e == kEpiPipeFusedP || e == kEpiPipeFusedJ || e == kEpiPipeFusedPJ; }
constexpr bool epi_prec(int e) { return e == kEpiPipeFusedJ || e == kEpiPipeFusedPJ; }
constexpr bool epi_recompute(int e) { return e == kEpiPipeFused || e == kEpiPipeFusedJ; }

struct CsrDev {
    const int* indptr;
    const int* col;
    const double* val;
    // Device-internal re-encoding of the column indices, built at prcg_set_csr when every
    // tile's columns span < 65536: col16[q] = col[q] - tile_base[tile of q].  Same indices,
    // 2 bytes instead of 4 on the HBM stream (12 -> 10 bytes per nonzero).  Lossless; the
    // arithmetic is untouched.  Null when not built.
    const unsigned short* col16;
    const unsigned char* col8;  // the same with 1 byte, when every tile spans < 256 columns (narrow bands)
    const int* tile_base;      // per tile, same indexing as the tile table
    // Device-internal re-encoding of the VALUES ("value dictionary", built at prcg_set_csr when
    // every streamed tile holds at most kDictMax distinct bit patterns -- stencils, constant
    // off-diagonals): vidx8[q] = index of val[q] in the tile's dictionary vdict[vd[t].x ..+vd[t].y).
    // 1 byte instead of 8 on the HBM stream; the dictionary entries ARE the original doubles, so
    // every product is bit-identical.  Null when not built.
    const unsigned char* vidx8;
    const double* vdict;
    const int2* vd;            // per tile (same indexing as the tile table): {first entry, count}
};
constexpr int kDictMax = 64;   // one dictionary entry per lane
// spare entries behind every gather-source vector: the largest tile-relative column offset (16 bit)
// added to a valid column of the tile never leaves the allocation
constexpr int kGatherPad = 65536;

// experiment knobs of the tile kernels, owned by the handle (read once in prcg_create / prcg_set_option)
struct TileKnobs { int per_cu = 0; int chunked = 0; };

// y = A x over tiles[0..ntiles).  x has ghost room; y has n_rows entries.
// partials: [grid][kPartialStride] doubles (slots 0..2 used by the epilogues) or null.
// returns the grid size used (needed to reduce the partials), <0 on launch failure.
// (A.tile_base / A.vd are indexed like `tiles`: the caller offsets all three together; the
//  narrow encodings are non-null in A only if every tile of the range qualified)
int launch_spmv(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps,
                const double* x, double* y, SpmvEpilogue epi,
                const double* ep_r, const double* ep_d, double* ep_st,
                double* partials, TileKnobs kn = TileKnobs{});

// [w u] = A [r s] on interleaved pairs.  write_mask: 1 = first, 2 = second, 3 = both.
int launch_spmm2(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps,
                 const double* rs, double* wu, int write_mask, TileKnobs kn = TileKnobs{});

// One launch per iteration of pipe_pr_cg / pipe_pr_m_cg on ONE GPU: [w u] = A [r s] with the
// vector update of the following iteration applied to each row as soon as its (w_i,u_i)
// exist.  w,u never reach memory; (r,s) are double-buffered (gathered from rs_old, written
// to rs_new); the 4 inner products ride along (partials[grid][0..4]).  Returns the grid.
// `prev`: if prev.nprev > 0 the inner products of the previous iteration are still block
// partials (prev.prev_partials[nprev][kPartialStride]); every block sums them itself in the
// fixed order and block 0 stores the result to prev.dots_prev_out -- no reduction launch
// between iterations.  Otherwise the reduced values are read from dots_prev.
int launch_pipe_fused(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps,
                      const FusedState& f, TileKnobs kn = TileKnobs{});

// ---- window tiles: row-per-lane kernels for bands and stencils (prcg_win.hip) ---------------
// Planned on the host by plan_window_tiles (prcg_plan.cpp).  The CSR arrays are the caller's;
// device-internal, lossless re-encodings: cw8 / cw16[q] = index of col[q] in the tile's staged
// window of the input vector (page * 64 + offset); vidx8 / vdict = per-tile value dictionary
// (<= kWinDictMax distinct bit patterns per tile; entries ARE the caller's doubles).
constexpr int kWinSlots = 1024;                 // nonzeros of one window tile that fit the wave's LDS slice
constexpr int kWinCapNnz = kWinSlots - 15;      // the stream starts at a multiple of 16 nonzeros
constexpr int kWinDictMax = 256;
#ifndef PRCG_WTILE_DEFINED
#define PRCG_WTILE_DEFINED
constexpr int kWinMaxPages = 12;
struct alignas(16) WTile {
    int rb, re, lo, hi;
    int geo, maxlen, vd_first, vd_count;
    int page_col[kWinMaxPages];
    int src_c, src_v, src_r, spare;
};
#endif
static_assert(sizeof(WTile) == 96, "the kernels read a window tile descriptor as six int4");
#ifndef PRCG_PATREC_DEFINED
#define PRCG_PATREC_DEFINED
// pattern tiles (prcg_plan.h: plan_window_patterns): constant-coefficient stencils without index streams
constexpr int kPatSlots = 16;
constexpr int kPatValues = 4;
struct alignas(8) PatRec {
    int nslots;
    unsigned vsel;
    short cb[kPatSlots];
    double val[kPatValues];
};
static_assert(sizeof(PatRec) == 72, "the kernels read a pattern record with scalar loads");
#endif
struct WinDev {
    const int* indptr;
    const double* val;
    const unsigned char* cw8;      // geometry 0, 1
    const unsigned short* cw16;    // geometry 2, 3
    const unsigned char* vidx8;    // null: plain values
    const double* vdict;
    const unsigned short* rel;     // row pointers relative to the tile's first nonzero (tile.src_r); geometry 5: the rows' slot masks
    const PatRec* pat;             // geometry 5: the pattern records (tile.src_c = pattern id)
    int sweep_waves, sweep_tiles;  // geometry 5, sweep table (prcg_plan.h: plan_sweep_tiles): the waves / tiles the carry bits assume (0: none)
    int big_ok;                    // short launches of the one-launch iteration may take big workgroups (PRCG_WIN_BIG=0: never)
    int order;                     // 1: XCD-chunked tile order (each XCD sweeps one contiguous eighth of the table), 0: chip-wide front
    int period;                    // > 1: tiles t and t + period read the same stream images (host: the launch picks a wave
                                   // count that is a multiple of it, so that a wave meets the same image tile after tile)
};
// geometry id of a class planned with rows_per_tile (64 | 128) whose tiles need at most most_pages
// pages: 0 = 64 rows / 2 pages / 8-bit indices, 1 = 64 / 4 / 8-bit, 2 = 128 / 8 / 16-bit, 3 = 128 / 12 / 16-bit,
// 4 = 64 / 8 / 16-bit (3-D stencils in 64-row tiles: images repeat with the period of a grid plane)
// 5 = 64 rows / 6 pages / PATTERN tiles (constant-coefficient stencils, no index streams; chosen by prcg_set_csr when every
//     tile qualifies, prcg_plan.h: plan_window_patterns)
constexpr int kWinPatGeom = 5, kWinPatPages = 6;
inline int win_geometry(int rows_per_tile, int most_pages) {
    if (rows_per_tile == 64) return most_pages <= 2 ? 0 : (most_pages <= 4 ? 1 : (most_pages <= 8 ? 4 : -1));
    if (rows_per_tile == 128) return most_pages <= 8 ? 2 : (most_pages <= 12 ? 3 : -1);
    return -1;
}
inline int win_max_pages(int rows_per_tile) { return rows_per_tile == 64 ? 8 : 12; }
// same contracts as launch_spmv / launch_spmm2 / launch_pipe_fused below; per_cu > 0 overrides the
// number of workgroups launched per CU (experiments)
int launch_win_spmv(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* x, double* y,
                    SpmvEpilogue epi, const double* ep_r, const double* ep_d, double* ep_st, double* partials, int per_cu);
int launch_win_spmm2(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* rs, double* wu,
                     int write_mask, int per_cu);
int launch_win_pipe_fused(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedState& f,
                          int per_cu);
// waves per workgroup of that launch (the inner products' summation order depends on it: prcg_debug_layout)
int win_fused_waves_per_block(int geom, bool value_dict, bool deferred, int ntiles, bool big_ok, int sweep_waves);
// Second of the TWO launches of a Hestenes-Stiefel iteration on a window operator (hs_cg.py:57-61,
// hs_pcg :120-124).  The first (launch_hs_update_xr with `prev`) left nu_k = <r~,r> as block partials;
// every workgroup of this launch sums them in the same fixed order, b_k = nu_k / nu_k1, and the window of
// the input vector is FORMED while it is staged: p = z + b_k p_old (z = r, or r~ with Jacobi; mul then add,
// two roundings, as the reference's `r_k + b_k * p_k1`), so p_k is never gathered from memory.  Then
// s = A p with mu = <p,s> as block partials (partials[grid][0]); the row's own p goes to p_new (other
// tiles still stage p_old: double-buffered).  hs.prev_partials / nprev: the update launch's partials
// (slots 3, 4); hs.dots_old: scalars of iteration k-1; hs.dots_prev_out: scalars of iteration k (nu, rr
// written by workgroup 0); coef_out[1] = b_k.  nprev == 0: nu_k is read from dots_prev_out instead.
int launch_win_hs(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* z,
                  const double* p_old, double* p_new, double* s, double* partials, double* coef_out,
                  const FusedPrev& hs, int per_cu);
// ONE launch per iteration of the non-pipelined predict-and-recompute variants (pr_cg.py:146-158; pr_pcg, m_pcg) on
// a window operator.  These variants have ONE reduction per iteration, and everything the product needs from the
// update is a linear combination of old vectors with coefficients known at the head of the launch (a, b from the
// previous launch's partials -- predicted nu, pr_cg.py:149-150): the staged window of the new direction is FORMED
// while it is parked, p = (z - a zs) + b p_old (the reference's `rt_k1 - a_k1 * st_k1` then `rt_k + b_k * p_k1`:
// mul, sub, mul, add, four roundings), s = A p follows, and the row's own x, r, (r~), p, s, (s~) and the five
// inner-product partials mu = p.s, dl = r.s~, gm = s~.s, nu = r~.r, r.r are written by the lane that summed the row.
// f.pr: the vectors (see FusedPrev::PrOne; d null without Jacobi); f.prev_partials / nprev / dots_prev_out: as
// launch_pipe_fused; f.dots_old: the reduced scalars of iteration k-1 when nprev == 0; coef_out: a, b, predicted nu.
int launch_win_pr_one(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                      int meurant, double* partials, double* coef_out, int per_cu);
// First of the TWO launches of a Chronopoulos-Gear iteration on a window operator (cg_cg.py:59-63, cg_pcg :116-121):
// a = nu_k1 / mu_k1 (f.dots_old), the staged window is the NEW residual formed while it is parked, r - a s (times d
// with Jacobi: r~ = M^-1 r), w = A r~ follows; the lane that summed row i writes x += a p, r (to f.pr.z_new: other
// tiles still stage the old r), r~ (f.pr.zs_new, Jacobi) and w, with the partials eta = w.r~ (slot 1), nu = r.r~ (3),
// r.r (4).  The second launch is launch_cg_update_ps with `prev`: it sums those partials in its prologue.
// f.pr: z_old = r, zs_old = s, p_old = p, z_new = the other r buffer, zs_new = r~ (Jacobi), x, d (Jacobi or null).
int launch_win_cg_w(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                    double* w_out, double* partials, double* coef_out, int per_cu);
// The same for Ghysels-Vanroose (gv_cg.py:65-75, gv_pcg :152-164): the staged window is the new w formed as w - a u
// (times d: w~ = M^-1 w), t = A w~ follows; the lane that summed row i writes x += a p, r -= a s, (r~ -= a s~), w (to
// f.pr.z_new), (w~ to f.pr.zs_new), t, and the partials eta = w.r~ (1), nu = r.r~ (3), r.r (4).
// f.pr: z_old = w, zs_old = u, p_old = p, z_new = the other w buffer, zs_new = w~, x, r, s, d, rt, st.
int launch_win_gv_w(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                    double* t_out, double* partials, double* coef_out, int per_cu);
// ONE launch per iteration of Chronopoulos-Gear (cg_cg.py:59-68, cg_pcg :116-137) or Ghysels-Vanroose (gv_cg.py:65-81,
// unpreconditioned).  Their reduction FOLLOWS the product it depends on, so an iteration cannot close inside its own
// launch -- but its tail can move into the next one: launch k+1 sums launch k's partials (nu_k, eta_k) in its prologue,
// derives b_k, mu_k = eta_k - (b_k / a_k1) nu_k and a_k = nu_k / mu_k, and applies the p, s (u) update of iteration k
// while it forms its window, which is the NEW vector of iteration k+1 expressed in old ones:
//   cg:  s_k = w + b s_old;  window  r_k1 = r - a (w + b s_old)   [times d];   w_k1 = A r~_k1
//   gv:  u_k = t + b u_old;  window  w_k1 = w - a (t + b u_old);               t_k1 = A w_k1
// (the reference's mul / add order: `w_k + b_k * s_k1` then `r_k1 - a_k1 * s_k1`).  The lane that summed row i writes
// x, p, (r, s), the three new buffers and the partials eta (slot 1), nu (3), r.r (4).  f.nprev == 0: p, s (u) are
// up to date (first launch of a call): no deferred update, a from f.dots_old = the complete scalars of the last
// iteration; else f.dots_old = the scalars of the iteration BEFORE the one whose partials are pending and
// f.dots_prev_out / f.lag.coef_prev receive that iteration's scalars and coefficients.  coef_out[0] = a.
int launch_win_cg_one(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f, int gv,
                      double* partials, double* coef_out, int per_cu);

// ---- sliced rows: lane-per-row kernels for medium-length rows (prcg_sell.hip) -----------------
// Planned on the host by plan_sell (prcg_plan.h): slices of up to 64 rows of one class; stored position u of the row in lane
// l at (u, l) of the slice -- val in chunks of two doubles, col16 in chunks of eight 16-bit column-DELTA codes (the lane's
// running column moves by code - 16384 per position; codes 0 / 65535 are skips: no nonzero) -- so that one wave
// instruction reads "positions u.. of all 64 rows" fully coalesced.  Lossless re-layout of the caller's CSR arrays.
// Slice descriptor: 8 int32 {first (smallest) row, that + rows, offset of the slice in val, offset in col16, longest stored
// row, smallest first column, first (row, stored length) pair in rows or -1, 0}.
struct SellDev {
    const int* indptr;
    const double* val;
    const unsigned short* col16;
    const int* rows;          // slices with rows_off >= 0 (a sorting window wider than a slice): (row, length) of lane l at pair rows_off + l (row -1: none)
    int nt;                   // 1: value / code streams read with nontemporal loads (operators far larger than the Infinity Cache)
    int run;                  // 1: a column code per nonzero; 3: a code per aligned run of three consecutive columns (prcg_plan.h)
    int gb, defer;            // order of a trip's gathers / the next trip's loads, and of a slice's row stores (prcg_sell.hip)
    const int* gran;          // non-null: WINDOW codes (prcg_plan.h) -- the first column of every granule, slice after slice
    int window;               // ... and the most granules a slice has (<= 64)
};
int launch_sell_spmv(hipStream_t st, const SellDev& A, const void* slices, int nslices, const double* x, double* y, SpmvEpilogue epi,
                     const double* ep_r, const double* ep_d, double* ep_st, double* partials, int per_cu);
int launch_sell_spmm2(hipStream_t st, const SellDev& A, const void* slices, int nslices, const double* rs, double* wu, int write_mask,
                      int per_cu);
int launch_sell_pipe_fused(hipStream_t st, const SellDev& A, const void* slices, int nslices, const FusedState& f, int per_cu);

// ---- small systems: the whole pipelined solve in one launch of one workgroup -------------
struct SmallArgs {
    int n, nnz;
    const int* indptr; const int* col; const double* val;
    double* xp;        // pairs (x,p), read at entry, written at exit
    double* rs;        // pairs (r,s), read at entry, written at exit
    double* dots;      // [max_iter+1][kPartialStride]: row k0 is read, rows k0+1..k0+iters written
    double* coef;      // [max_iter+1][4]
    int k0, iters, meurant;
    double* hs_p; double* hs_s;   // launch_small_hs: xp = x, rs = r, and the direction and its product
};
bool small_fits(int64_t n, int64_t nnz, int max_row_len, int* mode);
int launch_small_pipe_pr(hipStream_t st, const SmallArgs& a, int mode);
int launch_small_hs(hipStream_t st, const SmallArgs& a, int mode);      // Hestenes-Stiefel, same storage

// ---- mid-size systems: the whole pipelined solve in one launch of a few co-operating workgroups (prcg_medium.hip) ----
constexpr int kMedSlices = 4;                   // 64-row slices per wave at most
constexpr int kMedMaxGroups = 32;               // workgroups (one per CU, 16 waves each)
constexpr int kMedMaxWindow = 9728;             // (r,s) pairs of a workgroup's column window that fit its LDS (152 KB)
struct MediumArgs {
    int n, G;
    const int4* slices;                          // the sliced layout of the operator (prcg_plan.h: plan_sell)
    const double* val; const unsigned short* col16; const int* rows; const int* indptr;
    const int* wave_first;                       // [16 G + 1]: first slice of every wave
    const int2* wg_window;                       // [G]: {first column, columns} of the workgroup's window
    const int2* wg_own;                          // [G]: the workgroup's own rows [first, end) -- every row of its slices, nobody else's
    double* xp; double* rs;                      // pairs (x,p), (r,s): read at entry, written at exit
    double* exch;                                // [2][n] pairs: the exchange buffer
    double* slots;                               // [2][kMedMaxGroups][8] doubles: per workgroup four partial sums ... tag; "rows visible" tags
    double* dots; double* coef;                  // [max_iter+1][kPartialStride], [max_iter+1][4]
    int k0, iters, meurant;
    unsigned long long seq;                      // launch number: the tags of this launch are (seq << 24) + iteration
    unsigned* err;                               // set if a workgroup waited longer than its bound for the others
};
int launch_medium_pipe_pr(hipStream_t st, const MediumArgs& a, int window_pairs);

// ---- fused vector updates + inner products -----------------------------------------
struct PipeUpdateArgs {
    int64_t n;
    double* xp;       // pairs (x,p)
    double* rs;       // pairs (r,s)  -- when preconditioned: the plain pair, no ghosts
    double* rst;      // pairs (r~,s~) (preconditioned only; the SpMM input)
    double* wu;       // pairs (w,u)
    double* wt;       // w~ (preconditioned 'p' flavours only)
    const double* d;  // inverse diagonal (preconditioned only)
    const double* ut; // non-null: u~ = M^-1 u (and w~ in wt) were computed elsewhere (host-callback preconditioner)
    const double* dots_prev;   // kNumScalars doubles: mu, dl, gm, nu of iteration k-1
    double* coef_out;          // alpha, beta, nu_pred of this iteration
    double* partials;          // [grid][kPartialStride]
    double* final_out;         // non-null: the last block reduces the partials into final_out[0..5)
    unsigned* ticket;          // arrival counter of that hand-off (zero between launches)
    int meurant;      // nu prediction flavour
    int recompute_w;  // 'pr' flavours: w is overwritten by the following SpMM
};
int launch_pipe_update(hipStream_t st, const PipeUpdateArgs& a);
// the four (five) inner products of the current pipe state, no update (initialisation)
int launch_pipe_dots(hipStream_t st, const PipeUpdateArgs& a);

struct HsArgs {
    int64_t n;
    double* x; double* r; double* rt; double* p; const double* s; const double* d;
    const double* dots_prev; const double* dots_cur; double* coef_out; double* partials;
};
// prev_mu / nprev: mu = <p,s> of iteration k-1 still as block partials (slot 0) of the product launch -- every
// block sums them in the same fixed order and block 0 stores mu to dots_prev_w[0]; nprev == 0: a.dots_prev[0]
int launch_hs_update_xr(hipStream_t st, const HsArgs& a, const double* prev_mu = nullptr, int nprev = 0,
                        double* dots_prev_w = nullptr);   // x,r,(rt); partial nu (slot 3), rr (slot 4)
// prev_nu / nprev: nu_k, rr_k still as block partials (slots 3, 4) of launch_hs_update_xr; block 0 stores them
// to dots_cur_w[3], [4]
int launch_hs_update_p(hipStream_t st, const HsArgs& a, const double* prev_nu = nullptr, int nprev = 0,
                       double* dots_cur_w = nullptr);    // p = z + beta p
int launch_hs_init_dots(hipStream_t st, const HsArgs& a);   // nu, rr of the initial state

struct PrArgs {   // non-pipelined predict-and-recompute (pr_pcg / m_pcg)
    int64_t n;
    double* x; double* r; double* rt; double* p; const double* s; const double* st_;
    const double* dots_prev; double* coef_out; double* partials; int meurant; int precond;
};
int launch_pr_update(hipStream_t st, const PrArgs& a);      // x,r,rt,p; partial nu (3), rr (4)
int launch_pr_init_dots(hipStream_t st, const PrArgs& a);

struct CgArgs {   // Chronopoulos-Gear / Ghysels-Vanroose vector kernels
    int64_t n;
    double* x; double* r; double* rt; double* w; double* wt; double* p; double* s; double* st_; double* u;
    const double* t; const double* z;   // t = A w~ (gv); z = r~ or r
    const double* d;
    const double* dots_prev; const double* dots_cur; double* dots_cur_w; double* coef_out; double* partials;
};
// prev / nprev: eta, nu, r.r of this iteration still as block partials (slots 1, 3, 4) of launch_win_cg_w: every
// block sums them in the order of k_reduce_final, block 0 stores them to a.dots_cur_w
int launch_cg_update_ps(hipStream_t st, const CgArgs& a, const double* prev = nullptr, int nprev = 0);   // p,s,(s~),(u); mu by recurrence
int launch_gv_update1(hipStream_t st, const CgArgs& a, bool dots_only);   // x,r,(r~),w,(w~); partials eta(1), nu(3), rr(4)

// out[dst_first..+count) = fixed-order sum over blocks b of partials[b][src_first..+count)
void launch_reduce_final(hipStream_t st, const double* partials, int nparts, double* out,
                         int src_first, int dst_first, int count);

// ---- small utility kernels (strided so they can address one half of a pair array) ----
void launch_copy(hipStream_t st, double* dst, int ds, const double* src, int ss, int64_t n);
void launch_sub(hipStream_t st, double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n);
void launch_mul(hipStream_t st, double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n);
// partial[slot] = sum (a[i*as] - b[i])^2
int launch_diff_sq(hipStream_t st, const double* a, int as, const double* b, int64_t n, double* partials, int slot);
// partial[slot] = sum a_i * b_i
int launch_dot(hipStream_t st, const double* a, const double* b, int64_t n, double* partials, int slot);
// buf[j*nc + c] = v[idx[j]*nc + c]
void launch_pack(hipStream_t st, double* buf, const double* v, const int* idx, int64_t count, int nc);
// merged exchange for small halos: see k_gather_pack / k_gather_unpack
void launch_gather_pack(hipStream_t st, const double* partials, int nparts, double* slot, const double* rs,
                        const int* send_idx, int nsend);
void launch_gather_unpack(hipStream_t st, const double* gbuf, int slot_doubles, int nranks, double* dots_out,
                          double* rs_ghost, const int* ghost_src, int nghost, double* pub = nullptr, unsigned pub_value = 0,
                          hipEvent_t done = nullptr);
// peer exchange outside the iteration launches (session start, teacher forcing): send this rank's rows of `rs` to the
// neighbours' ghost areas of parity k & 1 and its slot for iteration k (contribute: the slot carries dots, else zeros --
// the engine lets rank 0 alone contribute the GLOBAL inner products of the state just set: the sum in rank order is
// then exactly that)
void launch_peer_push(hipStream_t st, const PeerDev* px, const double* rs, const double* dots, int k, int contribute);
// one wave: if nparts > 0 first adds this rank's block partials of iteration k (lane l: rows l, l + 64, ...; butterfly --
// as the communication wave of the next launch would) and sends the slot; then waits (bounded) until all ranks' slots
// of iteration k have arrived, adds them in rank order into dots_out[0..5) and, if pub, publishes them with counter
// k; *err = 1 on a timeout
void launch_peer_collect(hipStream_t st, const PeerDev* px, int k, const double* partials, int nparts, double* dots_out, double* pub,
                         unsigned* err);
// marks the window tiles that own rows of the send plan (bit 30 of WTile::geo)
void launch_flag_send_tiles(hipStream_t st, void* wtiles, const void* tile_send, int ntiles);
// every copy c: pub[8c .. 8c+5) = dots[0..5), then the copy's counter (byte 48 of the record) = value: what the
// deferred one-launch iteration waits for
constexpr int kPubCopies = 64;
constexpr int kPubDoubles = 8 * kPubCopies;
void launch_publish(hipStream_t st, const double* dots, double* pub, unsigned value);
// one launch of the streaming probes (prcg_stream_ceiling): mode 0 reads a[0..n_pairs) pairs; 1 / 2: a read and rewritten,
// b read, c written (plain / nontemporal stores)
void launch_stream_mix(hipStream_t st, const double* v, double* x, const double* r, double* rn, size_t n_rows, int kb);
void launch_stream_probe(hipStream_t st, int mode, double* a, double* b, double* c, size_t n_pairs);
// one wave on `st` waits (bounded, ~2 ms) for the record to reach `want`; *err = 1 if it does not
void launch_probe_wait(hipStream_t st, const double* pub, unsigned want, unsigned* err);

}  // namespace prcg
