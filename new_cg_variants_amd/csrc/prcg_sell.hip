// gfx950 (MI355X / CDNA4): "sliced rows" kernels for operators with medium-length rows that are no window
// operators -- assembled FEM matrices (3 dof x 27 neighbours = 81 nonzeros per row; SuiteSparse Queen_4147 has ~76).
//
// The CSR-adaptive tile kernels (prcg_kernels.hip) stream such a matrix in CSR order, park the products in LDS and let
// ONE LANE PER ROW add them -- with 81 nonzeros per row a 1021-nonzero tile holds 12 rows: 12 of 64 lanes work in the
// reduction phase, two dependent round trips per tile.  Here the rows keep their own lane for the whole product:
//
//   * the operator is re-laid once, on the host, in SLICES of 64 consecutive rows (prcg_plan.cpp: plan_sell): nonzero
//     u of row (rb + lane) sits at position (u, lane) of its slice -- values in chunks of two, 16-bit slice-relative
//     columns in chunks of four -- so that the wave's load of "nonzeros u..u+1 of all 64 rows" is ONE fully coalesced
//     16-byte-per-lane instruction (a device-internal, lossless re-layout of the caller's CSR arrays; rows shorter than
//     the slice's longest are padded and the padding is never multiplied);
//   * lane i walks row i left to right, 8 nonzeros per trip with the next trip's values and columns already in flight:
//     no LDS, no cross-lane step, the same sum as scipy's csr_matvec bit for bit
//     (numerical_experiments/cg_variants/pipe_pr_cg.py:69-70 calls `A @ v`);
//   * the input vector is gathered per nonzero from memory (L1 / L2: the columns of consecutive FEM rows are clustered);
//   * the row epilogues are those of the CSR-adaptive family (finish_row: store, inner-product partials, the fused
//     pipelined update), so every schedule that runs on CSR-adaptive tiles runs on slices.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <type_traits>

#include "prcg_device.hpp"
#include "prcg_kernels.h"

namespace prcg {
namespace {

typedef double d2_t __attribute__((ext_vector_type(2)));

struct SDesc { int rb, re, voff, coff, width, cbase, rows_off, flags; };

__device__ __forceinline__ SDesc read_sdesc(const int4* __restrict__ st, int t) {
    const int4 a = st[2 * t], b = st[2 * t + 1];
    SDesc d;
    d.rb = __builtin_amdgcn_readfirstlane(a.x); d.re = __builtin_amdgcn_readfirstlane(a.y);
    d.voff = __builtin_amdgcn_readfirstlane(a.z); d.coff = __builtin_amdgcn_readfirstlane(a.w);
    d.width = __builtin_amdgcn_readfirstlane(b.x); d.cbase = __builtin_amdgcn_readfirstlane(b.y);
    d.rows_off = __builtin_amdgcn_readfirstlane(b.z); d.flags = __builtin_amdgcn_readfirstlane(b.w);
    return d;
}

// One TRIP of every row of the slice: eight column codes (one 16-byte chunk per lane) and the 8 RUN stored positions they
// cover -- 4 RUN value chunks of 2 doubles per lane.  RUN = 1: a code per nonzero; RUN = 3: a code per aligned run of three
// consecutive columns (prcg_plan.h)
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int RUN> struct Trip { d2_t v[4 * RUN]; u4_t c; };

template <bool NT, int RUN>
__device__ __forceinline__ void load_trip(const SellDev& A, const SDesc& d, int u0, int lane, Trip<RUN>& T) {
    // chunk index of position u: values u / 2, codes u / (8 RUN).  Value chunks past the slice's width (the slice's last trip)
    // belong to the following slice: the lane reads its first chunk of the trip again instead (no branch around a load -- the
    // compiler's wait counts at a join assume the worst -- and no bytes from memory that nobody uses: 7 % of the stream)
    const int64_t vb = (int64_t)d.voff + ((int64_t)(u0 >> 1) * 64 + lane) * 2;
    const int64_t cb = (int64_t)d.coff + ((int64_t)(u0 / (8 * RUN)) * 64 + lane) * 8;
#pragma unroll
    for (int k = 0; k < 4 * RUN; ++k) {
        const int kk = (u0 + 2 * k < d.width) ? k : 0;                      // wave-uniform select
        const d2_t* q = reinterpret_cast<const d2_t*>(A.val + vb + (int64_t)kk * 128);
        T.v[k] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    const u4_t* q = reinterpret_cast<const u4_t*>(A.col16 + cb);
    T.c = NT ? __builtin_nontemporal_load(q) : *q;
}

// the row lane `lane` of slice d holds and its length: consecutive rows (lengths from the row pointers), or -- sorting
// window wider than a slice -- the slice's (row, length) pairs in lane order (row -1 behind the last: length 0)
__device__ __forceinline__ int2 slice_row(const SellDev& A, const SDesc& d, int lane) {
    if (d.rows_off < 0) {                                                   // wave-uniform
        const int row = d.rb + lane;
        const int r0 = row < d.re ? row : d.rb;
        const int a = A.indptr[r0], b = A.indptr[r0 + 1];
        return make_int2(row < d.re ? row : -1, row < d.re ? b - a : 0);
    }
    return reinterpret_cast<const int2*>(A.rows)[d.rows_off + lane];
}

// The wave's memory counter retires in issue order as far as a wait can tell: a wait for ANY load is a wait for everything
// issued before it.  Two orders follow from that (GB, DEFER; both measured, r04_sweeps.md B):
//   GB     0: the next trip's stream loads first, then this trip's gathers two codes at a time -- the first gather wait is a
//             wait for the whole next trip, and nothing of the stream is in flight for the wave while the trip is summed;
//             4 / 8: the gathers of the trip's first 4 / 8 codes, THEN the next trip's loads (they stay in flight);
//   DEFER  the fused iteration's row results of slice k are stored right in front of the first gathers of slice k + 1 (their
//             acknowledgement is covered by the gathers' round trip) instead of at the end of slice k, where the next
//             slice's first address computation -- a wait for everything -- found them as the youngest requests
template <int NV, int EPI, bool NT, int RUN, int GB, bool DEFER>
__global__ __launch_bounds__(kBlock) void k_sell_tiles(
    SellDev A, const int4* __restrict__ slices, int nslices,
    const void* __restrict__ xin_, void* __restrict__ yout_, int write_mask,
    const double* __restrict__ ep_r, const double* __restrict__ ep_d, double* __restrict__ ep_st,
    double* __restrict__ partials, double* __restrict__ aux, FusedPrev fz)
{
    using V = typename VecT<NV>::type;
    const V* __restrict__ X = reinterpret_cast<const V*>(xin_);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    Coefs cf = {0.0, 0.0, 0.0};
    const FusedRowPtrs fr{reinterpret_cast<double2*>(yout_), reinterpret_cast<double2*>(ep_st), reinterpret_cast<double2*>(fz.rs),
                          ep_d, fz.w, fz.wt, (write_mask & 8) != 0};
    if constexpr (epi_fused(EPI)) {
        // inner products of the previous iteration: one row of partials per workgroup of the previous launch, summed by
        // every workgroup of this one in the fixed 256-thread tree (as k_spmv_tiles / k_win_tiles do)
        if (fz.nprev > 0) {
            double dsum[5];
            sum_prev_partials<5, kWaves>(fz.prev_partials, fz.nprev, 0, dsum);
            if (blockIdx.x == 0 && threadIdx.x < 5) fz.dots_prev_out[threadIdx.x] = dsum[threadIdx.x];
            cf = predict(dsum, (write_mask >> 2) & 1);
        } else {
            cf = predict(ep_r, (write_mask >> 2) & 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
    }

    const int nblk = gridDim.x;
    const int W = nblk * kWaves;
    int t = xcd_remap(blockIdx.x, nblk) * kWaves + wv;

    Trip<RUN> cur, nxt;
    constexpr int TP = 8 * RUN;           // stored positions per trip
#if defined(PRCG_SELL_DIAG_LDSGATHER)
    __shared__ V diag_win[kWaves * 768];
    for (int i = threadIdx.x; i < kWaves * 768; i += kBlock) vzero(diag_win[i]);
    __syncthreads();
#endif
    // DEFER: the previous slice's row, sums and operands until they are stored
    int prow = -1;
    V psum; vzero(psum);
    FusedRowIn pq = {};
    V pown; vzero(pown);
    auto flush_row = [&]() {
        if constexpr (DEFER && epi_fused(EPI)) {
#if !defined(PRCG_SELL_DIAG_NOEPI)
            if (prow >= 0) fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(prow, psum, pq, pown, fr, cf, acc);
#endif
            prow = -1;
        }
    };
    SDesc d = {0, 0, 0, 0, 0, 0, -1, 0}, dn = {0, 0, 0, 0, 0, 0, -1, 0};
    int2 rl = make_int2(-1, 0);            // row and length of the lane's row in the CURRENT slice (requested a slice ahead)
    if (t < nslices) {
        d = read_sdesc(slices, t);
        load_trip<NT, RUN>(A, d, 0, lane, cur);
        rl = slice_row(A, d, lane);
        if (t + W < nslices) dn = read_sdesc(slices, t + W);
    }
    while (t < nslices) {
        const int row = rl.x;
        const bool active = row >= 0;
        const int rr = active ? row : d.rb;
        const int len = rl.y;
        // requested now, used at the end of the slice: the next slice's rows and lengths and, for the fused iteration, the
        // row's own operands (the row walk covers their latency)
        if (t + W < nslices) rl = slice_row(A, dn, lane);
        FusedRowIn q;
        V own; vzero(own);
        if constexpr (epi_fused(EPI)) {
#if defined(PRCG_SELL_DIAG_NOOWN)               // TIMING ONLY (wrong results): the row's own operands are not loaded
            q.xp = make_double2(1.0, (double)rr); own = V{};
            if constexpr (false) {
#else
            q.xp = fr.XP[rr];
            own = X[rr];
            {
#endif
            if constexpr (epi_prec(EPI)) { q.rs = fr.RS[rr]; q.d = fr.D[rr]; }
            if constexpr (!epi_recompute(EPI)) { q.w = fr.W[rr]; if constexpr (epi_prec(EPI)) q.wt = fr.WT[rr]; }
            }
        }
        V sum; vzero(sum);
        int colacc = d.cbase;
        for (int u0 = 0; u0 < d.width; u0 += TP) {                           // wave-uniform trip count
            // the next trip -- of this slice, or the first of the wave's next slice -- is requested before this one is used
            // (no branch around the loads: at a join the compiler's wait counts assume the path WITHOUT them, and the first
            //  gather wait would then cover part of the stream -- the wave's very last trip reads itself once more instead)
            const bool more = u0 + TP < d.width;
            const bool other = !more && t + W < nslices;
            auto request_next = [&]() {
                SDesc s = d;
                s.voff = other ? dn.voff : d.voff; s.coff = other ? dn.coff : d.coff; s.width = other ? dn.width : d.width;
                load_trip<NT, RUN>(A, s, more ? u0 + TP : other ? 0 : u0, lane, nxt);
            };
            if constexpr (GB == 0) request_next();
            // the lane's running column: every code moves it by code - 16384 (to the first column of its run); codes 0 and
            // 65535 only move it (skips)
            int code[8], col[8];
            code[0] = cur.c.x & 0xffffu; code[1] = cur.c.x >> 16; code[2] = cur.c.y & 0xffffu; code[3] = cur.c.y >> 16;
            code[4] = cur.c.z & 0xffffu; code[5] = cur.c.z >> 16; code[6] = cur.c.w & 0xffffu; code[7] = cur.c.w >> 16;
#pragma unroll
            for (int k = 0; k < 8; ++k) { colacc += code[k] - 16384; col[k] = colacc; }
            if (u0 == 0) flush_row();                                       // wave-uniform
            constexpr int CB = GB == 0 ? 2 : GB;                            // codes per batch of gathers
#pragma unroll
            for (int k2 = 0; k2 < 8; k2 += CB) {
                V g[CB * RUN];
#pragma unroll
                for (int k = 0; k < CB * RUN; ++k) {
                    const int c0 = col[k2 + k / RUN] + k % RUN;
#if defined(PRCG_SELL_DIAG_LDSGATHER)       // TIMING ONLY (wrong products): the gathers read an (unfilled) LDS window of the wave instead of memory
                    g[k] = diag_win[wv * 768 + ((c0 - d.cbase + 384) & 0x3ff) % 768];
#elif defined(PRCG_SELL_DIAG_NOGATHER) && PRCG_SELL_DIAG_NOGATHER == 2   // TIMING ONLY (wrong products): no gather instruction at all
                    g[k] = own; if constexpr (NV == 2) g[k].x += (double)c0; else g[k] += (double)c0;
#elif defined(PRCG_SELL_DIAG_NOGATHER)      // TIMING ONLY (wrong products): every gather hits the same 64 entries -- what the kernel costs without gather misses
                    g[k] = X[d.cbase + (c0 & 63)];
#else
                    g[k] = X[c0];                                           // (padding positions stay at the row's last column, a skip lands between
                                                                            //  two of the row's columns: valid entries, never used)
#endif
                }
                if constexpr (GB != 0) {
                    if (k2 == 0) {                                          // (the scheduler must not move the stream loads in front of the gathers)
                        __builtin_amdgcn_sched_barrier(0);
                        request_next();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int k = 0; k < CB * RUN; ++k) {
                    const int pos = k2 * RUN + k;                           // position inside the trip
                    const double a = (pos & 1) ? cur.v[pos >> 1].y : cur.v[pos >> 1].x;
                    if (u0 + pos < len && (unsigned)(code[k2 + k / RUN] - 1) < 65534u) vacc(sum, vmul(a, g[k]));   // left to right, product rounded, then added
                }
            }
            cur = nxt;
        }
        if (d.width == 0 && t + W < nslices) load_trip<NT, RUN>(A, dn, 0, lane, cur);
        if constexpr (epi_fused(EPI)) {
#if defined(PRCG_SELL_DIAG_NOEPI)               // TIMING ONLY (wrong results): nothing is stored per row
            if constexpr (NV == 2) { if (active) acc[0] += sum.x + sum.y + q.xp.x + q.xp.y + own.x + own.y; }
#else
            if constexpr (DEFER) {
                if (d.width == 0) flush_row();                              // (a slice without trips had no place for it)
                prow = active ? row : -1; psum = sum; pq = q; pown = own;
            } else {
                if (active) fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(row, sum, q, own, fr, cf, acc);
            }
#endif
        } else {
            if (active) finish_row<NV, EPI>(row, sum, yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
        }
        t += W;
        d = dn;
        if (t + W < nslices) dn = read_sdesc(slices, t + W);
    }

    flush_row();
    if constexpr (epi_fused(EPI)) { if constexpr (!epi_prec(EPI)) acc[4] = acc[3]; block_reduce_store<5>(acc, partials, 0); }
    else if constexpr (EPI == kEpiCG) block_reduce_store<5>(acc, partials, 0);
    else if constexpr (EPI != kEpiNone) {
        double a3[3] = {acc[0], acc[1], acc[2]};
        block_reduce_store<3>(a3, partials, 0);
    }
}

// ---- WINDOW codes (prcg_plan.h): the input-vector entries a slice touches are staged in LDS, the row walk reads them there ----
// Per nonzero an LDS read (its own counter, its own pipe) instead of a gather from memory: the gathers' INSTRUCTIONS, not their
// misses, cost the Queen-size stand-in 95 of 600 us (r04_sweeps.md B: every gather redirected to one line, same time; no gather
// instruction, or the gathers read from an unfilled LDS window, 509-513 against 658).  Every slice holds consecutive rows and
// names its window by granules of 16 consecutive entries; the wave's window is its own (no barrier: LDS traffic of ONE wave is
// processed in issue order).  Software pipeline per wave, all requests a whole slice ahead of their use:
//   slice t   : window pages (kSellWindowPages wave-wide loads, four granules each) sit in registers -> LDS;
//               the pages of slice t + W are requested with the granule starts `gbn` (lane g holds granule g's first column);
//               the granule starts of slice t + 2 W are requested;  then the trips, as k_sell_tiles walks them.
// Always all kSellWindowPages pages (granule index clamped: the last granule again): no branch around a load (the compiler's
// wait counts at a join assume the path without it).
// PAGES = 12 or 16 wave-wide page loads per slice (48 or 64 granules: SellPlan::window decides; the window lives in dynamic LDS,
// 4 waves x PAGES x 1 KB for pairs).

// (pages and window entries as native vectors: arrays of the HIP double2 class that live across the slice loop stay in scratch memory)
template <int NV> struct PageT;
template <> struct PageT<1> { using type = double; };
template <> struct PageT<2> { using type = d2_t; };
__device__ __forceinline__ double from_page(double w) { return w; }
__device__ __forceinline__ double2 from_page(d2_t w) { return make_double2(w.x, w.y); }

template <typename V, int PAGES>
__device__ __forceinline__ void request_pages(const V* __restrict__ X, int ng, int gb, int lane, V (&pg)[PAGES]) {
#pragma unroll
    for (int j = 0; j < PAGES; ++j) {
        int g = 4 * j + (lane >> 4);
        g = g < ng ? g : ng - 1;
        const int base = __builtin_amdgcn_ds_bpermute(g << 2, gb);
        pg[j] = X[base + (lane & 15)];
    }
}

template <int NV, int EPI, bool NT, int RUN, int PAGES>
__global__ __launch_bounds__(kBlock) void k_sell_win(
    SellDev A, const int4* __restrict__ slices, int nslices,
    const void* __restrict__ xin_, void* __restrict__ yout_, int write_mask,
    const double* __restrict__ ep_r, const double* __restrict__ ep_d, double* __restrict__ ep_st,
    double* __restrict__ partials, double* __restrict__ aux, FusedPrev fz)
{
    using V = typename VecT<NV>::type;
    const V* __restrict__ X = reinterpret_cast<const V*>(xin_);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    using PV = typename PageT<NV>::type;
    const PV* __restrict__ XPG = reinterpret_cast<const PV*>(xin_);
    extern __shared__ __align__(16) unsigned char sell_smem[];
    PV* const win = reinterpret_cast<PV*>(sell_smem) + wv * (PAGES * 64);

    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    Coefs cf = {0.0, 0.0, 0.0};
    const FusedRowPtrs fr{reinterpret_cast<double2*>(yout_), reinterpret_cast<double2*>(ep_st), reinterpret_cast<double2*>(fz.rs),
                          ep_d, fz.w, fz.wt, (write_mask & 8) != 0};
    if constexpr (epi_fused(EPI)) {
        if (fz.nprev > 0) {
            double dsum[5];
            sum_prev_partials<5, kWaves>(fz.prev_partials, fz.nprev, 0, dsum);
            if (blockIdx.x == 0 && threadIdx.x < 5) fz.dots_prev_out[threadIdx.x] = dsum[threadIdx.x];
            cf = predict(dsum, (write_mask >> 2) & 1);
        } else {
            cf = predict(ep_r, (write_mask >> 2) & 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
    }

    const int nblk = gridDim.x;
    const int W = nblk * kWaves;
    int t = xcd_remap(blockIdx.x, nblk) * kWaves + wv;
    if (t >= nslices) {                                   // (every wave joins the block's reduction)
        if constexpr (epi_fused(EPI)) { if constexpr (!epi_prec(EPI)) acc[4] = acc[3]; block_reduce_store<5>(acc, partials, 0); }
        else if constexpr (EPI == kEpiCG) block_reduce_store<5>(acc, partials, 0);
        else if constexpr (EPI != kEpiNone) { double a3[3] = {acc[0], acc[1], acc[2]}; block_reduce_store<3>(a3, partials, 0); }
        return;
    }

    constexpr int TP = 8 * RUN;
    Trip<RUN> cur, nxt;
    PV pg[PAGES];
    // granule starts of a slice, lane g <- granule g (clamped); a page load: lane l reads entry l % 16 of granule 4 j + l / 16
    auto granule_starts = [&](const SDesc& s) { const int ng = s.flags >> 8; return A.gran[s.cbase + (lane < ng ? lane : ng > 0 ? ng - 1 : 0)]; };   // (a slice of empty rows has none)
    SDesc d = read_sdesc(slices, t);
    SDesc dn = t + W < nslices ? read_sdesc(slices, t + W) : d;
    request_pages<PV, PAGES>(XPG, d.flags >> 8, granule_starts(d), lane, pg);
    int gbn = granule_starts(dn);
    load_trip<NT, RUN>(A, d, 0, lane, cur);
    int2 rl = slice_row(A, d, lane);
    while (t < nslices) {
        const int row = rl.x;
        const bool active = row >= 0;
        const int rr = active ? row : d.rb;
        const int len = rl.y;
        // this slice's window; then everything the next slices need
#pragma unroll
        for (int j = 0; j < PAGES; ++j) win[64 * j + lane] = pg[j];
        wave_lds_sync();
        request_pages<PV, PAGES>(XPG, dn.flags >> 8, gbn, lane, pg);
        const SDesc dnn = t + 2 * W < nslices ? read_sdesc(slices, t + 2 * W) : dn;
        gbn = granule_starts(dnn);
        rl = slice_row(A, dn, lane);
        FusedRowIn q;
        V own; vzero(own);
        if constexpr (epi_fused(EPI)) {
            q.xp = fr.XP[rr];
            own = X[rr];
            if constexpr (epi_prec(EPI)) { q.rs = fr.RS[rr]; q.d = fr.D[rr]; }
            if constexpr (!epi_recompute(EPI)) { q.w = fr.W[rr]; if constexpr (epi_prec(EPI)) q.wt = fr.WT[rr]; }
        }
        V sum; vzero(sum);
        for (int u0 = 0; u0 < d.width; u0 += TP) {                           // wave-uniform trip count
            const bool more = u0 + TP < d.width;
            {
                SDesc s = d;
                s.voff = more ? d.voff : dn.voff; s.coff = more ? d.coff : dn.coff; s.width = more ? d.width : dn.width;
                load_trip<NT, RUN>(A, s, more ? u0 + TP : 0, lane, nxt);      // (the wave's last slice: its own first trip once more)
            }
            int code[8];
            code[0] = cur.c.x & 0xffffu; code[1] = cur.c.x >> 16; code[2] = cur.c.y & 0xffffu; code[3] = cur.c.y >> 16;
            code[4] = cur.c.z & 0xffffu; code[5] = cur.c.z >> 16; code[6] = cur.c.w & 0xffffu; code[7] = cur.c.w >> 16;
#pragma unroll
            for (int k2 = 0; k2 < 8; k2 += 2) {
                V g[2 * RUN];
#pragma unroll
                for (int k = 0; k < 2 * RUN; ++k) g[k] = from_page(win[code[k2 + k / RUN] + k % RUN]);
#pragma unroll
                for (int k = 0; k < 2 * RUN; ++k) {
                    const int pos = k2 * RUN + k;
                    const double a = (pos & 1) ? cur.v[pos >> 1].y : cur.v[pos >> 1].x;
                    if (u0 + pos < len) vacc(sum, vmul(a, g[k]));               // left to right, product rounded, then added
                }
            }
            cur = nxt;
        }
        if (d.width == 0) load_trip<NT, RUN>(A, dn, 0, lane, cur);
        wave_lds_sync();                                                     // (the window is rewritten at the head of the next slice)
        if constexpr (epi_fused(EPI)) {
            if (active) fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(row, sum, q, own, fr, cf, acc);
        } else {
            if (active) finish_row<NV, EPI>(row, sum, yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
        }
        t += W;
        d = dn;
        dn = dnn;
    }

    if constexpr (epi_fused(EPI)) { if constexpr (!epi_prec(EPI)) acc[4] = acc[3]; block_reduce_store<5>(acc, partials, 0); }
    else if constexpr (EPI == kEpiCG) block_reduce_store<5>(acc, partials, 0);
    else if constexpr (EPI != kEpiNone) {
        double a3[3] = {acc[0], acc[1], acc[2]};
        block_reduce_store<3>(a3, partials, 0);
    }
}

// persistent grid: workgroups that are truly co-resident (the kernel uses no LDS beyond the reduction scratch: the
// occupancy API's register bound is the bound), at most `per_cu` per CU
template <typename K>
int sell_grid(K kernel, int nslices, int per_cu, size_t dyn_lds = 0) {
    // (per kernel: every instantiation has the same signature, hence the same K)
    static std::map<const void*, int> cache;
    static std::mutex mu;
    int cached = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(reinterpret_cast<const void*>(kernel));
        if (it != cache.end()) cached = it->second;
    }
    if (cached == 0) {
        int dev = 0, cus = 256, occ = 4;
        if (hipGetDevice(&dev) == hipSuccess) {
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kBlock, dyn_lds) != hipSuccess || occ < 1) occ = 2;
        }
        if (occ > 2) occ = 2;      // (s4b: 2 workgroups per CU 3955 it/s, 3: 3840, 4: 3813 -- more requests in flight cost bandwidth)
        cached = occ * 1024 + cus;
        std::lock_guard<std::mutex> lk(mu);
        cache[reinterpret_cast<const void*>(kernel)] = cached;
    }
    int occ = cached / 1024;
    const int cus = cached % 1024;
    if (per_cu >= 1 && per_cu <= 8) occ = per_cu;
    int g = (nslices + kWaves - 1) / kWaves;
    if (g > occ * cus) {
        // every wave the same number of slices: the fewest rounds the resident waves need, then the fewest waves for
        // those rounds (a last round that only a third of the waves take costs the launch 2 % at 30 rounds)
        const int waves = occ * cus * kWaves;
        const int rounds = (nslices + waves - 1) / waves;
        g = ((nslices + rounds - 1) / rounds + kWaves - 1) / kWaves;
    }
    if (g < 1) g = 1;
    return g;
}

template <int NV, int EPI>
int launch_sell(hipStream_t st, const SellDev& A, const void* slices, int nslices, const void* x, void* y, int write_mask,
                const double* ep_r, const double* ep_d, double* ep_st, double* partials, double* aux, FusedPrev fz, int per_cu)
{
    if (A.gran) {                                       // WINDOW codes
        auto pickw = [&](auto pages) {
            constexpr int P = decltype(pages)::value;
            return A.run == 3 ? (A.nt ? k_sell_win<NV, EPI, true, 3, P> : k_sell_win<NV, EPI, false, 3, P>)
                              : (A.nt ? k_sell_win<NV, EPI, true, 1, P> : k_sell_win<NV, EPI, false, 1, P>);
        };
        const int pages = A.window <= 48 ? 12 : 16;
        auto kw = pages == 12 ? pickw(std::integral_constant<int, 12>{}) : pickw(std::integral_constant<int, 16>{});
        const size_t lds = (size_t)kWaves * pages * 64 * sizeof(typename VecT<NV>::type);
        {
            static std::map<const void*, bool> raised;
            static std::mutex mu;
            std::lock_guard<std::mutex> lk(mu);
            if (!raised[reinterpret_cast<const void*>(kw)]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                raised[reinterpret_cast<const void*>(kw)] = true;
            }
        }
        const int gridw = sell_grid(kw, nslices, per_cu, lds);
        hipLaunchKernelGGL(kw, dim3(gridw), dim3(kBlock), lds, st, A, reinterpret_cast<const int4*>(slices), nslices, x, y, write_mask, ep_r,
                           ep_d, ep_st, partials, aux, fz);
        return hipGetLastError() == hipSuccess ? gridw : -1;
    }
    auto pick = [&](auto gb, auto df) {
        constexpr int G = decltype(gb)::value;
        constexpr bool D = decltype(df)::value;
        return A.run == 3 ? (A.nt ? k_sell_tiles<NV, EPI, true, 3, G, D> : k_sell_tiles<NV, EPI, false, 3, G, D>)
                          : (A.nt ? k_sell_tiles<NV, EPI, true, 1, G, D> : k_sell_tiles<NV, EPI, false, 1, G, D>);
    };
    using I0 = std::integral_constant<int, 0>;
    auto k = pick(I0{}, std::false_type{});
    if constexpr (EPI == kEpiPipeFused) {                // (the experiment's orders: the unpreconditioned pipelined iteration only)
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        const int sel = (A.gb == 8 ? 2 : A.gb == 4 ? 1 : 0) * 2 + (A.defer ? 1 : 0);
        switch (sel) {
        case 1: k = pick(I0{}, std::true_type{}); break;
        case 2: k = pick(I4{}, std::false_type{}); break;
        case 3: k = pick(I4{}, std::true_type{}); break;
        case 4: k = pick(I8{}, std::false_type{}); break;
        case 5: k = pick(I8{}, std::true_type{}); break;
        default: break;
        }
    }
    const int grid = sell_grid(k, nslices, per_cu);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), 0, st, A, reinterpret_cast<const int4*>(slices), nslices, x, y, write_mask, ep_r, ep_d,
                       ep_st, partials, aux, fz);
    return hipGetLastError() == hipSuccess ? grid : -1;
}

}  // namespace

int launch_sell_spmv(hipStream_t st, const SellDev& A, const void* slices, int nslices, const double* x, double* y, SpmvEpilogue epi,
                     const double* ep_r, const double* ep_d, double* ep_st, double* partials, int per_cu)
{
    if (nslices <= 0) return 0;
    const FusedPrev none{};
    switch (epi) {
    case kEpiNone: return launch_sell<1, kEpiNone>(st, A, slices, nslices, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiDotXY: return launch_sell<1, kEpiDotXY>(st, A, slices, nslices, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiPR: return launch_sell<1, kEpiPR>(st, A, slices, nslices, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiCG: return launch_sell<1, kEpiCG>(st, A, slices, nslices, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    default: break;
    }
    return -1;
}

int launch_sell_spmm2(hipStream_t st, const SellDev& A, const void* slices, int nslices, const double* rs, double* wu, int write_mask,
                      int per_cu)
{
    if (nslices <= 0) return 0;
    return launch_sell<2, kEpiNone>(st, A, slices, nslices, rs, wu, write_mask, nullptr, nullptr, nullptr, nullptr, nullptr, FusedPrev{}, per_cu);
}

int launch_sell_pipe_fused(hipStream_t st, const SellDev& A, const void* slices, int nslices, const FusedState& f, int per_cu)
{
    if (nslices <= 0) return 0;
    FusedPrev fz = f.prev;
    fz.rs = f.rs; fz.w = f.w; fz.wt = f.wt;
    const int mask = 3 | (f.meurant ? 4 : 0) | (f.stream_stores ? 8 : 0);
    if (f.dinv) {
        if (f.recompute_w)
            return launch_sell<2, kEpiPipeFusedJ>(st, A, slices, nslices, f.in_old, f.xp, mask, f.dots_prev, f.dinv, f.in_new, f.partials,
                                                  f.coef_out, fz, per_cu);
        return launch_sell<2, kEpiPipeFusedPJ>(st, A, slices, nslices, f.in_old, f.xp, mask, f.dots_prev, f.dinv, f.in_new, f.partials,
                                               f.coef_out, fz, per_cu);
    }
    if (f.recompute_w)
        return launch_sell<2, kEpiPipeFused>(st, A, slices, nslices, f.in_old, f.xp, mask, f.dots_prev, nullptr, f.in_new, f.partials,
                                             f.coef_out, fz, per_cu);
    return launch_sell<2, kEpiPipeFusedP>(st, A, slices, nslices, f.in_old, f.xp, mask, f.dots_prev, nullptr, f.in_new, f.partials,
                                          f.coef_out, fz, per_cu);
}

}  // namespace prcg
