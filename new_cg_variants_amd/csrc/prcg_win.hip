// gfx950 (MI355X / CDNA4): row-per-lane "window" kernels for operators whose tiles touch few
// column ranges (bands, stencils).  Same products and the same left-to-right row sums as the
// CSR-adaptive tile kernels (prcg_kernels.hip) and as scipy's csr_matvec
// (numerical_experiments/cg_variants/pipe_pr_cg.py:69-70 calls `A @ v`), different data path:
//
//   * the tile's val / col stream is read from HBM with 16-byte coalesced loads ONE TILE AHEAD
//     into a register image, then parked raw in the wave's LDS slice (CSR order);
//   * the entries of the input vector(s) the tile can touch -- at most PG pages of 64 consecutive
//     columns, planned on the host (prcg_plan.cpp: plan_window_tiles) -- are read with coalesced
//     loads one tile ahead as well and parked in LDS: NO per-nonzero gather from memory;
//   * lane i then walks row i: column (1 or 2 bytes: its index into the staged window), value
//     (or 1-byte dictionary index) and vector operand all come from LDS, products are added in
//     row order without FMA; nothing is written back to LDS, no cross-lane step;
//   * the row's epilogue (store, fused vector update, inner-product partials) follows in the
//     same lane; for the one-launch pipelined iteration (x,p) of the row was requested a tile
//     ahead and (r,s) of the row is read from the staged window;
//   * the 1- and 2-byte streams (window indices, value-dictionary indices, relative row pointers)
//     are read from the tile's IMAGE, which tiles with byte-identical streams share (prcg_plan.h:
//     share_window_streams): a band or a stencil finds them in L2, not in HBM;
//   * the non-pipelined variants FORM their staged window from old vectors while it is parked
//     (Hestenes-Stiefel p = z + b p_old, predict-and-recompute p = (r~ - a s~) + b p_old,
//     Chronopoulos-Gear r - a s, Ghysels-Vanroose w - a u): the vector a product needs is never
//     written and gathered again first.
//
// Every global load of the loop is issued a whole tile before its data is needed and there is
// no dependent second round trip, which is what the latency-bound CSR-adaptive form suffered
// from on narrow streams (profiles/r01_e_*: 70 % of wave cycles in s_waitcnt).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "prcg_device.hpp"
#include "prcg_kernels.h"

namespace prcg {
namespace {

#ifndef PRCG_WAIT_SPINS
#define PRCG_WAIT_SPINS (1u << 23)   // polls (~1 us apart) before a waiting wave gives up and flags the session
#endif
#ifndef PRCG_WIN_UNROLL
#define PRCG_WIN_UNROLL 8
#endif
#ifndef PRCG_NT_LOADS
#define PRCG_NT_LOADS 0
#endif
#ifndef PRCG_NT_VALUES
#define PRCG_NT_VALUES 0          // 1: the plain value stream (8 B per nonzero, read once per launch) with nontemporal loads (A/B builds)
#endif
#ifndef PRCG_WIN_UNIFORM_ROWS
#define PRCG_WIN_UNIFORM_ROWS 1  // 0: no scalar-value / scalar-offset row walk (A/B builds)
#endif
#ifndef PRCG_WIN_ROW_CACHE
#define PRCG_WIN_ROW_CACHE 1      // 0: the dictionary kernels re-read their shared stream images for every tile (A/B builds)
#endif
constexpr int kU = PRCG_WIN_UNROLL;
constexpr int kRelayTiles = 8;     // boundary tiles the communication wave of the peer exchange takes itself (no more: it is ONE wave)

// wave-uniform view of one tile descriptor
template <int PG>
struct WDesc {
    int rb, re, lo, hi, np, own, maxlen, vdf, vdc;
    int srcc, srcv, srcr;      // where the tile's (possibly shared) stream images start: window indices, value indices, row pointers
    int img;                   // image id: equal (and non-zero) for tiles that read identical streams, table and shape
    int send;                  // peer exchange: some of the tile's rows go to neighbours (bit 30 of `geo`, set by prcg_peer_connect)
    int pc[PG];
};

template <int PG>
__device__ __forceinline__ WDesc<PG> read_desc(const int4* __restrict__ wt, int t) {
    const int4 a = wt[t * 6 + 0], b = wt[t * 6 + 1], e = wt[t * 6 + 5];
    WDesc<PG> d;
    d.rb = __builtin_amdgcn_readfirstlane(a.x); d.re = __builtin_amdgcn_readfirstlane(a.y);
    d.lo = __builtin_amdgcn_readfirstlane(a.z); d.hi = __builtin_amdgcn_readfirstlane(a.w);
    const int geo = __builtin_amdgcn_readfirstlane(b.x);
    d.np = geo & 255; d.own = (geo >> 8) & 0xffff; d.send = (geo >> 30) & 1;
    d.maxlen = __builtin_amdgcn_readfirstlane(b.y);
    d.vdf = __builtin_amdgcn_readfirstlane(b.z); d.vdc = __builtin_amdgcn_readfirstlane(b.w);
    d.srcc = __builtin_amdgcn_readfirstlane(e.x); d.srcv = __builtin_amdgcn_readfirstlane(e.y);
    d.srcr = __builtin_amdgcn_readfirstlane(e.z); d.img = __builtin_amdgcn_readfirstlane(e.w);
#pragma unroll
    for (int q = 0; q < (PG + 3) / 4; ++q) {
        const int4 p = wt[t * 6 + 2 + q];
        if (4 * q + 0 < PG) d.pc[4 * q + 0] = __builtin_amdgcn_readfirstlane(p.x);
        if (4 * q + 1 < PG) d.pc[4 * q + 1] = __builtin_amdgcn_readfirstlane(p.y);
        if (4 * q + 2 < PG) d.pc[4 * q + 2] = __builtin_amdgcn_readfirstlane(p.z);
        if (4 * q + 3 < PG) d.pc[4 * q + 3] = __builtin_amdgcn_readfirstlane(p.w);
    }
    return d;
}

// register image of one tile, as loaded
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int NV> struct RegV;
template <> struct RegV<1> { using type = double; };
template <> struct RegV<2> { using type = d2_t; };
typedef double d3_t __attribute__((ext_vector_type(3)));
template <> struct RegV<3> { using type = d3_t; };
typedef double d4_t __attribute__((ext_vector_type(4)));
template <> struct RegV<4> { using type = d4_t; };
// components per staged window entry AS LOADED: the Hestenes-Stiefel product launch loads (z, p_old) and stages
// p = z + b p_old as one double
// (one-launch predict-and-recompute: (z, zs, p_old) -> p = (z - a zs) + b p_old)
// (Chronopoulos-Gear product launch: (r, s[, d]) -> r~ = [d] (r - a s))
constexpr int win_nw(int nv, int epi) {
    return (epi == kEpiHS || epi == kEpiCGW || epi == kEpiGVW) ? 2
         : (epi == kEpiPROneQ ? 4 : ((epi_pr_one(epi) || epi == kEpiCGWJ || epi == kEpiGVWJ || epi == kEpiCGOne || epi == kEpiGVOne) ? 3 : (epi == kEpiCGOneJ ? 4 : nv)));
}
// The one-launch pr / cg / gv kernels need the RAW window triple of the lane's own row beside the formed window entry; the
// row lies inside the tile's pages, which the wave holds raw in registers when it parks them: a cross-lane read instead of
// three more 8-byte loads per row (these kernels are bound by the number of vector-memory instructions)
#ifndef PRCG_ROW_FROM_PAGES
#define PRCG_ROW_FROM_PAGES 1
#endif
constexpr bool row_from_pages(int epi, int m, int pg) { return PRCG_ROW_FROM_PAGES && (epi_pr_one(epi) || epi_lag(epi)) && m == 1 && pg <= 4; }
template <int NV, int M, int PG, int CW, bool VD>
struct WRegs {
    d2_t v[VD ? 1 : kWinSlots / 128];   // plain values: nonzeros alo + st*128 + lane*2 .. +2
    u4_t vi;                              // dictionary indices: nonzeros alo + lane*16 .. +16
    u4_t c[CW / 8];                       // window indices: CW=8 like vi; CW=16: alo + k*512 + lane*8 .. +8
    double dv[4];                          // dictionary entries lane, lane+64, ...
    typename RegV<NV>::type w[PG];         // page p: column pc[p] + lane
    int s[M], e[M];                        // row pointers of rows rb + j*64 + lane
    d2_t xp[M];                         // fused iteration: (x,p) of those rows
    d2_t rsx[M];                        // ... Jacobi: the plain (r,s) of those rows
    double dd[M], ww[M], wwt[M];        // ... Jacobi: 1/diag; 'p' flavours: the stored w, w~
    d3_t zrow[M];                       // one-launch predict-and-recompute: (z, zs, p_old) of those rows; x in xp[].x, (r,s) in rsx[]
    // one-launch Chronopoulos-Gear / Ghysels-Vanroose: (z0, z1, z2) of those rows in zrow[]; x in xp[].x, p in xp[].y, gv's (r,s) in rsx[], d in dd[]
};

// Row cache of the dictionary kernels.  Tiles whose stream images are SHARED (prcg_plan.h: share_window_streams --
// a band keeps 3 images for 156,250 tiles) hand a wave the same bytes tile after tile: the window indices and the
// value-dictionary indices of the lane's own row(s) are then kept in registers, packed, across tiles, and a tile
// whose image equals the previous one's neither requests nor parks its index streams, row pointers and dictionary
// again, nor reads an index byte from LDS in the row walk (the wave's dependent chain index byte -> operand was what
// bound these kernels, profiles/r02_sweeps.md I).  With at most two distinct values in the tile's table the two
// entries are wave-uniform scalars and the value is a select, no LDS read.  Same products, same order.
// (64-row tiles with 1-byte window indices only: the 128-row stencil geometries rarely see the same image twice in a
//  row -- S2's images repeat every 729 tiles -- and lose occupancy to the cache registers: S2 -17 %, S1 -5 % when tried)
//  the 64-row geometry with 2-byte indices -- 3-D stencils -- caches rows of up to 8 nonzeros)
constexpr int win_row_cache_len(int m, int cw, bool vd) { return (vd && PRCG_WIN_ROW_CACHE && m == 1 && cw != 32) ? (cw == 8 ? 16 : 8) : 0; }
template <int M, int RL, int CW>
struct RowCache {
    unsigned c[M][RL > 0 ? RL * (CW / 8) / 4 : 1];   // window indices of row (rb + j*64 + lane), nonzero u at byte / half-word u
    unsigned v[M][RL > 0 ? RL / 4 : 1];              // value-dictionary indices, one byte each
    int len[M];                                       // length of that row (0: lane has no row)
    int img;                                          // which image the cache holds (WTile::spare; 0: none) -- wave-uniform
    double d0, d1;                                    // the table's entries when vdc <= 2
    // "uniform rows" (wave-uniform; a band or a stencil away from its edges): all 64 lanes own a full row, nonzero u of
    // every row has the same value index and sits at window index (lane + cbase[u]) -- value and window offset of
    // nonzero u are then SCALARS: no per-lane index extraction, no value select
    bool uni;
    unsigned cbase[RL > 0 ? RL / 4 : 1];              // lane 0's packed window indices = the offsets cbase[u]
    unsigned vmask;                                   // bit u: value index of nonzero u (vdc <= 2)
};
// Pattern tiles (CW == 32; prcg_plan.h: plan_window_patterns): the wave's current pattern, wave-uniform (scalar registers),
// re-read with scalar loads when a tile names another one (a stencil has a few patterns in all: grid edges)
struct PatState {
    int id, nslots;
    unsigned vsel;
    unsigned cb[kPatSlots / 2];            // two 16-bit signed offsets per word
    double val[kPatValues];
};
__device__ __forceinline__ void load_pattern(const PatRec* __restrict__ pat, int id, PatState& ps) {
    const unsigned* w = reinterpret_cast<const unsigned*>(pat + id);
    ps.id = id;
    ps.nslots = __builtin_amdgcn_readfirstlane((int)w[0]);
    ps.vsel = __builtin_amdgcn_readfirstlane(w[1]);
#pragma unroll
    for (int k = 0; k < kPatSlots / 2; ++k) ps.cb[k] = __builtin_amdgcn_readfirstlane(w[2 + k]);
#pragma unroll
    for (int k = 0; k < kPatValues; ++k) {
        const unsigned lo = __builtin_amdgcn_readfirstlane(w[2 + kPatSlots / 2 + 2 * k]), hi = __builtin_amdgcn_readfirstlane(w[3 + kPatSlots / 2 + 2 * k]);
        ps.val[k] = __hiloint2double((int)hi, (int)lo);
    }
}
// sweep tables (prcg_plan.h: plan_sweep_tiles): d.vdf = perm (3 bits per logical page: its LDS slot) | carry << 18 | through-perm << 24
__device__ __forceinline__ int pat_slot(int vdf, int p) { return (vdf >> (3 * p)) & 7; }
template <int PG, int M, int RL, int CW>
__device__ __forceinline__ bool same_image(const RowCache<M, RL, CW>& rc, const WDesc<PG>& d) {
    if constexpr (RL == 0) return false;
    return d.img != 0 && d.img == rc.img;
}

template <int NV, int EPI, int M, int PG, int CW, bool VD>
__device__ __forceinline__ void issue_loads(const WinDev& A, const WDesc<PG>& d, int lane,
                                            const typename VecT<NV>::type* __restrict__ X, const double* __restrict__ X2,
                                            const FusedRowPtrs& fr, const FusedPrev::PrOne& pr,
                                            WRegs<win_nw(NV, EPI), M, PG, CW, VD>& R, bool skip_img = false,
                                            const typename VecT<NV>::type* G = nullptr, int n_own = 0,
                                            const FusedPrev::Lag* lg = nullptr, bool carry_ok = false, bool need_rows = true) {
    constexpr bool FUSED = epi_fused(EPI);
    const int alo = d.lo & ~15;
    // the 1- and 2-byte streams are read from the tile's IMAGE (prcg_plan.h: share_window_streams), which tiles
    // with identical structure / value indices share: len elements from a 16-aligned start, the first lo % 16 padding.
    // branch-free: a lane whose chunk lies past the image re-reads its first chunk (hot line)
    const int len = (d.lo & 15) + (d.hi - d.lo);
    const int o16 = lane * 16 < len ? lane * 16 : 0;
    if constexpr (CW == 32) {
        // pattern tile: no index streams; the rows' slot masks unless every row has every slot (d.img)
        static_assert(M == 1, "pattern tiles have 64 rows");
        if (d.img == 0) R.s[0] = A.rel[d.srcr + lane];                      // wave-uniform
    } else
    if (!skip_img) {                                                        // wave-uniform
    if constexpr (VD) {
        R.vi = *reinterpret_cast<const u4_t*>(A.vidx8 + d.srcv + o16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (64 * k < d.vdc) {                               // wave-uniform
                const int i = 64 * k + lane;
                R.dv[k] = A.vdict[d.vdf + (i < d.vdc ? i : 0)];
            }
        }
    } else {
#pragma unroll
        for (int st = 0; st < kWinSlots / 128; ++st) {
            const int q = alo + st * 128 + lane * 2;
#if PRCG_NT_VALUES
            R.v[st] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(A.val + (q < d.hi ? q : alo)));   // read once per launch
#else
            R.v[st] = *reinterpret_cast<const d2_t*>(A.val + (q < d.hi ? q : alo));
#endif
        }
    }
    if constexpr (CW == 8) {
        R.c[0] = *reinterpret_cast<const u4_t*>(A.cw8 + d.srcc + o16);
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = k * 512 + lane * 8;
            R.c[k] = *reinterpret_cast<const u4_t*>(A.cw16 + d.srcc + (q < len ? q : 0));
        }
    }
    }
#pragma unroll
    for (int p = 0; p < PG; ++p) {
#ifdef PRCG_DEBUG_SKIP_PAGES
        if (p < d.np - PRCG_DEBUG_SKIP_PAGES) {                             // TIMING EXPERIMENT ONLY (wrong products): the last pages are not loaded
#else
        if (p < d.np && !(CW == 32 && carry_ok && ((d.vdf >> (18 + p)) & 1))) {   // wave-uniform (sweep tables: a page the wave's previous tile left in LDS)
#endif
#ifdef PRCG_DEBUG_PAGE1_LANES         // TIMING EXPERIMENT (correct only for bands of half-bandwidth < PRCG_DEBUG_PAGE1_LANES / 2): the tile's last page is loaded by its first lanes only
            if (p == d.np - 1 && p > 0 && lane >= PRCG_DEBUG_PAGE1_LANES) continue;
#endif
            if constexpr (EPI == kEpiHS) { R.w[p].x = X[d.pc[p] + lane]; R.w[p].y = X2[d.pc[p] + lane]; }
            else if constexpr (EPI == kEpiPROneQ) {                         // (z, zs) and (p, x): two 16-byte loads
                const d2_t a = reinterpret_cast<const d2_t*>(pr.q_old)[d.pc[p] + lane], b = reinterpret_cast<const d2_t*>(pr.px_old)[d.pc[p] + lane];
                R.w[p].x = a.x; R.w[p].y = a.y; R.w[p].z = b.x; R.w[p].w = b.y;
            }
            else if constexpr (epi_pr_one(EPI)) {
                R.w[p].x = pr.z_old[d.pc[p] + lane]; R.w[p].y = pr.zs_old[d.pc[p] + lane]; R.w[p].z = pr.p_old[d.pc[p] + lane];
            }
            else if constexpr (epi_lag(EPI)) {
                R.w[p].x = lg->z0[d.pc[p] + lane]; R.w[p].y = lg->z1[d.pc[p] + lane]; R.w[p].z = lg->z2[d.pc[p] + lane];
                if constexpr (EPI == kEpiCGOneJ) R.w[p].w = lg->d[d.pc[p] + lane];
            }
            else if constexpr (epi_cg_w(EPI) || epi_gv_w(EPI)) {
                R.w[p].x = pr.z_old[d.pc[p] + lane]; R.w[p].y = pr.zs_old[d.pc[p] + lane];
                if constexpr (EPI == kEpiCGWJ || EPI == kEpiGVWJ) R.w[p].z = pr.d[d.pc[p] + lane];
            }
            else {
#ifdef PRCG_DEBUG_PLAIN_GHOST
                if (false) {
#else
                if (G != nullptr && d.pc[p] >= n_own) {
#endif
                    // peer exchange: a page of ghost columns lies in this rank's exchange buffer (pages never straddle n_own),
                    // written by OTHER GPUs' stores: system-scope loads, which no cache of this GPU serves -- no acquire
                    // fence (and no invalidation of anybody's cached lines) needed
                    // (ONE 16-byte request per lane, sc0 sc1: two 8-byte system-scope atomic loads per lane made the boundary tile the
                    //  slowest step of the launch)
                    if constexpr (NV == 2) {
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(G)), 0, 0x7ffffff0, 0x00020000);
                        const u4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (d.pc[p] + lane) * 16, 0, 17);
                        R.w[p].x = __hiloint2double((int)raw.y, (int)raw.x);
                        R.w[p].y = __hiloint2double((int)raw.w, (int)raw.z);
                    } else {
                        const double* gp = reinterpret_cast<const double*>(G + d.pc[p] + lane);
                        R.w[p] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                } else {
                    R.w[p] = reinterpret_cast<const typename RegV<NV>::type*>(X)[d.pc[p] + lane];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int row = d.rb + j * 64 + lane;
        const int rr = row < d.re ? row : d.rb;
        const int jj = row < d.re ? j * 64 + lane : 0;
        if (CW != 32 && !skip_img) {
            R.s[j] = A.rel[d.srcr + jj];                 // row pointers relative to the tile's first nonzero
            R.e[j] = A.rel[d.srcr + jj + 1];
        }
        if constexpr (epi_lag(EPI)) {
            if constexpr (!row_from_pages(EPI, M, PG)) { R.zrow[j].x = lg->z0[rr]; R.zrow[j].y = lg->z1[rr]; R.zrow[j].z = lg->z2[rr]; }
            R.xp[j].x = lg->x[rr]; R.xp[j].y = lg->p[rr];
            if constexpr (EPI == kEpiCGOneJ) R.dd[j] = lg->d[rr];
            if constexpr (EPI == kEpiGVOne) { R.rsx[j].x = lg->r[rr]; R.rsx[j].y = lg->s[rr]; }
        }
        if constexpr (EPI == kEpiPROneQ) {
            if constexpr (!row_from_pages(EPI, M, PG)) {
                const d2_t a = reinterpret_cast<const d2_t*>(pr.q_old)[rr], b = reinterpret_cast<const d2_t*>(pr.px_old)[rr];
                R.zrow[j].x = a.x; R.zrow[j].y = a.y; R.zrow[j].z = b.x; R.xp[j].x = b.y;
            }
        } else
        if constexpr (epi_rowset(EPI)) {
            if constexpr (!row_from_pages(EPI, M, PG)) { R.zrow[j].x = pr.z_old[rr]; R.zrow[j].y = pr.zs_old[rr]; R.zrow[j].z = pr.p_old[rr]; }
            R.xp[j].x = pr.x[rr];
            if constexpr (EPI == kEpiPROneJ) { R.rsx[j].x = pr.r[rr]; R.rsx[j].y = pr.s[rr]; R.dd[j] = pr.d[rr]; }
            if constexpr (EPI == kEpiCGWJ) R.dd[j] = pr.d[rr];
            if constexpr (epi_gv_w(EPI)) { R.rsx[j].x = pr.r[rr]; R.rsx[j].y = pr.s[rr]; }
            if constexpr (EPI == kEpiGVWJ) { R.dd[j] = pr.d[rr]; R.ww[j] = pr.rt[rr]; R.wwt[j] = pr.st[rr]; }
        }
        if (FUSED && need_rows) {                                           // (a tile whose update is deferred asks for its rows later: phase B)
#if PRCG_NT_LOADS
            R.xp[j] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(fr.XP) + rr);       // read once per launch
#else
            R.xp[j] = reinterpret_cast<const d2_t*>(fr.XP)[rr];
#endif
            if constexpr (epi_prec(EPI)) {
                R.rsx[j] = reinterpret_cast<const d2_t*>(fr.RS)[rr];
                R.dd[j] = fr.D[rr];
            }
            if constexpr (!epi_recompute(EPI)) {
                R.ww[j] = fr.W[rr];
                if constexpr (epi_prec(EPI)) R.wwt[j] = fr.WT[rr];
            }
        }
    }
}

template <int WPB, int NQ>
__device__ __forceinline__ void win_block_reduce_store(double (&acc)[NQ], double* partials) {
    __shared__ double red[WPB][NQ];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double v = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < WPB; ++w) v += red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kPartialStride + threadIdx.x] = v;
    }
}

// per-wave constants of the tile loop
template <int NV>
struct WCtx {
    double* sv; unsigned char* svi; unsigned char* sc; double* sd; typename VecT<NV>::type* sw;
    const typename VecT<NV>::type* X; const double* X2;
    void* yout; int write_mask;
    const double* ep_r; const double* ep_d; double* ep_st;
    FusedRowPtrs fr;
    FusedPrev::PrOne pr;
    int lane;
    // direct peer exchange (deferred form only, else null): the ghost columns [n_own, ...) of the input vector are read
    // from this rank's exchange buffer (G = its ghost area of the iteration read, minus n_own), the rows the neighbours
    // need go to THEIR ghost areas of the iteration written (offset gout, in doubles, into every exchange buffer)
    const typename VecT<NV>::type* G; int n_own; const PeerDev* px; long long gout;
    const FusedPrev::Lag* lg; bool lagged;      // one-launch Chronopoulos-Gear / Ghysels-Vanroose: vectors; a deferred p, s (u) update is pending
    bool carry_ok;                              // sweep tables: this launch runs the waves the carry bits assume (else every page is loaded)
};

// One tile: park its image (R, requested DEPTH tiles ago) in LDS, request tile `dnext` into the
// freed registers, then lane i walks row i (and i + 64, ...).
// STASH: only the products -- the row sums go to `stash` (this wave's LDS, [M][64] pairs), the epilogue
// follows later (deferred form of the one-launch iteration).
// Direct peer exchange: the rows of tile t that neighbours need (px->tile_send / send_ent, planned on the host) go
// straight into the destinations' ghost areas.  newp[j] = the new input pair of row rb + j*64 + lane, still in
// registers; entry e is handled by lane e % 64, which fetches the pair from the lane that owns the row.
template <int M>
__device__ __forceinline__ void peer_send_rows(const PeerDev* px, int t, int rb, const double2 (&newp)[M], long long gout, int lane) {
    const int2 ts = px->tile_send[t];
    const int first = __builtin_amdgcn_readfirstlane(ts.x), end = __builtin_amdgcn_readfirstlane(ts.y);
    for (int e0 = first; e0 < end; e0 += 64) {                              // wave-uniform
        const int e = e0 + lane;
        const bool on = e < end;
        const int4 ent = px->send_ent[on ? e : first];
        const int rel = ent.x - rb, src = rel & 63;
        double2 v = make_double2(__shfl(newp[0].x, src, 64), __shfl(newp[0].y, src, 64));
        if constexpr (M > 1) {
            const double2 v1 = make_double2(__shfl(newp[1].x, src, 64), __shfl(newp[1].y, src, 64));
            if (rel >= 64) v = v1;
        }
        if (on) {
            double* dst = px->peer[ent.y] + gout + 2 * (long long)ent.z;
            peer_store(dst, v.x);
            peer_store(dst + 1, v.y);
        }
    }
}

__device__ __forceinline__ double uniform_double(double v) {       // the same value in every lane -> scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// `same_cur`: the tile's index streams, row pointers and dictionary were neither requested nor need parking -- its image
// is the one the row cache `rc` holds (see RowCache).  Returns whether the tile requested here (dnext) is in that case.
template <int NV, int EPI, int M, int PG, int CW, bool VD, int RL, bool STASH = false>
__device__ __forceinline__ bool win_step(const WinDev& A, const WCtx<NV>& c, WRegs<win_nw(NV, EPI), M, PG, CW, VD>& R,
                                         const WDesc<PG>& dcur, bool have_next, const WDesc<PG>& dnext,
                                         double (&acc)[5], const Coefs& cf, RowCache<M, RL, CW>& rc, bool same_cur,
                                         PatState& ps, double2* stash = nullptr, bool acquire_first = false, int tcur = 0,
                                         bool next_rows = true)
{
    using V = typename VecT<NV>::type;
    using RV = typename RegV<NV>::type;
    constexpr bool FUSED = epi_fused(EPI);
    const int lane = c.lane;
    const int pad = dcur.lo & 15;          // the staged stream starts at the 16-aligned nonzero below the tile's first
    // ---- park the tile's image in LDS (this is where the wave waits for ITS loads only: the
    //      loads of the tiles requested after it stay in flight) ----
    if constexpr (CW == 32) {
        if (dcur.srcc != ps.id) load_pattern(A.pat, dcur.srcc, ps);         // wave-uniform
    } else
    if (!same_cur) {                                                        // wave-uniform
    if constexpr (VD) {
        *reinterpret_cast<u4_t*>(c.svi + lane * 16) = R.vi;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (64 * k < dcur.vdc) c.sd[64 * k + lane] = R.dv[k];
    } else {
#pragma unroll
        for (int st = 0; st < kWinSlots / 128; ++st)
            *reinterpret_cast<d2_t*>(c.sv + st * 128 + lane * 2) = R.v[st];
    }
    if constexpr (CW == 8) {
        *reinterpret_cast<u4_t*>(c.sc + lane * 16) = R.c[0];
    } else if constexpr (CW == 16) {
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<u4_t*>(c.sc + (k * 512 + lane * 8) * 2) = R.c[k];
    }
    }
#pragma unroll
    for (int p = 0; p < PG; ++p) {
        if (p < dcur.np && !(CW == 32 && c.carry_ok && ((dcur.vdf >> (18 + p)) & 1))) {
            const int sp = CW == 32 ? pat_slot(dcur.vdf, p) : p;            // (sweep tables: the page's LDS slot; else its own index)
            // Hestenes-Stiefel: the direction is formed here, p = z + b p_old (hs_cg.py:60), never gathered
            if constexpr (EPI == kEpiHS) c.sw[sp * 64 + lane] = R.w[p].x + cf.bt * R.w[p].y;
            // predict-and-recompute: r~ -= a s~, then p = r~ + b p_old (pr_cg.py:148,151), formed here
            else if constexpr (epi_pr_one(EPI)) c.sw[sp * 64 + lane] = (R.w[p].x - cf.al * R.w[p].y) + cf.bt * R.w[p].z;
            // Chronopoulos-Gear: r -= a s, r~ = M^-1 r (cg_cg.py:60, cg_pcg :117-118), formed here
            // (Ghysels-Vanroose: w -= a u, w~ = M^-1 w, gv_cg.py:67 / :155,161 -- the same form)
            // one launch per Chronopoulos-Gear / Ghysels-Vanroose iteration: the deferred s = w + b s_old (u = t + b u_old) and
            // then the new residual r - a s (new w - a u): mul, add, mul, sub as cg_cg.py:66,60 / gv_cg.py:79,67
            else if constexpr (epi_lag(EPI)) {
                const double z2n = c.lagged ? R.w[p].y + cf.bt * R.w[p].z : R.w[p].z;
                const double v = R.w[p].x - cf.al * z2n;
                if constexpr (EPI == kEpiCGOneJ) c.sw[sp * 64 + lane] = R.w[p].w * v; else c.sw[sp * 64 + lane] = v;
            }
            else if constexpr (EPI == kEpiCGW || EPI == kEpiGVW) c.sw[sp * 64 + lane] = R.w[p].x - cf.al * R.w[p].y;
            else if constexpr (EPI == kEpiCGWJ || EPI == kEpiGVWJ) c.sw[sp * 64 + lane] = R.w[p].z * (R.w[p].x - cf.al * R.w[p].y);
            else reinterpret_cast<RV*>(c.sw)[sp * 64 + lane] = R.w[p];
        }
    }
    int rs_[M], re_[M];
    FusedRowIn fin[M];
    d3_t zr[M];
    double xq = 0.0;                        // packed predict-and-recompute, rows taken from the pages: the row's x
    if constexpr (row_from_pages(EPI, M, PG)) {
        // row rb + lane sits at window index own + lane: lane (own + lane) % 64 of page (own + lane) / 64 holds its raw triple
        const int p0 = dcur.own >> 6, src = (dcur.own + lane) & 63;
        const bool second = ((dcur.own & 63) + lane) >= 64;
        d3_t a = {0.0, 0.0, 0.0}, b = {0.0, 0.0, 0.0};
#pragma unroll
        for (int p = 0; p < PG; ++p) {
            if (p == p0) { a.x = R.w[p].x; a.y = R.w[p].y; a.z = R.w[p].z; }          // wave-uniform
            if (p == p0 + 1) { b.x = R.w[p].x; b.y = R.w[p].y; b.z = R.w[p].z; }
        }
        const double ax = __shfl(a.x, src, 64), ay = __shfl(a.y, src, 64), az = __shfl(a.z, src, 64);
        const double bx = __shfl(b.x, src, 64), by = __shfl(b.y, src, 64), bz = __shfl(b.z, src, 64);
        zr[0].x = second ? bx : ax; zr[0].y = second ? by : ay; zr[0].z = second ? bz : az;
        if constexpr (EPI == kEpiPROneQ) {                                  // the row's x: the fourth component of its page entry
            double aw = 0.0, bw = 0.0;
#pragma unroll
            for (int p = 0; p < PG; ++p) {
                if (p == p0) aw = R.w[p].w;
                if (p == p0 + 1) bw = R.w[p].w;
            }
            const double sa = __shfl(aw, src, 64), sb = __shfl(bw, src, 64);
            xq = second ? sb : sa;
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        rs_[j] = R.s[j]; re_[j] = R.e[j];
        if constexpr (epi_lag(EPI)) {
            if constexpr (!row_from_pages(EPI, M, PG)) zr[j] = R.zrow[j];
            fin[j].xp = make_double2(R.xp[j].x, R.xp[j].y);
            if constexpr (EPI == kEpiCGOneJ) fin[j].d = R.dd[j];
            if constexpr (EPI == kEpiGVOne) fin[j].rs = make_double2(R.rsx[j].x, R.rsx[j].y);
        }
        if constexpr (epi_rowset(EPI)) {
            if constexpr (!row_from_pages(EPI, M, PG)) zr[j] = R.zrow[j];
            if constexpr (EPI == kEpiPROneQ && row_from_pages(EPI, M, PG)) fin[j].xp.x = xq; else fin[j].xp.x = R.xp[j].x;
            if constexpr (EPI == kEpiPROneJ) { fin[j].rs = make_double2(R.rsx[j].x, R.rsx[j].y); fin[j].d = R.dd[j]; }
            if constexpr (EPI == kEpiCGWJ) fin[j].d = R.dd[j];
            if constexpr (epi_gv_w(EPI)) fin[j].rs = make_double2(R.rsx[j].x, R.rsx[j].y);
            if constexpr (EPI == kEpiGVWJ) { fin[j].d = R.dd[j]; fin[j].w = R.ww[j]; fin[j].wt = R.wwt[j]; }
        }
        if constexpr (FUSED) {
            fin[j].xp = make_double2(R.xp[j].x, R.xp[j].y);
            if constexpr (epi_prec(EPI)) { fin[j].rs = make_double2(R.rsx[j].x, R.rsx[j].y); fin[j].d = R.dd[j]; }
            if constexpr (!epi_recompute(EPI)) { fin[j].w = R.ww[j]; if constexpr (epi_prec(EPI)) fin[j].wt = R.wwt[j]; }
        }
    }
    wave_lds_sync();

    // ---- the row cache: after this tile it holds this tile's image (if its rows fit) ----
    bool fill = false, next_same = false;
    if constexpr (RL > 0) {
        if (!same_cur) {
            fill = dcur.maxlen <= RL && dcur.img != 0;
            rc.img = fill ? dcur.img : 0;
        }
    }

    // ---- request the tile DEPTH ahead (the image registers are free again) ----
    if (have_next) {
        // (deferred form, first tile of this wave that reads ghost rows: consumer side of the hand-off)
        if (acquire_first && !c.px) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        next_same = same_image<PG>(rc, dnext);
        issue_loads<NV, EPI, M, PG, CW, VD>(A, dnext, lane, c.X, c.X2, c.fr, c.pr, R, next_same, c.G, c.n_own, c.lg, c.carry_ok, next_rows);
    }

    const int last = pad + (dcur.hi - dcur.lo) - 1 > 0 ? pad + (dcur.hi - dcur.lo) - 1 : 0;
    if constexpr (RL > 0) {
        if (fill) {
            // the lane's own row(s): index bytes out of the parked streams, once per IMAGE instead of once per tile
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const bool active = dcur.rb + j * 64 + lane < dcur.re;
                const int o = rs_[j] + pad;
                const int len = active ? re_[j] - rs_[j] : 0;
                rc.len[j] = len;
#pragma unroll
                for (int w = 0; w < RL * (CW / 8) / 4; ++w) rc.c[j][w] = 0u;
#pragma unroll
                for (int w = 0; w < RL / 4; ++w) rc.v[j][w] = 0u;
#pragma unroll
                for (int u = 0; u < RL; ++u) {
                    int idx = o + u;
                    idx = idx < last ? idx : last;
                    unsigned cb, vb = c.svi[idx];
                    if constexpr (CW == 8) cb = c.sc[idx]; else cb = reinterpret_cast<const unsigned short*>(c.sc)[idx];
                    if (u >= len) { cb = 0u; vb = 0u; }
                    if constexpr (CW == 8) rc.c[j][u >> 2] |= cb << (8 * (u & 3)); else rc.c[j][u >> 1] |= cb << (16 * (u & 1));
                    rc.v[j][u >> 2] |= vb << (8 * (u & 3));
                }
            }
            rc.d0 = uniform_double(c.sd[0]);
            rc.d1 = uniform_double(c.sd[1]);
            if constexpr (M == 1 && CW == 8) {
                bool ok = rc.len[0] == dcur.maxlen;
                unsigned vm = 0u;
#pragma unroll
                for (int w = 0; w < RL / 4; ++w) rc.cbase[w] = __builtin_amdgcn_readfirstlane(rc.c[0][w]);
#pragma unroll
                for (int u = 0; u < RL; ++u) {
                    if (u < dcur.maxlen) {                                 // wave-uniform
                        const int ci = (rc.c[0][u >> 2] >> (8 * (u & 3))) & 255, c0 = (rc.cbase[u >> 2] >> (8 * (u & 3))) & 255;
                        const int vi = (rc.v[0][u >> 2] >> (8 * (u & 3))) & 255;
                        const int v0 = __builtin_amdgcn_readfirstlane(vi);
                        ok = ok && ci == c0 + lane && vi == v0;
                        vm |= (unsigned)(v0 & 1) << u;
                    }
                }
                rc.vmask = vm;
                rc.uni = PRCG_WIN_UNIFORM_ROWS && dcur.vdc <= 2 && dcur.re - dcur.rb == 64 && __builtin_amdgcn_ballot_w64(ok) == ~0ull;
            } else {
                rc.uni = false;
            }
        }
    }
    const bool cached = RL > 0 && (same_cur || fill);                  // wave-uniform

    // ---- lane i walks row i (and i + 64, ...) ----
    double2 newp[M];
#pragma unroll
    for (int j = 0; j < M; ++j) newp[j] = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int row = dcur.rb + j * 64 + lane;
        const bool active = row < dcur.re;
        if (j > 0 && dcur.rb + j * 64 >= dcur.re) break;              // wave-uniform
        const int o = rs_[j] + pad;
        const int len = active ? re_[j] - rs_[j] : 0;
        V sum; vzero(sum);
        if constexpr (RL > 0) {
            if (cached) {
                // indices from registers; the value is a select between the table's two entries, or one LDS read
                const int clen = rc.len[j];
                if (rc.uni) {
                    // every lane: a full row, scalar value, window entry lane + scalar offset
#pragma unroll
                    for (int u = 0; u < RL; ++u) {
                        if (u < dcur.maxlen) {                             // wave-uniform
                            const int cb = (rc.cbase[u >> 2] >> (8 * (u & 3))) & 255;
                            const double a = ((rc.vmask >> u) & 1u) ? rc.d1 : rc.d0;
                            const V g = c.sw[cb + lane];
                            vacc(sum, vmul(a, g));
                        }
                    }
                } else if (dcur.vdc <= 2) {
#pragma unroll
                    for (int q4 = 0; q4 < RL; q4 += 4) {
                        if (q4 < dcur.maxlen) {                            // wave-uniform
#pragma unroll
                            for (int u = q4; u < q4 + 4; ++u) {
                                int ci;
                                if constexpr (CW == 8) ci = (rc.c[j][u >> 2] >> (8 * (u & 3))) & 255; else ci = (rc.c[j][u >> 1] >> (16 * (u & 1))) & 65535;
                                const int vi = (rc.v[j][u >> 2] >> (8 * (u & 3))) & 255;
                                const double a = vi ? rc.d1 : rc.d0;
                                const V g = c.sw[ci];
                                if (u < clen) vacc(sum, vmul(a, g));
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < RL; q4 += 4) {
                        if (q4 < dcur.maxlen) {
#pragma unroll
                            for (int u = q4; u < q4 + 4; ++u) {
                                int ci;
                                if constexpr (CW == 8) ci = (rc.c[j][u >> 2] >> (8 * (u & 3))) & 255; else ci = (rc.c[j][u >> 1] >> (16 * (u & 1))) & 65535;
                                const int vi = (rc.v[j][u >> 2] >> (8 * (u & 3))) & 255;
                                const double a = c.sd[vi];
                                const V g = c.sw[ci];
                                if (u < clen) vacc(sum, vmul(a, g));
                            }
                        }
                    }
                }
            }
        }
        if constexpr (CW == 32) {
            // pattern tile: slot u of every row that has it is val[u] * window[lane + cb[u]] -- scalar value, scalar offset, no
            // index byte; rows at a grid edge skip the slots their mask lacks (same left-to-right sum over what the row has).
            // The per-tile decisions are taken ONCE (straight-line scalar code per slot: a 15-diagonal band lost 40 % to a
            // branch per decision and slot): the slots' window offsets as the tile's LDS slots have them (sweep tables), whether
            // every row is complete, whether the pattern has two values at most (a select) or up to four
            const bool full = dcur.img != 0;                                // wave-uniform
            const unsigned mk = full ? 0xffffu : (unsigned)rs_[j];
            unsigned pcb[kPatSlots / 2];
            if ((dcur.vdf >> 24) & 1) {
#pragma unroll
                for (int w = 0; w < kPatSlots / 2; ++w) {
                    pcb[w] = 0u;
                    if (2 * w >= ps.nslots) continue;                       // wave-uniform
                    const int lo16 = (int)(short)(ps.cb[w] & 0xffffu), hi16 = (int)ps.cb[w] >> 16;
                    const int plo = pat_slot(dcur.vdf, (lo16 >> 6) & 7) * 64 + (lo16 & 63), phi = pat_slot(dcur.vdf, (hi16 >> 6) & 7) * 64 + (hi16 & 63);
                    pcb[w] = ((unsigned)plo & 0xffffu) | ((unsigned)phi << 16);
                }
            } else {
#pragma unroll
                for (int w = 0; w < kPatSlots / 2; ++w) pcb[w] = ps.cb[w];
            }
            const bool two = (ps.vsel & 0xaaaaaaaau) == 0u;                 // selectors 0 / 1 only
            auto offset_of = [&](int u) { const unsigned w = pcb[u >> 1]; return (u & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu); };
            if (full && two) {
#pragma unroll
                for (int u = 0; u < kPatSlots; ++u) {
                    if (u < ps.nslots) {                                    // wave-uniform
                        const double a = ((ps.vsel >> (2 * u)) & 1u) ? ps.val[1] : ps.val[0];
                        const V g = c.sw[offset_of(u) + lane];
                        vacc(sum, vmul(a, g));
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kPatSlots; ++u) {
                    if (u < ps.nslots) {                                    // wave-uniform
                        const unsigned sel = (ps.vsel >> (2 * u)) & 3u;
                        const double a01 = (sel & 1u) ? ps.val[1] : ps.val[0], a23 = (sel & 1u) ? ps.val[3] : ps.val[2];
                        const double a = (sel & 2u) ? a23 : a01;
                        int idx = offset_of(u) + lane;
                        idx = idx < 0 ? 0 : (idx > PG * 64 - 1 ? PG * 64 - 1 : idx);     // lanes without the slot: any valid entry
                        const V g = c.sw[idx];
                        if ((mk >> u) & 1u) vacc(sum, vmul(a, g));
                    }
                }
            }
        }
        for (int j0 = 0; j0 < ((cached || CW == 32) ? 0 : dcur.maxlen); j0 += kU) {
            int ci[kU];
            double a[kU];
            V g[kU];
            // the row's next kU window indices (and value indices) are consecutive bytes of the staged stream at an
            // arbitrary byte offset: ONE unaligned 8-byte LDS read each (two for 2-byte indices) instead of kU byte reads
            // -- the LDS pipe is this kernel's busiest unit (profiles/r02_sweeps.md).  Slots past the row's end
            // hold the following rows' bytes or stale LDS: index 0 is substituted, the product is skipped below.
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                int idx = o + j0 + u;
                idx = idx < last ? idx : last;
                if constexpr (CW == 8) ci[u] = c.sc[idx]; else ci[u] = reinterpret_cast<const unsigned short*>(c.sc)[idx];
                if constexpr (VD) a[u] = c.sd[c.svi[idx]]; else a[u] = c.sv[idx];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) g[u] = c.sw[ci[u]];
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (j0 + u < len) vacc(sum, vmul(a[u], g[u]));
        }
        if constexpr (STASH) {
            stash[j * 64 + lane] = sum;
        } else if constexpr (FUSED) {
            // update k of the row while (A in)_i = sum is in registers; the row's own entry of the OLD input
            // pair array comes from the staged window, the new pair goes to the other array
            const double2 in_old = c.sw[active ? dcur.own + j * 64 + lane : 0];
            if (active) newp[j] = fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(row, sum, fin[j], in_old, c.fr, cf, acc);
        } else if constexpr (epi_pr_one(EPI)) {
            // the row's own update (pr_cg.py:146-148,151) with the same expressions as the staged window, then
            // s = A p, s~ = d s and the partials of mu, dl, gm, nu, r.r (pr_cg.py:152-157)
            if (active) {
                const double xn = fin[j].xp.x + cf.al * zr[j].z;            // x += a p_old
                const double zn = zr[j].x - cf.al * zr[j].y;                // r~ -= a s~   (r -= a s without Jacobi)
                const double pn = zn + cf.bt * zr[j].z;                     // p = r~ + b p_old
                const bool st = c.fr.stream;
                if constexpr (EPI == kEpiPROneQ) {
                    // the row's packed pairs (z, zs = s) and (p, x): two 16-byte stores, a contiguous kilobyte per wave each
                    store_pair(reinterpret_cast<double2*>(c.pr.q_new) + row, make_double2(zn, sum), st);
                    store_pair(reinterpret_cast<double2*>(c.pr.px_new) + row, make_double2(pn, xn), st);
                    acc[0] += pn * sum; acc[1] += zn * sum; acc[2] += sum * sum; acc[3] += zn * zn;
                } else {
                store_one(c.pr.x + row, xn, st);
                store_one(c.pr.z_new + row, zn, st);
                store_one(c.pr.p_new + row, pn, st);
                if constexpr (EPI == kEpiPROneJ) {
                    const double rn = fin[j].rs.x - cf.al * fin[j].rs.y;    // r -= a s
                    const double stn = fin[j].d * sum;                      // s~ = M^-1 s
                    store_one(c.pr.r + row, rn, st);
                    store_one(c.pr.s + row, sum, st);
                    store_one(c.pr.zs_new + row, stn, st);
                    acc[0] += pn * sum; acc[1] += rn * stn; acc[2] += stn * sum; acc[3] += zn * rn; acc[4] += rn * rn;
                } else {
                    store_one(c.pr.zs_new + row, sum, st);
                    acc[0] += pn * sum; acc[1] += zn * sum; acc[2] += sum * sum; acc[3] += zn * zn;
                }
                }
            }
        } else if constexpr (EPI == kEpiCGOne || EPI == kEpiCGOneJ) {
            // the row's own deferred p = r~ + b p, s = w + b s (cg_cg.py:65-66), then x += a p, r -= a s (:59-60), w = A r~ and
            // the partials of eta = w.r~, nu = r.r~, r.r (:61-63)
            if (active) {
                const double zt_old = EPI == kEpiCGOneJ ? fin[j].d * zr[j].x : zr[j].x;      // r~ of the iteration being closed
                const double pn = c.lagged ? zt_old + cf.bt * fin[j].xp.y : fin[j].xp.y;
                const double sn = c.lagged ? zr[j].y + cf.bt * zr[j].z : zr[j].z;
                const double xn = fin[j].xp.x + cf.al * pn;
                const double rn = zr[j].x - cf.al * sn;
                const double ztn = EPI == kEpiCGOneJ ? fin[j].d * rn : rn;
                c.lg->x[row] = xn;
                c.lg->p[row] = pn;
                c.lg->z0n[row] = rn;
                c.lg->z1n[row] = sum;
                c.lg->z2n[row] = sn;
                acc[3] += rn * ztn; acc[1] += sum * ztn; acc[4] += rn * rn;
            }
        } else if constexpr (EPI == kEpiGVOne) {
            // the row's own deferred p = r + b p, s = w + b s, u = t + b u (gv_cg.py:77-79), then x += a p, r -= a s,
            // w -= a u (:65-67), t = A w and the partials of eta = w.r, nu = r.r (:74-75)
            if (active) {
                const double pn = c.lagged ? fin[j].rs.x + cf.bt * fin[j].xp.y : fin[j].xp.y;
                const double sn = c.lagged ? zr[j].x + cf.bt * fin[j].rs.y : fin[j].rs.y;
                const double un = c.lagged ? zr[j].y + cf.bt * zr[j].z : zr[j].z;
                const double xn = fin[j].xp.x + cf.al * pn;
                const double rn = fin[j].rs.x - cf.al * sn;
                const double wn = zr[j].x - cf.al * un;
                c.lg->x[row] = xn;
                c.lg->p[row] = pn;
                c.lg->r[row] = rn;
                c.lg->s[row] = sn;
                c.lg->z0n[row] = wn;
                c.lg->z1n[row] = sum;
                c.lg->z2n[row] = un;
                acc[3] += rn * rn; acc[1] += wn * rn; acc[4] += rn * rn;
            }
        } else if constexpr (epi_cg_w(EPI)) {
            // the row's own x += a p, r -= a s, r~ = d r (cg_cg.py:59-60, :117-118), w = A r~ and the partials of
            // eta = w.r~, nu = r.r~, r.r (:61-63)
            if (active) {
                const double xn = fin[j].xp.x + cf.al * zr[j].z;
                const double rn = zr[j].x - cf.al * zr[j].y;
                double zn = rn;
                if constexpr (EPI == kEpiCGWJ) { zn = fin[j].d * rn; c.pr.zs_new[row] = zn; }
                c.pr.x[row] = xn;
                c.pr.z_new[row] = rn;
                reinterpret_cast<double*>(c.yout)[row] = sum;
                acc[3] += rn * zn; acc[1] += sum * zn; acc[4] += rn * rn;
            }
        } else if constexpr (epi_gv_w(EPI)) {
            // the row's own x += a p, r -= a s, r~ -= a s~, w -= a u, w~ = d w (gv_cg.py:65-67 / :152-161), t = A w~ and
            // the partials of eta = w.r~, nu = r.r~, r.r (:74-75 / :163-164)
            if (active) {
                const double xn = fin[j].xp.x + cf.al * zr[j].z;
                const double rn = fin[j].rs.x - cf.al * fin[j].rs.y;
                const double wn = zr[j].x - cf.al * zr[j].y;
                double zn = rn;
                c.pr.x[row] = xn;
                c.pr.r[row] = rn;
                c.pr.z_new[row] = wn;
                if constexpr (EPI == kEpiGVWJ) {
                    zn = fin[j].w - cf.al * fin[j].wt;                      // r~ -= a s~
                    c.pr.rt[row] = zn;
                    c.pr.zs_new[row] = fin[j].d * wn;                       // w~ = M^-1 w
                }
                reinterpret_cast<double*>(c.yout)[row] = sum;
                acc[1] += wn * zn; acc[3] += rn * zn; acc[4] += rn * rn;
            }
        } else if constexpr (EPI == kEpiHS) {
            // s = A p, mu += p_i s_i (hs_cg.py:61-62); the row's own p comes from the staged window and goes to p_new
            const double pn = c.sw[active ? dcur.own + j * 64 + lane : 0];
            if (active) {
                reinterpret_cast<double*>(c.yout)[row] = sum;
                c.ep_st[row] = pn;
                acc[0] += pn * sum;
            }
        } else {
            if (active) finish_row<NV, EPI>(row, sum, c.yout, c.write_mask, c.X, c.ep_r, c.ep_d, c.ep_st, acc, cf, c.fr);
        }
    }
    if constexpr (FUSED && !STASH) {
#ifndef PRCG_DEBUG_NO_SEND
        if (c.px && dcur.send) peer_send_rows<M>(c.px, tcur, dcur.rb, newp, c.gout, lane);
#endif
    }
    wave_lds_sync();     // the next tile's image must not land before every lane has finished reading
    return next_same;
}

// tiles in flight per wave (measured, profiles/r02_sweeps.md).  With shared stream images the dictionary kernels
// are no longer bound by bytes but by what one wave can overlap: MORE resident waves with ONE image each
// (<= 128 VGPRs: 4 waves per SIMD) beat fewer waves with two (S3: 6.2 k it/s at 8 workgroups per CU, depth 1, vs
// 4.6 k at 4, depth 2).  The plain 9-byte stream is at the memory system's rate for its read/write mix with ONE
// (tools/membench.hip "fused-like": more requests in flight cost bandwidth)
#ifndef PRCG_WIN_DEPTH_DICT
#define PRCG_WIN_DEPTH_DICT 1
#endif
#ifndef PRCG_WIN_DEPTH_PAT
#define PRCG_WIN_DEPTH_PAT 1        // pattern tiles (A/B builds: 2)
#endif
#ifndef PRCG_WIN_DEPTH_PLAIN
#define PRCG_WIN_DEPTH_PLAIN 1
#endif

// One launch over window tiles.  Persistent grid, wave `slot` takes tiles slot, slot + W, ...
// Template: NV vectors (1: SpMV, 2: the pipelined SpMM on (r,s) pairs), EPI row epilogue,
// M rows per lane (tile = up to 64*M rows), PG pages, CW bits per window index, VD value
// dictionary, WPB waves per workgroup (waves are independent: no workgroup barrier in the loop),
// DEPTH register images = tiles in flight per wave (see PRCG_WIN_DEPTH_* above: one image per wave and every
// resident wave for the dictionary streams, whose images come from L2; one for the plain stream, which is HBM bound).
// DEF > 0 (communicator sessions): the inner products that update k needs are still being reduced across
// the ranks when this launch starts.  Each wave therefore computes the products of its first DEF tiles
// (the part of the iteration that only needs the old vectors) with the sums parked in LDS, then waits
// for the publication of the reduced inner products (FusedPrev::pub), then applies the DEF deferred
// updates and carries on in the fused form -- the GPU counterpart of VecDotBegin ... KSP_MatMult ...
// VecDotEnd (scaling_experiments_petsc/cg_impls/pipeprcg.c:154-173) inside ONE launch.  The wait is
// bounded; a timeout sets *err and lets the wave continue (wrong numbers, never a hang).
// (the deferred dictionary kernels of the 64-row geometries are asked to fit four workgroups per CU -- 128 VGPRs: they
//  are bound by what one wave can overlap, and a few spilled scalars cost less than a quarter of the resident waves)
// Waves per workgroup.  Every workgroup of the one-launch iteration sums the previous launch's partial rows in its
// prologue (one row per workgroup): B workgroups read B rows each, and at 2048 two-wave workgroups that was ~7 us
// of a 160 us launch (S3), a quarter of a 37 us one (one eighth of S3).  Four-wave workgroups halve B at the same
// number of resident waves (S3 +4 %, S1 +14 %, S3/8 +25 %, profiles/r02_sweeps.md) -- unless a four-wave
// workgroup's LDS no longer lets two of them share a CU (the 12-page plain geometry: S2 plain -16 %), then two.
constexpr int lds_bytes_per_wave(int nv, int pg, int cw, bool vd) {
    if (cw == 32) return 16 + 16 + 16 + 16 + pg * 64 * 8 * nv;             // pattern tiles: the window only
    return (vd ? 16 : kWinSlots * 8) + (vd ? kWinSlots : 16) + kWinSlots * (cw / 8) + (vd ? kWinDictMax * 8 : 16) + pg * 64 * 8 * nv;
}
// (... as many of the four as the workgroup's LDS -- window, streams, the stashed sums -- lets a CU hold: asking for more only
//  draws a compiler warning)
constexpr int win_min_blocks(int m, bool vd, int def, int wpb = 4, int nv = 2, int pg = 2, int cw = 8) {
    if (!(def > 0 && vd && m == 1 && wpb <= 4)) return 1;
    const int fit = (160 * 1024) / (wpb * (lds_bytes_per_wave(nv, pg, cw, vd) + def * 64 * 16) + 256);
    return fit >= 4 ? 4 : (fit >= 1 ? fit : 1);
}
template <int NV, int EPI, int M, int PG, int CW, bool VD, int WPB, int DEPTH, int DEF = 0>
__global__ __launch_bounds__(64 * WPB, win_min_blocks(M, VD, DEF, WPB, NV, PG, CW)) void k_win_tiles(
    WinDev A, const int4* __restrict__ wt, int ntiles,
    const void* __restrict__ xin_, void* __restrict__ yout_, int write_mask,
    const double* __restrict__ ep_r, const double* __restrict__ ep_d, double* __restrict__ ep_st,
    double* __restrict__ partials, double* __restrict__ aux, FusedPrev fz)
{
    using V = typename VecT<NV>::type;
    constexpr bool FUSED = epi_fused(EPI);
    static_assert(!FUSED || NV == 2, "the fused iteration works on (r,s) pairs");
    static_assert(PG * 64 <= (CW == 8 ? 256 : 32768), "window index does not fit");
    static_assert(DEPTH >= 1 && DEPTH <= 3, "one to three tiles in flight per wave");
    static_assert(DEF == 0 || (FUSED && DEF % DEPTH == 0), "the deferred tiles are whole turns of the image ring");
    __shared__ __attribute__((aligned(16))) double2 s_stash[WPB][DEF > 0 ? DEF * M * 64 : 1];
    constexpr bool PAT = CW == 32;       // pattern tiles: no per-nonzero stream is parked at all
    static_assert(!PAT || (VD && M == 1), "pattern tiles: 64 rows, values in the pattern records");
    __shared__ __attribute__((aligned(16))) double s_val[WPB][VD ? 2 : kWinSlots];
    __shared__ __attribute__((aligned(16))) unsigned char s_vi[WPB][(VD && !PAT) ? kWinSlots : 16];
    __shared__ __attribute__((aligned(16))) unsigned char s_col[WPB][PAT ? 16 : kWinSlots * (CW / 8)];
    __shared__ __attribute__((aligned(16))) double s_dict[WPB][(VD && !PAT) ? kWinDictMax : 2];
    __shared__ __attribute__((aligned(16))) V s_win[WPB][PG * 64];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WCtx<NV> c{s_val[wv], s_vi[wv], s_col[wv], s_dict[wv], s_win[wv], reinterpret_cast<const V*>(xin_), ep_r,
                     yout_, write_mask, ep_r, ep_d, ep_st,
                     FusedRowPtrs{reinterpret_cast<double2*>(yout_), reinterpret_cast<double2*>(ep_st),
                                  reinterpret_cast<double2*>(fz.rs), ep_d, fz.w, fz.wt, (write_mask & 8) != 0},
                     fz.pr, lane, nullptr, 0, nullptr, 0, &fz.lag, fz.nprev > 0,
                     CW == 32 && DEF == 0 && A.sweep_waves > 0 && A.sweep_waves == (int)gridDim.x * WPB && A.order == 0};
    WCtx<NV>& cm = c;
    if constexpr (DEF > 0) {
        if (fz.px) {
            // direct peer exchange: this launch is iteration k = want + 1; it reads the ghost rows of iteration k - 1 from
            // its own exchange buffer and sends the rows of iteration k to the neighbours' buffers
            const PeerDev* px = fz.px;
            const int R = px->nranks;
            cm.px = px; cm.n_own = px->n_own;
            cm.G = reinterpret_cast<const V*>(px->mine + peer_ghost_off(R, px->ghost_cap, (int)(fz.want & 1u))) - px->n_own;
            cm.gout = peer_ghost_off(R, px->ghost_cap, (int)((fz.want + 1u) & 1u));
        }
    }

    const int nblk = gridDim.x;
    int W = nblk * WPB;
    int t = xcd_remap(blockIdx.x, nblk) * WPB + wv;
    int tend = ntiles;                       // the wave's tiles: t, t + W, ... below tend
    if constexpr (DEF == 0) {
        if (A.order == 1) {
            // XCD-chunked order: the workgroups of one XCD (round-robin dispatch: workgroup b runs on XCD b % 8) sweep ONE
            // contiguous eighth of the tile table, front by front, instead of every eighth tile-range of a chip-wide front.
            // A 3-D stencil's plane neighbours (+-365 tiles at S2) are then rows the SAME XCD staged a round earlier or will
            // own a round later -- they meet in its L2 instead of being read through the fabric by three XCDs (S2: 0.67 -> 0.39 GB read
            // per launch, but 2 % slower: opt-in, PRCG_WIN_ORDER=1).
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = nblk >> 3, r = nblk & 7;
            const int nbx = q + (xcd < r ? 1 : 0);
            const int before = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            const int lo = (int)((long long)ntiles * before / nblk);
            tend = (int)((long long)ntiles * (before + nbx) / nblk);
            W = nbx * WPB;
            t = lo + idx * WPB + wv;
        }
    }
    bool relay = false;
    if constexpr (DEF > 0) {
        if (fz.px) {
            // direct peer exchange: wave 0 of workgroup 0 is the launch's COMMUNICATION WAVE and takes no tiles -- it sends this
            // rank's partial sums of the previous launch to every rank, waits for everybody's, and publishes the sums
            // to the other waves, all while those compute the part of the iteration that needs neither
            relay = t == 0;
            W -= 1;
            t = relay ? ntiles : t - 1;
            // a handful of boundary tiles (a band: the rows next to the cuts) go to the communication wave, after it has
            // published: they read ghost rows with system-scope loads (the slowest tile step of the launch) and would
            // otherwise end the launch one tile step late for everybody; a stencil's plane of boundary tiles stays with all
            if (ntiles - fz.nt_int <= kRelayTiles && fz.nt_int < ntiles) {
                tend = fz.nt_int;
                if (relay) { t = fz.nt_int; W = 1; tend = ntiles; }
            }
        }
    }
    // ring of DEPTH images: image i holds tile t + i*W; dn = descriptor of the tile to request next
    WRegs<win_nw(NV, EPI), M, PG, CW, VD> R[DEPTH];
    WDesc<PG> d[DEPTH], dn = {};
    // row cache (one image in flight per wave only: the cache then describes the tile processed just before)
    constexpr int RL = DEPTH == 1 ? win_row_cache_len(M, CW, VD) : 0;
    RowCache<M, RL, CW> rc;
    rc.img = 0;
    PatState ps;
    ps.id = -1; ps.nslots = 0; ps.vsel = 0u;
    bool same[DEPTH];
    // deferred form: tiles from `safe` on read ghost rows that arrive with the publication -- their loads are
    // postponed (pend) until the wave has seen it
    const int safe = DEF > 0 ? (fz.nt_int < ntiles ? fz.nt_int : ntiles) : ntiles;
    bool pend[DEPTH];
    bool acquired = false;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    Coefs cf = {0.0, 0.0, 0.0};
    constexpr bool PR1 = epi_pr_one(EPI);
    if constexpr ((FUSED || PR1) && DEF == 0) {
        // inner products of the previous iteration: still one row of partials per block of the
        // previous launch -- every block of this launch sums them in the same fixed order
        // (thread t: rows t, t+B, ...; butterfly; waves in order), see prcg_kernels.hip
        if (fz.nprev > 0) {
            // (the 256-thread tree of k_reduce_final whatever this workgroup's size: the sums do not depend on whether an
            //  iteration's partials were reduced here or by the reduction launch that ends a prcg_iterate call)
            double dsum[5];
            sum_prev_partials<5, WPB>(fz.prev_partials, fz.nprev, 0, dsum);
            if (blockIdx.x == 0 && threadIdx.x < 5) fz.dots_prev_out[threadIdx.x] = dsum[threadIdx.x];
            cf = predict(dsum, (write_mask >> 2) & 1);
        } else {
            cf = predict(PR1 ? fz.dots_old : ep_r, (write_mask >> 2) & 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
    }

    if constexpr (epi_lag(EPI)) {
        if (fz.nprev > 0) {
            // the iteration whose partials are pending is closed here (what launch_cg_update_ps does in the two-launch
            // schedule, same expressions): b = nu / nu_before, mu = eta - (b / a_before) nu (cg_cg.py:64,67), a = nu / mu
            double m[5];
            sum_prev_partials<5, WPB>(fz.prev_partials, fz.nprev, 0, m);
            const double al_before = fz.dots_old[3] / fz.dots_old[0];
            const double bt = m[3] / fz.dots_old[3];
            const double mu = m[1] - (bt / al_before) * m[3];
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                fz.dots_prev_out[0] = mu; fz.dots_prev_out[1] = m[1]; fz.dots_prev_out[3] = m[3]; fz.dots_prev_out[4] = m[4];
                fz.lag.coef_prev[0] = al_before; fz.lag.coef_prev[1] = bt;
            }
            cf.bt = bt;
            cf.al = m[3] / mu;
        } else {
            cf.al = fz.dots_old[3] / fz.dots_old[0];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) aux[0] = cf.al;
    }
    if constexpr (epi_cg_w(EPI) || epi_gv_w(EPI)) {
        cf.al = fz.dots_old[3] / fz.dots_old[0];                            // a_k1 = nu_k1 / mu_k1   cg_cg.py:58
        if (blockIdx.x == 0 && threadIdx.x == 0) aux[0] = cf.al;
    }
    if constexpr (EPI == kEpiHS) {
        // nu_k = <r~,r> of the update launch just before: still its block partials (slots 3, 4)
        double nu;
        if (fz.nprev > 0) {
            double m[2];
            sum_prev_partials<2, WPB>(fz.prev_partials, fz.nprev, 3, m);
            nu = m[0];
            if (blockIdx.x == 0 && threadIdx.x == 0) { fz.dots_prev_out[3] = m[0]; fz.dots_prev_out[4] = m[1]; }
        } else {
            nu = fz.dots_prev_out[3];
        }
        cf.bt = nu / fz.dots_old[3];                                        // b_k = nu_k / nu_k1   hs_cg.py:59
        if (blockIdx.x == 0 && threadIdx.x == 0) aux[1] = cf.bt;
    }

    __shared__ double s_mine[5];
    if constexpr (DEF > 0) {
        if (fz.px && blockIdx.x == 0 && fz.nprev > 0) {
            // peer exchange, workgroup 0: this rank's partial sums of the previous launch (one row per workgroup; complete:
            // kernel boundary) in the fixed 256-thread tree -- all its waves, before they request their first tiles (vmcnt is
            // in order); then the communication wave goes on alone
            double mine[5];
            sum_prev_partials<5, WPB>(fz.prev_partials, fz.nprev, 0, mine);
            if (threadIdx.x < 5) s_mine[threadIdx.x] = mine[threadIdx.x];
            __syncthreads();
        }
    }

    // the wave's first tiles: requested AFTER the prologue (vmcnt is in order: partial rows requested behind the images would
    // wait for the images' HBM latency -- measured 1-4 % slower; requesting them right behind the partial rows, or the
    // descriptors before the prologue, gains nothing either: profiles/r03_sweeps.md I)
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
        d[i] = WDesc<PG>{};
        pend[i] = false;
        same[i] = false;
        if (t + i * W < tend) {
            d[i] = read_desc<PG>(wt, t + i * W);
            if (t + i * W < safe) issue_loads<NV, EPI, M, PG, CW, VD>(A, d[i], lane, c.X, c.X2, c.fr, c.pr, R[i], false, c.G, c.n_own, c.lg, c.carry_ok,
                                                                      /* the wave's first DEF tiles are stashed: their rows are asked for in phase B */ DEF == 0);
            else pend[i] = true;
        }
    }
    if (t + DEPTH * W < tend) dn = read_desc<PG>(wt, t + DEPTH * W);

    if constexpr (DEF > 0) {
        static_assert(DEPTH == 1, "the deferred form keeps one image per wave");
        // ---- phase A: products of the first DEF tiles, sums parked in LDS (a loop, not unrolled: one copy of the tile
        //      step here and one in the main loop -- five unrolled copies outgrew the instruction cache) ----
        int n_def = 0;
        const int t_first = t;
#pragma unroll 1
        for (int it = 0; it < DEF; ++it) {
            if (t >= safe) break;                                   // (stops for good at the first tile that must wait)
            const WDesc<PG> dcur = d[0];
            const int tnext = t + W;
            const bool have_next = tnext < safe;
            d[0] = dn;
            pend[0] = tnext < tend && !have_next;
            same[0] = win_step<NV, EPI, M, PG, CW, VD, RL, true>(A, c, R[0], dcur, have_next, d[0], acc, cf, rc, same[0], ps,
                                                                 s_stash[wv] + it * M * 64, false, 0, /* next tile fused: */ it + 1 >= DEF);
            if (tnext + W < tend) dn = read_desc<PG>(wt, tnext + W);
            t += W;
            ++n_def;
        }
        // The deferred rows' operands -- (x,p) and the old input pair of the rows, old vectors all -- are requested for a whole
        // chunk of tiles at once (one memory round trip per chunk, not per tile), the first chunk BEFORE the wait for the
        // other ranks' inner products: the round trip and the wait overlap
        constexpr int CH = (epi_prec(EPI) || !epi_recompute(EPI)) ? (DEF >= 4 ? DEF / 2 : DEF) : (DEF > 4 ? 4 : DEF);
        FusedRowIn q[CH][M];
        double2 io[CH][M];
        int rbB[CH], reB[CH], sendB[CH];                               // rows of the chunk's tiles (wave-uniform, re-read: scalar loads)
        auto request_rows = [&](int c0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                rbB[i] = reB[i] = sendB[i] = 0;
                if (c0 + i < n_def) {
                    const int ti = t_first + (c0 + i) * W;
                    const int4 a4 = wt[ti * 6 + 0];
                    const int geo = wt[ti * 6 + 1].x;
                    rbB[i] = __builtin_amdgcn_readfirstlane(a4.x); reB[i] = __builtin_amdgcn_readfirstlane(a4.y);
                    sendB[i] = (__builtin_amdgcn_readfirstlane(geo) >> 30) & 1;
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        const int row = rbB[i] + j * 64 + lane;
                        const int rr = row < reB[i] ? row : rbB[i];
                        q[i][j].xp = c.fr.XP[rr];
                        io[i][j] = c.X[rr];
                        if constexpr (epi_prec(EPI)) { q[i][j].rs = c.fr.RS[rr]; q[i][j].d = c.fr.D[rr]; }
                        if constexpr (!epi_recompute(EPI)) { q[i][j].w = c.fr.W[rr]; if constexpr (epi_prec(EPI)) q[i][j].wt = c.fr.WT[rr]; }
                    }
                }
            }
        };
        if (n_def > 0) request_rows(0);
        // ---- the communication wave of the peer exchange ----
        if (relay) {
            const PeerDev* px = fz.px;
            double tot[5];
            if (fz.nprev > 0) {
                // this rank's partial sums of the previous launch (s_mine: summed by the whole workgroup, below) go as the rank's
                // slot into EVERY rank's exchange buffer
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < 5; ++q) v = lane == q ? s_mine[q] : v;
                peer_send_slot(px, (int)fz.want, v);
            }
            // all R slots of iteration `want` in this rank's buffer -> sums in rank order -> the publication record
            const bool ok = peer_collect(px, (int)fz.want, PRCG_WAIT_SPINS, tot);
            if (!ok && lane == 0) {
                __hip_atomic_store(fz.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (fz.err_host) __hip_atomic_store(fz.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            double* mine = const_cast<double*>(fz.pub) + (size_t)lane * 8;
#pragma unroll
            for (int q = 0; q < 5; ++q) __hip_atomic_store(mine + q, tot[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 6), fz.want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fz.dots_prev_out) {                                                      // the history's copy
#pragma unroll
                for (int q = 0; q < 5; ++q) if (lane == q) fz.dots_prev_out[q] = tot[q];
            }
            cf = predict(tot, (write_mask >> 2) & 1);
            if (lane == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
        } else {
            // ---- wait for the reduced inner products of the previous iteration ----
            // this wave's copy of the record (the copies spread the pollers over the L2 channels)
            const double* rec = fz.pub + (size_t)((blockIdx.x * WPB + wv) & (kPubCopies - 1)) * 8;
            const unsigned* cnt = reinterpret_cast<const unsigned*>(rec + 6);
            unsigned spins = 0;
            bool timed_out = __hip_atomic_load(fz.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;   // sticky: never wait twice
            while (!timed_out && (int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fz.want) < 0) {
                __builtin_amdgcn_s_sleep(32);                                 // ~1 us between polls
                if (++spins > PRCG_WAIT_SPINS) timed_out = true;             // ~10 s: a stalled peer, not a slow one
            }
            if (timed_out && lane == 0) {
                __hip_atomic_store(fz.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fz.err_host) __hip_atomic_store(fz.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // the payload was stored, and had left its CU, before the counter was: keep the payload loads behind the
            // counter load (compiler: wavefront-scope acquire emits no cache operation; hardware: vmem returns in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double dp[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dp[q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cf = predict(dp, (write_mask >> 2) & 1);
            if (!fz.px && blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
        }
        // the postponed requests.  Consumer side of the release / acquire hand-off, paid only by the waves that read
        // ghost rows (an agent-scope acquire invalidates the CU's L1: ~1.7 us each, serialised per CU): what the wave
        // loads from here on is what the publisher wrote before publishing
        if (pend[0]) {
            if (!acquired) { if (!fz.px) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); acquired = true; }
            same[0] = same_image<PG>(rc, d[0]);
            issue_loads<NV, EPI, M, PG, CW, VD>(A, d[0], lane, c.X, c.X2, c.fr, c.pr, R[0], same[0], c.G, c.n_own, c.lg, c.carry_ok);
            pend[0] = false;
        }
        // ---- phase B: the deferred updates (the first chunk's operands were requested before the wait) ----
#pragma unroll 1
        for (int c0 = 0; c0 < n_def; c0 += CH) {
            if (c0 > 0) request_rows(c0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (c0 + i < n_def) {
                    double2 newp[M];
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        const int row = rbB[i] + j * 64 + lane;
                        newp[j] = make_double2(0.0, 0.0);
                        if (row < reB[i])
                            newp[j] = fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(row, s_stash[wv][((c0 + i) * M + j) * 64 + lane],
                                                                                          q[i][j], io[i][j], c.fr, cf, acc);
                    }
                    if (fz.px && sendB[i]) peer_send_rows<M>(fz.px, t_first + (c0 + i) * W, rbB[i], newp, c.gout, lane);
                }
            }
        }
    }

    while (t < tend) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (t < tend) {
                const WDesc<PG> dcur = d[i];
                const int tnext = t + DEPTH * W;
                const bool have_next = tnext < tend;
                d[i] = dn;
                const bool acq = DEF > 0 && have_next && tnext >= safe && !acquired;
                same[i] = win_step<NV, EPI, M, PG, CW, VD, RL>(A, c, R[i], dcur, have_next, d[i], acc, cf, rc, same[i], ps, nullptr, acq, t);
                if (acq) acquired = true;
                // descriptor of the tile after that one: loaded now, looked at one step later
                if (tnext + W < tend) dn = read_desc<PG>(wt, tnext + W);
                t += W;
            }
        }
    }

    if constexpr (FUSED) { if constexpr (!epi_prec(EPI)) acc[4] = acc[3]; win_block_reduce_store<WPB, 5>(acc, partials); }
    else if constexpr (PR1) { if constexpr (EPI != kEpiPROneJ) acc[4] = acc[3]; win_block_reduce_store<WPB, 5>(acc, partials); }
    else if constexpr (epi_cg_w(EPI) || epi_gv_w(EPI) || epi_lag(EPI)) win_block_reduce_store<WPB, 5>(acc, partials);
    else if constexpr (EPI == kEpiCG) win_block_reduce_store<WPB, 5>(acc, partials);
    else if constexpr (EPI != kEpiNone) {
        double a3[3] = {acc[0], acc[1], acc[2]};
        win_block_reduce_store<WPB, 3>(a3, partials);
    }
}

constexpr int waves_per_block(int nv, int pg, int cw, bool vd) {
#ifdef PRCG_WIN_WPB
    return PRCG_WIN_WPB;
#else
    return 4 * lds_bytes_per_wave(nv, pg, cw, vd) <= 80 * 1024 ? 4 : 2;
#endif
}
// SHORT launches (few rounds of tiles: one rank's share of a strong-scaling run, S1) are dominated by what every launch pays
// once -- tools/fixed_cost.py: 12.5 us + 2.3 us per round with 1024 four-wave workgroups, 9.4 us with 256 sixteen-wave ones:
// the B x B reads of the prologue.  For them ALL the waves that stream best on one CU (16 dictionary, 8 plain, fewer where
// their LDS slices do not fit) form ONE workgroup; B <= 256 also lets the first tiles travel while the partial rows are
// summed (prev_partials_request / _finish).  Long launches keep the small workgroups (S3: 3 % faster with them).
constexpr int waves_per_block_big(int nv, int pg, int cw, bool vd) {
    int fit = (160 * 1024 - 1024) / lds_bytes_per_wave(nv, pg, cw, vd);
    const int resident = vd ? 16 : 8;
    if (fit > resident) fit = resident;
    fit -= fit % 4;
    const int base = waves_per_block(nv, pg, cw, vd);
    // (never fewer resident waves than the small workgroups give: 2-wave workgroups of the 12-page plain geometry pack 6)
    int packed = (160 * 1024) / (base * lds_bytes_per_wave(nv, pg, cw, vd)) * base;
    if (packed > resident) packed = resident;
    return (fit > base && fit >= packed) ? fit : base;
}
#ifndef PRCG_WIN_BIG_ROUNDS
#define PRCG_WIN_BIG_ROUNDS 12     // launches of at most this many rounds of tiles take the big workgroups
#endif
// (one definition for the launch and for prcg_debug_layout: the summation order of the inner products depends on it)
inline bool win_big_workgroups(int ntiles, bool vd) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus = c > 0 ? c : 256;
    }
    const long waves = (long)cus * (vd ? 16 : 8);
    return (long)ntiles <= (long)PRCG_WIN_BIG_ROUNDS * waves;
}
// ... of the deferred form: one wave on EACH SIMD of the CU (see defer_grid_per_cu) for the 64-row geometries; the 128-row
// ones (12 KB of window per wave + the stashed sums) take two-wave workgroups: a four-wave one needs 86 KB of LDS and would
// be alone on its CU
// (64-row geometries whose four-wave workgroup would be alone on its CU -- plain values with eight 2-byte-index pages:
//  22 KB per wave with the stash -- take two-wave workgroups too: three of them share a CU)
#ifndef PRCG_DEFER_WPB
#define PRCG_DEFER_WPB 4
#endif
constexpr int wpb_defer(int m, int nv = 2, int pg = 2, int cw = 8, bool vd = true) {
    if (PRCG_DEFER_WPB > 4 && m == 1 && vd && PRCG_DEFER_WPB * (lds_bytes_per_wave(nv, pg, cw, vd) + 3 * 64 * 16) <= 159 * 1024) return PRCG_DEFER_WPB;
    return (m == 1 && 4 * (lds_bytes_per_wave(nv, pg, cw, vd) + 3 * 64 * 16) <= 80 * 1024) ? 4 : 2;
}

// Workgroups per CU.  Upper bounds: what is truly co-resident (160 KiB of LDS per CU; the occupancy API
// knows the register limit) -- a persistent strided grid with queued workgroups grows a serial tail (S3
// dictionary kernel, 8 resident: 6.2 k it/s, 10 launched: 4.7 k).  Below that bound (profiles/r02_sweeps.md):
// the dictionary kernels, whose stream images come from L2, want every resident wave (16 per CU at <= 128
// VGPRs); the plain-value kernels are at the memory system's mixed read/write rate with 8 waves per CU and
// lose with more (S3 plain: 4 workgroups per CU 2.70 k it/s, 6 per CU 2.52 k).
template <typename K>
int win_grid(K kernel, int ntiles, int per_cu_override, int tuned, int wpb) {
    int dev = 0, cus = 256, occ = 4;
    if (hipGetDevice(&dev) == hipSuccess) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 64 * wpb, 0) != hipSuccess || occ < 1) occ = 4;
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.sharedSizeBytes > 0) {
            const int by_lds = (int)((160 * 1024) / fa.sharedSizeBytes);
            if (by_lds >= 1 && by_lds < occ) occ = by_lds;
        }
        if (occ > 32 / wpb) occ = 32 / wpb;
        if (occ > tuned) occ = tuned;
    }
    if (per_cu_override >= 1 && per_cu_override <= 32) occ = per_cu_override;
    int g = (ntiles + wpb - 1) / wpb;
    if (g > occ * cus) g = occ * cus;
    if (g < 1) g = 1;
    return g;
}

// The deferred form WAITS inside the launch for kernels of the communication stream (reduction, pack /
// unpack, RCCL): those must be able to become resident on a chip whose every CU already holds our
// persistent workgroups, or nobody ever publishes what the waves wait for.  Measured footprint of the
// largest of them, rcclGenericKernel of RCCL 2.26 (rocprofv3 --kernel-trace): 256 threads = one wave per
// SIMD, 132 VGPRs, 19,968 bytes of LDS.  With 4-wave workgroups every one of OUR workgroups also puts
// exactly one wave on each SIMD, so b workgroups per CU leave 512 - b * vgprs registers on every SIMD;
// 160 KiB of LDS per CU.  The bound below keeps room for one such workgroup on every CU
// (2-wave workgroups do not: two of them can land on the same SIMD pair and fill its register file --
// seen as a stalled reduction with the 250-register 128-row geometry).
int defer_grid_per_cu(const void* kernel, bool guest, int wpb) {
    // (direct peer exchange: nothing else has to run beside the launch -- only its own residency counts)
    const int kGuestVgprs = guest ? 144 : 0, kGuestLds = guest ? 24 * 1024 : 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return 1;
    const int vg = ((fa.numRegs + 7) / 8) * 8;
    const int lds = (int)fa.sharedSizeBytes;
    int best = 1;
    for (int b = 2; b <= 16 / wpb; ++b) {
        const bool regs_ok = ((b * wpb + 3) / 4) * vg + kGuestVgprs <= 512;       // waves per SIMD x registers
        const bool lds_ok = b * lds + kGuestLds <= 160 * 1024;
        if (regs_ok && lds_ok) best = b;
    }
    return best;
}

// 64-row tiles whose update waits for the other ranks' inner products.  One eighth of S3 on one GPU (nothing to wait for:
// the pure cost of deferring), peer exchange with a loopback halo: 2 tiles 30.1 us per iteration, 3 tiles 31.1, 4 tiles
// 32.5 (plain one-launch schedule 25.2); a round of tiles takes ~4 us there, the exchange chain of a real 8-rank run
// (sum, stores over xGMI, the slowest rank's skew, publication) is estimated at 6-8 us: three rounds of cover
#ifndef PRCG_DEFER_TILES
#define PRCG_DEFER_TILES 3
#endif
constexpr int kDeferTiles = PRCG_DEFER_TILES;     // 64-row tiles whose update waits for the reduction (M = 2: half as many); 6 KB of LDS per wave

template <int NV, int EPI, int M, int PG, int CW, bool vd, bool DEFER, bool BIG = false>
int launch_win_v(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, const void* x, void* y, int write_mask,
                 const double* ep_r, const double* ep_d, double* ep_st, double* partials, double* aux, FusedPrev fz,
                 int per_cu, hipEvent_t done);

template <int NV, int EPI, int M, int PG, int CW, bool DEFER = false>
int launch_win_g(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, const void* x, void* y, int write_mask,
                 const double* ep_r, const double* ep_d, double* ep_st, double* partials, double* aux, FusedPrev fz,
                 int per_cu, hipEvent_t done = nullptr)
{
    if constexpr (epi_fused(EPI) && !DEFER) {
        // the one-launch pipelined iteration of a SHORT launch: big workgroups (see waves_per_block_big)
        // (a sweep table keeps the workgroup size that divides its waves: the carry bits are worth more than the prologue)
        constexpr int kBigW = waves_per_block_big(win_nw(NV, EPI), PG, CW, true);
        if (A.big_ok && win_big_workgroups(ntiles, A.vidx8 != nullptr) && !(CW == 32 && A.sweep_waves > 0 && A.sweep_waves % kBigW != 0)) {
            if (A.vidx8 != nullptr || CW == 32)
                return launch_win_v<NV, EPI, M, PG, CW, true, DEFER, true>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials,
                                                                           aux, fz, per_cu, done);
            if constexpr (CW != 32)
                return launch_win_v<NV, EPI, M, PG, CW, false, DEFER, true>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials,
                                                                            aux, fz, per_cu, done);
        }
    }
    if (A.vidx8 != nullptr || CW == 32)
        return launch_win_v<NV, EPI, M, PG, CW, true, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz,
                                                             per_cu, done);
    if constexpr (CW != 32)
        return launch_win_v<NV, EPI, M, PG, CW, false, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz,
                                                              per_cu, done);
    return -1;
}

template <int NV, int EPI, int M, int PG, int CW, bool vd, bool DEFER, bool BIG>
int launch_win_v(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, const void* x, void* y, int write_mask,
                 const double* ep_r, const double* ep_d, double* ep_st, double* partials, double* aux, FusedPrev fz,
                 int per_cu, hipEvent_t done)
{
    constexpr int DEF = DEFER ? (M == 1 ? kDeferTiles : 2) : 0;     // (128-row tiles: two stashed tiles = 16 KB of LDS per workgroup)
    constexpr int WPB = DEFER ? wpb_defer(M, win_nw(NV, EPI), PG, CW, vd) : (BIG ? waves_per_block_big(win_nw(NV, EPI), PG, CW, vd) : waves_per_block(win_nw(NV, EPI), PG, CW, vd));
    auto k = k_win_tiles<NV, EPI, M, PG, CW, vd, WPB, ((CW == 32 && !DEFER) ? PRCG_WIN_DEPTH_PAT : (vd ? PRCG_WIN_DEPTH_DICT : PRCG_WIN_DEPTH_PLAIN)), DEF>;
    // (residency is a property of the kernel, not of the call: cached per instantiation and device)
    static int cached_ntiles_cap[2][2][16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cap = cached_ntiles_cap[vd ? 1 : 0][fz.px ? 1 : 0][dev & 15];
    int tuned = (vd ? 16 : 8) / WPB;        // resident waves per CU that stream best (see win_grid)
    if (tuned < 1) tuned = 1;
    if (DEFER) tuned = defer_grid_per_cu(reinterpret_cast<const void*>(k), fz.px == nullptr, WPB);
    if (cap == 0) cap = win_grid(k, 1 << 30, 0, tuned, WPB);
    int grid = per_cu >= 1 ? win_grid(k, ntiles, per_cu, tuned, WPB) : cap;
    const int need = (ntiles + WPB - 1) / WPB;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (per_cu < 1) {
        // every wave takes the same number of tiles: with few tiles per wave (S1: 7813 tiles on 3072 resident waves)
        // the last, partly filled round of the strided loop would cost a whole tile time
        const int waves = grid * WPB, rounds = (ntiles + waves - 1) / waves;
        grid = ((ntiles + rounds - 1) / rounds + WPB - 1) / WPB;
    }
    if (!DEFER && per_cu < 1 && CW == 32 && A.sweep_waves > 0 && A.sweep_waves % WPB == 0 && ntiles == A.sweep_tiles && A.order == 0) {
        // a sweep table: the carry bits assume exactly these waves (slot s takes tiles s, s + waves, ...)
        grid = A.sweep_waves / WPB;
    }
    if (!DEFER && per_cu < 1 && A.period > 1) {
        // the stream images repeat every `period` tiles: with a wave count that is a multiple of the period (and of the
        // workgroup size) every wave meets the SAME image tile after tile and keeps its rows decoded in registers
        long unit = A.period;
        while (unit % WPB) unit += A.period;
        const long waves = (long)grid * WPB, aligned = waves / unit * unit;
        if (aligned > 0 && aligned * 4 >= waves * 3) grid = (int)(aligned / WPB);
    }
    if (done)
        hipExtLaunchKernelGGL(k, dim3(grid), dim3(64 * WPB), 0, st, nullptr, done, 0, A, reinterpret_cast<const int4*>(tiles), ntiles,
                              x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz);
    else
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WPB), 0, st, A, reinterpret_cast<const int4*>(tiles), ntiles, x, y,
                           write_mask, ep_r, ep_d, ep_st, partials, aux, fz);
    return hipGetLastError() == hipSuccess ? grid : -1;
}

template <int NV, int EPI, bool DEFER = false>
int launch_win(int geom, hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, const void* x, void* y,
               int write_mask, const double* ep_r, const double* ep_d, double* ep_st, double* partials, double* aux,
               FusedPrev fz, int per_cu, hipEvent_t done = nullptr)
{
    switch (geom) {
    case 0: return launch_win_g<NV, EPI, 1, 2, 8, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    case 1: return launch_win_g<NV, EPI, 1, 4, 8, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    case 2: return launch_win_g<NV, EPI, 2, 8, 16, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    case 3: return launch_win_g<NV, EPI, 2, 12, 16, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    case 4: return launch_win_g<NV, EPI, 1, 8, 16, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    case 5: return launch_win_g<NV, EPI, 1, kWinPatPages, 32, DEFER>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, aux, fz, per_cu, done);
    default: return -1;
    }
}

}  // namespace

int win_fused_waves_per_block(int geom, bool value_dict, bool deferred, int ntiles, bool big_ok, int sweep_waves) {
    if (deferred) {
        switch (geom) {
        case 0: return wpb_defer(1, 2, 2, 8, value_dict);
        case 1: return wpb_defer(1, 2, 4, 8, value_dict);
        case 4: return wpb_defer(1, 2, 8, 16, value_dict);
        case 5: return wpb_defer(1, 2, kWinPatPages, 32, true);
        default: return wpb_defer(2);
        }
    }
    const bool big = big_ok && win_big_workgroups(ntiles, value_dict);
    switch (geom) {
    case 0: return big ? waves_per_block_big(2, 2, 8, value_dict) : waves_per_block(2, 2, 8, value_dict);
    case 1: return big ? waves_per_block_big(2, 4, 8, value_dict) : waves_per_block(2, 4, 8, value_dict);
    case 2: return big ? waves_per_block_big(2, 8, 16, value_dict) : waves_per_block(2, 8, 16, value_dict);
    case 3: return big ? waves_per_block_big(2, 12, 16, value_dict) : waves_per_block(2, 12, 16, value_dict);
    case 4: return big ? waves_per_block_big(2, 8, 16, value_dict) : waves_per_block(2, 8, 16, value_dict);
    case 5: return (big && !(sweep_waves > 0 && sweep_waves % waves_per_block_big(2, kWinPatPages, 32, true) != 0))
                       ? waves_per_block_big(2, kWinPatPages, 32, true) : waves_per_block(2, kWinPatPages, 32, true);
    default: return 0;
    }
}

int launch_win_spmv(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* x, double* y,
                    SpmvEpilogue epi, const double* ep_r, const double* ep_d, double* ep_st, double* partials, int per_cu)
{
    if (ntiles <= 0) return 0;
    const FusedPrev none{};
    switch (epi) {
    case kEpiNone: return launch_win<1, kEpiNone>(geom, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiDotXY: return launch_win<1, kEpiDotXY>(geom, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiPR: return launch_win<1, kEpiPR>(geom, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    case kEpiCG: return launch_win<1, kEpiCG>(geom, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, nullptr, none, per_cu);
    default: break;
    }
    return -1;
}

int launch_win_hs(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* z,
                  const double* p_old, double* p_new, double* s, double* partials, double* coef_out,
                  const FusedPrev& hs, int per_cu)
{
    if (ntiles <= 0) return 0;
    return launch_win<1, kEpiHS>(geom, st, A, tiles, ntiles, z, s, 3, p_old, nullptr, p_new, partials, coef_out, hs, per_cu);
}

int launch_win_pr_one(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                      int meurant, double* partials, double* coef_out, int per_cu)
{
    if (ntiles <= 0) return 0;
    const int mask = 3 | ((meurant & 1) ? 4 : 0) | ((meurant & 2) ? 8 : 0);      // (bit 1 of `meurant`: streaming stores)
    if (f.pr.d)
        return launch_win<1, kEpiPROneJ>(geom, st, A, tiles, ntiles, f.pr.z_old, f.pr.zs_new, mask, nullptr, f.pr.d, nullptr, partials,
                                         coef_out, f, per_cu);
    if (f.pr.q_old)
        return launch_win<1, kEpiPROneQ>(geom, st, A, tiles, ntiles, f.pr.q_old, f.pr.q_new, mask, nullptr, nullptr, nullptr, partials,
                                         coef_out, f, per_cu);
    return launch_win<1, kEpiPROne>(geom, st, A, tiles, ntiles, f.pr.z_old, f.pr.zs_new, mask, nullptr, nullptr, nullptr, partials,
                                    coef_out, f, per_cu);
}

int launch_win_cg_w(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                    double* w_out, double* partials, double* coef_out, int per_cu)
{
    if (ntiles <= 0) return 0;
    if (f.pr.d)
        return launch_win<1, kEpiCGWJ>(geom, st, A, tiles, ntiles, f.pr.z_old, w_out, 3, nullptr, f.pr.d, nullptr, partials, coef_out, f,
                                       per_cu);
    return launch_win<1, kEpiCGW>(geom, st, A, tiles, ntiles, f.pr.z_old, w_out, 3, nullptr, nullptr, nullptr, partials, coef_out, f,
                                  per_cu);
}

int launch_win_gv_w(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f,
                    double* t_out, double* partials, double* coef_out, int per_cu)
{
    if (ntiles <= 0) return 0;
    if (f.pr.d)
        return launch_win<1, kEpiGVWJ>(geom, st, A, tiles, ntiles, f.pr.z_old, t_out, 3, nullptr, f.pr.d, nullptr, partials, coef_out, f,
                                       per_cu);
    return launch_win<1, kEpiGVW>(geom, st, A, tiles, ntiles, f.pr.z_old, t_out, 3, nullptr, nullptr, nullptr, partials, coef_out, f,
                                  per_cu);
}

int launch_win_cg_one(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedPrev& f, int gv,
                      double* partials, double* coef_out, int per_cu)
{
    if (ntiles <= 0) return 0;
    if (gv)
        return launch_win<1, kEpiGVOne>(geom, st, A, tiles, ntiles, f.lag.z0, f.lag.z1n, 3, nullptr, nullptr, nullptr, partials, coef_out, f, per_cu);
    if (f.lag.d)
        return launch_win<1, kEpiCGOneJ>(geom, st, A, tiles, ntiles, f.lag.z0, f.lag.z1n, 3, nullptr, f.lag.d, nullptr, partials, coef_out, f, per_cu);
    return launch_win<1, kEpiCGOne>(geom, st, A, tiles, ntiles, f.lag.z0, f.lag.z1n, 3, nullptr, nullptr, nullptr, partials, coef_out, f, per_cu);
}

int launch_win_spmm2(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const double* rs, double* wu,
                     int write_mask, int per_cu)
{
    if (ntiles <= 0) return 0;
    return launch_win<2, kEpiNone>(geom, st, A, tiles, ntiles, rs, wu, write_mask, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   FusedPrev{}, per_cu);
}

int launch_win_pipe_fused(hipStream_t st, const WinDev& A, const WTile* tiles, int ntiles, int geom, const FusedState& f,
                          int per_cu)
{
    if (ntiles <= 0) return 0;
    FusedPrev fz = f.prev;
    fz.rs = f.rs; fz.w = f.w; fz.wt = f.wt;
    const int mask = 3 | (f.meurant ? 4 : 0) | (f.stream_stores ? 8 : 0);
    if (f.deferred) {
        if (f.dinv) {
            if (f.recompute_w)
                return launch_win<2, kEpiPipeFusedJ, true>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv,
                                                           f.in_new, f.partials, f.coef_out, fz, per_cu, f.done);
            return launch_win<2, kEpiPipeFusedPJ, true>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv,
                                                        f.in_new, f.partials, f.coef_out, fz, per_cu, f.done);
        }
        if (f.recompute_w)
            return launch_win<2, kEpiPipeFused, true>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr,
                                                      f.in_new, f.partials, f.coef_out, fz, per_cu, f.done);
        return launch_win<2, kEpiPipeFusedP, true>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr,
                                                   f.in_new, f.partials, f.coef_out, fz, per_cu, f.done);
    }
    if (f.dinv) {
        if (f.recompute_w)
            return launch_win<2, kEpiPipeFusedJ>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv, f.in_new,
                                                 f.partials, f.coef_out, fz, per_cu, f.done);
        return launch_win<2, kEpiPipeFusedPJ>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv, f.in_new,
                                              f.partials, f.coef_out, fz, per_cu, f.done);
    }
    if (f.recompute_w)
        return launch_win<2, kEpiPipeFused>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr, f.in_new,
                                            f.partials, f.coef_out, fz, per_cu, f.done);
    return launch_win<2, kEpiPipeFusedP>(geom, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr, f.in_new,
                                         f.partials, f.coef_out, fz, per_cu, f.done);
}

}  // namespace prcg
