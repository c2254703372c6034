// gfx950 (MI355X / CDNA4) kernels of the predict-and-recompute CG hot path.
//
// Everything here is HBM-bandwidth-bound fp64 streaming work (0.17-0.25 flop/byte):
// no MFMA.  What matters is 16-byte coalesced loads of val/col_ind, enough bytes in
// flight per CU, LDS-staged per-wavefront row reductions and wave64 shuffle reductions.
//
// Arithmetic contract (built with -ffp-contract=off): every update is `a + c*b` with
// the product rounded first and every SpMV row is summed left to right without FMA,
// exactly as NumPy / scipy._sparsetools.csr_matvec do in the reference
// (numerical_experiments/cg_variants/pipe_pr_cg.py:61-75, hs_cg.py:54-61), so vectors
// agree bit for bit with the reference given the same scalars; inner products use a
// fixed reduction tree (thread-sequential -> wave xor-butterfly -> waves in order ->
// blocks in order), deterministic run to run.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "prcg_kernels.h"
#include "prcg_device.hpp"

namespace prcg {
namespace {

// ======================================================================================
// CSR-adaptive SpMV / two-vector SpMM.
// One wavefront per tile, persistent grid: wave `slot` handles tiles slot, slot+W, ...
// so the chip sweeps the matrix as one front (stencil neighbours are fetched while
// their lines are still in L2 / Infinity Cache).
//   stream phase : 16-B loads of 4 column indices and 4 values per lane, gather of
//                  x[col] (NV=2: one 16-B gather of the interleaved pair), products
//                  into this wave's LDS slice;
//   reduce phase : lane i sums row i of the tile sequentially from LDS and stores it.
// A row longer than a tile is summed by the whole wave (partial sums + butterfly).
// ======================================================================================
#ifndef PRCG_ROWSUM_UNROLL
#define PRCG_ROWSUM_UNROLL 8
#endif

template <int NV>
__device__ __forceinline__ typename VecT<NV>::type lds_row_sum(const typename VecT<NV>::type* my, int s, int e) {
    using V = typename VecT<NV>::type;
    V sum; vzero(sum);
    int q = s;
#if PRCG_ROWSUM_UNROLL >= 16
    for (; q + 16 <= e; q += 16) {
        V p[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = my[q + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) vacc(sum, p[i]);
    }
#endif
#if PRCG_ROWSUM_UNROLL >= 8
    for (; q + 8 <= e; q += 8) {   // 8 LDS reads in flight (medium rows: the read latency, not the adds, is the chain)
        const V p0 = my[q], p1 = my[q + 1], p2 = my[q + 2], p3 = my[q + 3];
        const V p4 = my[q + 4], p5 = my[q + 5], p6 = my[q + 6], p7 = my[q + 7];
        vacc(sum, p0); vacc(sum, p1); vacc(sum, p2); vacc(sum, p3);
        vacc(sum, p4); vacc(sum, p5); vacc(sum, p6); vacc(sum, p7);
    }
#endif
    for (; q + 4 <= e; q += 4) {   // 4 LDS reads in flight, adds stay in row order
        const V p0 = my[q], p1 = my[q + 1], p2 = my[q + 2], p3 = my[q + 3];
        vacc(sum, p0); vacc(sum, p1); vacc(sum, p2); vacc(sum, p3);
    }
    for (; q < e; ++q) vacc(sum, my[q]);
    return sum;
}

// ---- register image of one tile's val/col stream ------------------------------------
// Each lane holds 4 CONSECUTIVE nonzeros per step.  The image keeps the words AS LOADED
// (column words of the encoding in use, values or dictionary indices); they are decoded only
// when the tile is processed, one loop phase after the loads were issued -- a decode next to
// the load would make the compiler wait for the load right there and serialise the pipeline.
template <int C16> struct ColWord;
template <> struct ColWord<0> { int4 w; };          // the int32 indices given
template <> struct ColWord<16> { uint2 w; };        // 4 x 16-bit offsets from the tile's smallest column
template <> struct ColWord<8> { unsigned w; };      // 4 x 8-bit offsets

__device__ __forceinline__ int4 decode_cols(const ColWord<0>& c, int) { return c.w; }
// (entries of the chunk that belong to a neighbouring tile were encoded against ANOTHER base;
//  decoded against this one they still land inside the vector's kGatherPad spare entries, and
//  their products are never read)
__device__ __forceinline__ int4 decode_cols(const ColWord<16>& c, int b) {
    return make_int4(b + (int)(c.w.x & 0xffffu), b + (int)(c.w.x >> 16), b + (int)(c.w.y & 0xffffu), b + (int)(c.w.y >> 16));
}
__device__ __forceinline__ int4 decode_cols(const ColWord<8>& c, int b) {
    return make_int4(b + (int)(c.w & 255u), b + (int)((c.w >> 8) & 255u), b + (int)((c.w >> 16) & 255u), b + (int)(c.w >> 24));
}

// VD = true: the values arrive as 1-byte dictionary indices (4 per lane and step, one 32-bit
// load) plus ONE dictionary entry per lane and tile
template <int STEPS, int C16, bool VD>
struct MatRegs {
    ColWord<C16> c[STEPS];
    double2 va[STEPS], vb[STEPS];
};
template <int STEPS, int C16>
struct MatRegs<STEPS, C16, true> {
    ColWord<C16> c[STEPS];
    unsigned vi[STEPS];
    double dv;
};

// Tile descriptor: RawDesc = the words as loaded (two tiles ahead), TileDesc = the same made
// wave-uniform -- again one loop phase later, so that nobody waits for the load where it is issued.
struct RawDesc { int4 d; int base; int2 v; };
struct TileDesc { int rb, re, lo, hi, base, vo, vc; };

// (the three tables are separate __restrict__ const kernel arguments: wave-uniform loads from
//  them become scalar loads, which retire on their own counter instead of queueing behind the
//  vector loads and stores of the tile before)
template <int C16, bool VD>
__device__ __forceinline__ RawDesc read_raw(const int4* __restrict__ T4, const int* __restrict__ tbase,
                                            const int2* __restrict__ vdp, int t) {
    RawDesc r;
    r.d = T4[t];
    r.base = 0; r.v = make_int2(0, 0);
    if constexpr (C16 != 0) r.base = tbase[t];
    if constexpr (VD) r.v = vdp[t];
    return r;
}
__device__ __forceinline__ TileDesc cook(const RawDesc& r) {
    TileDesc o;
    o.rb = __builtin_amdgcn_readfirstlane(r.d.x); o.re = __builtin_amdgcn_readfirstlane(r.d.y);
    o.lo = __builtin_amdgcn_readfirstlane(r.d.z); o.hi = __builtin_amdgcn_readfirstlane(r.d.w);
    o.base = __builtin_amdgcn_readfirstlane(r.base);
    o.vo = __builtin_amdgcn_readfirstlane(r.v.x); o.vc = __builtin_amdgcn_readfirstlane(r.v.y);
    return o;
}

// Branch-free 16-byte (4-byte, 8-byte) loads of the tile's column words and values: a lane
// whose chunk lies past the tile re-reads the tile's first chunk (one hot line); its products
// land in LDS slots nobody reads.
// C16: width of the streamed column encoding: 0 = the int32 given, 16 / 8 = tile-relative offsets
template <int STEPS, int C16, bool VD>
__device__ __forceinline__ void load_tile_stream(const CsrDev& A, const TileDesc& d, int lane, MatRegs<STEPS, C16, VD>& m) {
    const int alo = d.lo & ~3;
    if constexpr (VD) m.dv = lane < d.vc ? A.vdict[d.vo + lane] : 0.0;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int base = alo + st * 256 + lane * 4;
        const int lb = base < d.hi ? base : alo;
        if constexpr (C16 == 8) m.c[st].w = *reinterpret_cast<const unsigned*>(A.col8 + lb);
        else if constexpr (C16 == 16) m.c[st].w = *reinterpret_cast<const uint2*>(A.col16 + lb);
        else m.c[st].w = *reinterpret_cast<const int4*>(A.col + lb);
        if constexpr (VD) {
            // 4 dictionary indices in 4 bytes (bytes of a neighbouring tile index ITS dictionary;
            // the lookup masks them into range and those products are never read)
            m.vi[st] = *reinterpret_cast<const unsigned*>(A.vidx8 + lb);
        } else {
            m.va[st] = *reinterpret_cast<const double2*>(A.val + lb);
            m.vb[st] = *reinterpret_cast<const double2*>(A.val + lb + 2);
        }
    }
}

// One tile: gather x[col], products -> this wave's LDS slice, then one lane per row sums
// its row left to right.  `cur` holds the tile's val/col stream (loaded one tile ago).
template <int NV, int EPI, int STEPS, int C16, bool VD>
__device__ __forceinline__ void process_tile(
    const CsrDev& A, const TileDesc& d, int lane, const MatRegs<STEPS, C16, VD>& cur,
    typename VecT<NV>::type* my, double* dict, const typename VecT<NV>::type* __restrict__ X,
    void* __restrict__ yout_, int write_mask, const double* __restrict__ ep_r,
    const double* __restrict__ ep_d, double* __restrict__ ep_st, double (&acc)[5], const Coefs& cf,
    const FusedRowPtrs& fr)
{
    using V = typename VecT<NV>::type;
    constexpr int kCap = 256 * STEPS - 3;
    const int rb = d.rb, re = d.re, lo = d.lo, hi = d.hi;
    if (hi - lo > kCap) {
        // ---- long row: the planner gives it a tile of its own (re == rb+1) ----------
        V sum; vzero(sum);
        for (int q = lo + lane; q < hi; q += 64) vacc(sum, vmul(A.val[q], X[A.col[q]]));
        sum = vwave_sum(sum);
        if (lane == 0) finish_row<NV, EPI>(rb, sum, yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
        return;
    }
    const int alo = lo & ~3;   // 16-B aligned start; head slots < lo are never read
    // row pointers of the first two row batches (branch-free, clamped: issued now, waited
    // for only in the reduce phase)
    const int row0 = rb + lane, row1 = rb + 64 + lane;
    const int* ip0 = A.indptr + (row0 < re ? row0 : rb);
    const int* ip1 = A.indptr + (row1 < re ? row1 : rb);
    const int s0r = ip0[0], e0r = ip0[1], s1r = ip1[0], e1r = ip1[1];
    if constexpr (VD) {
        // the tile's dictionary: one entry per lane into this wave's LDS slot
        dict[lane] = cur.dv;
        wave_lds_sync();
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int4 cc = decode_cols(cur.c[st], d.base);
        const V g0 = X[cc.x], g1 = X[cc.y], g2 = X[cc.z], g3 = X[cc.w];
        double a0, a1, a2, a3;
        if constexpr (VD) {
            const unsigned v = cur.vi[st];
            a0 = dict[v & 63u]; a1 = dict[(v >> 8) & 63u]; a2 = dict[(v >> 16) & 63u]; a3 = dict[(v >> 24) & 63u];   // & 63: in range whatever the byte
        } else {
            a0 = cur.va[st].x; a1 = cur.va[st].y; a2 = cur.vb[st].x; a3 = cur.vb[st].y;
        }
        const int o = st * 256 + lane * 4;
        my[o + 0] = vmul(a0, g0);
        my[o + 1] = vmul(a1, g1);
        my[o + 2] = vmul(a2, g2);
        my[o + 3] = vmul(a3, g3);
    }
    wave_lds_sync();
    if (row0 < re) finish_row<NV, EPI>(row0, lds_row_sum<NV>(my, s0r - alo, e0r - alo), yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
    if (row1 < re) finish_row<NV, EPI>(row1, lds_row_sum<NV>(my, s1r - alo, e1r - alo), yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
    for (int row = rb + 128 + lane; row < re; row += 64) {
        const int s = A.indptr[row] - alo;
        const int e = A.indptr[row + 1] - alo;
        finish_row<NV, EPI>(row, lds_row_sum<NV>(my, s, e), yout_, write_mask, X, ep_r, ep_d, ep_st, acc, cf, fr);
    }
    wave_lds_sync();
}

// Software pipeline, two tiles deep: while a wave gathers / reduces tile t, the val/col
// stream of tile t+W is already in flight (second register image) and the descriptor of
// tile t+2W is being fetched.  The dependent chain per tile is then just
// gather -> LDS -> row sums, and every wave keeps HBM loads outstanding all the time.
// Everything loaded ahead stays RAW until the phase that needs it (see MatRegs / RawDesc).
template <int NV, int EPI, int STEPS, int C16, bool VD>
__global__ __launch_bounds__(kBlock) void k_spmv_tiles(
    CsrDev A, const Tile* __restrict__ tiles, const int* __restrict__ tbase, const int2* __restrict__ vdp, int ntiles,
    const void* __restrict__ xin_, void* __restrict__ yout_, int write_mask,
    const double* __restrict__ ep_r, const double* __restrict__ ep_d,
    double* __restrict__ ep_st, double* __restrict__ partials, int chunked, double* __restrict__ aux,
    FusedPrev fz)
{
    using V = typename VecT<NV>::type;
    constexpr int kSlots = 256 * STEPS;
    constexpr int kCap = kSlots - 3;
    __shared__ V prod[kWaves][kSlots];
    __shared__ double dict_lds[VD ? kWaves : 1][VD ? kDictMax : 1];

    const V* __restrict__ X = reinterpret_cast<const V*>(xin_);
    const int4* __restrict__ T4 = reinterpret_cast<const int4*>(tiles);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    V* my = prod[wv];
    double* dict = dict_lds[VD ? wv : 0];

    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    Coefs cf = {0.0, 0.0, 0.0};
    const FusedRowPtrs fr{reinterpret_cast<double2*>(yout_), reinterpret_cast<double2*>(ep_st), reinterpret_cast<double2*>(fz.rs),
                          ep_d, fz.w, fz.wt, (write_mask & 8) != 0};
    if constexpr (epi_fused(EPI)) {
        // ep_r = the reduced inner products of the previous iteration; bit 2 of write_mask =
        // Meurant's prediction; aux = where alpha, beta, nu_pred of this iteration are kept
        if (fz.nprev > 0) {
            // The previous launch left one row of partial inner products per block.  Every
            // block of THIS launch sums them itself, in the same fixed order (thread t: rows
            // t, t+256, ...; butterfly; waves in order), so no separate reduction launch is
            // needed and all blocks get bit-identical coefficients.  The kernel boundary
            // makes the partials visible; nothing is exchanged inside a launch.
            __shared__ double redp[kWaves][5];
            __shared__ double dsum[5];
            double tot[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j = threadIdx.x; j < fz.nprev; j += kBlock) {
#pragma unroll
                for (int q = 0; q < 5; ++q) tot[q] += fz.prev_partials[(size_t)j * kPartialStride + q];
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const double v = wave_sum(tot[q]);
                if (lane == 0) redp[wv][q] = v;
            }
            __syncthreads();
            if (threadIdx.x < 5) {
                double v = redp[0][threadIdx.x];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) v += redp[w][threadIdx.x];
                dsum[threadIdx.x] = v;
                if (blockIdx.x == 0) fz.dots_prev_out[threadIdx.x] = v;    // history / get_scalars
            }
            __syncthreads();
            cf = predict(dsum, (write_mask >> 2) & 1);
        } else {
            cf = predict(ep_r, (write_mask >> 2) & 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = cf.al; aux[1] = cf.bt; aux[2] = cf.nup; }
    }

    // Tile order.  strided (default): wave `slot` takes tiles slot, slot+W, ... so the whole
    // chip sweeps the matrix as one front (stencil neighbours are fetched while their
    // lines are still in L2 / Infinity Cache).  chunked: wave `slot` owns the contiguous
    // tile range [slot*T, (slot+1)*T) -- measured slower on S3 (profiles/r01_sweeps.md).
    const int nblk = gridDim.x;
    const int W = nblk * kWaves;
    const int slot = xcd_remap(blockIdx.x, nblk) * kWaves + wv;
    int t, tend, step;
    if (chunked) {
        const int T = (ntiles + W - 1) / W;
        t = slot * T;
        tend = t + T < ntiles ? t + T : ntiles;
        step = 1;
    } else {
        t = slot; tend = ntiles; step = W;
    }

    MatRegs<STEPS, C16, VD> m0, m1;
    TileDesc d0 = {0, 0, 0, 0, 0, 0, 0}, d1 = {0, 0, 0, 0, 0, 0, 0};
    RawDesc rn = {make_int4(0, 0, 0, 0), 0, make_int2(0, 0)};     // raw descriptor of the tile after the current one
    if (t < tend) {
        d0 = cook(read_raw<C16, VD>(T4, tbase, vdp, t));
        if (d0.hi - d0.lo <= kCap) load_tile_stream<STEPS, C16, VD>(A, d0, lane, m0);
        if (t + step < tend) rn = read_raw<C16, VD>(T4, tbase, vdp, t + step);
    }

    while (t < tend) {
        // ---- even phase: tile t lives in m0 / d0; rn = raw descriptor of tile t+step (loaded a phase ago) ----
        {
            if (t + step < tend) {
                d1 = cook(rn);
                if (d1.hi - d1.lo <= kCap) load_tile_stream<STEPS, C16, VD>(A, d1, lane, m1);
            }
            const int t2 = t + 2 * step;
            rn = read_raw<C16, VD>(T4, tbase, vdp, t2 < tend ? t2 : t);
            process_tile<NV, EPI, STEPS, C16, VD>(A, d0, lane, m0, my, dict, X, yout_, write_mask, ep_r, ep_d, ep_st, acc, cf, fr);
            t += step;
        }
        if (t >= tend) break;
        // ---- odd phase: tile t lives in m1 / d1 ----
        {
            if (t + step < tend) {
                d0 = cook(rn);
                if (d0.hi - d0.lo <= kCap) load_tile_stream<STEPS, C16, VD>(A, d0, lane, m0);
            }
            const int t2 = t + 2 * step;
            rn = read_raw<C16, VD>(T4, tbase, vdp, t2 < tend ? t2 : t);
            process_tile<NV, EPI, STEPS, C16, VD>(A, d1, lane, m1, my, dict, X, yout_, write_mask, ep_r, ep_d, ep_st, acc, cf, fr);
            t += step;
        }
    }

    if constexpr (epi_fused(EPI)) { if constexpr (!epi_prec(EPI)) acc[4] = acc[3]; block_reduce_store<5>(acc, partials, 0); }
    else if constexpr (EPI == kEpiCG) block_reduce_store<5>(acc, partials, 0);
    else if constexpr (EPI != kEpiNone) {
        double a3[3] = {acc[0], acc[1], acc[2]};
        block_reduce_store<3>(a3, partials, 0);
    }
}

// ======================================================================================
// Fused vector updates + inner products.  Block b owns `trips` consecutive 512-element
// trips; thread t handles elements 2t, 2t+1 of each trip with 16-byte accesses.
// ======================================================================================
// All state of the pipelined variants is stored as 16-byte pairs -- XP = (x,p), RS = (r,s),
// WU = (w,u), RSt = (r~,s~) -- so that every global access of this kernel is one fully
// coalesced 16-byte-per-lane instruction.  Thread t of block b handles elements
// base + t and base + 256 + t of each 512-element trip (two independent elements in
// flight).
template <bool PREC, bool DOTS_ONLY>
__global__ __launch_bounds__(kBlock) void k_pipe_update(PipeUpdateArgs a, int trips) {
    Coefs c = {0.0, 0.0, 0.0};
    if constexpr (!DOTS_ONLY) {
        c = predict(a.dots_prev, a.meurant);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.coef_out[0] = c.al; a.coef_out[1] = c.bt; a.coef_out[2] = c.nup;
        }
    }
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // mu, dl, gm, nu, rr
    const int64_t n = a.n;
    double2* __restrict__ XP = reinterpret_cast<double2*>(a.xp);
    double2* __restrict__ RS = reinterpret_cast<double2*>(a.rs);
    double2* __restrict__ RST = reinterpret_cast<double2*>(a.rst);
    double2* __restrict__ WU = reinterpret_cast<double2*>(a.wu);
    double* __restrict__ WT = a.wt;
    const double* __restrict__ D = a.d;
    const bool recompute_w = a.recompute_w != 0;

    int64_t base = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x;
    for (int j = 0; j < trips; ++j, base += kElemsPerTrip) {
        if (base >= n) break;
        // ---- loads of both elements first ----
        double2 xp[2], rs[2], wu[2], rst[2];
        double dv[2], wt[2], utv[2];
        bool ok[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t ie = base + e * kBlock;
            ok[e] = ie < n;
            const int64_t il = ok[e] ? ie : base;      // clamped: branch-free loads
            xp[e] = XP[il];
            rs[e] = RS[il];
            if constexpr (!DOTS_ONLY) wu[e] = WU[il];
            if constexpr (PREC) {
                rst[e] = RST[il];
                if constexpr (!DOTS_ONLY) {
                    if (a.ut) {              // preconditioner applied elsewhere (host callback): u~ and w~ are vectors
                        dv[e] = 0.0; utv[e] = a.ut[il]; wt[e] = WT[il];
                    } else {
                        dv[e] = D[il];
                        utv[e] = 0.0;
                        wt[e] = recompute_w ? 0.0 : WT[il];
                    }
                }
            }
        }
        // ---- arithmetic + stores, element 0 then element 1 (this order is part of the
        //      reduction tree the tests emulate) ----
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!ok[e]) continue;
            const int64_t ie = base + e * kBlock;
            if constexpr (DOTS_ONLY) {
                if constexpr (PREC) {
                    acc[0] += xp[e].y * rs[e].y; acc[1] += rs[e].x * rst[e].y; acc[2] += rst[e].y * rs[e].y;
                    acc[3] += rst[e].x * rs[e].x; acc[4] += rs[e].x * rs[e].x;
                } else {
                    acc[0] += xp[e].y * rs[e].y; acc[1] += rs[e].x * rs[e].y; acc[2] += rs[e].y * rs[e].y;
                    acc[3] += rs[e].x * rs[e].x;
                }
            } else {
                const double xn = xp[e].x + c.al * xp[e].y;             // x += a p
                const double rn = rs[e].x - c.al * rs[e].y;             // r -= a s
                const double wn = wu[e].x - c.al * wu[e].y;             // w -= a u
                if constexpr (PREC) {
                    const double ut = a.ut ? utv[e] : dv[e] * wu[e].y;  // u~ = M^-1 u
                    const double wtv = a.ut ? wt[e] : (recompute_w ? dv[e] * wu[e].x : wt[e]);   // w~
                    const double rtn = rst[e].x - c.al * rst[e].y;      // r~ -= a s~
                    const double wtn = wtv - c.al * ut;                 // w~ -= a u~
                    const double pn = rtn + c.bt * xp[e].y;             // p = r~ + b p
                    const double sn = wn + c.bt * rs[e].y;              // s = w + b s
                    const double stn = wtn + c.bt * rst[e].y;           // s~ = w~ + b s~
                    XP[ie] = make_double2(xn, pn);
                    RS[ie] = make_double2(rn, sn);
                    RST[ie] = make_double2(rtn, stn);
                    if (!recompute_w) { a.wu[2 * ie] = wn; WT[ie] = wtn; }
                    acc[0] += pn * sn; acc[1] += rn * stn; acc[2] += stn * sn;
                    acc[3] += rtn * rn; acc[4] += rn * rn;
                } else {
                    const double pn = rn + c.bt * xp[e].y;              // p = r + b p
                    const double sn = wn + c.bt * rs[e].y;              // s = w + b s
                    XP[ie] = make_double2(xn, pn);
                    RS[ie] = make_double2(rn, sn);
                    if (!recompute_w) a.wu[2 * ie] = wn;
                    acc[0] += pn * sn; acc[1] += rn * sn; acc[2] += sn * sn; acc[3] += rn * rn;
                }
            }
        }
    }
    if constexpr (!PREC) acc[4] = acc[3];
    if (a.final_out) block_reduce_store_final<5>(acc, a.partials, a.ticket, a.final_out);
    else block_reduce_store<5>(acc, a.partials, 0);
}

// ---- Hestenes-Stiefel (hs_cg.py:54-61, hs_pcg :116-124) ------------------------------
template <bool PREC, bool DOTS_ONLY>
__global__ __launch_bounds__(kBlock) void k_hs_update_xr(HsArgs a, int trips, const double* __restrict__ prev_mu, int nprev,
                                                         double* dots_prev_w) {
    double al = 0.0;
    if constexpr (!DOTS_ONLY) {
        double mu;
        if (nprev > 0) {
            // mu = <p,s> of the previous iteration: still the product launch's block partials
            double m[1];
            sum_prev_partials<1, kWaves>(prev_mu, nprev, 0, m);
            mu = m[0];
            if (blockIdx.x == 0 && threadIdx.x == 0) dots_prev_w[0] = mu;
        } else {
            mu = a.dots_prev[0];
        }
        al = a.dots_prev[3] / mu;                               // a_k1 = nu/mu
        if (blockIdx.x == 0 && threadIdx.x == 0) a.coef_out[0] = al;
    }
    double acc[2] = {0.0, 0.0};   // nu, rr
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
        if (i + 1 < a.n) {
            // the thread's two elements as 16-byte loads and stores (same expressions, same order of the sums: element i, then i + 1)
            double2 rn = *reinterpret_cast<const double2*>(a.r + i);
            if constexpr (!DOTS_ONLY) {
                const double2 x2 = *reinterpret_cast<const double2*>(a.x + i), p2 = *reinterpret_cast<const double2*>(a.p + i);
                const double2 s2 = *reinterpret_cast<const double2*>(a.s + i);
                *reinterpret_cast<double2*>(a.x + i) = make_double2(x2.x + al * p2.x, x2.y + al * p2.y);
                rn = make_double2(rn.x - al * s2.x, rn.y - al * s2.y);
                *reinterpret_cast<double2*>(a.r + i) = rn;
            }
            if constexpr (PREC) {
                const double2 d2 = *reinterpret_cast<const double2*>(a.d + i);
                const double2 z = make_double2(d2.x * rn.x, d2.y * rn.y);      // r~ = M^-1 r
                *reinterpret_cast<double2*>(a.rt + i) = z;
                acc[0] += rn.x * z.x; acc[1] += rn.x * rn.x;
                acc[0] += rn.y * z.y; acc[1] += rn.y * rn.y;
            } else {
                acc[0] += rn.x * rn.x;
                acc[0] += rn.y * rn.y;
            }
        } else if (i < a.n) {
            double rn = a.r[i];
            if constexpr (!DOTS_ONLY) {
                a.x[i] = a.x[i] + al * a.p[i];
                rn = rn - al * a.s[i];
                a.r[i] = rn;
            }
            if constexpr (PREC) {
                const double z = a.d[i] * rn;
                a.rt[i] = z;
                acc[0] += rn * z; acc[1] += rn * rn;
            } else {
                acc[0] += rn * rn;
            }
        }
    }
    if constexpr (!PREC) acc[1] = acc[0];
    block_reduce_store<2>(acc, a.partials, 3);
}

__global__ __launch_bounds__(kBlock) void k_hs_update_p(HsArgs a, int trips, const double* __restrict__ prev_nu, int nprev,
                                                        double* dots_cur_w) {
    double nu;
    if (nprev > 0) {
        double m[2];
        sum_prev_partials<2, kWaves>(prev_nu, nprev, 3, m);     // nu_k, rr_k: still the update launch's block partials
        nu = m[0];
        if (blockIdx.x == 0 && threadIdx.x == 0) { dots_cur_w[3] = m[0]; dots_cur_w[4] = m[1]; }
    } else {
        nu = a.dots_cur[3];
    }
    const double bt = nu / a.dots_prev[3];                      // b_k = nu_k / nu_k1
    if (blockIdx.x == 0 && threadIdx.x == 0) a.coef_out[1] = bt;
    const double* __restrict__ z = a.rt ? a.rt : a.r;
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
        if (i + 1 < a.n) {                                      // the thread's two elements as 16-byte loads and stores
            const double2 z2 = *reinterpret_cast<const double2*>(z + i), p2 = *reinterpret_cast<const double2*>(a.p + i);
            *reinterpret_cast<double2*>(a.p + i) = make_double2(z2.x + bt * p2.x, z2.y + bt * p2.y);     // p = r~ + b p
        } else if (i < a.n) {
            a.p[i] = z[i] + bt * a.p[i];
        }
    }
}

// ---- non-pipelined predict-and-recompute (pr_cg.py:146-151) ---------------------------
template <bool PREC, bool DOTS_ONLY>
__global__ __launch_bounds__(kBlock) void k_pr_update(PrArgs a, int trips) {
    Coefs c = {0.0, 0.0, 0.0};
    if constexpr (!DOTS_ONLY) {
        c = predict(a.dots_prev, a.meurant);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.coef_out[0] = c.al; a.coef_out[1] = c.bt; a.coef_out[2] = c.nup;
        }
    }
    double acc[2] = {0.0, 0.0};   // nu = r~.r, rr
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
        if (i + 1 < a.n) {
            // the thread's two elements as 16-byte loads and stores (same expressions, sums in the order element i, then i + 1)
            double2 rn = *reinterpret_cast<const double2*>(a.r + i);
            double2 rtn = rn;
            if constexpr (PREC) rtn = *reinterpret_cast<const double2*>(a.rt + i);
            if constexpr (!DOTS_ONLY) {
                const double2 x2 = *reinterpret_cast<const double2*>(a.x + i), p2 = *reinterpret_cast<const double2*>(a.p + i);
                const double2 s2 = *reinterpret_cast<const double2*>(a.s + i);
                *reinterpret_cast<double2*>(a.x + i) = make_double2(x2.x + c.al * p2.x, x2.y + c.al * p2.y);
                rn = make_double2(rn.x - c.al * s2.x, rn.y - c.al * s2.y);
                *reinterpret_cast<double2*>(a.r + i) = rn;
                if constexpr (PREC) {
                    const double2 st2 = *reinterpret_cast<const double2*>(a.st_ + i);
                    rtn = make_double2(rtn.x - c.al * st2.x, rtn.y - c.al * st2.y);
                    *reinterpret_cast<double2*>(a.rt + i) = rtn;
                } else {
                    rtn = rn;
                }
                *reinterpret_cast<double2*>(a.p + i) = make_double2(rtn.x + c.bt * p2.x, rtn.y + c.bt * p2.y);
            }
            acc[0] += rtn.x * rn.x; acc[1] += rn.x * rn.x;
            acc[0] += rtn.y * rn.y; acc[1] += rn.y * rn.y;
        } else if (i < a.n) {
            double rn = a.r[i];
            double rtn = PREC ? a.rt[i] : rn;
            if constexpr (!DOTS_ONLY) {
                a.x[i] = a.x[i] + c.al * a.p[i];
                rn = rn - c.al * a.s[i];
                a.r[i] = rn;
                if constexpr (PREC) {
                    rtn = rtn - c.al * a.st_[i];
                    a.rt[i] = rtn;
                } else {
                    rtn = rn;
                }
                a.p[i] = rtn + c.bt * a.p[i];
            }
            acc[0] += rtn * rn; acc[1] += rn * rn;
        }
    }
    block_reduce_store<2>(acc, a.partials, 3);
}

// ---- competitor baselines: Chronopoulos-Gear (cg_cg.py:59-68) and Ghysels-Vanroose
//      (gv_cg.py:65-81).  mu is a RECURRENCE here, not an inner product:
//      mu_k = eta_k - (b_k / a_k1) nu_k, computed by thread 0 and stored with the scalars. ----
__global__ __launch_bounds__(kBlock) void k_cg_update_ps(CgArgs a, int trips, const double* __restrict__ prev, int nprev) {
    // dots_prev: nu_k1 (slot 3), mu_k1 (slot 0); dots_cur: nu_k (3), eta_k (1)
    const double al = a.dots_prev[3] / a.dots_prev[0];          // a_k1
    double nu, eta;
    if (nprev > 0) {
        double m[5];
        sum_prev_partials<5, kWaves>(prev, nprev, 0, m);        // the product launch's block partials
        nu = m[3]; eta = m[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.dots_cur_w[1] = m[1]; a.dots_cur_w[3] = m[3]; a.dots_cur_w[4] = m[4]; }
    } else {
        nu = a.dots_cur[3]; eta = a.dots_cur[1];
    }
    const double bt = nu / a.dots_prev[3];                      // b_k
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.coef_out[0] = al; a.coef_out[1] = bt;
        a.dots_cur_w[0] = eta - (bt / al) * nu;                 // mu_k
    }
    const double* __restrict__ z = a.z;
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    auto axpby2 = [bt](double* __restrict__ y, const double* __restrict__ x, int64_t at) {      // y = x + b y on two elements, 16 bytes each way
        const double2 x2 = *reinterpret_cast<const double2*>(x + at), y2 = *reinterpret_cast<const double2*>(y + at);
        *reinterpret_cast<double2*>(y + at) = make_double2(x2.x + bt * y2.x, x2.y + bt * y2.y);
    };
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
        if (i + 1 < a.n) {
            axpby2(a.p, z, i);                                  // p = r~ + b p
            axpby2(a.s, a.w, i);                                // s = w + b s
            if (a.st_) axpby2(a.st_, a.wt, i);                  // s~ = w~ + b s~   (gv_pcg)
            if (a.u) axpby2(a.u, a.t, i);                       // u = t + b u      (gv)
        } else if (i < a.n) {
            a.p[i] = z[i] + bt * a.p[i];
            a.s[i] = a.w[i] + bt * a.s[i];
            if (a.st_) a.st_[i] = a.wt[i] + bt * a.st_[i];
            if (a.u) a.u[i] = a.t[i] + bt * a.u[i];
        }
    }
}

// gv: x += a p; r -= a s; r~ -= a s~; w -= a u; w~ = M^-1 w; partials nu = r.r~, eta = w.r~, r.r
template <bool PREC, bool DOTS_ONLY>
__global__ __launch_bounds__(kBlock) void k_gv_update1(CgArgs a, int trips) {
    double al = 0.0;
    if constexpr (!DOTS_ONLY) al = a.dots_prev[3] / a.dots_prev[0];
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // slots: -, eta, -, nu, rr
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t ie = i + e;
            if (ie >= a.n) break;
            double rn = a.r[ie], wn = a.w[ie];
            double zn = PREC ? a.rt[ie] : rn;
            if constexpr (!DOTS_ONLY) {
                a.x[ie] = a.x[ie] + al * a.p[ie];
                rn = rn - al * a.s[ie];
                a.r[ie] = rn;
                wn = wn - al * a.u[ie];
                a.w[ie] = wn;
                if constexpr (PREC) {
                    zn = zn - al * a.st_[ie];
                    a.rt[ie] = zn;
                    if (a.d) a.wt[ie] = a.d[ie] * wn;        // (null: w~ = M^-1 w is applied by the caller afterwards)
                } else {
                    zn = rn;
                }
            }
            // gv_pcg's initial eta is w.r, its loop's eta is w.r~ (gv_cg.py:107 vs :163)
            acc[1] += wn * ((DOTS_ONLY && PREC) ? rn : zn);
            acc[3] += rn * zn; acc[4] += rn * rn;
        }
    }
    block_reduce_store<5>(acc, a.partials, 0);
}

// ======================================================================================
// Small systems (n <= 4096): the whole pipelined solve in ONE launch of ONE workgroup.
// nos7 (n=729) or bcsstk03 (n=112) cannot fill a chip; with one launch per kernel they are
// launch-bound (~12 us per iteration).  Here 1024 threads keep (r,s) in LDS (double-buffered,
// like the one-launch schedule above), x and p of their own rows in registers, the matrix
// in registers or LDS, and iterate `iters` times with two
// barriers per iteration.  Same arithmetic per element; rows are summed left to right.
// Inner products: thread-sequential over its rows (row = tid, tid+1024, ...), wave butterfly,
// 16 waves in order.  Every thread reaches every barrier (trip counts are uniform).
// ======================================================================================
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's
// outstanding GLOBAL stores (vmcnt(0)); thread 0 streams the per-iteration scalars to memory,
// and waiting for their acknowledgement twice per iteration cost ~3 us of a 4.5 us iteration.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

static_assert(kPartialStride == 8, "k_small_pipe_pr indexes the scalar history (PRCG_NUM_SCALARS doubles per iteration) with kPartialStride");
constexpr int kSmallThreads = 1024;
constexpr int kSmallRows = 4;                    // rows per thread -> n <= 4096
constexpr int kSmallMaxN = kSmallThreads * kSmallRows;

// MODE 0: matrix in LDS (any row length, up to 4 rows per thread)
// MODE 1: matrix in REGISTERS: n <= 1024 (one row per thread) and every row <= kSmallRegLen
//         nonzeros -- the paper's small test matrices (nos7: rows of 4-7, bcsstk03: 4-6).
constexpr int kSmallRegLen = 8;

template <int MODE>
__global__ __launch_bounds__(kSmallThreads) void k_small_pipe_pr(SmallArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ROWS = MODE == 1 ? 1 : kSmallRows;
    const int n = a.n, nnz = a.nnz;
    double2* rsA = reinterpret_cast<double2*>(smem);
    double2* rsB = rsA + n;
    double* red = reinterpret_cast<double*>(rsB + n);          // [16 waves][4]
    double* lval = red + 64;
    int* lcol = reinterpret_cast<int*>(lval + (MODE == 0 ? nnz : 0));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    const double2* __restrict__ XPg = reinterpret_cast<const double2*>(a.xp);
    const double2* __restrict__ RSg = reinterpret_cast<const double2*>(a.rs);
    double xr[ROWS], pr[ROWS];
    int rbeg[ROWS], rend[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int row = tid + j * kSmallThreads;
        xr[j] = 0.0; pr[j] = 0.0; rbeg[j] = 0; rend[j] = 0;
        if (row < n) {
            const double2 xp = XPg[row];
            xr[j] = xp.x; pr[j] = xp.y;
            rsA[row] = RSg[row];
            rbeg[j] = a.indptr[row]; rend[j] = a.indptr[row + 1];
        }
    }
    // MODE 1: this thread's row, padded with (0.0, own column) -- adding +-0.0 products at the
    // END of the row sum cannot change it (x + 0.0 == x, and a finite entry times 0.0 is 0.0;
    // inf/nan entries poison the row either way)... except that 0.0 * inf = nan: so padded
    // slots are skipped by a length test instead of being multiplied.
    double rv[kSmallRegLen];
    int rc[kSmallRegLen];
    int rlen = 0;
    if constexpr (MODE == 1) {
        rlen = rend[0] - rbeg[0];
#pragma unroll
        for (int q = 0; q < kSmallRegLen; ++q) {
            const bool ok = q < rlen;
            rv[q] = ok ? a.val[rbeg[0] + q] : 0.0;
            rc[q] = ok ? a.col[rbeg[0] + q] : 0;
        }
    } else {
        for (int q = tid; q < nnz; q += kSmallThreads) { lval[q] = a.val[q]; lcol[q] = a.col[q]; }
    }
    // inner products of the incoming state
    double mu = a.dots[(size_t)a.k0 * kPartialStride + 0], dl = a.dots[(size_t)a.k0 * kPartialStride + 1];
    double gm = a.dots[(size_t)a.k0 * kPartialStride + 2], nu = a.dots[(size_t)a.k0 * kPartialStride + 3];
    __syncthreads();

    double2* cur = rsA;
    double2* nxt = rsB;
    for (int it = 1; it <= a.iters; ++it) {
        // coefficients (pipe_pr_cg.py:64-66,75), identical in every thread
        const double al = nu / mu;
        const double a2 = al * al;
        const double nup = a.meurant ? (-nu + a2 * gm) : ((nu - (2 * al) * dl) + a2 * gm);
        const double bt = nup / nu;
        if (tid == 0) {
            double* cf = a.coef + (size_t)(a.k0 + it) * 4;
            cf[0] = al; cf[1] = bt; cf[2] = nup;
        }
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int row = tid + j * kSmallThreads;
            if (row < n) {
                double wr = 0.0, us = 0.0;                       // (A r)_i, (A s)_i, left to right
                if constexpr (MODE == 1) {
                    double2 g[kSmallRegLen];
#pragma unroll
                    for (int q = 0; q < kSmallRegLen; ++q) g[q] = cur[rc[q]];      // all gathers in flight
#pragma unroll
                    for (int q = 0; q < kSmallRegLen; ++q)
                        if (q < rlen) { wr += rv[q] * g[q].x; us += rv[q] * g[q].y; }
                } else {
                    for (int q = rbeg[j]; q < rend[j]; ++q) {
                        const double v = lval[q];
                        const double2 g = cur[lcol[q]];
                        wr += v * g.x; us += v * g.y;
                    }
                }
                const double2 rs = cur[row];
                xr[j] = xr[j] + al * pr[j];                      // x += a p
                const double rn = rs.x - al * rs.y;              // r -= a s
                const double wn = wr - al * us;                  // w -= a u
                const double pn = rn + bt * pr[j];               // p = r + b p
                const double sn = wn + bt * rs.y;                // s = w + b s
                pr[j] = pn;
                nxt[row] = make_double2(rn, sn);
                acc0 += pn * sn; acc1 += rn * sn; acc2 += sn * sn; acc3 += rn * rn;
            }
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2); acc3 = wave_sum(acc3);
        if (lane == 0) { red[wv * 4 + 0] = acc0; red[wv * 4 + 1] = acc1; red[wv * 4 + 2] = acc2; red[wv * 4 + 3] = acc3; }
        lds_barrier();                                           // nxt complete, red complete
        // 16 wave partials -> every wave sums them with the same butterfly (lanes 0..15 hold them)
        mu = wave_sum16(red[(lane & 15) * 4 + 0]);
        dl = wave_sum16(red[(lane & 15) * 4 + 1]);
        gm = wave_sum16(red[(lane & 15) * 4 + 2]);
        nu = wave_sum16(red[(lane & 15) * 4 + 3]);
        if (tid == 0) {
            double* d = a.dots + (size_t)(a.k0 + it) * kPartialStride;
            d[0] = mu; d[1] = dl; d[2] = gm; d[3] = nu; d[4] = nu;
        }
        lds_barrier();                                           // everybody has read red before it is rewritten
        double2* tmp = cur; cur = nxt; nxt = tmp;
    }

    double2* XPo = reinterpret_cast<double2*>(a.xp);
    double2* RSo = reinterpret_cast<double2*>(a.rs);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int row = tid + j * kSmallThreads;
        if (row < n) { XPo[row] = make_double2(xr[j], pr[j]); RSo[row] = cur[row]; }
    }
}

// ---- Hestenes-Stiefel, the whole solve of a SMALL system in one launch of one workgroup (hs_cg.py:54-62) ----
// BASELINE config 1 is bcsstk03 (n = 112) under hs_cg: two launches and two kernel boundaries per iteration cost it 9-10 us each.
// Same storage as k_small_pipe_pr (matrix in registers or LDS, the gather source in LDS: here the direction p, double-buffered);
// x, r, p, s of a thread's rows live in its registers.  Per iteration, with the expressions and roundings of k_hs_update_xr /
// k_hs_update_p / the row sums of csr_matvec:  a = nu / mu;  x += a p;  r -= a s;  nu' = r.r;  b = nu' / nu;  p = r + b p;
// s = A p;  mu = p.s  -- two workgroup-wide sums (the two reductions that make Hestenes-Stiefel what it is), three barriers.
// Scalars as the launches leave them: dots[k] = {mu_k, -, -, nu_k, rr_k}, coef[k] = {a_k, b_k}.
template <int MODE>
__global__ __launch_bounds__(kSmallThreads) void k_small_hs(SmallArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ROWS = MODE == 1 ? 1 : kSmallRows;
    const int n = a.n, nnz = a.nnz;
    double* pA = reinterpret_cast<double*>(smem);
    double* pB = pA + n;
    double* red = pB + n + (n & 1);                              // [16 waves]
    double* lval = red + 16;
    int* lcol = reinterpret_cast<int*>(lval + (MODE == 0 ? nnz : 0));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double xr[ROWS], rr_[ROWS], pr[ROWS], sr[ROWS];
    int rbeg[ROWS], rend[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int row = tid + j * kSmallThreads;
        xr[j] = 0.0; rr_[j] = 0.0; pr[j] = 0.0; sr[j] = 0.0; rbeg[j] = 0; rend[j] = 0;
        if (row < n) {
            xr[j] = a.xp[row]; rr_[j] = a.rs[row]; pr[j] = a.hs_p[row]; sr[j] = a.hs_s[row];
            pA[row] = pr[j];
            rbeg[j] = a.indptr[row]; rend[j] = a.indptr[row + 1];
        }
    }
    double rv[kSmallRegLen];
    int rc[kSmallRegLen];
    int rlen = 0;
    if constexpr (MODE == 1) {
        rlen = rend[0] - rbeg[0];
#pragma unroll
        for (int q = 0; q < kSmallRegLen; ++q) {
            const bool ok = q < rlen;
            rv[q] = ok ? a.val[rbeg[0] + q] : 0.0;
            rc[q] = ok ? a.col[rbeg[0] + q] : 0;
        }
    } else {
        for (int q = tid; q < nnz; q += kSmallThreads) { lval[q] = a.val[q]; lcol[q] = a.col[q]; }
    }
    double mu = a.dots[(size_t)a.k0 * kPartialStride + 0], nu = a.dots[(size_t)a.k0 * kPartialStride + 3];
    __syncthreads();

    double* cur = pA;                                            // holds p_{k-1}
    double* nxt = pB;
    for (int it = 1; it <= a.iters; ++it) {
        const double al = nu / mu;                               // a = nu / mu                       hs_cg.py:55
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            if (tid + j * kSmallThreads < n) {
                xr[j] = xr[j] + al * pr[j];                      // x += a p
                rr_[j] = rr_[j] - al * sr[j];                    // r -= a s
                acc += rr_[j] * rr_[j];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wv] = acc;
        lds_barrier();
        const double nun = wave_sum16(red[lane & 15]);           // nu' = r.r                          :58
        const double bt = nun / nu;                              // b = nu' / nu                       :59
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int row = tid + j * kSmallThreads;
            if (row < n) { pr[j] = rr_[j] + bt * pr[j]; nxt[row] = pr[j]; }     // p = r + b p           :60
        }
        lds_barrier();                                           // the new direction is complete (and red has been read)
        double accm = 0.0;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int row = tid + j * kSmallThreads;
            if (row < n) {
                double sp = 0.0;                                 // (A p)_i, left to right
                if constexpr (MODE == 1) {
                    double g[kSmallRegLen];
#pragma unroll
                    for (int q = 0; q < kSmallRegLen; ++q) g[q] = nxt[rc[q]];
#pragma unroll
                    for (int q = 0; q < kSmallRegLen; ++q)
                        if (q < rlen) sp += rv[q] * g[q];
                } else {
                    for (int q = rbeg[j]; q < rend[j]; ++q) sp += lval[q] * nxt[lcol[q]];
                }
                sr[j] = sp;                                      // s = A p                            :61
                accm += pr[j] * sp;
            }
        }
        accm = wave_sum(accm);
        if (lane == 0) red[wv] = accm;
        lds_barrier();
        mu = wave_sum16(red[lane & 15]);                         // mu = p.s                           :62
        if (tid == 0) {
            double* cf = a.coef + (size_t)(a.k0 + it) * 4;
            cf[0] = al; cf[1] = bt;
            double* d = a.dots + (size_t)(a.k0 + it) * kPartialStride;
            d[0] = mu; d[3] = nun; d[4] = nun;
        }
        nu = nun;
        lds_barrier();                                           // everybody has read red before it is rewritten
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int row = tid + j * kSmallThreads;
        if (row < n) { a.xp[row] = xr[j]; a.rs[row] = rr_[j]; a.hs_p[row] = pr[j]; a.hs_s[row] = sr[j]; }
    }
}

// ---- fixed-order final reduction of per-block partials --------------------------------
constexpr int kFinalThreads = 256;   // same tree as the fused last-block reduction
__global__ __launch_bounds__(kFinalThreads) void k_reduce_final(
    const double* __restrict__ partials, int nparts, double* __restrict__ out,
    int src_first, int dst_first, int count)
{
    __shared__ double red[kFinalThreads / 64][kPartialStride];
    double acc[kPartialStride];
#pragma unroll
    for (int q = 0; q < kPartialStride; ++q) acc[q] = 0.0;
    for (int j = threadIdx.x; j < nparts; j += kFinalThreads) {
#pragma unroll
        for (int q = 0; q < kPartialStride; ++q)
            if (q < count) acc[q] += partials[(size_t)j * kPartialStride + src_first + q];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < kPartialStride; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < count) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < kFinalThreads / 64; ++w) v += red[w][threadIdx.x];
        out[dst_first + threadIdx.x] = v;
    }
}

// ---- merged exchange (small halos) -----------------------------------------------------
// One all-gather per iteration carries both the rank's five partial inner products and the
// (r,s) rows its neighbours need.  Slot layout per rank: 8 doubles (sums 0..4, 3 pad), then
// the packed rows as pairs.
// pack: the fixed-order reduction of k_reduce_final, written into the slot header, plus the rows.
__global__ __launch_bounds__(kFinalThreads) void k_gather_pack(
    const double* __restrict__ partials, int nparts, double* __restrict__ slot,
    const double2* __restrict__ rs, const int* __restrict__ send_idx, int nsend)
{
    __shared__ double red[kFinalThreads / 64][kPartialStride];
    double acc[kPartialStride];
#pragma unroll
    for (int q = 0; q < kPartialStride; ++q) acc[q] = 0.0;
    for (int j = threadIdx.x; j < nparts; j += kFinalThreads) {
#pragma unroll
        for (int q = 0; q < kPartialStride; ++q)
            if (q < 5) acc[q] += partials[(size_t)j * kPartialStride + q];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < kPartialStride; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < 5) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < kFinalThreads / 64; ++w) v += red[w][threadIdx.x];
        slot[threadIdx.x] = v;
    }
    double2* rows = reinterpret_cast<double2*>(slot + 8);
    for (int j = threadIdx.x; j < nsend; j += kFinalThreads) rows[j] = rs[send_idx[j]];
}

// unpack: global sums = the ranks' partial sums added in rank order (the same bits on every
// rank, whatever algorithm RCCL picked for the transport); ghost rows copied into place.
// Publication for waves that WAIT inside a running launch (cdna_hip_programming.md Guideline 16, the
// all-write-through form): payload and counter are stored with agent-scope atomic stores (sc1: they
// leave the XCD's L2), the storing wave drains them before the counter goes out, and the readers
// (k_win_tiles, deferred form) use agent-scope atomic loads for the counter AND the payload.
// The record is REPLICATED kPubCopies times, one copy per 64-byte line: thousands of waves poll, and they
// spread over the copies (one hot word would saturate its L2 channel and delay the very store they wait for).
__device__ __forceinline__ void publish(double* pub, double v, int nvals, unsigned value) {
    // called by ONE wave; lane q < nvals holds value q; lane c writes copy c
    const int lane = threadIdx.x & 63;
    double* mine = pub + (size_t)lane * 8;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double vq = __shfl(v, q, 64);
        if (q < nvals) __hip_atomic_store(mine + q, vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every lane's payload has left before its counter does
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 6), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kFinalThreads) void k_gather_unpack(
    const double* __restrict__ gbuf, int slot_doubles, int nranks, double* __restrict__ dots_out,
    double2* __restrict__ rs_ghost, const int* __restrict__ ghost_src, int nghost, double* pub, unsigned pub_value)
{
    double v = 0.0;
    if ((int)threadIdx.x < 5) {
        v = gbuf[threadIdx.x];
        for (int r = 1; r < nranks; ++r) v += gbuf[(size_t)r * slot_doubles + threadIdx.x];
        dots_out[threadIdx.x] = v;
    }
    const double2* g2 = reinterpret_cast<const double2*>(gbuf);
    for (int j = threadIdx.x; j < nghost; j += kFinalThreads) rs_ghost[j] = g2[ghost_src[j]];
    if (pub) {
        // the ghost rows are part of what the waiting launch reads after the publication (its boundary tiles):
        // plain stores of every wave drained -> barrier -> agent-scope release -> publication
        // (cdna_hip_programming.md Guideline 16: producer side of the release / acquire form)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish(pub, v, 5, pub_value);
        }
    }
}

// ---- direct peer exchange outside the iteration launches (PeerDev, prcg_kernels.h) ----
__global__ __launch_bounds__(256) void k_peer_push(const PeerDev* __restrict__ px, const double2* __restrict__ rs,
                                                   const double* __restrict__ dots, int k, int contribute) {
    const int R = px->nranks;
    const long long gout = peer_ghost_off(R, px->ghost_cap, k & 1);
    for (int e = threadIdx.x; e < px->n_send; e += 256) {
        const int4 ent = px->send_ent[e];
        const double2 v = rs[ent.x];
        double* dst = px->peer[ent.y] + gout + 2 * (long long)ent.z;
        peer_store(dst, v.x);
        peer_store(dst + 1, v.y);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        // rank 0 contributes the (global) inner products of state k, the others zeros: added in rank order that is dots
        const double v = (threadIdx.x < 5 && contribute) ? dots[threadIdx.x] : 0.0;
        peer_send_slot(px, k, v);
    }
}
__global__ __launch_bounds__(256) void k_peer_collect(const PeerDev* __restrict__ px, int k, const double* __restrict__ partials, int nparts,
                                                      double* __restrict__ dots_out, double* pub, unsigned* err) {
    __shared__ double s_mine[5];
    if (nparts > 0) {      // the same 256-thread tree as workgroup 0 of a following launch would use
        double mine[5];
        sum_prev_partials<5, 4>(partials, nparts, 0, mine);
        if (threadIdx.x < 5) s_mine[threadIdx.x] = mine[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double tot[5];
    if (nparts > 0) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) v = lane == q ? s_mine[q] : v;
        peer_send_slot(px, k, v);
    }
    const bool ok = peer_collect(px, k, 1u << 24, tot);
    if (!ok && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) v = lane == q ? tot[q] : v;
    if (lane < 5 && dots_out) dots_out[lane] = v;
    if (pub) publish(pub, v, 5, (unsigned)k);
}

__global__ __launch_bounds__(64) void k_publish(const double* __restrict__ dots, double* pub, unsigned value) {
    const double v = threadIdx.x < 5 ? dots[threadIdx.x] : 0.0;
    publish(pub, v, 5, value);
}

// ---- utilities -------------------------------------------------------------------------
__global__ void k_copy(double* dst, int ds, const double* src, int ss, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i * ds] = src[i * ss];
}
__global__ void k_sub(double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i * ds] = a[i * as] - b[i * bs];
}
__global__ void k_mul(double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i * ds] = a[i * as] * b[i * bs];
}
__global__ __launch_bounds__(kBlock) void k_diff_sq(const double* __restrict__ a, int as, const double* __restrict__ b,
                                                    int64_t n, double* partials, int slot, int trips) {
    double acc[1] = {0.0};
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t ie = i + e;
            if (ie >= n) break;
            const double dlt = a[ie * as] - b[ie];
            acc[0] += dlt * dlt;
        }
    }
    block_reduce_store<1>(acc, partials, slot);
}
__global__ __launch_bounds__(kBlock) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                double* partials, int slot, int trips) {
    double acc[1] = {0.0};
    int64_t i = ((int64_t)blockIdx.x * trips) * kElemsPerTrip + threadIdx.x * 2;
    for (int j = 0; j < trips; ++j, i += kElemsPerTrip) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t ie = i + e;
            if (ie >= n) break;
            acc[0] += a[ie] * b[ie];
        }
    }
    block_reduce_store<1>(acc, partials, slot);
}
__global__ void k_pack(double* buf, const double* v, const int* idx, int64_t count, int nc) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = idx[j];
        for (int c = 0; c < nc; ++c) buf[j * nc + c] = v[src * nc + c];
    }
}

struct Chunking { int grid; int trips; };
Chunking chunking(int64_t n) {
    const int64_t total = (n + kElemsPerTrip - 1) / kElemsPerTrip;
    int64_t grid = total < kMaxGridBlocks ? total : kMaxGridBlocks;
    if (grid < 1) grid = 1;
    const int64_t trips = (total + grid - 1) / grid;
    grid = trips > 0 ? (total + trips - 1) / trips : 1;
    if (grid < 1) grid = 1;
    return {(int)grid, (int)(trips > 0 ? trips : 1)};
}

int util_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// Persistent grid for the tile kernels = blocks that are truly co-resident.  The strided
// tile assignment assumes every block runs from the start; a block that has to queue
// behind the others turns into a serial tail (measured: 32 KiB blocks at the occupancy
// API's 5 per CU run 17 % slower than at 4 per CU -- only 4 are resident).  The LDS of a
// CU is handed out per half (80 KiB each), so residency = 2 * floor(80 KiB / block LDS),
// further capped by the API and by 8 blocks (32 waves) per CU.
template <int TAG, typename K>
int tile_grid(K kernel, int ntiles, int per_cu_override) {
    static int caps[16] = {};   // per TAG (the kernels share one function-pointer type) and device
    int devid = 0;
    (void)hipGetDevice(&devid);
    int& cap = caps[devid & 15];
    if (cap == 0) {
        int dev = 0, cus = 256, occ = 4;
        if (hipGetDevice(&dev) == hipSuccess) {
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kBlock, 0) != hipSuccess || occ < 1) occ = 4;
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.sharedSizeBytes > 0) {
                const int by_lds = 2 * (int)((80 * 1024) / fa.sharedSizeBytes);
                if (by_lds >= 1 && by_lds < occ) occ = by_lds;
            }
            if (occ > 8) occ = 8;
            // Measured on MI355X (profiles/r01_sweeps.md): beyond ~12 resident waves per CU
            // of this two-deep pipeline the extra outstanding streams cost more HBM
            // efficiency than the latency hiding they add (S3: 3 blocks/CU 5.2 TB/s,
            // 4 blocks/CU 4.6 TB/s at 512-slot tiles; 256-slot tiles, which are chosen for
            // short-row operators, likewise: S2 3/CU 3236 it/s vs 4/CU 3065, S1 35.2 k vs 34.1 k).
            // TAG = NV*1000 + EPI*100 + STEPS*10 + (column bytes on the stream: 0 = int32, 1, 2);
            // with the narrower column stream one more resident block pays (S3: 2394 vs 2317 it/s)
            const int steps_ = (TAG / 10) % 10, narrow_ = TAG % 10;
            // (codes 4..6: value dictionary beside int32 / 8-bit / 16-bit columns -- lighter stream, fewer
            //  registers: 4 blocks also at 256-slot tiles, S2 4103 vs 3666 it/s, S1 36.6 k vs 35.4 k)
            const int tuned = steps_ == 4 ? 2 : (((steps_ == 2 && narrow_ != 0) || narrow_ >= 4) ? 4 : 3);
            if (occ > tuned) occ = tuned;
        }
        cap = occ * cus;
    }
    int g = (ntiles + kWaves - 1) / kWaves;
    int lim = cap;
    if (per_cu_override >= 1 && per_cu_override <= 16) {      // experiment knob PRCG_GRID_PER_CU (read into the handle)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid);
        lim = per_cu_override * cus;
    }
    if (g > lim) g = lim;
    if (g < 1) g = 1;
    return g;
}

}  // namespace

// ---- launch wrappers -------------------------------------------------------------------
#define PRCG_LAUNCH_OK() (hipGetLastError() == hipSuccess)

template <int NV, int EPI, int STEPS>
int launch_tiles(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, const void* x, void* y,
                 int write_mask, const double* ep_r, const double* ep_d, double* ep_st, double* partials,
                 TileKnobs kn, double* aux = nullptr, FusedPrev fz = FusedPrev{})
{
    const int cw = A.tile_base == nullptr ? 0 : (A.col8 ? 8 : (A.col16 ? 16 : 0));
    const bool vd = A.vidx8 != nullptr;   // value dictionary
    auto k = cw == 8 ? (vd ? k_spmv_tiles<NV, EPI, STEPS, 8, true> : k_spmv_tiles<NV, EPI, STEPS, 8, false>)
                     : (cw == 16 ? (vd ? k_spmv_tiles<NV, EPI, STEPS, 16, true> : k_spmv_tiles<NV, EPI, STEPS, 16, false>)
                                 : (vd ? k_spmv_tiles<NV, EPI, STEPS, 0, true> : k_spmv_tiles<NV, EPI, STEPS, 0, false>));
    int grid;
    if (cw == 8) grid = vd ? tile_grid<NV * 1000 + EPI * 100 + STEPS * 10 + 5>(k, ntiles, kn.per_cu) : tile_grid<NV * 1000 + EPI * 100 + STEPS * 10 + 1>(k, ntiles, kn.per_cu);
    else if (cw == 16) grid = vd ? tile_grid<NV * 1000 + EPI * 100 + STEPS * 10 + 6>(k, ntiles, kn.per_cu) : tile_grid<NV * 1000 + EPI * 100 + STEPS * 10 + 2>(k, ntiles, kn.per_cu);
    else grid = vd ? tile_grid<NV * 1000 + EPI * 100 + STEPS * 10 + 4>(k, ntiles, kn.per_cu) : tile_grid<NV * 1000 + EPI * 100 + STEPS * 10>(k, ntiles, kn.per_cu);
    const int chunked = kn.chunked;   // experiment knob PRCG_TILE_ORDER=chunk (measured slower: 4.2-4.7 vs 4.6-4.9 TB/s)
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), 0, st, A, tiles, A.tile_base, A.vd, ntiles, x, y, write_mask, ep_r, ep_d, ep_st,
                       partials, chunked, aux, fz);
    return PRCG_LAUNCH_OK() ? grid : -1;
}

template <int NV, int EPI>
int launch_tiles_steps(int steps, hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, const void* x,
                       void* y, int write_mask, const double* ep_r, const double* ep_d, double* ep_st,
                       double* partials, TileKnobs kn, double* aux = nullptr, FusedPrev fz = FusedPrev{})
{
    switch (steps) {
    case 1: return launch_tiles<NV, EPI, 1>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, kn, aux, fz);
    case 2: return launch_tiles<NV, EPI, 2>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, kn, aux, fz);
    case 4: return launch_tiles<NV, EPI, 4>(st, A, tiles, ntiles, x, y, write_mask, ep_r, ep_d, ep_st, partials, kn, aux, fz);
    default: return -1;
    }
}

int launch_spmv(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps,
                const double* x, double* y, SpmvEpilogue epi,
                const double* ep_r, const double* ep_d, double* ep_st, double* partials, TileKnobs kn)
{
    if (ntiles <= 0) return 0;
    switch (epi) {
    case kEpiNone: return launch_tiles_steps<1, kEpiNone>(steps, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, kn);
    case kEpiDotXY: return launch_tiles_steps<1, kEpiDotXY>(steps, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, kn);
    case kEpiPR: return launch_tiles_steps<1, kEpiPR>(steps, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, kn);
    case kEpiCG: return launch_tiles_steps<1, kEpiCG>(steps, st, A, tiles, ntiles, x, y, 3, ep_r, ep_d, ep_st, partials, kn);
    default: break;
    }
    return -1;
}

int launch_spmm2(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps,
                 const double* rs, double* wu, int write_mask, TileKnobs kn)
{
    if (ntiles <= 0) return 0;
    return launch_tiles_steps<2, kEpiNone>(steps, st, A, tiles, ntiles, rs, wu, write_mask, nullptr, nullptr, nullptr,
                                           nullptr, kn);
}

int launch_pipe_fused(hipStream_t st, const CsrDev& A, const Tile* tiles, int ntiles, int steps, const FusedState& f, TileKnobs kn)
{
    if (ntiles <= 0) return 0;
    FusedPrev fz = f.prev;
    fz.rs = f.rs; fz.w = f.w; fz.wt = f.wt;
    const int mask = 3 | (f.meurant ? 4 : 0) | (f.stream_stores ? 8 : 0);
    if (f.dinv) {
        if (f.recompute_w)
            return launch_tiles_steps<2, kEpiPipeFusedJ>(steps, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv,
                                                         f.in_new, f.partials, kn, f.coef_out, fz);
        return launch_tiles_steps<2, kEpiPipeFusedPJ>(steps, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, f.dinv,
                                                      f.in_new, f.partials, kn, f.coef_out, fz);
    }
    if (f.recompute_w)
        return launch_tiles_steps<2, kEpiPipeFused>(steps, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr,
                                                    f.in_new, f.partials, kn, f.coef_out, fz);
    return launch_tiles_steps<2, kEpiPipeFusedP>(steps, st, A, tiles, ntiles, f.in_old, f.xp, mask, f.dots_prev, nullptr,
                                                 f.in_new, f.partials, kn, f.coef_out, fz);
}

size_t small_lds_bytes(int n, int nnz, int mode) {
    return (size_t)2 * n * 16 + 64 * 8 + (mode == 0 ? (size_t)nnz * 12 + 16 : 0);
}
// mode 1 (matrix in registers) if n <= 1024 and the longest row has <= 8 nonzeros; mode 0
// (matrix in LDS) if it fits beside the vectors; otherwise the multi-launch schedule is
// faster (measured: bcsstk14, 63 k nonzeros re-read through L2 by one CU: 58 us/iteration).
bool small_fits(int64_t n, int64_t nnz, int max_row_len, int* mode) {
    if (n < 1 || n > kSmallMaxN) return false;
    if (n <= kSmallThreads && max_row_len <= kSmallRegLen) { *mode = 1; return true; }
    const size_t cap = 156 * 1024;
    if (nnz < (1 << 20) && small_lds_bytes((int)n, (int)nnz, 0) <= cap && max_row_len <= 64) { *mode = 0; return true; }
    return false;
}
// (xp = x, rs = r, hs_p = p, hs_s = s: separate arrays, as the Hestenes-Stiefel sessions hold them)
int launch_small_hs(hipStream_t st, const SmallArgs& a, int mode) {
    const size_t lds = small_lds_bytes(a.n, a.nnz, mode);        // (sized for the pipelined solver's pairs: more than the directions need)
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_hs<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_hs<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (mode == 1) hipLaunchKernelGGL(k_small_hs<1>, dim3(1), dim3(kSmallThreads), lds, st, a);
    else hipLaunchKernelGGL(k_small_hs<0>, dim3(1), dim3(kSmallThreads), lds, st, a);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
int launch_small_pipe_pr(hipStream_t st, const SmallArgs& a, int mode) {
    const size_t lds = small_lds_bytes(a.n, a.nnz, mode);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_pipe_pr<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_pipe_pr<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (mode == 1) hipLaunchKernelGGL(k_small_pipe_pr<1>, dim3(1), dim3(kSmallThreads), lds, st, a);
    else hipLaunchKernelGGL(k_small_pipe_pr<0>, dim3(1), dim3(kSmallThreads), lds, st, a);
    return PRCG_LAUNCH_OK() ? 1 : -1;
}

int launch_pipe_update(hipStream_t st, const PipeUpdateArgs& a) {
    const Chunking c = chunking(a.n);
    // preconditioned: the inverse diagonal, or u~ / w~ as vectors (host-callback preconditioner)
    if (a.d || a.ut) hipLaunchKernelGGL((k_pipe_update<true, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    else             hipLaunchKernelGGL((k_pipe_update<false, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_pipe_dots(hipStream_t st, const PipeUpdateArgs& a) {
    const Chunking c = chunking(a.n);
    if (a.d || a.ut) hipLaunchKernelGGL((k_pipe_update<true, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    else             hipLaunchKernelGGL((k_pipe_update<false, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}

int launch_hs_update_xr(hipStream_t st, const HsArgs& a, const double* prev_mu, int nprev, double* dots_prev_w) {
    const Chunking c = chunking(a.n);
    if (a.d) hipLaunchKernelGGL((k_hs_update_xr<true, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, prev_mu, nprev, dots_prev_w);
    else     hipLaunchKernelGGL((k_hs_update_xr<false, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, prev_mu, nprev, dots_prev_w);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_hs_init_dots(hipStream_t st, const HsArgs& a) {
    const Chunking c = chunking(a.n);
    const double* none = nullptr;
    double* nonew = nullptr;
    if (a.d) hipLaunchKernelGGL((k_hs_update_xr<true, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, none, 0, nonew);
    else     hipLaunchKernelGGL((k_hs_update_xr<false, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, none, 0, nonew);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_hs_update_p(hipStream_t st, const HsArgs& a, const double* prev_nu, int nprev, double* dots_cur_w) {
    const Chunking c = chunking(a.n);
    hipLaunchKernelGGL(k_hs_update_p, dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, prev_nu, nprev, dots_cur_w);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}

int launch_pr_update(hipStream_t st, const PrArgs& a) {
    const Chunking c = chunking(a.n);
    if (a.precond) hipLaunchKernelGGL((k_pr_update<true, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    else           hipLaunchKernelGGL((k_pr_update<false, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_pr_init_dots(hipStream_t st, const PrArgs& a) {
    const Chunking c = chunking(a.n);
    if (a.precond) hipLaunchKernelGGL((k_pr_update<true, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    else           hipLaunchKernelGGL((k_pr_update<false, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}

int launch_cg_update_ps(hipStream_t st, const CgArgs& a, const double* prev, int nprev) {
    const Chunking c = chunking(a.n);
    hipLaunchKernelGGL(k_cg_update_ps, dim3(c.grid), dim3(kBlock), 0, st, a, c.trips, prev, nprev);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_gv_update1(hipStream_t st, const CgArgs& a, bool dots_only) {
    const Chunking c = chunking(a.n);
    if (a.rt) {      // preconditioned (a.d null: w~ = M^-1 w is applied by the caller afterwards)
        if (dots_only) hipLaunchKernelGGL((k_gv_update1<true, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
        else hipLaunchKernelGGL((k_gv_update1<true, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    } else {
        if (dots_only) hipLaunchKernelGGL((k_gv_update1<false, true>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
        else hipLaunchKernelGGL((k_gv_update1<false, false>), dim3(c.grid), dim3(kBlock), 0, st, a, c.trips);
    }
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}

void launch_reduce_final(hipStream_t st, const double* partials, int nparts, double* out,
                         int src_first, int dst_first, int count) {
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(kFinalThreads), 0, st, partials, nparts, out, src_first, dst_first, count);
}

void launch_copy(hipStream_t st, double* dst, int ds, const double* src, int ss, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_copy, dim3(util_grid(n)), dim3(256), 0, st, dst, ds, src, ss, n);
}
void launch_sub(hipStream_t st, double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sub, dim3(util_grid(n)), dim3(256), 0, st, dst, ds, a, as, b, bs, n);
}
void launch_mul(hipStream_t st, double* dst, int ds, const double* a, int as, const double* b, int bs, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_mul, dim3(util_grid(n)), dim3(256), 0, st, dst, ds, a, as, b, bs, n);
}
int launch_diff_sq(hipStream_t st, const double* a, int as, const double* b, int64_t n, double* partials, int slot) {
    const Chunking c = chunking(n);
    hipLaunchKernelGGL(k_diff_sq, dim3(c.grid), dim3(kBlock), 0, st, a, as, b, n, partials, slot, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
int launch_dot(hipStream_t st, const double* a, const double* b, int64_t n, double* partials, int slot) {
    const Chunking c = chunking(n);
    hipLaunchKernelGGL(k_dot, dim3(c.grid), dim3(kBlock), 0, st, a, b, n, partials, slot, c.trips);
    return PRCG_LAUNCH_OK() ? c.grid : -1;
}
void launch_gather_pack(hipStream_t st, const double* partials, int nparts, double* slot, const double* rs,
                        const int* send_idx, int nsend) {
    hipLaunchKernelGGL(k_gather_pack, dim3(1), dim3(kFinalThreads), 0, st, partials, nparts, slot,
                       reinterpret_cast<const double2*>(rs), send_idx, nsend);
}
void launch_gather_unpack(hipStream_t st, const double* gbuf, int slot_doubles, int nranks, double* dots_out,
                          double* rs_ghost, const int* ghost_src, int nghost, double* pub, unsigned pub_value, hipEvent_t done) {
    if (done)      // the launch's own completion signal (no marker packet behind it)
        hipExtLaunchKernelGGL(k_gather_unpack, dim3(1), dim3(kFinalThreads), 0, st, nullptr, done, 0, gbuf, slot_doubles, nranks,
                              dots_out, reinterpret_cast<double2*>(rs_ghost), ghost_src, nghost, pub, pub_value);
    else
        hipLaunchKernelGGL(k_gather_unpack, dim3(1), dim3(kFinalThreads), 0, st, gbuf, slot_doubles, nranks, dots_out,
                           reinterpret_cast<double2*>(rs_ghost), ghost_src, nghost, pub, pub_value);
}
// One wave that waits (bounded, ~2 ms) for copy 0 of a publication record to reach `want`: the probe
// prcg_solve_begin uses to find out whether a kernel of the communication stream can run WHILE a kernel of
// the compute stream waits for it (two HIP streams may share one hardware queue, which is in order).
__global__ __launch_bounds__(64) void k_probe_wait(const double* pub, unsigned want, unsigned* err) {
    const unsigned* cnt = reinterpret_cast<const unsigned*>(pub + 6);
    unsigned spins = 0;
    bool ok = false;
    while (!(ok = (int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) && ++spins < 4096u)
        __builtin_amdgcn_s_sleep(16);
    if (!ok && threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- what the memory system delivers for the iteration's own byte mix (bench.py: roofline.stream_ceiling_GBps) ----
// mode 0: pure 16-byte-per-lane read (8 loads in flight per thread); mode 1 / 2: per row one pair read and rewritten in
// place, one pair read from one array and written to another -- 2 x 16 B in, 2 x 16 B out, the vector traffic of the
// one-launch pipelined iteration with nothing else -- with plain (1) or nontemporal (2) stores.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_read(const double2* __restrict__ a, size_t n2, double* out) {
    double s = 0.0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;                                           // (keeps the loads alive)
}
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_stream_pairs(double2* __restrict__ xp, const double2* __restrict__ rs, double2* __restrict__ rsn, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* X = reinterpret_cast<d2*>(xp);
    const d2* R = reinterpret_cast<const d2*>(rs);
    d2* Rn = reinterpret_cast<d2*>(rsn);
    for (; i < n2; i += UNROLL * stride) {
        d2 x[UNROLL], r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * stride < n2) { x[u] = X[i + u * stride]; r[u] = R[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * stride < n2) {
            const d2 xo = x[u], ro = r[u];
            const d2 xn = {xo.x + 0.5 * xo.y, ro.x + 0.25 * xo.y}, rn = {ro.x - 0.5 * ro.y, xo.y + 0.25 * ro.y};
            if (NT) { __builtin_nontemporal_store(xn, X + i + u * stride); __builtin_nontemporal_store(rn, Rn + i + u * stride); }
            else { X[i + u * stride] = xn; Rn[i + u * stride] = rn; }
        }
    }
}
// pure read as a streaming kernel should issue it (tools/readpat.hip: 6.5-7.1 TB/s where the grid-stride form above with eight
// 1 KB pieces per wave in flight and 16+ waves per CU reads 5.1-5.5): every wave walks contiguous 4 KB chunks, four nontemporal
// 16-byte loads in flight, eight waves per CU
__global__ __launch_bounds__(256) void k_stream_read_chunks(const double2* __restrict__ a_, size_t n2, double* out) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* a = reinterpret_cast<const d2*>(a_);
    const int lane = threadIdx.x & 63;
    const size_t W = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t chunks = n2 / 256;                                          // 256 pairs = 4 KB
    double s = 0.0;
    for (size_t c = w; c < chunks; c += W) {
        const d2* base = a + c * 256 + lane;
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(base + u * 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y;
    }
    if (s == 123.456) out[0] = s;
}
// the byte MIX of a one-launch iteration with no arithmetic to speak of (tools/mixbench.hip): per piece of 64 rows a wave reads
// `kb` KB of an operator stream (nontemporal, read once: 16 one-KB loads in flight, the next 16 requested before these are used),
// reads the rows' two pairs and writes one in place and one to another array with nontemporal stores.  What the memory system
// delivers for THIS mix is the ceiling of a kernel that moves it: 0.32 GB of row results beside a 1.3 GB stream cost 55-90 us
// (read / write turnarounds), beside a 3 GB stream 76 us -- in the mix itself, not in the kernel.
__global__ __launch_bounds__(256) void k_stream_mix(const double2* __restrict__ V_, double2* __restrict__ X_, const double2* __restrict__ R_,
                                                    double2* __restrict__ Rn_, size_t n_rows, int kb) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* V = reinterpret_cast<const d2*>(V_);
    d2* X = reinterpret_cast<d2*>(X_);
    const d2* R = reinterpret_cast<const d2*>(R_);
    d2* Rn = reinterpret_cast<d2*>(Rn_);
    const int lane = threadIdx.x & 63;
    const size_t W = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t pieces = n_rows / 64;
    constexpr int CH = 16;
    const int nch = (kb + CH - 1) / CH;
    for (size_t p = w; p < pieces; p += W) {
        const d2* base = V + p * (size_t)kb * 64 + lane;
        const d2 x = X[p * 64 + lane], r = R[p * 64 + lane];
        d2 a[CH], b[CH];
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < CH; ++u) a[u] = __builtin_nontemporal_load(base + (size_t)(u < kb ? u : 0) * 64);
        for (int c = 1; c <= nch; ++c) {
            if (c < nch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) { const int q = c * CH + u; b[u] = __builtin_nontemporal_load(base + (size_t)(q < kb ? q : 0) * 64); }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) s += a[u].x + a[u].y;
#pragma unroll
            for (int u = 0; u < CH; ++u) a[u] = b[u];
        }
        const d2 xn = {x.x + 0.5 * x.y + s, r.x + 0.25 * x.y}, rn = {r.x - 0.5 * r.y, x.y + 0.25 * r.y};
        __builtin_nontemporal_store(xn, X + p * 64 + lane);
        __builtin_nontemporal_store(rn, Rn + p * 64 + lane);
    }
}
void launch_stream_mix(hipStream_t st, const double* v, double* x, const double* r, double* rn, size_t n_rows, int kb) {
    hipLaunchKernelGGL(k_stream_mix, dim3(512), dim3(256), 0, st, reinterpret_cast<const double2*>(v), reinterpret_cast<double2*>(x),
                       reinterpret_cast<const double2*>(r), reinterpret_cast<double2*>(rn), n_rows, kb);
}
void launch_stream_probe(hipStream_t st, int mode, double* a, double* b, double* c, size_t n_pairs) {
    const dim3 grid(kMaxGridBlocks), block(256);
    if (mode == 3) { hipLaunchKernelGGL(k_stream_read_chunks, dim3(512), block, 0, st, reinterpret_cast<const double2*>(a), n_pairs, c); return; }
    if (mode == 0) hipLaunchKernelGGL(k_stream_read<8>, grid, block, 0, st, reinterpret_cast<const double2*>(a), n_pairs, c);
    else if (mode == 1) hipLaunchKernelGGL((k_stream_pairs<4, false>), grid, block, 0, st, reinterpret_cast<double2*>(a), reinterpret_cast<const double2*>(b), reinterpret_cast<double2*>(c), n_pairs);
    else hipLaunchKernelGGL((k_stream_pairs<4, true>), grid, block, 0, st, reinterpret_cast<double2*>(a), reinterpret_cast<const double2*>(b), reinterpret_cast<double2*>(c), n_pairs);
}

void launch_probe_wait(hipStream_t st, const double* pub, unsigned want, unsigned* err) {
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, st, pub, want, err);
}

// bit 30 of a window tile's `geo` word: some of its rows go to neighbours (the iteration launch then looks at tile_send)
__global__ void k_flag_send_tiles(int* __restrict__ wtiles, const int2* __restrict__ tile_send, int ntiles) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
        const int2 ts = tile_send[t];
        int& geo = wtiles[(size_t)t * 24 + 4];
        geo = ts.y > ts.x ? (geo | (1 << 30)) : (geo & ~(1 << 30));
    }
}
void launch_flag_send_tiles(hipStream_t st, void* wtiles, const void* tile_send, int ntiles) {
    if (ntiles > 0) hipLaunchKernelGGL(k_flag_send_tiles, dim3((ntiles + 255) / 256), dim3(256), 0, st, static_cast<int*>(wtiles), static_cast<const int2*>(tile_send), ntiles);
}
void launch_peer_push(hipStream_t st, const PeerDev* px, const double* rs, const double* dots, int k, int contribute) {
    hipLaunchKernelGGL(k_peer_push, dim3(1), dim3(256), 0, st, px, reinterpret_cast<const double2*>(rs), dots, k, contribute);
}
void launch_peer_collect(hipStream_t st, const PeerDev* px, int k, const double* partials, int nparts, double* dots_out, double* pub,
                         unsigned* err) {
    hipLaunchKernelGGL(k_peer_collect, dim3(1), dim3(256), 0, st, px, k, partials, nparts, dots_out, pub, err);
}
void launch_publish(hipStream_t st, const double* dots, double* pub, unsigned value) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, dots, pub, value);
}
void launch_pack(hipStream_t st, double* buf, const double* v, const int* idx, int64_t count, int nc) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_pack, dim3(util_grid(count)), dim3(256), 0, st, buf, v, idx, count, nc);
}

}  // namespace prcg
