// gfx950 (MI355X / CDNA4): the whole pipelined predict-and-recompute solve of a MID-SIZE system in ONE launch of a FEW
// co-operating workgroups (numerical_experiments/cg_variants/pipe_pr_cg.py:61-75; the matrices of figure_gen.py:245-339
// that are too large for the one-workgroup solver -- bcsstk14 ... bcsstk18, bcsstm25 -- and too small to fill the chip:
// one launch per iteration costs them 8-10 us each, nearly all of it launch boundary and prologue).
//
//   * G <= 32 workgroups of 16 waves, one per CU, all resident; wave w owns up to kMedSlices SLICES of 64 rows (the sliced
//     layout of prcg_plan.h: plan_sell -- lane per row, values and 16-bit column-delta codes transposed so that the wave's
//     loads are coalesced); x and p of a row live in the lane's registers for the whole solve;
//   * the (r,s) pairs of ALL rows live in a double-buffered EXCHANGE array in global memory: iteration k gathers from
//     buffer k & 1 and writes the rows' new pairs to the other one.  A workgroup first stages the WINDOW of columns its rows
//     touch (one contiguous range: these matrices are banded) in LDS with coalesced loads, then every lane walks its row
//     with LDS gathers -- the same left-to-right sum as scipy's csr_matvec, bit for bit;
//   * ONE all-to-all per iteration: a workgroup's four partial inner products and the iteration number go to its 64-byte
//     SLOT; every workgroup polls all G slots and adds them in slot order (the same bits everywhere).  Seeing all G tags of
//     iteration k also means every workgroup's rows of iteration k are stored and nobody still reads buffer k & 1.
//     Hand-off (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "hand-offs measured with sc1 loads", first row):
//     every exchanged byte is stored with sc1 (agent scope), every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup
//     barrier behind which ONE lane stores the tag; the consumer polls the tags with sc1 loads, joins a workgroup
//     barrier, and every load of exchanged bytes is an sc1 load.  Every spin is bounded.
#include <hip/hip_runtime.h>

#include "prcg_device.hpp"
#include "prcg_kernels.h"

namespace prcg {
namespace {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
constexpr int kMedThreads = 1024, kMedWaves = 16;
constexpr int kScope = 16;        // sc1: agent scope (one GPU) -- the coherence point is the fabric side of the L2s, not memory

__device__ __forceinline__ double2 ld_pair_sc(const double* base, long long idx) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7ffffff0, 0x00020000);
    const u4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx * 16), 0, kScope);
    return make_double2(__hiloint2double((int)raw.y, (int)raw.x), __hiloint2double((int)raw.w, (int)raw.z));
}
__device__ __forceinline__ void st_pair_sc(double* base, long long idx, double2 v) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
    u4_t raw;
    raw.x = (unsigned)__double2loint(v.x); raw.y = (unsigned)__double2hiint(v.x);
    raw.z = (unsigned)__double2loint(v.y); raw.w = (unsigned)__double2hiint(v.y);
    __builtin_amdgcn_raw_buffer_store_b128(raw, rsrc, (int)(idx * 16), 0, kScope);
}
__device__ __forceinline__ double ld_sc(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct MDesc { int rb, re, voff, coff, width, cbase, rows_off; };
__device__ __forceinline__ MDesc read_mdesc(const int4* __restrict__ st, int t) {
    const int4 a = st[2 * t], b = st[2 * t + 1];
    MDesc d;
    d.rb = __builtin_amdgcn_readfirstlane(a.x); d.re = __builtin_amdgcn_readfirstlane(a.y);
    d.voff = __builtin_amdgcn_readfirstlane(a.z); d.coff = __builtin_amdgcn_readfirstlane(a.w);
    d.width = __builtin_amdgcn_readfirstlane(b.x); d.cbase = __builtin_amdgcn_readfirstlane(b.y);
    d.rows_off = __builtin_amdgcn_readfirstlane(b.z);
    return d;
}

// One wave: wait (bounded) until the tag word (double 7) of every one of the G lines at `lines` equals `tag`.  false on a timeout.
__device__ __forceinline__ bool wait_tags(const double* lines, int G, unsigned long long tag) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
    if (lane < G) {
        const unsigned long long* tp = reinterpret_cast<const unsigned long long*>(lines + (size_t)lane * 8 + 7);
        unsigned spins = 0;
        // (>=: a workgroup that is ahead may already have posted the next iteration's tag -- its flag -- in the same place)
        while (__hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tag) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    // whatever was stored (and drained) before the tags is loaded behind the tag loads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
// One lane: the line's tag, behind everything this lane stored before (the other waves' stores: drained before the barrier
// the caller has just passed)
__device__ __forceinline__ void post_tag(double* line, unsigned long long tag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(line + 7), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// products (A r)_i, (A s)_i of the lane's row of slice d, left to right, gathered from the LDS window
__device__ __forceinline__ void row_products(const MediumArgs& a, const MDesc& d, int lane, int len, const double2* win, int cmin, int wlen,
                                             double& wr, double& us) {
    wr = 0.0; us = 0.0;
    int colacc = d.cbase;
    for (int u0 = 0; u0 < d.width; u0 += 8) {
        const long long vb = (long long)d.voff + ((long long)(u0 >> 1) * 64 + lane) * 2;
        const long long cb = (long long)d.coff + ((long long)(u0 >> 3) * 64 + lane) * 8;
        d2_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kk = (u0 + 2 * k < d.width) ? k : 0;
            v[k] = *reinterpret_cast<const d2_t*>(a.val + vb + (long long)kk * 128);
        }
        const u4_t c = *reinterpret_cast<const u4_t*>(a.col16 + cb);
        int code[8];
        code[0] = c.x & 0xffffu; code[1] = c.x >> 16; code[2] = c.y & 0xffffu; code[3] = c.y >> 16;
        code[4] = c.z & 0xffffu; code[5] = c.z >> 16; code[6] = c.w & 0xffffu; code[7] = c.w >> 16;
#pragma unroll
        for (int h4 = 0; h4 < 8; h4 += 4) {
            double2 gg[4];
            bool real[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                colacc += code[h4 + k] - 16384;
                real[k] = u0 + h4 + k < len && (unsigned)(code[h4 + k] - 1) < 65534u;
                int wi = colacc - cmin;
                wi = (wi >= 0 && wi < wlen) ? wi : 0;                        // (padding / skips: never used)
                gg[k] = win[wi];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double av = ((h4 + k) & 1) ? v[(h4 + k) >> 1].y : v[(h4 + k) >> 1].x;
                if (real[k]) { wr += av * gg[k].x; us += av * gg[k].y; }
            }
        }
    }
}

}  // namespace

// Per iteration and workgroup (pipe_pr_cg.py:61-75 re-phased as scaling_experiments_mpi4py/cg_variants/pipe_pr_cg.py:58-83 does:
// the products of an iteration's (r,s) overlap the reduction of its inner products):
//   1. a, b from the global sums of the previous iteration;  x, p, r, s of the own rows (registers; the new (r,s) into the
//      LDS window in place and, write-through, into exchange buffer it & 1); partial sums;
//   2. barrier; wave 0 posts the workgroup's sums (slot + tag); every wave: products of its INTERIOR slices (all columns
//      among the workgroup's own rows: LDS only) -- the row stores drain meanwhile;
//   3. drain, barrier, "rows visible" flag; wait for every workgroup's flag; stage the window's columns outside the own rows
//      from the exchange buffer; barrier; products of the BOUNDARY slices;
//   4. wait for every workgroup's sums: the same chain over the G slots in every workgroup.
__global__ __launch_bounds__(kMedThreads) void k_medium_pipe_pr(MediumArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2* win = reinterpret_cast<double2*>(smem);                       // the workgroup's window of (r,s) pairs
    __shared__ double s_red[kMedWaves][4];
    __shared__ double s_bc[4];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, G = a.G;
    const int gw = g * kMedWaves + wv;
    const int s_first = a.wave_first[gw], s_end = a.wave_first[gw + 1];
    const int cmin = a.wg_window[g].x, wlen = a.wg_window[g].y;
    const int own_lo = a.wg_own[g].x, own_hi = a.wg_own[g].y;            // rows [own_lo, own_hi) belong to this workgroup
    const int4* __restrict__ slices = a.slices;
    // slots: [2][G][8] four partial sums ... tag, by iteration parity -- a workgroup that has all sums of iteration `it` goes on
    // and posts its sums of it + 1 while another still reads those of `it` (it cannot post it + 2 before that one has posted its
    // flag of it + 1, i.e. has finished reading); then [G][8]: "rows of iteration it are visible" tags (monotone, waited for with >=)
    double* const sums_all = a.slots;
    double* flags = a.slots + (size_t)2 * kMedMaxGroups * 8;

    double xr[kMedSlices], pr[kMedSlices], wr[kMedSlices], us[kMedSlices];
    int row[kMedSlices], len[kMedSlices];
    bool bnd[kMedSlices];
    double* exch0 = a.exch;
    double* exch1 = a.exch + 2 * (size_t)a.n;
#pragma unroll
    for (int j = 0; j < kMedSlices; ++j) {
        row[j] = -1; len[j] = 0; xr[j] = 0.0; pr[j] = 0.0; wr[j] = 0.0; us[j] = 0.0; bnd[j] = false;
        if (s_first + j < s_end) {
            const MDesc d = read_mdesc(slices, s_first + j);
            bnd[j] = __builtin_amdgcn_readfirstlane(slices[2 * (s_first + j) + 1].w) != 0;
            if (d.rows_off < 0) {
                const int r = d.rb + lane;
                if (r < d.re) { row[j] = r; len[j] = a.indptr[r + 1] - a.indptr[r]; }
            } else {
                const int2 e = reinterpret_cast<const int2*>(a.rows)[d.rows_off + lane];
                row[j] = e.x; len[j] = e.y;
            }
            if (row[j] >= 0) {
                const double2 xp = reinterpret_cast<const double2*>(a.xp)[row[j]];
                xr[j] = xp.x; pr[j] = xp.y;
                const double2 rs = reinterpret_cast<const double2*>(a.rs)[row[j]];
                win[row[j] - cmin] = rs;
                st_pair_sc(exch0, row[j], rs);                               // the incoming (r,s) into buffer 0
            }
        }
    }
    const unsigned long long tag0 = a.seq << 24;
    bool alive = true;
    double mu = a.dots[(size_t)a.k0 * kPartialStride + 0], dl = a.dots[(size_t)a.k0 * kPartialStride + 1];
    double gm = a.dots[(size_t)a.k0 * kPartialStride + 2], nu = a.dots[(size_t)a.k0 * kPartialStride + 3];

    int done = 0;
    for (int it = 0; it <= a.iters && alive; ++it) {
        double* mine = (it & 1) ? exch1 : exch0;                              // where this iteration's rows went
        const unsigned long long tag = tag0 + (unsigned long long)it;
        double* sums = sums_all + (size_t)(it & 1) * kMedMaxGroups * 8;
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        if (it > 0) {
            // ---- 1. coefficients (pipe_pr_cg.py:64-66,75) and the own rows' update; w, u from the products of iteration it - 1
            const double al = nu / mu;
            const double a2 = al * al;
            const double nup = a.meurant ? (-nu + a2 * gm) : ((nu - (2 * al) * dl) + a2 * gm);
            const double bt = nup / nu;
            if (g == 0 && tid == 0) {
                double* cf = a.coef + (size_t)(a.k0 + it) * 4;
                cf[0] = al; cf[1] = bt; cf[2] = nup;
            }
#pragma unroll
            for (int j = 0; j < kMedSlices; ++j) {
                if (row[j] >= 0) {
                    const double2 rs = win[row[j] - cmin];
                    xr[j] = xr[j] + al * pr[j];                              // x += a p
                    const double rn = rs.x - al * rs.y;                      // r -= a s
                    const double wn = wr[j] - al * us[j];                    // w -= a u
                    const double pn = rn + bt * pr[j];                       // p = r + b p
                    const double sn = wn + bt * rs.y;                        // s = w + b s
                    pr[j] = pn;
                    win[row[j] - cmin] = make_double2(rn, sn);               // (only this lane reads or writes its row's entry here)
                    st_pair_sc(mine, row[j], make_double2(rn, sn));
                    acc0 += pn * sn; acc1 += rn * sn; acc2 += sn * sn; acc3 += rn * rn;
                }
            }
            acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2); acc3 = wave_sum(acc3);
            if (lane == 0) { s_red[wv][0] = acc0; s_red[wv][1] = acc1; s_red[wv][2] = acc2; s_red[wv][3] = acc3; }
        }
        __syncthreads();                                                      // the own rows of the window and s_red are complete
        if (it > 0 && tid < 64) {
            // ---- 2a. the workgroup's four sums: the 16 wave partials by one butterfly (lanes 0..15 hold them) -> its slot
            double w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = wave_sum(lane < kMedWaves ? s_red[lane][q] : 0.0);
            if (lane == 0) {
                double* slot = sums + (size_t)g * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) st_sc(slot + q, w4[q]);
                post_tag(slot, tag);
            }
        }
        if (it == a.iters) {                                                  // (the last update needs no products behind it)
            if (tid < 64) alive = wait_tags(sums, G, tag);
            done = it;
            break;
        }
        // ---- 2b. products of the interior slices: (r,s) of iteration `it` from the window's own rows
#pragma unroll
        for (int j = 0; j < kMedSlices; ++j)
            if (s_first + j < s_end && !bnd[j]) row_products(a, read_mdesc(slices, s_first + j), lane, len[j], win, cmin, wlen, wr[j], us[j]);
        // ---- 3. the rows of iteration `it` are visible to the other workgroups; theirs to this one
        if (G > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's row stores have left
            __syncthreads();
            if (tid == 0) post_tag(flags + (size_t)g * 8, tag);
            if (tid < 64) {
                const bool ok = wait_tags(flags, G, tag);
                if (lane == 0) s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            alive = s_ok != 0;
            // the window's columns outside the own rows, from the exchange buffer (coalesced)
            for (int i = tid; i < wlen; i += kMedThreads) {
                const int c = cmin + i;
                if (c < own_lo || c >= own_hi) win[i] = ld_pair_sc(mine, c);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kMedSlices; ++j)
                if (s_first + j < s_end && bnd[j]) row_products(a, read_mdesc(slices, s_first + j), lane, len[j], win, cmin, wlen, wr[j], us[j]);
        }
        // ---- 4. the global sums of iteration `it` (it == 0: the incoming state's, from the scalar history)
        if (it > 0) {
            if (tid < 64) {
                const bool ok = wait_tags(sums, G, tag);
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = lane < G ? ld_sc(sums + (size_t)lane * 8 + q) : 0.0;
                // slot order: a chain over the G slots -- the same in every workgroup
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double t = __shfl(v[q], 0, 64);
                    for (int r = 1; r < G; ++r) t += __shfl(v[q], r, 64);
                    if (lane == 0) s_bc[q] = t;
                }
                if (lane == 0) s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            alive = alive && s_ok != 0;
            mu = s_bc[0]; dl = s_bc[1]; gm = s_bc[2]; nu = s_bc[3];
            if (alive && g == 0 && tid == 0) {
                double* dd = a.dots + (size_t)(a.k0 + it) * kPartialStride;
                dd[0] = mu; dd[1] = dl; dd[2] = gm; dd[3] = nu; dd[4] = nu;
            }
            done = it;
        } else {
            __syncthreads();                                                  // (boundary products read the window: before the next update writes it)
        }
    }
    if (done == a.iters && a.iters > 0) {
        // the last iteration's sums: every workgroup has them posted; workgroup 0 writes them to the history
        __syncthreads();
        if (g == 0 && tid < 64) {
            const double* sums = sums_all + (size_t)(a.iters & 1) * kMedMaxGroups * 8;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = lane < G ? ld_sc(sums + (size_t)lane * 8 + q) : 0.0;
            double t4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double t = __shfl(v[q], 0, 64);
                for (int r = 1; r < G; ++r) t += __shfl(v[q], r, 64);
                t4[q] = t;
            }
            if (lane == 0 && alive) {
                double* dd = a.dots + (size_t)(a.k0 + a.iters) * kPartialStride;
                dd[0] = t4[0]; dd[1] = t4[1]; dd[2] = t4[2]; dd[3] = t4[3]; dd[4] = t4[3];
            }
        }
    }
    if (!alive && tid == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // state out: x, p from the registers, (r,s) of the own rows from the window
#pragma unroll
    for (int j = 0; j < kMedSlices; ++j) {
        if (row[j] >= 0) {
            reinterpret_cast<double2*>(a.xp)[row[j]] = make_double2(xr[j], pr[j]);
            reinterpret_cast<double2*>(a.rs)[row[j]] = win[row[j] - cmin];
        }
    }
}

int launch_medium_pipe_pr(hipStream_t st, const MediumArgs& a, int window_pairs) {
    const size_t lds = (size_t)window_pairs * 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_medium_pipe_pr), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_medium_pipe_pr, dim3(a.G), dim3(kMedThreads), lds, st, a);
    return hipGetLastError() == hipSuccess ? a.G : -1;
}

}  // namespace prcg
