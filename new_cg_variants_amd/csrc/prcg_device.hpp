// Device-side helpers shared by the kernel translation units of libprcg.so (gfx950 only).
// Included by prcg_kernels.hip (CSR-adaptive tile kernels, vector updates) and prcg_win.hip
// (row-per-lane window kernels).  Everything here is internal linkage on purpose.
#pragma once
#include <hip/hip_runtime.h>

#include "prcg_kernels.h"

namespace prcg {
namespace {


constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kElemsPerTrip = kBlock * 2;   // update kernels: 2 elements per thread per trip

// The butterfly v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: every lane ends with the same sum, added in the same tree.  The steps
// inside a row of 16 lanes are data-parallel-primitive moves (VALU, no trip through the LDS crossbar as __shfl_xor makes): row_ror:8 IS
// lane ^ 8; row_ror:4 hands a lane the value of lane ^ 4 or of (lane ^ 4) ^ 8 -- the same bits, since the ^ 8 step has been done;
// quad_perm for ^ 2 and ^ 1.  Bit for bit the shuffle form (checked on 262,144 random inputs incl. zeros of both signs).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {          // the last four steps: lanes of one row of 16
    v += dpp_move<0x128>(v);     // row_ror:8
    v += dpp_move<0x124>(v);     // row_ror:4
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    return row_sum(v);
}
// ... of values that only lanes 0..15 hold (the others would add +0.0: no partial sum here is -0.0, accumulators start at +0.0),
// the same bits as wave_sum(lane < 16 ? v : 0.0) in every lane
__device__ __forceinline__ double wave_sum16(double v) {
    v = row_sum(v);
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Blocks b and b+8 share an XCD (round-robin dispatch).  Give each XCD a contiguous
// range of work items so neighbouring tiles (which gather overlapping x entries) meet
// in the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int xcd = b & 7, idx = b >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// LDS traffic of ONE wave is processed in issue order; only the compiler has to be
// kept from moving the row reads above the product writes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NV> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = double2; };

__device__ __forceinline__ double vmul(double a, double g) { return a * g; }
__device__ __forceinline__ double2 vmul(double a, double2 g) { return make_double2(a * g.x, a * g.y); }
__device__ __forceinline__ void vacc(double& s, double p) { s += p; }
__device__ __forceinline__ void vacc(double2& s, double2 p) { s.x += p.x; s.y += p.y; }
__device__ __forceinline__ void vzero(double& s) { s = 0.0; }
__device__ __forceinline__ void vzero(double2& s) { s.x = 0.0; s.y = 0.0; }
__device__ __forceinline__ double vwave_sum(double v) { return wave_sum(v); }
__device__ __forceinline__ double2 vwave_sum(double2 v) { return make_double2(wave_sum(v.x), wave_sum(v.y)); }

// block-level combine of per-lane accumulators -> partials[block][slot0 + q]
template <int NQ>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NQ], double* partials, int slot0) {
    __shared__ double red[kWaves][NQ];
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
        for (int w = 1; w < kWaves; ++w) v += red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kPartialStride + slot0 + threadIdx.x] = v;
    }
}

// Same, plus: the block that finishes LAST sums all block partials in a fixed order and
// writes the final values -- no separate reduction launch.  Which block is last varies
// from run to run, the order of the summation does not, so the result is reproducible.
// Hand-off protocol (cdna_hip_programming.md Guideline 16, counter form): every block
// publishes its partials with plain stores -> s_waitcnt vmcnt(0) -> barrier -> one lane:
// agent-scope release fence, wait, relaxed agent-scope ticket; the block that draws the
// last ticket: agent-scope acquire fence, wait, barrier, plain loads.  Nobody spins.
template <int NQ>
__device__ __forceinline__ void block_reduce_store_final(double (&acc)[NQ], double* partials, unsigned* ticket,
                                                         double* __restrict__ final_out) {
    __shared__ double red[kWaves][NQ];
    __shared__ int s_last;
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
        for (int w = 1; w < kWaves; ++w) v += red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kPartialStride + threadIdx.x] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (tk == gridDim.x - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last block: thread t sums partials t, t+256, ...; butterfly; waves in order ----
    double tot[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) tot[q] = 0.0;
    const int nparts = gridDim.x;
    for (int j = threadIdx.x; j < nparts; j += kBlock) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) tot[q] += partials[(size_t)j * kPartialStride + q];
    }
    __syncthreads();   // red[] is reused
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double v = wave_sum(tot[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double v = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[w][threadIdx.x];
        final_out[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) *ticket = 0u;   // ready for the next launch (kernel boundary orders it)
}

// Every block of a launch sums the block partials the PREVIOUS launch left (slots slot0 .. slot0 + NQ) itself, in
// the order of k_reduce_final (a 256-thread tree: thread v rows v, v + 256, ...; butterfly over each 64 lanes; the
// four wave sums in order) whatever the block's own size: a block of NWAVES < 4 waves lets each thread play
// 4 / NWAVES of the tree's threads.  The same order in every block and in the separate reduction launch, so all
// blocks hold the same bits, and the schedules with and without reduction launches give the same numbers.
template <int NQ, int NWAVES>
__device__ __forceinline__ void sum_prev_partials(const double* __restrict__ prev, int nprev, int slot0, double (&out)[NQ]) {
    constexpr int kTreeWaves = 4;
    static_assert(kTreeWaves % NWAVES == 0 || NWAVES % kTreeWaves == 0, "1, 2, 4 or a multiple of 4 waves per block");
    constexpr int R = NWAVES >= kTreeWaves ? 1 : kTreeWaves / NWAVES;
    constexpr int TW = NWAVES >= kTreeWaves ? kTreeWaves : NWAVES;         // waves of this block that play tree threads
    __shared__ double redp[kTreeWaves][NQ];
    __shared__ double tot_s[NQ];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (NWAVES <= kTreeWaves || wv < kTreeWaves) {                          // (larger blocks: their first four waves are the tree)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double tot[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) tot[q] = 0.0;
        for (int j = threadIdx.x + r * 64 * TW; j < nprev; j += 64 * kTreeWaves) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) tot[q] += prev[(size_t)j * kPartialStride + slot0 + q];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double v = wave_sum(tot[q]);
            if (lane == 0) redp[wv + r * TW][q] = v;
        }
    }
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double v = redp[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kTreeWaves; ++w) v += redp[w][threadIdx.x];
        tot_s[threadIdx.x] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = tot_s[q];
}

struct Coefs { double al, bt, nup; };

__device__ __forceinline__ Coefs predict(const double* __restrict__ dp, int meurant) {
    // a_k1 = nu/mu; nu_k = nu - 2 a dl + a^2 gm (or Meurant's -nu + a^2 gm); b_k = nu_k/nu
    // (numerical_experiments/cg_variants/pipe_pr_cg.py:64-66,75; Python evaluates
    //  ((nu - (2a)dl) + (a^2)gm) left to right)
    const double mu = dp[0], dl = dp[1], gm = dp[2], nu = dp[3];
    Coefs c;
    c.al = nu / mu;
    const double a2 = c.al * c.al;
    c.nup = meurant ? (-nu + a2 * gm) : ((nu - (2 * c.al) * dl) + a2 * gm);
    c.bt = c.nup / nu;
    return c;
}

// Extra per-row operands of the one-launch iteration (see FusedState): kernel arguments
struct FusedRowPtrs {
    double2* XP; double2* IN_NEW; double2* RS; const double* D; double* W; double* WT;
    bool stream;      // streaming stores for the row results (see store_pair)
};
// the row's operands as loaded (ahead of time where the kernel can)
struct FusedRowIn { double2 xp, rs; double d, w, wt; };

// Update k of row `row` while (A in.x, A in.y)_row = sum is still in registers.
//   unpreconditioned (pipe_pr_cg.py:61-74): in = (r,s);  w_prev = A r (recomputed) or the stored recurrence
//   Jacobi           (pipe_pr_cg.py:169-186): in = (r~,s~); w~ = d w and u~ = d u as preconditioner(w), preconditioner(u)
// Same expressions, same order, no FMA as k_pipe_update (the two-kernel schedule) -- vectors agree bit for bit.
// The iteration's row results are written once and not read again before the next launch.  When the vectors are far
// larger than the caches (S3, S2: 640 MB of pairs per launch against 256 MB of Infinity Cache) streaming (nontemporal)
// stores keep them from displacing the stream images and window pages: S3 7.7 -> 8.8 k it/s, the same 2-read 2-write
// mix in tools/membench.hip 6.0 -> 6.7 TB/s.  When they fit (S1, one eighth of S3) the next launch finds them in the
// cache and plain stores win (S1 -8 % with streaming stores): `stream` is chosen per operator by the engine.
__device__ __forceinline__ void store_pair(double2* p, const double2& v, bool stream) {
    if (stream) {                                                           // wave-uniform
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        d2v_t w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<d2v_t*>(p));
    } else {
        *p = v;
    }
}

__device__ __forceinline__ void store_one(double* p, double v, bool stream) {
    if (stream) __builtin_nontemporal_store(v, p); else *p = v;
}

// Returns the row's new SpMM input pair (what went to IN_NEW): the peer exchange sends it on to the neighbours.
template <bool PREC, bool RECOMP>
__device__ __forceinline__ double2 fused_row_update(int row, const double2& sum, const FusedRowIn& q, const double2& in_old,
                                                    const FusedRowPtrs& f, const Coefs& cf, double (&acc)[5])
{
    const double us = sum.y;                                   // u = A s  (A s~)
    const double wprev = RECOMP ? sum.x : q.w;                 // w = A r  (A r~), or the recurrence
    const double xn = q.xp.x + cf.al * q.xp.y;                 // x += a p
    if constexpr (!PREC) {
        const double rn = in_old.x - cf.al * in_old.y;         // r -= a s
        const double wn = wprev - cf.al * us;                  // w -= a u
        const double pn = rn + cf.bt * q.xp.y;                 // p = r + b p
        const double sn = wn + cf.bt * in_old.y;               // s = w + b s
        store_pair(f.XP + row, make_double2(xn, pn), f.stream);
        store_pair(f.IN_NEW + row, make_double2(rn, sn), f.stream);
        if constexpr (!RECOMP) f.W[row] = wn;
        acc[0] += pn * sn; acc[1] += rn * sn; acc[2] += sn * sn; acc[3] += rn * rn;
        return make_double2(rn, sn);
    } else {
        const double ut = q.d * us;                            // u~ = M^-1 u
        const double wtprev = RECOMP ? q.d * sum.x : q.wt;     // w~ = M^-1 w, or the recurrence
        const double rn = q.rs.x - cf.al * q.rs.y;             // r -= a s
        const double rtn = in_old.x - cf.al * in_old.y;        // r~ -= a s~
        const double wn = wprev - cf.al * us;                  // w -= a u
        const double wtn = wtprev - cf.al * ut;                // w~ -= a u~
        const double pn = rtn + cf.bt * q.xp.y;                // p = r~ + b p
        const double sn = wn + cf.bt * q.rs.y;                 // s = w + b s
        const double stn = wtn + cf.bt * in_old.y;             // s~ = w~ + b s~
        store_pair(f.XP + row, make_double2(xn, pn), f.stream);
        store_pair(f.RS + row, make_double2(rn, sn), f.stream);
        store_pair(f.IN_NEW + row, make_double2(rtn, stn), f.stream);
        if constexpr (!RECOMP) { f.W[row] = wn; f.WT[row] = wtn; }
        acc[0] += pn * sn; acc[1] += rn * stn; acc[2] += stn * sn; acc[3] += rtn * rn; acc[4] += rn * rn;
        return make_double2(rtn, stn);
    }
}

// ---- direct peer exchange: device side (PeerDev, prcg_kernels.h) ----
__device__ __forceinline__ void peer_store(double* dst, double v) {           // leaves the chip: system scope, write-through
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long peer_counter(const PeerDev* px, int k) { return px->epoch | (unsigned long long)(unsigned)(k + 1); }
// One wave: lane q < R waits (bounded) for rank q's slot of iteration k in THIS rank's buffer, then the five sums are
// added in rank order (the same bits on every rank).  Returns lane-uniform sums in out[0..5); false on a timeout.
__device__ __forceinline__ bool peer_collect(const PeerDev* px, int k, unsigned max_spins, double (&out)[5]) {
    const int lane = threadIdx.x & 63;
    const int R = px->nranks;
    const double* slot = px->mine + peer_slot_off(R, k & 1, lane < R ? lane : 0);
    const unsigned long long want = peer_counter(px, k);
    const unsigned long long* cnt = reinterpret_cast<const unsigned long long*>(slot + 7);
    bool ok = true;
    if (lane < R) {
        unsigned spins = 0;
        while ((long long)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > max_spins) { ok = false; break; }
        }
    }
    // the sums were stored (and had left their chip) before the counter was: order the loads behind the counter load
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = lane < R ? __hip_atomic_load(slot + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double t = __shfl(v[q], 0, 64);
        for (int r = 1; r < R; ++r) t += __shfl(v[q], r, 64);
        out[q] = t;
    }
    return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
// One wave: this rank's slot of iteration k (lane q < 5 holds sum q in v) into EVERY rank's buffer, counter last
__device__ __forceinline__ void peer_send_slot(const PeerDev* px, int k, double v) {
    const int lane = threadIdx.x & 63;
    const int R = px->nranks;
    double vq[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) vq[q] = __shfl(v, q, 64);
    if (lane < R) {
        double* slot = px->peer[lane] + peer_slot_off(R, k & 1, px->rank);
#pragma unroll
        for (int q = 0; q < 5; ++q) peer_store(slot + q, vq[q]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the sums have arrived before the counter leaves
    if (lane < R) {
        double* slot = px->peer[lane] + peer_slot_off(R, k & 1, px->rank);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + 7), peer_counter(px, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// per-row epilogue: store y (and the fused extras)
template <int NV, int EPI>
__device__ __forceinline__ void finish_row(int row, const typename VecT<NV>::type& sum, void* __restrict__ yout_,
                                           int write_mask, const typename VecT<NV>::type* __restrict__ X,
                                           const double* __restrict__ ep_r, const double* __restrict__ ep_d,
                                           double* __restrict__ ep_st, double (&acc)[5], const Coefs& cf,
                                           const FusedRowPtrs& fr)
{
    if constexpr (NV == 1) {
        double* Y = reinterpret_cast<double*>(yout_);
        Y[row] = sum;
        if constexpr (EPI == kEpiDotXY) acc[0] += X[row] * sum;
        if constexpr (EPI == kEpiPR) {
            const double stv = ep_d ? ep_d[row] * sum : sum;
            if (ep_st) ep_st[row] = stv;
            acc[0] += X[row] * sum; acc[1] += ep_r[row] * stv; acc[2] += stv * sum;
        }
        if constexpr (EPI == kEpiCG) {
            // Chronopoulos-Gear: w = A r~ with nu = r.r~, eta = w.r~ (cg_cg.py:61-63), r.r for the history
            // (stored in the scalar-slot layout: eta -> 1, nu -> 3, r.r -> 4)
            const double rv = ep_r[row], zv = X[row];
            acc[3] += rv * zv; acc[1] += sum * zv; acc[4] += rv * rv;
        }
    } else if constexpr (epi_fused(EPI)) {
        // one-launch iteration on the CSR-adaptive tiles: the row's operands are loaded here
        // (the window kernels request them a tile ahead and call fused_row_update themselves)
        FusedRowIn q;
        q.xp = fr.XP[row];
        const double2 in_old = X[row];
        if constexpr (epi_prec(EPI)) { q.rs = fr.RS[row]; q.d = fr.D[row]; }
        if constexpr (!epi_recompute(EPI)) { q.w = fr.W[row]; if constexpr (epi_prec(EPI)) q.wt = fr.WT[row]; }
        fused_row_update<epi_prec(EPI), epi_recompute(EPI)>(row, sum, q, in_old, fr, cf, acc);
    } else {
        if (write_mask == 3) {
            reinterpret_cast<double2*>(yout_)[row] = sum;
        } else {
            double* Y = reinterpret_cast<double*>(yout_);
            if (write_mask & 1) Y[2 * (size_t)row] = sum.x;
            if (write_mask & 2) Y[2 * (size_t)row + 1] = sum.y;
        }
    }
}


}  // namespace
}  // namespace prcg
