// The byte mix of the one-launch iteration on an operator with plain values, with no arithmetic to speak of: per piece of 64 rows a
// wave reads VAL_KB KB of a value stream (nontemporal, read once), reads the rows' (x,p) and (r,s) pairs and writes (x,p) in place and
// (r,s) to another array (nontemporal stores) -- S3 with plain values: 7.5 KB of values + 0.94 KB of column bytes per 64 rows.
//   hipcc --offload-arch=gfx950 -O3 tools/mixbench.hip -o tools/mixbench ; tools/mixbench [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int VAL_KB, bool WRITE, bool AHEAD>
__global__ __launch_bounds__(256) void k(const d2* __restrict__ V, d2* __restrict__ X, const d2* __restrict__ R, d2* __restrict__ Rn, long n) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long pieces = n / 64;
    double keep = 0.0;
    d2 v[VAL_KB > 0 ? VAL_KB : 1], x, r;
    auto req = [&](long p) {
#pragma unroll
        for (int u = 0; u < VAL_KB; ++u) v[u] = __builtin_nontemporal_load(V + (p * VAL_KB + u) * 64 + lane);
        x = X[p * 64 + lane]; r = R[p * 64 + lane];
    };
    long p = w;
    if (AHEAD && p < pieces) req(p);
    for (; p < pieces; p += W) {
        if (!AHEAD) req(p);
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < VAL_KB; ++u) s += v[u].x + v[u].y;
        const d2 xo = x, ro = r;
        const long i = p * 64 + lane;
        if (AHEAD && p + W < pieces) req(p + W);                 // the next piece is requested before this one is stored (as the kernels do)
        const d2 xn = {xo.x + 0.5 * xo.y + s, ro.x + 0.25 * xo.y}, rn = {ro.x - 0.5 * ro.y, xo.y + 0.25 * ro.y};
        if (WRITE) { __builtin_nontemporal_store(xn, X + i); __builtin_nontemporal_store(rn, Rn + i); }
        else keep += xn.x + rn.y;
    }
    if (keep == 123.456) Rn[0].x = keep;
}
// the same mix with the row results of BATCH consecutive pieces stored together: a wave takes BATCH neighbouring pieces one after
// the other (their results: 2 x BATCH KB contiguous per array) and issues the 2 x BATCH stores behind the last one
template <int VAL_KB, int BATCH>
__global__ __launch_bounds__(256) void kb(const d2* __restrict__ V, d2* __restrict__ X, const d2* __restrict__ R, d2* __restrict__ Rn, long n) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long groups = n / 64 / BATCH;
    for (long g = w; g < groups; g += W) {
        d2 xn[BATCH], rn[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const long p = g * BATCH + b;
            d2 v[VAL_KB];
#pragma unroll
            for (int u = 0; u < VAL_KB; ++u) v[u] = __builtin_nontemporal_load(V + (p * VAL_KB + u) * 64 + lane);
            const d2 x = X[p * 64 + lane], r = R[p * 64 + lane];
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < VAL_KB; ++u) s += v[u].x + v[u].y;
            xn[b] = d2{x.x + 0.5 * x.y + s, r.x + 0.25 * x.y}; rn[b] = d2{r.x - 0.5 * r.y, x.y + 0.25 * r.y};
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) __builtin_nontemporal_store(xn[b], X + (g * BATCH + b) * 64 + lane);
#pragma unroll
        for (int b = 0; b < BATCH; ++b) __builtin_nontemporal_store(rn[b], Rn + (g * BATCH + b) * 64 + lane);
    }
}
template <int VAL_KB, int BATCH>
double runb(const d2* V, d2* X, d2* R, d2* Rn, long n, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((kb<VAL_KB, BATCH>), dim3(grid), dim3(256), 0, 0, V, X, R, Rn, n);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kb<VAL_KB, BATCH>), dim3(grid), dim3(256), 0, 0, V, X, R, Rn, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10 * 1e3;
}
template <int VAL_KB, bool WRITE, bool AHEAD>
double run(const d2* V, d2* X, d2* R, d2* Rn, long n, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<VAL_KB, WRITE, AHEAD>), dim3(grid), dim3(256), 0, 0, V, X, R, Rn, n);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<VAL_KB, WRITE, AHEAD>), dim3(grid), dim3(256), 0, 0, V, X, R, Rn, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10 * 1e3;
}
int main(int argc, char** argv) {
    if (argc > 1 && atol(argv[1]) == 0) {         // `mixbench 0 [order]`: only the Queen-size mix, once -- is its speed a property of the process?
        const long n2 = 4102893;
        const int order = argc > 2 ? atoi(argv[2]) : 0;
        d2 *V2 = nullptr, *X, *R, *Rn;
        if (order == 0) CK(hipMalloc(&V2, (n2 / 64 + 1) * 46 * 1024));
        CK(hipMalloc(&X, (n2 + 64) * 16)); CK(hipMalloc(&R, (n2 + 64) * 16)); CK(hipMalloc(&Rn, (n2 + 64) * 16));
        if (order != 0) CK(hipMalloc(&V2, (n2 / 64 + 1) * 46 * 1024));
        CK(hipMemset(V2, 0, (n2 / 64 + 1) * 46 * 1024)); CK(hipMemset(X, 0, (n2 + 64) * 16)); CK(hipMemset(R, 0, (n2 + 64) * 16)); CK(hipMemset(Rn, 0, (n2 + 64) * 16));
        printf("s4b mix, stores of 1 / 4 / 8 / 16 neighbouring pieces together: %.1f / %.1f / %.1f / %.1f us\n", runb<46, 1>(V2, X, R, Rn, n2, 512),
               runb<46, 4>(V2, X, R, Rn, n2, 512), runb<46, 8>(V2, X, R, Rn, n2, 512), runb<46, 16>(V2, X, R, Rn, n2, 512));
        {   // the written array Rn at other offsets from X (same allocation order, one arena): does the RELATIVE placement of the
            // streams that are accessed in lock step at equal offsets matter?
            char* arena; const size_t vb = (size_t)(n2 + 64) * 16, room = (size_t)160 << 20;
            CK(hipMalloc(&arena, 3 * (vb + room))); CK(hipMemset(arena, 0, 3 * (vb + room)));
            const size_t offs[] = {0, 256, 4096, 65536, (size_t)1 << 20, ((size_t)1 << 20) + 4096, (size_t)8 << 20, ((size_t)33 << 20) + 8192};
            printf("   arena, (R, Rn) shifted by k x offset against X:");
            for (size_t o : offs) {
                d2* Xa = reinterpret_cast<d2*>(arena);
                d2* Ra = reinterpret_cast<d2*>(arena + ((vb + room) & ~(size_t)0x1fffff) + o);
                d2* Rna = reinterpret_cast<d2*>(arena + (2 * (vb + room) & ~(size_t)0x1fffff) + 2 * o);
                if ((char*)Ra + vb > arena + 2 * (vb + room) - ((size_t)2 << 20) || (char*)Rna + vb > arena + 3 * (vb + room)) { printf("  %zu: out of the arena", o); continue; }
                printf("  %zu: %.1f", o, run<46, true, true>(V2, Xa, Ra, Rna, n2, 512));
            }
            printf("\n");
        }
        {   // six placements of the three vector arrays, all kept alive: allocated one by one, and as ONE allocation each
            const size_t vb = (size_t)(n2 + 64) * 16;
            printf("   placements, one by one:");
            for (int c = 0; c < 6; ++c) {
                d2 *a, *b, *cc; CK(hipMalloc(&a, vb)); CK(hipMalloc(&b, vb)); CK(hipMalloc(&cc, vb));
                CK(hipMemset(a, 0, vb)); CK(hipMemset(b, 0, vb)); CK(hipMemset(cc, 0, vb));
                printf("  %.1f", run<46, true, true>(V2, a, b, cc, n2, 512));
            }
            printf("\n   placements, one allocation for the three:");
            for (int c = 0; c < 6; ++c) {
                char* ar; const size_t step = (vb + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
                CK(hipMalloc(&ar, 3 * step)); CK(hipMemset(ar, 0, 3 * step));
                printf("  %.1f", run<46, true, true>(V2, reinterpret_cast<d2*>(ar), reinterpret_cast<d2*>(ar + step), reinterpret_cast<d2*>(ar + 2 * step), n2, 512));
            }
            printf("\n");
        }
        printf("s4b mix (order %d): %.1f us   without stores %.1f   V %p X %p R %p Rn %p\n", order, run<46, true, true>(V2, X, R, Rn, n2, 512),
               run<46, false, true>(V2, X, R, Rn, n2, 512), (void*)V2, (void*)X, (void*)R, (void*)Rn);
        return 0;
    }
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    d2 *V, *X, *R, *Rn;
    CK(hipMalloc(&V, (n / 64 + 1) * 8 * 1024)); CK(hipMalloc(&X, (n + 64) * 16)); CK(hipMalloc(&R, (n + 64) * 16)); CK(hipMalloc(&Rn, (n + 64) * 16));
    CK(hipMemset(V, 0, (n / 64 + 1) * 8 * 1024)); CK(hipMemset(X, 0, (n + 64) * 16)); CK(hipMemset(R, 0, (n + 64) * 16)); CK(hipMemset(Rn, 0, (n + 64) * 16));
    printf("%ld rows: vectors 2r + 2w of 16 B per row = %.3f GB, value stream 8 KB per 64 rows = %.3f GB; us per pass\n", n, 64.0 * n * 1e-9, n / 64 * 8192e-9);
    for (int grid : {512, 1024}) {
        printf("  grid %4d  request ahead:  vectors only %.1f   + values %.1f   values, no stores %.1f  |  request at use:  vectors only %.1f   + values %.1f   values, no stores %.1f\n", grid,
               run<0, true, true>(V, X, R, Rn, n, grid), run<8, true, true>(V, X, R, Rn, n, grid), run<8, false, true>(V, X, R, Rn, n, grid),
               run<0, true, false>(V, X, R, Rn, n, grid), run<8, true, false>(V, X, R, Rn, n, grid), run<8, false, false>(V, X, R, Rn, n, grid));
        fflush(stdout);
    }
    // the Queen-size FEM-like stand-in (s4b): 46 KB of values + codes per 64 rows, 4.1 M rows
    {
        const long n2 = 4102893;
        d2* V2; CK(hipMalloc(&V2, (n2 / 64 + 1) * 46 * 1024)); CK(hipMemset(V2, 0, (n2 / 64 + 1) * 46 * 1024));
        printf("%ld rows: vectors %.3f GB, stream 46 KB per 64 rows = %.3f GB; us per pass\n", n2, 64.0 * n2 * 1e-9, n2 / 64 * 46 * 1024e-9);
        for (int grid : {256, 512}) {
            printf("  grid %4d  request ahead:  + stream %.1f   stream, no stores %.1f  |  request at use:  + stream %.1f   stream, no stores %.1f\n", grid,
                   run<46, true, true>(V2, X, R, Rn, n2, grid), run<46, false, true>(V2, X, R, Rn, n2, grid),
                   run<46, true, false>(V2, X, R, Rn, n2, grid), run<46, false, false>(V2, X, R, Rn, n2, grid));
            fflush(stdout);
        }
    }
    return 0;
}
