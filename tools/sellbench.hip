// What the memory system delivers for the STREAM of the sliced-row kernels (prcg_sell.hip), with nothing else: every wave reads
// its slices (64 lanes x width nonzeros: 8-byte values in 16-byte chunks, 2-byte columns in 8- or 16-byte chunks), trip by trip
// (8 nonzeros per lane), DEPTH trips requested ahead.  Sweeps prefetch depth, workgroups per CU, the width of the column loads and
// nontemporal loads.   hipcc --offload-arch=gfx950 -O3 tools/sellbench.hip -o tools/sellbench ; tools/sellbench [GB] [width]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

template <int COLW> struct Trip { d2_t v[4]; u4_t c; };

template <int COLW, bool NT>
__device__ __forceinline__ void load_trip(const double* __restrict__ val, const unsigned short* __restrict__ col, size_t vbase, size_t cbase,
                                          int trip, int lane, Trip<COLW>& T) {
    const size_t vb = vbase + ((size_t)trip * 4 * 64 + lane) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2_t* q = reinterpret_cast<const d2_t*>(val + vb + (size_t)k * 128);
        T.v[k] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (COLW == 16) {
        const u4_t* q = reinterpret_cast<const u4_t*>(col + cbase + ((size_t)trip * 64 + lane) * 8);
        T.c = NT ? __builtin_nontemporal_load(q) : *q;
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u2_t* q = reinterpret_cast<const u2_t*>(col + cbase + ((size_t)(trip * 2 + k) * 64 + lane) * 4);
            const u2_t c = NT ? __builtin_nontemporal_load(q) : *q;
            if (k == 0) { T.c.x = c.x; T.c.y = c.y; } else { T.c.z = c.x; T.c.w = c.y; }
        }
    }
}

// slices of `trips` trips each; wave w takes slices w, w + W, ...; the stream of a wave = its slices' trips in order
template <int DEPTH, int COLW, bool NT>
__global__ __launch_bounds__(256) void k_sell_stream(const double* __restrict__ val, const unsigned short* __restrict__ col, int nslices, int trips,
                                                     double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int W = gridDim.x * 4;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t vslice = (size_t)trips * 8 * 64, cslice = (size_t)trips * 8 * 64;
    const long total = w < nslices ? ((long)(nslices - 1 - w) / W + 1) * trips : 0;       // trips of this wave
    Trip<COLW> T[DEPTH];
    double acc = 0.0;
    unsigned cacc = 0;
    auto addr = [&](long g, size_t& vb, size_t& cb, int& tr) {
        const long s = g / trips; tr = (int)(g - s * trips);
        const size_t sl = (size_t)w + (size_t)s * W;
        vb = sl * vslice; cb = sl * cslice;
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < total) { size_t vb, cb; int tr; addr(i, vb, cb, tr); load_trip<COLW, NT>(val, col, vb, cb, tr, lane, T[i]); }
    for (long g = 0; g < total; g += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (g + i < total) {
                const Trip<COLW> cur = T[i];
                if (g + i + DEPTH < total) { size_t vb, cb; int tr; addr(g + i + DEPTH, vb, cb, tr); load_trip<COLW, NT>(val, col, vb, cb, tr, lane, T[i]); }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += cur.v[k].x + cur.v[k].y;
                cacc += cur.c.x ^ cur.c.y ^ cur.c.z ^ cur.c.w;
            }
        }
    }
    if (acc == 123.456 || cacc == 0x12345u) out[0] = acc + cacc;
}

template <int DEPTH, int COLW, bool NT>
double run(const double* val, const unsigned short* col, int nslices, int trips, double* out, int per_cu, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = per_cu * 256;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_sell_stream<DEPTH, COLW, NT>), dim3(grid), dim3(256), 0, 0, val, col, nslices, trips, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_sell_stream<DEPTH, COLW, NT>), dim3(grid), dim3(256), 0, 0, val, col, nslices, trips, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 3.3;
    const int width = argc > 2 ? atoi(argv[2]) : 88;            // nonzeros per row (padded to whole trips)
    const int trips = (width + 7) / 8;
    const size_t per_slice = (size_t)trips * 8 * 64 * 10;
    const int nslices = (int)(gb * 1e9 / per_slice);
    double* val; unsigned short* col; double* out;
    CK(hipMalloc(&val, (size_t)nslices * trips * 8 * 64 * 8 + 4096));
    CK(hipMalloc(&col, (size_t)nslices * trips * 8 * 64 * 2 + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(val, 0, (size_t)nslices * trips * 8 * 64 * 8)); CK(hipMemset(col, 0, (size_t)nslices * trips * 8 * 64 * 2));
    const double bytes = (double)nslices * per_slice;
    printf("sell stream: %d slices x %d trips (%.2f GB: 8 B values + 2 B columns per nonzero)\n", nslices, trips, bytes * 1e-9);
    for (int per_cu : {1, 2, 3, 4}) {
#define ROW(D, C, N) { const double ms = run<D, C, N>(val, col, nslices, trips, out, per_cu, 10); \
        printf("  wg/CU %d depth %d col-load %2d B nt %d : %.4f ms  %.2f TB/s\n", per_cu, D, C, N, ms, bytes / ms * 1e-9); fflush(stdout); }
        ROW(1, 8, false) ROW(2, 8, false) ROW(3, 8, false) ROW(1, 16, false) ROW(2, 16, false) ROW(3, 16, false) ROW(4, 16, false)
        ROW(2, 16, true)
    }
    return 0;
}
