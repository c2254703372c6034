// Which READ pattern the memory system likes for a stream much larger than the Infinity Cache: how the bytes are dealt to the
// waves (1 KB pieces round-robin = a chip-wide front, or contiguous chunks of C bytes per wave), how many 16-byte loads a wave
// keeps in flight, how many waves per CU, plain or nontemporal loads.
//   hipcc --offload-arch=gfx950 -O3 tools/readpat.hip -o tools/readpat ; tools/readpat [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));

// the array is `pieces` pieces of 1 KB (64 lanes x 16 B); chunk = CH consecutive pieces; wave w takes chunks w, w + W, ...
// and walks each piece by piece with U loads in flight (U divides CH or CH == 1: then the U loads are U chunks = round-robin pieces)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2_t* __restrict__ a, long pieces, int ch, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long chunks = pieces / ch;
    double acc = 0.0;
    if (ch == 1) {
        long c = w;
        for (; c + (U - 1) * W < chunks; c += U * W) {
            d2_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const d2_t* q = a + (c + u * W) * 64 + lane; v[u] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
        }
        for (; c < chunks; c += W) { const d2_t v = a[c * 64 + lane]; acc += v.x + v.y; }
    } else {
        for (long c = w; c < chunks; c += W) {
            const d2_t* base = a + c * ch * 64 + lane;
            for (int p = 0; p < ch; p += U) {
                d2_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const d2_t* q = base + (long)(p + u) * 64; v[u] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
            }
        }
    }
    if (acc == 123.456) out[0] = acc;
}

template <int U, bool NT>
double run(const d2_t* a, long pieces, int ch, double* out, int per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = per_cu * 256, reps = 8;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, a, pieces, ch, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, a, pieces, ch, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.0;
    const long pieces = ((long)(gb * 1e9 / 1024) / 4096) * 4096;
    d2_t* a; double* out;
    CK(hipMalloc(&a, pieces * 1024)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 0, pieces * 1024));
    const double bytes = (double)pieces * 1024;
    printf("read patterns over %.2f GB (TB/s); columns: waves per CU 4 / 8 / 16 / 32\n", bytes * 1e-9);
    const int chs[] = {1, 4, 16, 64, 256, 1024};
    for (int nt = 0; nt < 2; ++nt)
        for (int ch : chs)
            for (int u : {2, 4, 8}) {
                if (ch != 1 && ch < u) continue;
                printf("  %s chunk %4d KB  in flight %d :", nt ? "nt   " : "plain", ch, u);
                for (int per_cu : {1, 2, 4, 8}) {
                    double ms = 0;
#define R(U) ms = nt ? run<U, true>(a, pieces, ch, out, per_cu) : run<U, false>(a, pieces, ch, out, per_cu)
                    if (u == 2) R(2); else if (u == 4) R(4); else R(8);
                    printf("  %.2f", bytes / ms * 1e-9);
                }
                printf("\n"); fflush(stdout);
            }
    return 0;
}
