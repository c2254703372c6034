// Is the speed of a large stream a property of the ALLOCATION (r04_sweeps.md D)?  Several buffers of the size of a Queen-size value
// stream, alive at once; each read whole and in regions of 128 MB with the same kernel (contiguous 4 KB chunks per wave, four
// nontemporal 16-byte loads in flight, 8 waves per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/regionscan.hip -o tools/regionscan ; tools/regionscan [GB per buffer] [buffers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_read(const d2_t* __restrict__ a, long pieces, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long chunks = pieces / 4;
    double acc = 0.0;
    for (long c = w; c < chunks; c += W) {
        const d2_t* base = a + c * 4 * 64 + lane;
        d2_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(base + u * 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 123.456) out[0] = acc;
}

double rate(const char* p, size_t bytes, double* out, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long pieces = (long)(bytes / 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, 0, reinterpret_cast<const d2_t*>(p), pieces, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, 0, reinterpret_cast<const d2_t*>(p), pieces, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)bytes * reps / (ms * 1e-3) * 1e-12;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.9;
    const int nbuf = argc > 2 ? atoi(argv[2]) : 4;
    const size_t region = (size_t)512 << 20;
    const size_t bytes = ((size_t)(gb * 1e9) / region) * region;
    double* out; CK(hipMalloc(&out, 64));
    std::vector<char*> bufs(nbuf);
    for (int b = 0; b < nbuf; ++b) { CK(hipMalloc(&bufs[b], bytes)); CK(hipMemset(bufs[b], 0, bytes)); }
    CK(hipDeviceSynchronize());
    printf("%d buffers of %.2f GB; TB/s whole, then per region of 512 MB (a region alone is read from HBM too: twice the Infinity Cache)\n", nbuf, bytes * 1e-9);
    for (int pass = 0; pass < 2; ++pass)
        for (int b = 0; b < nbuf; ++b) {
            printf("  buffer %d (%p) whole %.2f | regions", b, (void*)bufs[b], rate(bufs[b], bytes, out, 6));
            for (size_t o = 0; o + region <= bytes; o += region) printf(" %.2f", rate(bufs[b] + o, region, out, 12));
            printf("\n"); fflush(stdout);
        }
    return 0;
}
