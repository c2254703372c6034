// Period of back-to-back dependent launches on one stream (what a kernel boundary costs): empty kernel, a kernel whose
// every workgroup reads one line and writes one, at several grid sizes.   hipcc --offload-arch=gfx950 -O2 tools/launch_gap.hip -o tools/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(const double* __restrict__ in, double* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x * 8] = in[blockIdx.x * 8] + 1.0;
}
// grid barrier inside one persistent launch: `iters` rounds of (arrive on a counter, spin until all arrived)
__global__ void k_grid_barrier(unsigned* ctr, int iters, int nblk) {
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)it * (unsigned)nblk;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
int main() {
    double *a, *b; unsigned* ctr;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&ctr, 64);
    hipMemset(a, 0, 1 << 20); hipMemset(b, 0, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    const int N = 5000;
    for (int grid : {1, 256, 1024, 2048}) {
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 200; ++i) { if (which) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, a, b); else hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st); }
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { if (which) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, i & 1 ? a : b, i & 1 ? b : a); else hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st); }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("%s grid %5d x 256 threads: %6.2f us per launch\n", which ? "touch" : "empty", grid, us);
        }
    }
    for (int grid : {256, 512, 1024}) {
        const int iters = 2000;
        hipMemset(ctr, 0, 64);
        hipLaunchKernelGGL(k_grid_barrier, dim3(grid), dim3(256), 0, st, ctr, 10, grid); hipStreamSynchronize(st);
        hipMemset(ctr, 0, 64);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_grid_barrier, dim3(grid), dim3(256), 0, st, ctr, iters, grid);
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("grid barrier (one counter, agent-scope atomics) %5d workgroups: %6.2f us per round\n", grid, us);
    }
    return 0;
}
