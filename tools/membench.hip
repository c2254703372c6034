// Streaming ceilings on the attached GPU: read-only, copy, and the SpMV-shaped
// two-stream (4-byte + 8-byte) read.  hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n2, double* out) {
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
    for (; i < n2; i += stride) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
// block-contiguous chunks instead of grid-stride
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read_chunk(const double2* __restrict__ a, size_t n2, double* out) {
    double s = 0.0;
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
    size_t i = lo + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    }
    for (; i < hi; i += 256) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) b[i] = a[i];
}
// copy with UNROLL 16-byte loads in flight per thread before the first store (the guide's float4 copy: 6.29 TB/s);
// NT: nontemporal loads and stores
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy_u(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* A = reinterpret_cast<const d2*>(a);
    d2* B = reinterpret_cast<d2*>(b);
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(A + i + u * stride) : A[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], B + i + u * stride); else B[i + u * stride] = v[u]; }
    }
    for (; i < n2; i += stride) B[i] = A[i];
}
// The byte mix of the one-launch iteration with the dictionary (S3): per row (x,p) read and written in place,
// (r,s) read from one array and written to another: 2 x 16 B in, 2 x 16 B out, nothing else.  n rows.
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_pairs_u(double2* __restrict__ xp, const double2* __restrict__ rs, double2* __restrict__ rsn, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* X = reinterpret_cast<d2*>(xp);
    const d2* R = reinterpret_cast<const d2*>(rs);
    d2* Rn = reinterpret_cast<d2*>(rsn);
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        d2 x[UNROLL], r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { x[u] = X[i + u * stride]; r[u] = R[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            d2 xo = x[u], ro = r[u];
            d2 xn = {xo.x + 0.5 * xo.y, ro.x + 0.25 * xo.y}, rn = {ro.x - 0.5 * ro.y, xo.y + 0.25 * ro.y};
            if (NT) { __builtin_nontemporal_store(xn, X + i + u * stride); __builtin_nontemporal_store(rn, Rn + i + u * stride); }
            else { X[i + u * stride] = xn; Rn[i + u * stride] = rn; }
        }
    }
}
// read/modify/write like the CG update: 3 pair arrays read, 2 written
__global__ __launch_bounds__(256) void k_update_like(double2* __restrict__ a, double2* __restrict__ b, const double2* __restrict__ c, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) {
        double2 x = a[i], y = b[i], z = c[i];
        a[i] = make_double2(x.x + 0.5 * x.y, y.x + 0.25 * x.y);
        b[i] = make_double2(y.x - 0.5 * y.y, z.x + 0.25 * y.y);
    }
}
// The byte mix of the one-launch pipelined iteration on S3 with plain values, no arithmetic to speak of:
// per 64-row tile a wave reads 7.5 KB of an 8-byte stream, 1 KB of a 1-byte stream, (x,p) and (r,s) of its
// rows (1 KB each) and writes both pairs back to (x,p) and a second (r,s) array: 203 bytes per row.
// DEPTH tiles in flight per wave, persistent strided tiles like k_win_tiles.
template <int DEPTH>
__global__ __launch_bounds__(128) void k_fused_like(const double2* __restrict__ val, const uint4* __restrict__ col,
                                                    double2* __restrict__ xp, const double2* __restrict__ rs,
                                                    double2* __restrict__ rsn, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int W = gridDim.x * 2;
    int t = blockIdx.x * 2 + (threadIdx.x >> 6);
    double2 v[DEPTH][8], x[DEPTH], r[DEPTH];
    uint4 c[DEPTH];
    auto issue = [&](int i, int tt) {
#pragma unroll
        for (int st = 0; st < 8; ++st) v[i][st] = val[(size_t)tt * 480 + (st * 64 + lane < 480 ? st * 64 + lane : 0)];
        c[i] = col[(size_t)tt * 60 + (lane < 60 ? lane : 0)];
        x[i] = xp[(size_t)tt * 64 + lane];
        r[i] = rs[(size_t)tt * 64 + lane];
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) if (t + i * W < ntiles) issue(i, t + i * W);
    while (t < ntiles) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (t < ntiles) {
                double s = (double)(c[i].x & 1u);
#pragma unroll
                for (int st = 0; st < 8; ++st) s += v[i][st].x + v[i][st].y;
                const double2 xo = x[i], ro = r[i];
                if (t + DEPTH * W < ntiles) issue(i, t + DEPTH * W);
                xp[(size_t)t * 64 + lane] = make_double2(xo.x + s, xo.y + ro.x);
                rsn[(size_t)t * 64 + lane] = make_double2(ro.x - s, ro.y + xo.y);
                t += W;
            }
        }
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB per buffer
    const size_t n2 = bytes / 16;
    double2 *a, *b, *c; double* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes)); CK(hipMemset(c, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name, double gbytes) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms;
        }
        printf("%-34s mean %.3f ms  %.0f GB/s   best %.0f GB/s\n", name, tot / 10, gbytes / (tot / 10) * 1e3, gbytes / best * 1e3);
    };
    const double G = bytes / 1e9;
    for (int g : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) {
        char nm[64];
        snprintf(nm, 64, "read stride u1 grid %d", g); time([&] { hipLaunchKernelGGL(k_read<1>, dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, G);
        snprintf(nm, 64, "read stride u4 grid %d", g); time([&] { hipLaunchKernelGGL(k_read<4>, dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, G);
        snprintf(nm, 64, "read stride u8 grid %d", g); time([&] { hipLaunchKernelGGL(k_read<8>, dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, G);
        snprintf(nm, 64, "read chunk  u4 grid %d", g); time([&] { hipLaunchKernelGGL(k_read_chunk<4>, dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, G);
    }
    time([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, n2); }, "copy grid 2048 (r+w bytes)", 2 * G);
    time([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n2); }, "copy grid 8192 (r+w bytes)", 2 * G);
    for (int g : {256 * 2, 256 * 4, 256 * 8}) {
        char nm[64];
        snprintf(nm, 64, "copy u4 grid %d (r+w bytes)", g); time([&] { hipLaunchKernelGGL((k_copy_u<4, false>), dim3(g), dim3(256), 0, 0, a, b, n2); }, nm, 2 * G);
        snprintf(nm, 64, "copy u8 grid %d (r+w bytes)", g); time([&] { hipLaunchKernelGGL((k_copy_u<8, false>), dim3(g), dim3(256), 0, 0, a, b, n2); }, nm, 2 * G);
        snprintf(nm, 64, "copy u4 nt grid %d (r+w bytes)", g); time([&] { hipLaunchKernelGGL((k_copy_u<4, true>), dim3(g), dim3(256), 0, 0, a, b, n2); }, nm, 2 * G);
        snprintf(nm, 64, "copy u8 nt grid %d (r+w bytes)", g); time([&] { hipLaunchKernelGGL((k_copy_u<8, true>), dim3(g), dim3(256), 0, 0, a, b, n2); }, nm, 2 * G);
    }
    {   // the dictionary kernel's mix at S3's size: 1e7 rows, 64 B per row = 0.64 GB per launch
        const size_t rows = 10000000;
        for (int g : {256 * 2, 256 * 4, 256 * 8}) {
            char nm[64];
            snprintf(nm, 64, "pairs 2r+2w u2 grid %d (S3 rows)", g); time([&] { hipLaunchKernelGGL((k_pairs_u<2, false>), dim3(g), dim3(256), 0, 0, a, b, c, rows); }, nm, 64.0 * rows / 1e9);
            snprintf(nm, 64, "pairs 2r+2w u4 grid %d (S3 rows)", g); time([&] { hipLaunchKernelGGL((k_pairs_u<4, false>), dim3(g), dim3(256), 0, 0, a, b, c, rows); }, nm, 64.0 * rows / 1e9);
            snprintf(nm, 64, "pairs 2r+2w u4 nt grid %d (S3 rows)", g); time([&] { hipLaunchKernelGGL((k_pairs_u<4, true>), dim3(g), dim3(256), 0, 0, a, b, c, rows); }, nm, 64.0 * rows / 1e9);
        }
    }
    time([&] { hipLaunchKernelGGL(k_update_like, dim3(2048), dim3(256), 0, 0, a, b, c, n2); }, "update-like 3r+2w grid 2048", 5 * G);
    time([&] { hipLaunchKernelGGL(k_update_like, dim3(8192), dim3(256), 0, 0, a, b, c, n2); }, "update-like 3r+2w grid 8192", 5 * G);
    {   // S3-shaped: 1e7 rows
        const int ntiles = 10000000 / 64;
        const double gb = 203.0 * 64 * ntiles / 1e9;
        for (int per_cu : {2, 4, 6, 8}) {
            char nm[64];
            snprintf(nm, 64, "fused-like depth1 %d wg/CU", per_cu);
            time([&] { hipLaunchKernelGGL(k_fused_like<1>, dim3(per_cu * 256), dim3(128), 0, 0, a, (const uint4*)b, c, c + (1 << 26), c + (1 << 25), ntiles); }, nm, gb);
            snprintf(nm, 64, "fused-like depth2 %d wg/CU", per_cu);
            time([&] { hipLaunchKernelGGL(k_fused_like<2>, dim3(per_cu * 256), dim3(128), 0, 0, a, (const uint4*)b, c, c + (1 << 26), c + (1 << 25), ntiles); }, nm, gb);
        }
    }
    return 0;
}
