// What misaligned row-result stores cost: the one-launch iteration's vector traffic (per row one pair read and rewritten in place, one
// pair read from one array and written to another, nontemporal stores) in pieces of ROWS rows per wave-instruction -- 64 rows are
// whole 128-byte lines, 62 rows (the pattern tiles of a 7-point stencil) start and end inside a line that another wave completes.
//   hipcc --offload-arch=gfx950 -O3 tools/storealign.hip -o tools/storealign ; tools/storealign [rows_total]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int ROWS, bool NT, bool WRITE>
__global__ __launch_bounds__(256) void k(d2* __restrict__ X, const d2* __restrict__ R, d2* __restrict__ Rn, long n) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long pieces = n / ROWS;
    double keep = 0.0;
    for (long p = w; p < pieces; p += W) {
        const long i = p * ROWS + lane;
        if (lane < ROWS) {
            const d2 x = X[i], r = R[i];
            const d2 xn = {x.x + 0.5 * x.y, r.x + 0.25 * x.y}, rn = {r.x - 0.5 * r.y, x.y + 0.25 * r.y};
            if (WRITE) {
                if (NT) { __builtin_nontemporal_store(xn, X + i); __builtin_nontemporal_store(rn, Rn + i); }
                else { X[i] = xn; Rn[i] = rn; }
            } else keep += xn.x + rn.y;
        }
    }
    if (keep == 123.456) Rn[0].x = keep;
}
template <int ROWS, bool NT, bool WRITE>
double run(d2* X, d2* R, d2* Rn, long n, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ROWS, NT, WRITE>), dim3(grid), dim3(256), 0, 0, X, R, Rn, n);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ROWS, NT, WRITE>), dim3(grid), dim3(256), 0, 0, X, R, Rn, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10077696;
    d2 *X, *R, *Rn;
    CK(hipMalloc(&X, (n + 64) * 16)); CK(hipMalloc(&R, (n + 64) * 16)); CK(hipMalloc(&Rn, (n + 64) * 16));
    CK(hipMemset(X, 0, (n + 64) * 16)); CK(hipMemset(R, 0, (n + 64) * 16)); CK(hipMemset(Rn, 0, (n + 64) * 16));
    printf("%ld rows, us per pass (2 x 16 B read + 2 x 16 B written per row = %.3f GB; read-only: half)\n", n, 64.0 * n * 1e-9);
    for (int grid : {512, 1024, 1536}) {
        printf("  grid %4d   64 rows: nt %.1f plain %.1f read-only %.1f |  62 rows: nt %.1f plain %.1f read-only %.1f |  63 rows nt %.1f   60 rows nt %.1f   56 rows nt %.1f\n", grid,
               run<64, true, true>(X, R, Rn, n, grid), run<64, false, true>(X, R, Rn, n, grid), run<64, true, false>(X, R, Rn, n, grid),
               run<62, true, true>(X, R, Rn, n, grid), run<62, false, true>(X, R, Rn, n, grid), run<62, true, false>(X, R, Rn, n, grid),
               run<63, true, true>(X, R, Rn, n, grid), run<60, true, true>(X, R, Rn, n, grid), run<56, true, true>(X, R, Rn, n, grid));
        fflush(stdout);
    }
    return 0;
}
