// Host-side planning: CSR-adaptive tiling of a row block.  No GPU calls in this file,
// so the CPU test-suite can exercise it through the C-ABI (prcg_plan_tiles).
#include "prcg_plan.h"

namespace prcg {

// Greedy packing of consecutive rows of one class into wave tiles.
//  - a tile never exceeds cap_nnz nonzeros nor cap_rows rows;
//  - a row with more than cap_nnz nonzeros becomes a tile of its own (the kernel then
//    sums it with the whole wave);
//  - rows of different class (interior vs. touching ghost columns) never share a tile,
//    so the interior tiles can run while the halo is still in flight.
void plan_tiles(int64_t n, const int32_t* indptr, const uint8_t* row_class,
                int cap_nnz, int cap_rows,
                std::vector<Tile>& class0, std::vector<Tile>& class1)
{
    class0.clear();
    class1.clear();
    int64_t r = 0;
    while (r < n) {
        const uint8_t cls = row_class ? (row_class[r] != 0) : 0;
        std::vector<Tile>& out = cls ? class1 : class0;
        // the run of rows of this class
        int64_t run_end = r + 1;
        if (row_class) {
            while (run_end < n && ((row_class[run_end] != 0) == cls)) ++run_end;
        } else {
            run_end = n;
        }
        while (r < run_end) {
            int64_t e = r;
            int64_t nn = 0;
            while (e < run_end && (e - r) < cap_rows) {
                const int64_t len = (int64_t)indptr[e + 1] - indptr[e];
                if (nn + len > cap_nnz) break;
                nn += len;
                ++e;
            }
            if (e == r) e = r + 1;   // a single long row
            out.push_back(Tile{(int)r, (int)e, (int)indptr[r], (int)indptr[e]});
            r = e;
        }
    }
}

int plan_gather_sources(int rank, int T, const double* tab, int n_peers, const int32_t* peer_rank,
                        const int64_t* recv_ptr, int64_t slot, int32_t* src) {
    for (int q = 0; q < n_peers; ++q) {
        const int64_t want = recv_ptr[q + 1] - recv_ptr[q];
        if (want == 0) continue;
        const int pr = peer_rank[q];
        const double* pt = tab + (size_t)pr * T;
        int64_t off = -1;
        for (int e = 0; e < (int)pt[0]; ++e)
            if ((int)pt[1 + 3 * e] == rank && (int64_t)pt[3 + 3 * e] == want) { off = (int64_t)pt[2 + 3 * e]; break; }
        if (off < 0) return 1 + q;
        const int64_t first = ((int64_t)pr * slot + 8) / 2 + off;
        if (first + want >= (int64_t)INT32_MAX) return 1 + q;
        for (int64_t i = 0; i < want; ++i) src[(size_t)(recv_ptr[q] + i)] = (int32_t)(first + i);
    }
    return 0;
}

}  // namespace prcg
