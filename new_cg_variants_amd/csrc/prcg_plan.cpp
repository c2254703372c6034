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

}  // namespace prcg
