// Sanitizer harness (CPU only): the window planning pipeline of prcg_set_csr -- plan_window_tiles,
// plan_window_dict, share_window_streams, plan_window_patterns, plan_sweep_tiles -- behind one C entry point, built with ASan/UBSan by
// tools/run_plan_asan.py and driven over every golden matrix and the synthetic operators.
#include "../new_cg_variants_amd/csrc/prcg_plan.h"

#include <cstring>
using namespace prcg;

extern "C" int plan_all(long n, long ncols, const int* indptr, const int* indices, const double* data, int rows, int share) {
    {   // sliced rows (plan_sell): every slice's trip-wise reads stay inside the arrays
        SellPlan sp;
        if (plan_sell(n, indptr, indices, data, nullptr, 1.25, sp)) {
            for (const auto* v : {&sp.s0, &sp.s1})
                for (const auto& t : *v) {
                    const long trips = (t.width + 7) / 8;
                    if (t.voff < 0 || (size_t)t.voff + (size_t)trips * 4 * 128 > sp.val.size() + 0) return -6;
                    if (t.coff < 0 || (size_t)t.coff + (size_t)trips * 2 * 256 > sp.col.size() + 0) return -7;
                }
        }
    }
    if (rows == 64) {
        // pattern tiles (plan_window_patterns) on the 64-row / 6-page tiling, and the sweep table (plan_sweep_tiles): every tile's
        // masks inside the store, every pattern id valid, every slot of a present row inside the tile's pages
        for (int sweep = 0; sweep < 2; ++sweep) {
            std::vector<WTile> tl;
            std::vector<uint16_t> cw;
            if (sweep) {
                SweepPlan sw;
                if (n != ncols || !plan_sweep_tiles(n, ncols, indptr, indices, 6, 512, sw)) continue;
                tl.swap(sw.tiles); cw.swap(sw.cw);
            } else {
                WinPlan wq;
                plan_window_tiles(n, ncols, indptr, indices, nullptr, 64, 1009, 6, wq);
                if (!wq.ok0 || !wq.ok1) continue;
                tl = wq.t0; tl.insert(tl.end(), wq.t1.begin(), wq.t1.end());
                cw.swap(wq.cw);
            }
            std::vector<PatRec> pats;
            std::vector<uint16_t> masks;
            if (!plan_window_patterns(tl, indptr, cw.data(), data, pats, masks)) continue;
            for (const auto& t : tl) {
                if (t.src_c < 0 || (size_t)t.src_c >= pats.size()) return -8;
                if (!t.spare && (t.src_r < 0 || (size_t)t.src_r + 64 > masks.size())) return -9;
                const PatRec& p = pats[(size_t)t.src_c];
                const int np = t.geo & 255;
                for (int lane = 0; lane < t.re - t.rb; ++lane) {
                    const unsigned mk = t.spare ? 0xffffu : masks[(size_t)t.src_r + lane];
                    for (int u = 0; u < p.nslots; ++u)
                        if (((mk >> u) & 1u) && (lane + p.cb[u] < 0 || lane + p.cb[u] >= np * 64)) return -10;
                }
            }
        }
    }
    WinPlan wp;
    plan_window_tiles(n, ncols, indptr, indices, nullptr, rows, 1009, rows == 64 ? 4 : 12, wp);
    if (!wp.ok0 || !wp.ok1) return 0;
    std::vector<WTile> wall(wp.t0);
    wall.insert(wall.end(), wp.t1.begin(), wp.t1.end());
    const long nnz = indptr[n];
    std::vector<uint8_t> wvidx((size_t)nnz + 32, 0);
    std::vector<double> wvdict;
    const bool vd = plan_window_dict(wall, data, 256, wvidx, wvdict);
    std::vector<uint8_t> vstore;
    std::vector<uint16_t> rstore, cstore;
    share_window_streams<uint16_t>(wall, indptr, wp.cw.data(), vd ? wvidx.data() : nullptr, share != 0, cstore, vstore, rstore);
    std::vector<uint8_t> c8w(wp.cw.size()), c8s;
    for (size_t q = 0; q < wp.cw.size(); ++q) c8w[q] = (uint8_t)wp.cw[q];
    share_window_streams<uint8_t>(wall, indptr, c8w.data(), vd ? wvidx.data() : nullptr, share != 0, c8s, vstore, rstore);
    // every descriptor must stay inside the stores
    for (const auto& t : wall) {
        const int pad = t.lo & 15, len = t.hi - t.lo;
        if (t.src_c < 0 || (size_t)t.src_c + pad + len > c8s.size()) return -2;
        if (vd && (t.src_v < 0 || (size_t)t.src_v + pad + len > vstore.size())) return -3;
        if (t.src_r < 0 || (size_t)t.src_r + (t.re - t.rb) + 1 > rstore.size()) return -4;
        if (vd && (t.vd_first < 0 || (size_t)t.vd_first + t.vd_count > wvdict.size())) return -5;
    }
    return 1 + (vd ? 1 : 0);
}
