// Host-side planning: CSR-adaptive tiling of a row block.  No GPU calls in this file,
// so the CPU test-suite can exercise it through the C-ABI (prcg_plan_tiles).
#include "prcg_plan.h"

#include <algorithm>
#include <atomic>
#include <array>
#include <map>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

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

namespace {

// pages + column encoding of tiles [first, last) of `tiles`; returns the most pages a tile needed,
// or -1 if one needed more than max_pages
// n_own: owned columns [0, n_own) -- a page never serves columns on both sides of n_own (the ghost columns may live in
// another buffer than the owned ones: the peer-exchange schedule reads them from the rank's exchange buffer)
int window_pages(std::vector<WTile>& tiles, size_t first, size_t last, int64_t n_own, int64_t n_cols, const int32_t* indptr,
                 const int32_t* indices, int max_pages, uint16_t* cw) {
    auto page_end = [n_own](int32_t start) { return (int64_t)start < n_own ? std::min<int64_t>((int64_t)start + 64, n_own) : (int64_t)start + 64; };
    std::vector<int32_t> cols;
    int most = 0;
    for (size_t ti = first; ti < last; ++ti) {
        WTile& t = tiles[ti];
        cols.assign(indices + t.lo, indices + t.hi);
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        // greedy cover of {needed columns} U [rb, re) by pages of 64 consecutive columns; a page
        // opened inside the tile's own rows continues the previous one, so that the own rows are
        // contiguous in the window (the fused epilogues read (r,s) of their row from it)
        int np = 0;
        int32_t page[kWinMaxPages];
        size_t ci = 0;
        int32_t own = t.rb;                   // next own row still to be covered
        int maxlen = 0;
        for (int r = t.rb; r < t.re; ++r) maxlen = std::max(maxlen, indptr[r + 1] - indptr[r]);
        bool fail = false;
        for (;;) {
            // smallest uncovered column among the needed ones and the own rows
            while (ci < cols.size() && np > 0 && cols[ci] < page_end(page[np - 1])) ++ci;
            if (np > 0 && own < page_end(page[np - 1])) own = (int32_t)std::max<int64_t>(own, page_end(page[np - 1]));
            int32_t c;
            const bool have_col = ci < cols.size(), have_own = own < t.re;
            if (!have_col && !have_own) break;
            if (have_col && have_own) c = std::min(cols[ci], own);
            else c = have_col ? cols[ci] : own;
            if (np == max_pages || np == kWinMaxPages) { fail = true; break; }
            // keep the page inside the vector where possible (its 64 entries are loaded unconditionally); a page
            // pushed behind the previous one may still end up to 63 entries past n_cols -- every vector that feeds
            // a product is allocated with spare entries behind its end (kGatherPad, prcg_engine.cpp)
            // (a page of ghost columns is not pulled back below n_own; it may end past n_cols like a pushed one)
            if ((int64_t)c + 64 > n_cols) c = (int32_t)std::max<int64_t>(n_cols - 64, (int64_t)c >= n_own ? n_own : 0);
            if (np > 0 && c < page_end(page[np - 1])) c = (int32_t)page_end(page[np - 1]);   // the columns a page SERVES never overlap
            page[np++] = c;
        }
        if (fail) return -1;
        {
            const int p = (int)(std::upper_bound(page, page + np, (int32_t)t.rb) - page) - 1;
            t.geo = np | ((p * 64 + (t.rb - page[p])) << 8);
        }
        t.maxlen = maxlen;
        for (int p = 0; p < kWinMaxPages; ++p) t.page_col[p] = p < np ? page[p] : 0;
        most = std::max(most, np);
        for (int32_t q = t.lo; q < t.hi; ++q) {
            const int32_t col = indices[q];
            int p = (int)(std::upper_bound(page, page + np, col) - page) - 1;
            // (col lies inside page p: pages are opened at uncovered columns and cover 64 -- or up to n_own)
            cw[q] = (uint16_t)(p * 64 + (col - page[p]));
        }
    }
    return most;
}

bool window_class(std::vector<WTile>& tiles, int64_t n_own, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                  int max_pages, uint16_t* cw, int* most_pages) {
    if (tiles.empty()) { *most_pages = 0; return true; }
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (tiles.size() < 4096) nt = 1;
    std::vector<int> res(nt, 0);
    std::vector<std::thread> th;
    const size_t per = (tiles.size() + nt - 1) / nt;
    for (unsigned i = 0; i < nt; ++i) {
        const size_t a = std::min(tiles.size(), i * per), b = std::min(tiles.size(), a + per);
        if (nt == 1) res[0] = window_pages(tiles, a, b, n_own, n_cols, indptr, indices, max_pages, cw);
        else th.emplace_back([&, a, b, i] { res[i] = window_pages(tiles, a, b, n_own, n_cols, indptr, indices, max_pages, cw); });
    }
    for (auto& t : th) t.join();
    int most = 0;
    for (int r : res) { if (r < 0) return false; most = std::max(most, r); }
    *most_pages = most;
    return true;
}

}  // namespace

void plan_window_tiles(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                       const uint8_t* row_class, int rows_per_tile, int cap_nnz, int max_pages, WinPlan& out) {
    out.t0.clear(); out.t1.clear();
    out.ok0 = out.ok1 = true;
    out.pages0 = out.pages1 = 0;
    out.cw.assign((size_t)indptr[n] + 32, 0);
    int64_t r = 0;
    while (r < n) {
        const uint8_t cls = row_class ? (row_class[r] != 0) : 0;
        std::vector<WTile>& dst = cls ? out.t1 : out.t0;
        int64_t run_end = n;
        if (row_class) { run_end = r + 1; while (run_end < n && ((row_class[run_end] != 0) == cls)) ++run_end; }
        while (r < run_end) {
            int64_t e = r, nn = 0;
            while (e < run_end && (e - r) < rows_per_tile) {
                const int64_t len = (int64_t)indptr[e + 1] - indptr[e];
                if (nn + len > cap_nnz) break;
                nn += len;
                ++e;
            }
            if (e == r) {                 // one row longer than a tile: not a window operator
                (cls ? out.ok1 : out.ok0) = false;
                e = r + 1;
            }
            WTile t{};
            t.rb = (int)r; t.re = (int)e; t.lo = indptr[r]; t.hi = indptr[e];
            dst.push_back(t);
            r = e;
        }
    }
    if (n_cols < 64) out.ok0 = out.ok1 = false;     // a page must fit inside the vector
    if (out.ok0) out.ok0 = window_class(out.t0, n, n_cols, indptr, indices, max_pages, out.cw.data(), &out.pages0);
    if (out.ok1) out.ok1 = window_class(out.t1, n, n_cols, indptr, indices, max_pages, out.cw.data(), &out.pages1);
}

namespace {

uint64_t hash_bytes(const void* p, size_t n, uint64_t h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) { h = (h ^ b[i]) * 0x100000001B3ull; }
    return h ^ (h >> 32);
}

}  // namespace

bool plan_window_dict(std::vector<WTile>& tiles, const double* data, int dict_max,
                      std::vector<uint8_t>& vidx, std::vector<double>& vdict) {
    constexpr int kHash = 1024;           // open addressing, <= 256 live keys
    uint64_t keys[kHash];
    int16_t slot_of[kHash];
    std::unordered_map<uint64_t, std::vector<int64_t>> tables;
    for (auto& t : tiles) {
        if (vdict.size() & 1) vdict.push_back(0.0);            // 16-byte aligned table start
        for (int i = 0; i < kHash; ++i) slot_of[i] = -1;
        const size_t first = vdict.size();
        int count = 0;
        for (int32_t q = t.lo; q < t.hi; ++q) {
            uint64_t bits;
            memcpy(&bits, &data[q], sizeof bits);
            uint32_t hsh = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 54);   // 10 bits
            while (slot_of[hsh] >= 0 && keys[hsh] != bits) hsh = (hsh + 1) & (kHash - 1);
            if (slot_of[hsh] < 0) {
                if (count == dict_max) return false;
                keys[hsh] = bits;
                slot_of[hsh] = (int16_t)count++;
                vdict.push_back(data[q]);
            }
            vidx[q] = (uint8_t)slot_of[hsh];
        }
        t.vd_first = (int)first;
        t.vd_count = count;
        // a table identical to one already stored (constant coefficients: every tile of a stencil) is shared
        {
            const uint64_t key = hash_bytes(vdict.data() + first, (size_t)count * sizeof(double), 0x9AE16A3B2F90404Full + (uint64_t)count);
            auto it = tables.find(key);
            bool shared = false;
            if (it != tables.end())
                for (int64_t pos : it->second)
                    if (memcmp(vdict.data() + pos, vdict.data() + first, (size_t)count * sizeof(double)) == 0) {
                        t.vd_first = (int)pos;
                        vdict.resize(first);
                        shared = true;
                        break;
                    }
            if (!shared) tables[key].push_back((int64_t)first);
        }
        if (vdict.size() >= (size_t)INT32_MAX - 1024) return false;
    }
    return true;
}

namespace {

// one tile's pattern record and slot masks; false: the tile is no pattern tile
bool pattern_of_tile(const WTile& t, const int32_t* indptr, const uint16_t* cw, const double* data, PatRec& rec, uint16_t (&m)[64],
                     bool& full) {
    if (t.re - t.rb > 64 || t.re < t.rb) return false;
    if (t.re == t.rb) {                                                   // an empty tile (padding of a sweep table): no slots, no rows
        memset(&rec, 0, sizeof rec);
        memset(m, 0, sizeof m);
        full = false;
        return true;
    }
    // slots: the distinct d = window index - lane over the tile; one value per slot; `succ`: slot a directly precedes slot b
    // in some row (the rows' own order is what the sum follows -- in a row block ghost columns are numbered behind the
    // owned ones, so a row's columns need not ascend)
    int nd = 0;
    int dv[kPatSlots];
    uint64_t bits[kPatSlots];
    unsigned succ[kPatSlots];
    for (int u = 0; u < kPatSlots; ++u) succ[u] = 0u;
    for (int r = t.rb; r < t.re; ++r) {
        const int lane = r - t.rb;
        int before = -1;
        unsigned seen = 0u;
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int d = (int)cw[q] - lane;
            uint64_t b;
            memcpy(&b, &data[q], sizeof b);
            int u = 0;
            while (u < nd && dv[u] != d) ++u;
            if (u == nd) {
                if (nd == kPatSlots) return false;
                dv[nd] = d; bits[nd] = b; ++nd;
            } else if (bits[u] != b) {
                return false;                                         // the slot's value differs between rows
            }
            if (seen & (1u << u)) return false;                       // a column twice in one row
            seen |= 1u << u;
            if (before >= 0) succ[before] |= 1u << u;
            before = u;
        }
    }
    // one order of the slots that every row follows: topological (ties: ascending d); a cycle = rows disagree
    int order[kPatSlots];
    {
        unsigned placed = 0u;
        for (int k = 0; k < nd; ++k) {
            int pick = -1;
            for (int u = 0; u < nd; ++u) {
                if (placed & (1u << u)) continue;
                bool ready = true;
                for (int v = 0; v < nd && ready; ++v)
                    if (!(placed & (1u << v)) && v != u && (succ[v] & (1u << u))) ready = false;
                if (ready && (pick < 0 || dv[u] < dv[pick])) pick = u;
            }
            if (pick < 0) return false;
            order[k] = pick;
            placed |= 1u << pick;
        }
    }
    memset(&rec, 0, sizeof rec);
    rec.nslots = nd;
    int nvals = 0;
    uint64_t vbits[kPatValues];
    for (int k = 0; k < nd; ++k) {
        const int u = order[k];
        if (dv[u] < -32768 || dv[u] > 32767) return false;
        rec.cb[k] = (short)dv[u];
        int v = 0;
        while (v < nvals && vbits[v] != bits[u]) ++v;
        if (v == nvals) {
            if (nvals == kPatValues) return false;
            vbits[nvals++] = bits[u];
        }
        rec.vsel |= (unsigned)v << (2 * k);
    }
    for (int v = 0; v < nvals; ++v) memcpy(&rec.val[v], &vbits[v], sizeof(double));
    // masks of the rows (lanes past the tile's last row: 0)
    memset(m, 0, sizeof m);
    full = (t.re - t.rb) == 64;
    const unsigned all = nd >= 16 ? 0xffffu : ((1u << nd) - 1u);
    for (int r = t.rb; r < t.re; ++r) {
        const int lane = r - t.rb;
        unsigned mk = 0;
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int d = (int)cw[q] - lane;
            int k = 0;
            while (rec.cb[k] != d) ++k;
            mk |= 1u << k;
        }
        m[lane] = (uint16_t)mk;
        if (mk != all) full = false;
    }
    return true;
}

}  // namespace

bool plan_window_patterns(std::vector<WTile>& tiles, const int32_t* indptr, const uint16_t* cw, const double* data,
                          std::vector<PatRec>& patterns, std::vector<uint16_t>& masks) {
    struct Plan { int pat; int64_t mask_at; int full; int nslots; };
    PatRec rec;
    uint16_t m[64];
    bool full = false;
    // a few tiles first (a band whose diagonal varies only near one end fails at that end)
    if (tiles.empty()) return false;
    for (size_t k = 0; k <= 8; ++k) {
        const size_t ti = (tiles.size() - 1) * k / 8;
        if (!pattern_of_tile(tiles[ti], indptr, cw, data, rec, m, full)) return false;
    }
    std::vector<Plan> plan(tiles.size());
    std::unordered_map<uint64_t, std::vector<int>> pat_index;            // hash of a record -> pattern ids
    std::unordered_map<uint64_t, std::vector<int64_t>> mask_index;        // hash of 64 masks -> positions
    std::vector<PatRec> pats;
    std::vector<uint16_t> mstore;
    for (size_t ti = 0; ti < tiles.size(); ++ti) {
        if (!pattern_of_tile(tiles[ti], indptr, cw, data, rec, m, full)) return false;
        // identical records / mask images are stored once
        int pid = -1;
        {
            const uint64_t key = hash_bytes(&rec, sizeof rec, 0x51ED270B7A1F3C55ull);
            auto& v = pat_index[key];
            for (int id : v) if (memcmp(&pats[id], &rec, sizeof rec) == 0) { pid = id; break; }
            if (pid < 0) { pid = (int)pats.size(); pats.push_back(rec); v.push_back(pid); }
        }
        int64_t at = 0;
        if (!full) {
            const uint64_t key = hash_bytes(m, sizeof m, 0x2545F4914F6CDD1Dull);
            auto& v = mask_index[key];
            at = -1;
            for (int64_t pos : v) if (memcmp(mstore.data() + pos, m, sizeof m) == 0) { at = pos; break; }
            if (at < 0) { at = (int64_t)mstore.size(); mstore.insert(mstore.end(), m, m + 64); v.push_back(at); }
        }
        if (at > INT32_MAX - 64 || pats.size() > (size_t)1 << 20) return false;
        plan[ti] = Plan{pid, at, full ? 1 : 0, rec.nslots};
    }
    for (size_t ti = 0; ti < tiles.size(); ++ti) {
        tiles[ti].src_c = plan[ti].pat;
        tiles[ti].src_v = 0;
        tiles[ti].src_r = (int)plan[ti].mask_at;
        tiles[ti].spare = plan[ti].full;
        tiles[ti].maxlen = plan[ti].nslots;
        // LDS slot of each page: as a sweep planner set it (bit 25), else every page in its own order
        if (!(tiles[ti].vd_first & (1 << 25))) {
            int perm = 1 << 25;
            for (int p = 0; p < 6; ++p) perm |= p << (3 * p);
            tiles[ti].vd_first = perm;
        }
        tiles[ti].vd_count = 0;
    }
    if (mstore.empty()) mstore.assign(64, 0);
    patterns.swap(pats);
    masks.swap(mstore);
    return true;
}

bool plan_sweep_tiles(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, int max_pages, int max_waves,
                      SweepPlan& out) {
    if (n < 4096 || n_cols != n || max_pages > 6 || max_pages < 3) return false;
    {   // cheap look first: a row in the middle must reach at least 256 rows away, symmetrically, with n a multiple of that
        const int64_t r = n / 2;
        int64_t lo_o = 0, hi_o = 0;
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            lo_o = std::min<int64_t>(lo_o, (int64_t)indices[q] - r);
            hi_o = std::max<int64_t>(hi_o, (int64_t)indices[q] - r);
        }
        if (hi_o < 256 || lo_o != -hi_o || n % hi_o != 0) return false;
    }
    // the operator's distinct offsets col - row
    int nd = 0;
    int64_t off[kPatSlots];
    for (int64_t r = 0; r < n; ++r) {
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int64_t o = (int64_t)indices[q] - r;
            int u = 0;
            while (u < nd && off[u] != o) ++u;
            if (u == nd) {
                if (nd == kPatSlots) return false;
                off[nd++] = o;
            }
        }
    }
    if (nd < 3) return false;
    std::sort(off, off + nd);
    const int64_t Z = off[nd - 1];
    if (off[0] != -Z || Z < 256 || Z > (1 << 24) || n % Z != 0 || n / Z < 8) return false;
    // clusters of offsets that one page serves (gap <= 16 joins); the own cluster holds 0
    int ng = 0, gmin[8], gmax[8], own = -1;
    for (int u = 0; u < nd; ++u) {
        if (ng > 0 && off[u] - gmax[ng - 1] <= 16) { gmax[ng - 1] = (int)off[u]; }
        else {
            if (ng == max_pages) return false;
            gmin[ng] = gmax[ng] = (int)off[u]; ++ng;
        }
    }
    int span = 0;
    for (int g = 0; g < ng; ++g) {
        if (gmin[g] <= 0 && gmax[g] >= 0) own = g;
        span = std::max(span, gmax[g] - gmin[g]);
    }
    if (own < 0 || span > 16) return false;
    const int R = 64 - span;                                             // rows per tile: a cluster's columns fit one page
    // a cluster's page starts at rb + centre + (own cluster's minimum): the own page of one plane is the z page of the next
    int centre[8];
    for (int g = 0; g < ng; ++g) {
        // any centre in [gmax - max_own, gmin - min_own] lets the page serve the cluster; the middle of the cluster makes the
        // z clusters' centres exactly +-Z (their pages then ARE the own pages of the neighbouring planes)
        const int lo_c = gmax[g] - gmax[own], hi_c = gmin[g] - gmin[own];
        if (lo_c > hi_c) return false;                                            // wider than the own cluster
        int c = (gmin[g] + gmax[g]) / 2;
        c = c < lo_c ? lo_c : (c > hi_c ? hi_c : c);
        centre[g] = c;
        if (g > 0 && centre[g] - centre[g - 1] < 64) return false;               // pages of neighbouring clusters must not meet
    }
    const int64_t P = n / Z;
    const int C = (int)((Z + R - 1) / R);
    // every block of rows of a plane (C of them) is swept by `chunks` waves, each over a run of consecutive planes: as many
    // chunks as max_waves allows with runs of >= 8 planes; runs differ by one plane at most (the shorter ones end with an
    // empty tile), the waves are rounded up to a multiple of 4 by slots that hold empty tiles only
    int chunks = (int)std::min<int64_t>(max_waves / C, P / 8);
    if (chunks < 1 || (int64_t)C * chunks < max_waves / 8) return false;
    const int W = (C * chunks + 3) & ~3;
    const int64_t K = (P + chunks - 1) / chunks;
    const int64_t ntab = (int64_t)W * K;
    if (ntab >= (1 << 26)) return false;
    out.tiles.assign((size_t)ntab, WTile{});
    out.cw.assign((size_t)indptr[n] + 32, 0);
    out.waves = W; out.plane = (int)Z; out.rows_per_tile = R; out.chunks = chunks; out.most_pages = 0;
    std::atomic<bool> ok{true};
    auto sweep = [&](int s_begin, int s_end, int* most) {
        int64_t held[6];
        for (int s = s_begin; s < s_end && ok; ++s) {
            if (s >= C * chunks) continue;                                // padding slot: empty tiles only
            const int chunk = s / C, j = s % C;
            const int64_t p_first = P * chunk / chunks, p_end = P * (chunk + 1) / chunks;     // this wave's planes
            for (int p = 0; p < 6; ++p) held[p] = INT64_MIN;              // page start each LDS slot of this wave holds
            for (int64_t k = 0; k < K; ++k) {
                WTile& t = out.tiles[(size_t)(k * W + s)];
                const int64_t plane = p_first + k;
                if (plane >= p_end) continue;                             // a run one plane shorter: empty tile
                const int64_t rb = plane * Z + (int64_t)j * R, re = std::min(rb + R, (plane + 1) * Z);
                t.rb = (int)rb; t.re = (int)re; t.lo = indptr[rb]; t.hi = indptr[re];
                // logical pages: the clusters this tile has nonzeros in, ascending
                bool has[8] = {false, false, false, false, false, false, false, false};
                int maxlen = 0;
                for (int64_t r = rb; r < re; ++r) {
                    maxlen = std::max(maxlen, indptr[r + 1] - indptr[r]);
                    for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
                        const int64_t o = (int64_t)indices[q] - r;
                        int g = 0;
                        while (g < ng && !(o >= gmin[g] && o <= gmax[g])) ++g;
                        if (g == ng) { ok = false; return; }
                        has[g] = true;
                    }
                }
                has[own] = true;                                          // the fused epilogues read the row's own entry from the window
                int np = 0, lp_of[8];
                int64_t start[6];
                bool natural = true;
                for (int g = 0; g < ng; ++g) {
                    lp_of[g] = -1;
                    if (!has[g]) continue;
                    const int64_t st = rb + centre[g] + gmin[own];
                    if (st < 0) { natural = false; break; }               // at the very start of the vector: pages by the greedy cover, below
                    if (st + 64 > n_cols + 63) { ok = false; return; }
                    if (np > 0 && st < start[np - 1] + 64) { ok = false; return; }
                    lp_of[g] = np; start[np++] = st;
                }
                if (!natural) {
                    // the first tiles of the vector (a cluster's page would begin before column 0): the greedy cover of the row-order
                    // tilings, pages in their own order (no slot table, nothing carried in or out)
                    std::vector<WTile> one(1, t);
                    const int got = window_pages(one, 0, 1, n, n_cols, indptr, indices, max_pages, out.cw.data());
                    if (got < 0) { ok = false; return; }
                    t = one[0];
                    int perm = 0;
                    for (int p = 0; p < got; ++p) { perm |= p << (3 * p); held[p] = t.page_col[p]; }
                    t.vd_first = perm | (1 << 25);
                    t.vd_count = 0;
                    *most = std::max(*most, got);
                    continue;
                }
                const bool through_perm = true;
                for (int64_t r = rb; r < re; ++r)
                    for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
                        const int64_t o = (int64_t)indices[q] - r;
                        int g = 0;
                        while (!(o >= gmin[g] && o <= gmax[g])) ++g;
                        const int64_t w = (int64_t)indices[q] - start[lp_of[g]];
                        if (w < 0 || w > 63) { ok = false; return; }
                        out.cw[q] = (uint16_t)(lp_of[g] * 64 + w);
                    }
                // LDS slots: a page the wave's previous tile left behind stays where it is
                int slot[6], used = 0, carry = 0;
                for (int p = 0; p < np; ++p) {
                    slot[p] = -1;
                    if (!through_perm) continue;
                    for (int sl = 0; sl < 6; ++sl)
                        if (held[sl] == start[p] && !(used & (1 << sl))) { slot[p] = sl; used |= 1 << sl; carry |= 1 << p; break; }
                }
                if (!through_perm) { for (int p = 0; p < np; ++p) slot[p] = p; used = (1 << np) - 1; carry = 0; }
                for (int p = 0; p < np; ++p) {
                    if (slot[p] >= 0) continue;
                    int sl = 0;
                    while (used & (1 << sl)) ++sl;
                    slot[p] = sl; used |= 1 << sl;
                }
                for (int p = 0; p < np; ++p) held[slot[p]] = start[p];
                int perm = 0;
                for (int p = 0; p < np; ++p) perm |= slot[p] << (3 * p);
                t.vd_first = perm | (carry << 18) | (through_perm ? (1 << 24) : 0) | (1 << 25);
                t.vd_count = 0;
                t.geo = np | ((slot[lp_of[own]] * 64 + (int)(rb - start[lp_of[own]])) << 8);
                t.maxlen = maxlen;
                for (int p = 0; p < kWinMaxPages; ++p) t.page_col[p] = p < np ? (int)start[p] : 0;
                *most = std::max(*most, np);
            }
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    std::vector<int> most(nt, 0);
    std::vector<std::thread> th;
    const int per = (W + (int)nt - 1) / (int)nt;
    for (unsigned i = 0; i < nt; ++i) {
        const int a = std::min(W, (int)i * per), b = std::min(W, a + per);
        if (a < b) th.emplace_back([&, a, b, i] { sweep(a, b, &most[i]); });
    }
    for (auto& t : th) t.join();
    if (!ok) { out = SweepPlan{}; return false; }
    for (int m : most) out.most_pages = std::max(out.most_pages, m);
    return true;
}

namespace {

// images already in a store: hash -> positions of their first element
using ImageIndex = std::unordered_map<uint64_t, std::vector<int64_t>>;

// position of `img` (len elements) in `store`, at an address congruent to `pad` modulo `align`: an identical image
// already stored there if share, else appended
template <typename T>
int64_t place_image(std::vector<T>& store, ImageIndex& index, const T* img, int len, int pad, int align, bool share,
                    int64_t* n_images) {
    const uint64_t key = hash_bytes(img, (size_t)len * sizeof(T), 0xCBF29CE484222325ull + (uint64_t)pad);
    if (share) {
        auto it = index.find(key);
        if (it != index.end())
            for (int64_t pos : it->second)
                if ((pos % align) == pad && (size_t)pos + len <= store.size() && memcmp(store.data() + pos, img, (size_t)len * sizeof(T)) == 0)
                    return pos;
    }
    const size_t s = store.size();
    size_t pos = s - (s % align) + pad;
    if (pos < s) pos += align;
    store.resize(pos, T(0));
    store.insert(store.end(), img, img + len);
    if (share) index[key].push_back((int64_t)pos);
    ++*n_images;
    return (int64_t)pos;
}

}  // namespace

template <typename CW>
StreamStats share_window_streams(std::vector<WTile>& tiles, const int32_t* indptr, const CW* cw_in, const uint8_t* vidx_in,
                                 bool share, std::vector<CW>& cw_store, std::vector<uint8_t>& vidx_store,
                                 std::vector<uint16_t>& rel_store) {
    StreamStats st;
    ImageIndex ic, iv, ir;
    std::map<std::array<int, 8>, int> ids;
    cw_store.clear(); vidx_store.clear(); rel_store.clear();
    std::vector<uint16_t> rel;
    for (auto& t : tiles) {
        const int len = t.hi - t.lo, pad = t.lo & 15;
        t.src_c = (int)(place_image(cw_store, ic, cw_in + t.lo, len, pad, 16, share, &st.cw_images) - pad);
        t.src_v = vidx_in ? (int)(place_image(vidx_store, iv, vidx_in + t.lo, len, pad, 16, share, &st.vidx_images) - pad) : 0;
        rel.resize((size_t)(t.re - t.rb) + 1);
        for (int r = t.rb; r <= t.re; ++r) rel[(size_t)(r - t.rb)] = (uint16_t)(indptr[r] - t.lo);
        t.src_r = (int)place_image(rel_store, ir, rel.data(), (int)rel.size(), 0, 1, share, &st.rel_images);
        // image id: tiles that read the same bytes for all three streams, with the same table and the same shape, carry
        // the same non-zero number (the dictionary kernels keep a tile's decoded rows in registers while it repeats)
        const std::array<int, 8> key{t.src_c, t.src_v, t.src_r, t.vd_first, t.vd_count, len, t.re - t.rb, t.maxlen};
        auto it = ids.find(key);
        if (it == ids.end()) it = ids.emplace(key, (int)ids.size() + 1).first;
        t.spare = it->second;
    }
    // the kernels' 16-byte loads of the last image may run past its end; a lane of a short tile reads rel[0], rel[1]
    cw_store.resize(cw_store.size() + 32, CW(0));
    vidx_store.resize(vidx_store.size() + 32, 0);
    rel_store.resize(rel_store.size() + 8, 0);
    return st;
}
template StreamStats share_window_streams<uint8_t>(std::vector<WTile>&, const int32_t*, const uint8_t*, const uint8_t*, bool,
                                                   std::vector<uint8_t>&, std::vector<uint8_t>&, std::vector<uint16_t>&);
template StreamStats share_window_streams<uint16_t>(std::vector<WTile>&, const int32_t*, const uint16_t*, const uint8_t*, bool,
                                                    std::vector<uint16_t>&, std::vector<uint8_t>&, std::vector<uint16_t>&);

namespace {

// rows of one class run [a, b) in slice order for a sorting window of `sigma` rows: windows of sigma consecutive rows,
// inside a window the rows by descending length (stable: rows of equal length keep their order, so an operator whose
// rows all have the window's length keeps consecutive rows); sigma <= 64: no sorting
void sell_run_order(const int32_t* indptr, int64_t a, int64_t b, int sigma, int trip, std::vector<int32_t>& perm) {
    const size_t first = perm.size();
    for (int64_t r = a; r < b; ++r) perm.push_back((int32_t)r);
    if (sigma <= 64) return;
    for (int64_t w = a; w < b; w += sigma) {
        int32_t* lo = perm.data() + first + (w - a);
        int32_t* hi = perm.data() + first + (std::min<int64_t>(w + sigma, b) - a);
        // by descending number of TRIPS, not of nonzeros: a wave walks a slice trip by trip (`trip` positions each), rows of 57 and of
        // 64 nonzeros cost it the same eight -- and within a class the rows keep their order, so a slice's lanes hold longer runs of
        // neighbouring rows (whose gathers share cache lines: what a sorted slice loses against consecutive rows, r04_sweeps.md B)
        std::stable_sort(lo, hi, [&](int32_t x, int32_t y) { return (indptr[x + 1] - indptr[x] + trip - 1) / trip > (indptr[y + 1] - indptr[y] + trip - 1) / trip; });
    }
}

// padded nonzeros of the sliced layout for a sorting window of sigma rows
int64_t sell_padded(int64_t n, const int32_t* indptr, const uint8_t* row_class, int sigma, int trip) {
    int64_t tot = 0, r = 0;
    std::vector<int32_t> perm;
    while (r < n) {
        int64_t e = r + 1;
        if (row_class) { while (e < n && (row_class[e] != 0) == (row_class[r] != 0)) ++e; } else e = n;
        perm.clear();
        sell_run_order(indptr, r, e, sigma, trip, perm);
        for (size_t i = 0; i < perm.size(); i += 64) {
            int w = 0;
            for (size_t j = i; j < std::min(perm.size(), i + 64); ++j) w = std::max(w, indptr[perm[j] + 1] - indptr[perm[j]]);
            tot += (int64_t)((w + 1) & ~1) * 64;
        }
        r = e;
    }
    return tot;
}

// The dominant FAR column offset of the operator, in rows: a 3-D discretisation in natural ordering couples row r to rows
// r +- (one grid plane); 0 if no offset beyond 1024 rows carries a sizeable share of the nonzeros.  (Sampled.)
int64_t sell_far_stride(int64_t n, const int32_t* indptr, const int32_t* indices) {
    if (n < 16384) return 0;
    const int64_t bins = (n + 63) / 64 + 1;
    std::vector<int64_t> cnt((size_t)bins, 0);
    int64_t sampled = 0;
    const int64_t step = std::max<int64_t>(1, n / 20000);
    for (int64_t r = 0; r < n; r += step)
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int64_t c = indices[q];
            if (c >= n) continue;                                  // ghost columns: numbered behind the owned ones
            const int64_t d = c > r ? c - r : r - c;
            ++cnt[(size_t)(d >> 6)];
            ++sampled;
        }
    if (sampled == 0) return 0;
    // smoothed over +-6 bins (the plane's own line neighbours spread the peak), offsets of 1024 rows and more
    int64_t best = 0, best_bin = -1;
    for (int64_t k = 16; k < bins; ++k) {
        int64_t s = 0;
        for (int64_t j = std::max<int64_t>(16, k - 6); j <= std::min(bins - 1, k + 6); ++j) s += cnt[(size_t)j];
        if (s > best) { best = s; best_bin = k; }
    }
    if (best_bin < 0 || best * 100 < sampled * 15) return 0;
    // centre of mass of the peak, from the sampled offsets themselves
    double num = 0.0, den = 0.0;
    for (int64_t r = 0; r < n; r += step)
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int64_t c = indices[q];
            if (c >= n) continue;
            const int64_t d = c > r ? c - r : r - c;
            const int64_t k = d >> 6;
            if (k >= best_bin - 6 && k <= best_bin + 6) { num += (double)d; den += 1.0; }
        }
    return den > 0.0 ? (int64_t)(num / den + 0.5) : 0;
}

}  // namespace

bool plan_sell(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, const uint8_t* row_class,
               const SellOptions& opt, SellPlan& out) {
    out = SellPlan{};
    const int64_t nnz = indptr[n];
    // a slice is as wide as its longest row and ONE wave walks it: an operator with a few rows far longer than the rest (the
    // random-offset stand-in s4: one row of 46,694 nonzeros against a mean of 76) would end its launch on that one wave
    // (s4: 4.97 ms against 1.05 ms on the CSR-adaptive kernels, which sum a long row with the whole wave) -- refused
    {
        int32_t longest = 0;
        for (int64_t r = 0; r < n; ++r) longest = std::max(longest, indptr[r + 1] - indptr[r]);
        if (longest > 1024 && (int64_t)longest * n > 16 * std::max<int64_t>(nnz, 1)) return false;
    }
    // --- column RUNS: if every row consists of aligned runs of 3 consecutive columns (an assembled matrix with three unknowns
    // per node and full 3 x 3 blocks: Queen_4147's structure), ONE code per run is stored instead of one per nonzero
    // (8 + 2/3 instead of 10 bytes per nonzero)
    int run = 1;
    if (opt.allow_runs && nnz >= 3) {
        bool ok3 = true;
        for (int64_t r = 0; r < n && ok3; ++r) {
            const int32_t lo = indptr[r], len = indptr[r + 1] - lo;
            if (len % 3) { ok3 = false; break; }
            for (int32_t q = 0; q < len; q += 3)
                if (indices[lo + q + 1] != indices[lo + q] + 1 || indices[lo + q + 2] != indices[lo + q] + 2) { ok3 = false; break; }
        }
        if (ok3) run = 3;
    }
    out.run = run;
    // --- the sorting window (SELL-C-sigma): 64 (rows stay consecutive: coalesced row operands, the smallest gather footprint)
    // while that pads by at most opt.target64; else the smallest of 256, 1024, 4096 whose padding is within opt.target of
    // the nonzeros, else the one that pads least
    int sigma = opt.sigma;
    bool prefer_window = false;              // consecutive rows with WINDOW codes although a sorting window would pad less
    if (sigma <= 0) {
        int64_t best_pad = -1;
        int64_t pads[5];
        const int cands[5] = {64, 256, 1024, 4096, 16384};
        for (int i = 0; i < 5; ++i) {
            pads[i] = sell_padded(n, indptr, row_class, cands[i], 8 * run);
            if (best_pad < 0 || pads[i] < best_pad) best_pad = pads[i];
            if (i == 0 && (double)pads[0] <= opt.target64 * (double)std::max<int64_t>(nnz, 1)) break;
        }
        sigma = 64;
        if ((double)pads[0] > opt.target64 * (double)std::max<int64_t>(nnz, 1)) {
            int chosen = 0;
            for (int i = 1; i < 5; ++i)
                if ((double)pads[i] <= opt.target * (double)std::max<int64_t>(nnz, 1) || (double)pads[i] <= 1.01 * (double)best_pad) { sigma = cands[i]; chosen = i; break; }
            // sorted slices hold rows from anywhere in their window: their operands are gathered from memory, and those gathers
            // cost more than the padding of consecutive rows does while that stays below ~1.4 x (s4c at Queen size: 30 % more
            // bytes, 9 % less time: r04_sweeps.md B) -- consecutive rows are tried first up to opt.window_prefer x the sorted bytes
            if (opt.window_granules > 0 && chosen > 0 && (double)pads[0] <= opt.window_prefer * (double)pads[chosen] &&
                (double)pads[0] <= opt.window_max_overhead * (double)std::max<int64_t>(nnz, 1))
                prefer_window = true;
        }
    }
    if (sigma < 64) sigma = 64;
    const int sigma_sorted = sigma;
    // pass 1: slices, widths, offsets.  A row's columns are stored as 16-bit DELTAS (prcg_plan.h); a gap too wide for one
    // costs the row skip entries, i.e. stored positions: stored length = nonzeros + skips.  With WINDOW codes (tried first
    // when the rows stay consecutive) there are no skips; the attempt ends at the first slice with too many granules
    std::vector<SellSlice> all;
    std::vector<uint8_t> cls_of;
    std::vector<int32_t>& perm = out.rows;        // (row, stored length) pairs in lane order, 64 pairs per slice that has them
    std::vector<int32_t> rrun;
    auto skips_of = [&](int32_t row, int32_t base) {       // extra stored POSITIONS of the row: `run` per skip code
        int sk = 0;
        int64_t prev = base;
        for (int32_t q = indptr[row]; q < indptr[row + 1]; q += run) {
            int64_t dlt = (int64_t)indices[q] - prev;
            if (dlt > kSellDeltaMax) sk += (int)((dlt - kSellDeltaMax + kSellSkipFwd - 1) / kSellSkipFwd);
            else if (dlt < kSellDeltaMin) sk += (int)((kSellDeltaMin - dlt + kSellSkipBack - 1) / kSellSkipBack);
            prev = indices[q];
        }
        return sk * run;
    };
    int64_t voff = 0, coff = 0;
    bool windowed = opt.window_granules > 0 && (sigma == 64 || prefer_window) && opt.window_granules * kSellGranule <= 65536;
    std::vector<int32_t> cols;
    for (int attempt = 0; attempt < 2; ++attempt) {
        all.clear(); cls_of.clear(); perm.clear(); out.gran.clear(); out.window = 0;
        voff = 0; coff = 0;
        bool failed = false;
        sigma = windowed ? 64 : sigma_sorted;
        out.sigma = sigma;
        int64_t r = 0;
        while (r < n && !failed) {
            const uint8_t cls = row_class ? (row_class[r] != 0) : 0;
            int64_t e = r + 1;
            if (row_class) { while (e < n && (row_class[e] != 0) == cls) ++e; } else e = n;
            rrun.clear();
            sell_run_order(indptr, r, e, sigma, 8 * run, rrun);
            for (size_t i = 0, take = 64; i < rrun.size() && !failed; i += take) {
                take = 64;
                size_t je = std::min(rrun.size(), i + take);
                int32_t rmin = INT32_MAX, cbase = INT32_MAX;
                for (size_t j = i; j < je; ++j) {
                    rmin = std::min(rmin, rrun[j]);
                    if (indptr[rrun[j] + 1] > indptr[rrun[j]]) cbase = std::min(cbase, indices[indptr[rrun[j]]]);
                }
                if (cbase == INT32_MAX) cbase = 0;
                int width = 0, flags = 0;
                bool any_skip = false;
                if (windowed) {
                    // the slice's granules: the sorted distinct (run-first) columns, each covered together with its run.  A slice
                    // whose rows touch more than the kernels' window holds is cut: half the rows, and so on (its other lanes idle)
                    const size_t g0 = out.gran.size();
                    int ng = 0;
                    for (;;) {
                        cols.clear();
                        width = 0;
                        for (size_t j = i; j < je; ++j) {
                            for (int32_t q = indptr[rrun[j]]; q < indptr[rrun[j] + 1]; q += run) cols.push_back(indices[q]);
                            width = std::max(width, indptr[rrun[j] + 1] - indptr[rrun[j]]);
                        }
                        std::sort(cols.begin(), cols.end());
                        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
                        out.gran.resize(g0);
                        for (int32_t c : cols)
                            if (out.gran.size() == g0 || c + run - 1 > out.gran.back() + kSellGranule - 1) out.gran.push_back(c);
                        ng = (int)(out.gran.size() - g0);
                        if (ng <= opt.window_granules || take == 1) break;
                        take /= 2;
                        je = std::min(rrun.size(), i + take);
                    }
                    if (ng > opt.window_granules || out.gran.size() >= (size_t)INT32_MAX / 2) { failed = true; break; }
                    out.window = std::max(out.window, ng);
                    cbase = (int)g0;
                    flags = 2 | (ng << 8);
                } else {
                    for (size_t j = i; j < je; ++j) {
                        const int sk = skips_of(rrun[j], cbase);
                        any_skip |= sk > 0;
                        width = std::max(width, indptr[rrun[j] + 1] - indptr[rrun[j]] + sk);
                    }
                }
                const int w2 = (width + 1) & ~1, w8 = ((width + run - 1) / run + 7) & ~7;      // value slots (even), codes (whole 16-byte chunks)
                if (voff + (int64_t)w2 * 64 >= (int64_t)INT32_MAX - 4096 || coff + (int64_t)w8 * 64 >= (int64_t)INT32_MAX - 4096) return false;
                int rows_off = -1;
                if (sigma > 64 || any_skip) {
                    rows_off = (int)(perm.size() / 2);
                    for (size_t j = i; j < je; ++j) { perm.push_back(rrun[j]); perm.push_back(indptr[rrun[j] + 1] - indptr[rrun[j]] + skips_of(rrun[j], cbase)); }
                    for (size_t j = je; j < i + 64; ++j) { perm.push_back(-1); perm.push_back(0); }
                }
                // (rows_off < 0: rb .. re are the slice's rows; else re - rb is their count and rb the smallest of them)
                const int first = rows_off < 0 ? rrun[i] : rmin;
                all.push_back(SellSlice{first, first + (int)(je - i), (int)voff, (int)coff, width, cbase, rows_off, flags});
                cls_of.push_back(cls);
                voff += (int64_t)w2 * 64;
                coff += (int64_t)w8 * 64;
            }
            r = e;
        }
        // (cut slices idle lanes: the padding bound holds for what was built, not for what was estimated)
        if (windowed && !failed && (double)voff > std::max(opt.max_overhead, prefer_window ? opt.window_max_overhead : 0.0) * (double)std::max<int64_t>(nnz, 1)) failed = true;
        if (!failed) break;
        windowed = false;                      // (second attempt: delta codes, the sorting window chosen above)
    }
    if (getenv("PRCG_PLAN_DEBUG"))
        fprintf(stderr, "plan_sell: sigma %d, %zu slices, padded / nnz = %.4f\n", sigma, all.size(), (double)voff / (double)std::max<int64_t>(nnz, 1));
    if (!windowed && (double)voff > opt.max_overhead * (double)std::max<int64_t>(nnz, 1)) return false;
    out.padded_nnz = voff;
    out.col_entries = coff;
    // pass 2: fill (padding: value 0, delta 0 -- the column stays where it is)
    out.val.assign((size_t)voff + 1024, 0.0);
    out.col.assign((size_t)coff + 1024, windowed ? (uint16_t)0 : (uint16_t)kSellDeltaBias);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (all.size() < 1024) nt = 1;
    auto fill = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            const SellSlice& sl = all[i];
            for (int l = 0; l < sl.re - sl.rb; ++l) {
                const int row = sl.rows_off < 0 ? sl.rb + l : perm[2 * ((size_t)sl.rows_off + l)];
                const int32_t lo = indptr[row];
                const int len = indptr[row + 1] - lo;
                int64_t prev = sl.cbase;
                int u = 0;                                  // stored position (a multiple of `run` at every run start)
                auto put = [&](uint16_t code) { const int c = u / run; out.col[(size_t)sl.coff + ((size_t)(c >> 3) * 64 + l) * 8 + (c & 7)] = code; };
                if (sl.flags & 2) {                         // WINDOW codes: 16 g + offset inside the last granule that starts at or before the column
                    const int32_t* g0 = out.gran.data() + sl.cbase;
                    const int ng = sl.flags >> 8;
                    for (int q = 0; q < len; q += run) {
                        const int g = (int)(std::upper_bound(g0, g0 + ng, indices[lo + q]) - g0) - 1;
                        put((uint16_t)(g * kSellGranule + (indices[lo + q] - g0[g])));
                        for (int e2 = 0; e2 < run; ++e2, ++u)
                            out.val[(size_t)sl.voff + ((size_t)(u >> 1) * 64 + l) * 2 + (u & 1)] = data[lo + q + e2];
                    }
                    continue;
                }
                for (int q = 0; q < len; q += run) {
                    int64_t dlt = (int64_t)indices[lo + q] - prev;
                    while (dlt > kSellDeltaMax) { put(kSellCodeSkipFwd); u += run; prev += kSellSkipFwd; dlt -= kSellSkipFwd; }
                    while (dlt < kSellDeltaMin) { put(kSellCodeSkipBack); u += run; prev -= kSellSkipBack; dlt += kSellSkipBack; }
                    put((uint16_t)(dlt + kSellDeltaBias));
                    for (int e2 = 0; e2 < run; ++e2, ++u)
                        out.val[(size_t)sl.voff + ((size_t)(u >> 1) * 64 + l) * 2 + (u & 1)] = data[lo + q + e2];
                    prev = indices[lo + q];
                }
            }
        }
    };
    if (nt == 1) fill(0, all.size());
    else {
        std::vector<std::thread> th;
        const size_t per = (all.size() + nt - 1) / nt;
        for (unsigned i = 0; i < nt; ++i) {
            const size_t a = std::min(all.size(), i * per), b = std::min(all.size(), a + per);
            th.emplace_back(fill, a, b);
        }
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < all.size(); ++i) (cls_of[i] ? out.s1 : out.s0).push_back(all[i]);
    // --- processing order of the class-0 slices (any order is correct; the table IS the order: the waves of a launch
    // take consecutive entries).  A 3-D discretisation couples row r to r +- one grid plane: in row order the slices that
    // read a row as their plane neighbour are a plane apart -- processed in another XCD (each has its own L2) or long after
    // the row has left the L2, so every vector entry crosses the fabric ~7 times (s4b, r03: 1.14 x the bytes that must
    // move).  Here groups of opt.planes consecutive planes are INTERLEAVED: the slices at the same place of planes
    // z, z+1, ... are neighbours in the table, hence processed at the same time by neighbouring workgroups of one XCD.
    out.stride_rows = opt.planes > 1 ? sell_far_stride(n, indptr, indices) : 0;
    if (out.stride_rows > 0 && out.s0.size() >= 4096) {
        const int64_t beta = out.stride_rows;
        const int G = opt.planes;
        const int64_t bin = sigma;                         // rows per place: one slice (or one sorting window of slices)
        std::vector<std::array<int64_t, 4>> key(out.s0.size());
        for (size_t i = 0; i < out.s0.size(); ++i) {
            const int64_t rb = out.s0[i].rb;
            const int64_t z = rb / beta, y = rb - z * beta;
            key[i] = {z / G, y / bin, z, (int64_t)i};
        }
        std::sort(key.begin(), key.end());
        std::vector<SellSlice> ord(out.s0.size());
        for (size_t i = 0; i < key.size(); ++i) ord[i] = out.s0[(size_t)key[i][3]];
        out.s0.swap(ord);
        out.planes = G;
    }
    return true;
}

bool plan_medium(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, int max_groups, int max_slices,
                 int max_window, MediumPlan& out) {
    out = MediumPlan{};
    if (n < 1 || n > (int64_t)max_groups * 16 * max_slices * 64) return false;
    SellOptions so;
    so.max_overhead = 8.0;           // (short ragged rows pad heavily; the whole operator stays in the L2 anyway)
    so.planes = 0;
    so.allow_runs = false;           // (the few-workgroup solver reads one code per nonzero)
    if (!plan_sell(n, indptr, indices, data, nullptr, so, out.sell)) return false;
    const int nsl = (int)out.sell.s0.size();
    // slices per wave; a workgroup (16 waves) holds whole sorting windows, so that its rows are one contiguous range
    const int per_window = std::max(1, out.sell.sigma / 64);
    int per = 1;
    for (; per <= max_slices; ++per)
        if ((16 * per) % per_window == 0 && (int64_t)max_groups * 16 * per >= nsl) break;
    if (per > max_slices) return false;
    // (no more workgroups than the slices need; each at least one wave's worth)
    int G = (nsl + 16 * per - 1) / (16 * per);
    if (G < 1) G = 1;
    // spread over as many workgroups as give every wave a slice: a smaller `per` where the windows allow it
    const int W = G * 16;
    out.groups = G;
    out.wave_first.resize((size_t)W + 1);
    for (int w = 0; w <= W; ++w) out.wave_first[(size_t)w] = std::min(nsl, w * per);
    out.window.assign((size_t)2 * G, 0);
    out.own.assign((size_t)2 * G, 0);
    auto slice_row = [&](const SellSlice& sl, int l) { return sl.rows_off < 0 ? sl.rb + l : out.sell.rows[2 * ((size_t)sl.rows_off + l)]; };
    int64_t prev_hi = 0;
    for (int g = 0; g < G; ++g) {
        int64_t rmin = INT64_MAX, rmax = -1, cmin = INT64_MAX, cmax = -1;
        const int s_lo = out.wave_first[(size_t)g * 16], s_hi = out.wave_first[(size_t)(g + 1) * 16];
        for (int si = s_lo; si < s_hi; ++si) {
            const SellSlice& sl = out.sell.s0[(size_t)si];
            for (int l = 0; l < sl.re - sl.rb; ++l) {
                const int row = slice_row(sl, l);
                rmin = std::min<int64_t>(rmin, row); rmax = std::max<int64_t>(rmax, row);
                for (int32_t q = indptr[row]; q < indptr[row + 1]; ++q) {
                    if (indices[q] < 0 || indices[q] >= n) return false;
                    cmin = std::min<int64_t>(cmin, indices[q]); cmax = std::max<int64_t>(cmax, indices[q]);
                }
            }
        }
        if (rmax < 0) { rmin = prev_hi; rmax = prev_hi - 1; }
        if (rmin < prev_hi) return false;                 // (the workgroups' row ranges must not interleave)
        prev_hi = rmax + 1;
        out.own[(size_t)2 * g] = (int32_t)rmin;
        out.own[(size_t)2 * g + 1] = (int32_t)(rmax + 1);
        cmin = std::min(cmin, rmin); cmax = std::max(cmax, rmax);
        if (cmax < 0) { cmin = 0; cmax = 0; }
        // (a skip code lands between two columns of a row, a padded position stays on the row's last column: inside the window)
        const int64_t wlen = cmax - cmin + 1;
        if (wlen > max_window) return false;
        out.window[(size_t)2 * g] = (int32_t)cmin;
        out.window[(size_t)2 * g + 1] = (int32_t)wlen;
        out.window_pairs = std::max(out.window_pairs, (int)wlen);
        for (int si = s_lo; si < s_hi; ++si) {
            SellSlice& sl = out.sell.s0[(size_t)si];
            bool outside = false;
            for (int l = 0; l < sl.re - sl.rb && !outside; ++l) {
                const int row = slice_row(sl, l);
                for (int32_t q = indptr[row]; q < indptr[row + 1]; ++q)
                    if (indices[q] < rmin || indices[q] > rmax) { outside = true; break; }
            }
            sl.flags = outside ? 1 : 0;
        }
    }
    return true;
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
