#pragma once
#include <stdint.h>

#include <vector>

namespace prcg {

#ifndef PRCG_TILE_DEFINED
#define PRCG_TILE_DEFINED
struct alignas(16) Tile { int row_begin, row_end, nnz_begin, nnz_end; };
#endif

void plan_tiles(int64_t n, const int32_t* indptr, const uint8_t* row_class,
                int cap_nnz, int cap_rows,
                std::vector<Tile>& class0, std::vector<Tile>& class1);

// ---- window tiles (row-per-lane kernels, prcg_win.hip) -----------------------------------
// A window tile is a run of at most rows_per_tile consecutive rows of one class with at most
// cap_nnz nonzeros whose columns (plus the tile's own rows) are covered by at most max_pages
// PAGES of 64 consecutive columns.  The kernel stages the pages of the input vector in LDS once
// per tile and each lane walks its own row; the column of a nonzero is streamed as its index into
// that staged window (page * 64 + offset) -- 1 byte when max_pages <= 4, else 2.
#ifndef PRCG_WTILE_DEFINED
#define PRCG_WTILE_DEFINED
constexpr int kWinMaxPages = 12;
struct alignas(16) WTile {
    int rb, re, lo, hi;                  // rows [rb,re), nonzeros [lo,hi)
    int geo, maxlen, vd_first, vd_count;      // geo = pages in use | (window index of row rb) << 8; longest row;
                                         // value dictionary {first entry, count}
    int page_col[kWinMaxPages];          // first column of each page
    // where the kernel reads the tile's encoded streams (share_window_streams): 16-aligned start of the
    // window-index image / of the value-index image (elements), start of the relative row pointers
    int src_c, src_v, src_r, spare;      // spare: image id (share_window_streams), equal for tiles that read identical streams
};
#endif
struct WinPlan {
    std::vector<WTile> t0, t1;           // interior tiles, tiles touching ghost columns
    std::vector<uint16_t> cw;            // per nonzero: index into its tile's staged window
    bool ok0 = false, ok1 = false;       // every tile of the class qualified
    int pages0 = 0, pages1 = 0;          // most pages any tile of the class needs
};
// n_cols: owned + ghost columns.  Tiles of a class that does not qualify are still listed (their
// cw entries are unspecified); the caller then keeps the CSR-adaptive kernels for that class.
void plan_window_tiles(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                       const uint8_t* row_class, int rows_per_tile, int cap_nnz, int max_pages,
                       WinPlan& out);
// Per-tile value dictionaries (at most dict_max distinct bit patterns per tile, each tile's table
// starting at an even entry): vidx[q] = index of data[q] in its tile's table.  Returns false (and
// leaves the tiles' vd_* zero) if some tile needs more entries.
bool plan_window_dict(std::vector<WTile>& tiles, const double* data, int dict_max,
                      std::vector<uint8_t>& vidx, std::vector<double>& vdict);

// Tile images.  The kernels read three encoded streams per tile: the window indices of its nonzeros
// (cw, 1 or 2 bytes each), the value-dictionary indices (vidx, 1 byte each; absent with plain values)
// and the row pointers relative to the tile's first nonzero (rel, 2 bytes per row + 1).  Tiles whose
// image of a stream is byte-identical SHARE one copy: a band or a stencil repeats the same few images
// over and over (the structure is translation invariant, and so are the value indices when the
// coefficients repeat), so the kernels find them in L2 instead of streaming them from HBM.  Lossless:
// what a tile reads is exactly what it would have read from its own copy.
//   cw_in / vidx_in: per nonzero, as plan_window_tiles / plan_window_dict wrote them (vidx_in null:
//   plain values).  On return the *_store vectors hold the images (the cw / vidx image of a tile starts
//   at a multiple of 16 elements and is preceded by lo % 16 elements of padding, as in the per-nonzero
//   layout, so the kernels' 16-byte loads stay aligned), and every tile's src_c / src_v / src_r is set.
//   share = false stores every tile's own image (the layout then equals the per-nonzero one).
struct StreamStats { int64_t cw_images = 0, vidx_images = 0, rel_images = 0; };
template <typename CW>
StreamStats share_window_streams(std::vector<WTile>& tiles, const int32_t* indptr, const CW* cw_in, const uint8_t* vidx_in,
                                 bool share, std::vector<CW>& cw_store, std::vector<uint8_t>& vidx_store,
                                 std::vector<uint16_t>& rel_store);

// ---- pattern tiles: constant-coefficient stencils without index streams --------------------------------
// A 64-row window tile is a PATTERN tile if its rows share one sequence of at most kPatSlots "slots": slot u is the
// nonzero at window index (lane + cb[u]) with value val[u], the same for every row that has it; a row is then its
// 16-bit presence mask over the slots (rows at the edge of a grid lack neighbours).  Sorted, duplicate-free rows only
// (a row's slots ascend with its columns, so the left-to-right sum of csr_matvec is the sum over its present slots in
// slot order); at most kPatValues distinct bit patterns among a tile's values.  Lossless: (pattern, masks, pages)
// reproduce every nonzero's column and value bits exactly -- tests/test_abi_and_planning.py rebuilds the matrix.
// A 5- / 7- / 9-point stencil with constant coefficients qualifies tile by tile (few patterns in all); an operator is
// taken as a whole or not at all.  The kernels then read NO per-nonzero stream: per tile the pattern (scalar loads, a few
// records for the whole operator) and, for tiles with incomplete rows, 128 bytes of masks (shared between identical
// tiles like the other stream images).
#ifndef PRCG_PATREC_DEFINED
#define PRCG_PATREC_DEFINED
constexpr int kPatSlots = 16;
constexpr int kPatValues = 4;
struct alignas(8) PatRec {
    int nslots;                 // U
    unsigned vsel;              // 2 bits per slot: which of val[] the slot's value is
    short cb[kPatSlots];        // slot u sits at window index lane + cb[u]  (may be negative for lanes without the slot)
    double val[kPatValues];     // the caller's doubles, bit for bit
};
static_assert(sizeof(PatRec) == 72, "the kernels read a pattern record with scalar loads");
#endif
// tiles: window tiles planned with 64 rows per tile (geo, page_col set), cw: their window indices.  On success every
// tile's src_c = pattern id, src_r = start of its 64 masks in `masks`, spare = 1 if all 64 rows have every slot (no
// mask needed) else 0, maxlen = number of slots.  Returns false if some tile does not qualify (tiles unchanged then).
bool plan_window_patterns(std::vector<WTile>& tiles, const int32_t* indptr, const uint16_t* cw, const double* data,
                          std::vector<PatRec>& patterns, std::vector<uint16_t>& masks);

// ---- sweep order for stencils on a regular grid (pattern tiles only) ----------------------------------------
// A 3-D (2-D) stencil reaches one grid plane (line) up and down: rows r - Z and r + Z.  In row order a wave's consecutive
// tiles are far apart and every tile stages its z-, own and z+ pages anew (S2: 6 pages of (r,s) pairs per 64 rows against
// a band's 2 -- what bounds it, profiles/r03_sweeps.md K).  Here the tile TABLE is ordered so that the tiles one wave takes
// one after the other (slot, slot + waves, slot + 2 waves, ...) are the same block of rows in CONSECUTIVE planes, and the
// pages are cut so that the own page of plane k IS the z- page of plane k+1 and its z+ page the own page: a page that the
// wave's previous tile left in LDS is not loaded again (`carry`), only its place in the window changes (`perm`: logical
// page -> LDS slot).  Requirements: no ghost columns, n a multiple of Z, every row's offsets from the operator's <= 16
// distinct ones, offset clusters (z-, y-, own, y+, z+, ...) at most kWinPatPages, tiles of 64 - (width of the widest
// cluster) rows inside one plane.  The launch must run exactly `waves` waves (the kernel checks and otherwise ignores the
// carry bits: every page is loaded, which is always correct).
//   tile.vd_first = perm (3 bits per logical page) | carry << 18 | (1 << 24 if slots are addressed through perm) | 1 << 25;
//   tile.geo's own-row index is PHYSICAL (slot * 64 + offset); cw[q] = logical page * 64 + offset inside it.
// Empty tiles (rb == re) pad the table where a plane's blocks do not fill the last workgroup.
struct SweepPlan {
    std::vector<WTile> tiles;      // table order
    std::vector<uint16_t> cw;      // per nonzero (CSR order)
    int waves = 0;                 // waves of the launch the carry bits assume
    int plane = 0, rows_per_tile = 0, chunks = 0, most_pages = 0;
};
bool plan_sweep_tiles(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, int max_pages, int max_waves,
                      SweepPlan& out);

// ---- sliced rows (prcg_sell.hip) ---------------------------------------------------------------
// Slices of up to 64 rows of one class (interior slices first).  Within a slice of width w (its longest STORED row),
// stored position u of the row in lane l holds the value val[voff + ((u/2)*64 + l)*2 + u%2] and the 16-bit code
// col16[coff + ((u/8)*64 + l)*8 + u%8].  The codes of a row are column DELTAS: starting from the slice's `cbase` (the
// smallest first column of its rows), every position moves the lane's running column by code - 16384 and -- unless the
// code is 0 or 65535 -- names a nonzero at the column reached.  Codes 0 / 65535 are SKIPS (move by -16384 / +49151, no
// nonzero, value slot unused): a gap between consecutive columns of a row (or an unsorted row's step back) wider than
// one code costs the row extra stored positions, so any operator can be encoded, and one whose rows' sorted columns lie
// within 49,150 of each other -- an assembled FEM matrix of any size: a grid plane of 111 x 111 nodes x 3 is 37,000 --
// needs none.  (A fixed 16-bit offset from a per-slice base, round 3's encoding, refused every operator whose slices
// span 65,536 columns: Queen_4147's size.)  Shorter rows are padded with value 0 / code 16384 (column stays; never
// multiplied: the kernels mask by the stored length).  A TRIP (positions 8 j .. 8 j + 7) is what the wave reads with four
// 16-byte value loads and one 16-byte code load per lane.  The arrays end with a whole trip of padding.
// RUNS (SellPlan::run = 3): when every row of the operator consists of aligned runs of three consecutive columns -- three
// unknowns per node, full 3 x 3 blocks: what an assembled 3-D elasticity matrix such as Queen_4147 is -- ONE code is stored
// per run: code c of a row covers its stored positions 3 c .. 3 c + 2, moves the running column to the run's FIRST column
// (delta from the previous run's first column) and the positions are the columns +0, +1, +2; a skip code costs a whole run
// of unused value slots; code c sits at col16[coff + ((c/8)*64 + l)*8 + c%8]: 8 + 2/3 bytes per nonzero instead of 10.
// Which rows: rows_off = -1: the slice holds the consecutive rows [rb, re), their stored lengths are the row pointers'
// (no skips).  Otherwise lane l holds row rows[2 * (rows_off + l)] with stored length rows[2 * (rows_off + l) + 1]
// (64 pairs per slice; row -1, length 0 behind the last) and rb is the smallest of them: slices of a sorting window wider
// than 64 (SELL-C-sigma: every window of sigma consecutive rows of one class is sorted by descending number of TRIPS --
// ceil(length / (8 run)), what a wave pays for a row; stable, so rows of one trip count stay neighbours -- before it is cut
// into slices, so that rows of similar length share a slice and the padding stays small for operators whose row lengths vary), and slices with a row that needs skips.  Every row is still summed left to right
// by ONE lane: the products are scipy's bit for bit whatever the order.
// WINDOW codes (SellPlan::window > 0; every slice holds consecutive rows): the kernel stages the input-vector entries a slice's
// rows touch in LDS and the row walk reads them there -- per nonzero an LDS read instead of a gather from memory (the gathers'
// instructions, not their misses, cost the Queen-size stand-in 95 of 600 us: r04_sweeps.md B).  The entries are described by
// GRANULES of 16 consecutive columns (first column in `gran`; granules may overlap): granule g of the slice is staged at window
// entries 16 g .. 16 g + 15, four granules per wave-wide load.  A code is then the WINDOW INDEX of the (run's first) column,
// 16 g + (column - gran[g]); a run never straddles a granule; no deltas, no skips, padding code 0.  SellSlice::cbase = the
// slice's first entry in `gran`, flags = 2 | granules << 8.  Chosen when EVERY slice needs at most
// SellOptions::window_granules granules (an assembled 3-D matrix in natural ordering: 9 lines of neighbours per slice, 45
// granules for 3 unknowns per node); otherwise delta codes as above.
// Returns false (nothing built) if the operator does not qualify: padding would exceed `max_overhead` x nnz.
constexpr int kSellGranule = 16;
constexpr int kSellDeltaBias = 16384;
constexpr int kSellDeltaMin = -16383, kSellDeltaMax = 49150;       // deltas one code can hold (codes 1 .. 65534)
constexpr int kSellSkipFwd = 49151, kSellSkipBack = 16384;         // what the skip codes 65535 / 0 move the column by
constexpr uint16_t kSellCodeSkipFwd = 65535, kSellCodeSkipBack = 0;
struct SellSlice { int rb, re, voff, coff, width, cbase, rows_off, flags; };
static_assert(sizeof(SellSlice) == 32, "the kernels read a slice descriptor as two int4");
struct SellOptions {
    double max_overhead = 1.25;      // most padded nonzeros per nonzero
    int sigma = 0;                   // sorting window in rows (64: none); 0: 64 if that pads <= target64, else the smallest of 256,
    double target64 = 1.06;          //    1024, 4096, 16384 that pads <= target or within 1 % of the one that pads least
    double target = 1.04;
    int planes = 8;                  // class-0 slices of this many consecutive grid planes are interleaved in the table (<= 1: row order)
    bool allow_runs = true;          // operators whose rows are aligned runs of 3 consecutive columns store one code per run
    int window_granules = 0;         // > 0: slices of consecutive rows (sigma 64) whose columns fit this many 16-entry granules get WINDOW codes
    double window_prefer = 1.32;     //    ... preferred to a sorting window while consecutive rows pad at most this much more than sorted ones
    double window_max_overhead = 1.40;   //    and at most this much over the nonzeros
};
struct SellPlan {
    std::vector<SellSlice> s0, s1;       // interior slices (in PROCESSING order), slices touching ghost columns
    std::vector<double> val;
    std::vector<uint16_t> col;
    std::vector<int32_t> rows;           // (row, stored length) pairs in lane order of the slices that have them
    int64_t col_entries = 0;             // entries of col in use
    int64_t padded_nnz = 0;
    int sigma = 64;
    int64_t stride_rows = 0;             // the operator's dominant far column offset (a grid plane), 0: none found
    int planes = 0;                      // > 0: the class-0 table interleaves groups of this many planes
    int run = 1;                         // 3: one column code per aligned run of three consecutive columns (see below), else one per nonzero
    int window = 0;                      // > 0: WINDOW codes (below); the most granules a slice has
    std::vector<int32_t> gran;           // first column of every granule, slice after slice (slice: gran[cbase .. cbase + (flags >> 8)))
};
bool plan_sell(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, const uint8_t* row_class,
               const SellOptions& opt, SellPlan& out);

// ---- mid-size systems (prcg_medium.hip): the sliced layout + which wave of which workgroup owns which slices ----
// G workgroups of 16 waves; wave w (0 .. 16 G) owns the slices [wave_first[w], wave_first[w+1]) -- at most `max_slices`;
// workgroup g stages the columns [window[2g], window[2g] + window[2g+1]) of the (r,s) pairs in LDS: every column its rows
// touch and its own rows [own[2g], own[2g+1]) -- exactly the rows of its slices (workgroups are cut at sorting-window
// boundaries); a slice with a column outside its workgroup's own rows is flagged (SellSlice::flags = 1: its products wait for
// the other workgroups' rows).  false: the system does not qualify (too many slices, or a window beyond `max_window` pairs).
struct MediumPlan {
    SellPlan sell;
    std::vector<int32_t> wave_first, window, own;
    int groups = 0, window_pairs = 0;
};
bool plan_medium(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, int max_groups, int max_slices,
                 int max_window, MediumPlan& out);

// Merged exchange (small halos ride on the one all-gather per iteration):
// every rank contributes a slot of `slot` doubles = 8 (partial sums) + 2 x its packed send rows;
// `tab` holds every rank's send table, `T` doubles per rank: [n_peers, (peer, first row of the
// list, rows) ...].  For rank `rank` with peers `peer_rank[0..n_peers)` and receive segments
// recv_ptr, fill src[j] = index (in 16-byte pairs) of ghost j in the gathered buffer.
// Returns 0, or 1 + q if peer q's table has no list of the expected length for this rank.
int plan_gather_sources(int rank, int T, const double* tab, int n_peers, const int32_t* peer_rank,
                        const int64_t* recv_ptr, int64_t slot, int32_t* src);

}  // namespace prcg
