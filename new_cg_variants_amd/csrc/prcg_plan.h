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

// Merged exchange (small halos ride on the one all-gather per iteration):
// every rank contributes a slot of `slot` doubles = 8 (partial sums) + 2 x its packed send rows;
// `tab` holds every rank's send table, `T` doubles per rank: [n_peers, (peer, first row of the
// list, rows) ...].  For rank `rank` with peers `peer_rank[0..n_peers)` and receive segments
// recv_ptr, fill src[j] = index (in 16-byte pairs) of ghost j in the gathered buffer.
// Returns 0, or 1 + q if peer q's table has no list of the expected length for this rank.
int plan_gather_sources(int rank, int T, const double* tab, int n_peers, const int32_t* peer_rank,
                        const int64_t* recv_ptr, int64_t slot, int32_t* src);

}  // namespace prcg
