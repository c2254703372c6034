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

}  // namespace prcg
