/*
 * prcg_test.h -- planner, diagnostic and test entry points of libprcg.so.
 *
 * NOT part of the boundary a maintainer of the reference binds (that is include/prcg.h): these exist so that the CPU
 * test-suite can exercise the host planners without a GPU, so that tests/device_order.py can rebuild a launch's
 * summation order, and so that the peer exchange's plumbing can be rehearsed with several processes on one GPU.
 * Each stands for something prcg_set_csr / prcg_solve_begin run internally; none is needed to solve a system.
 */
#ifndef PRCG_TEST_H
#define PRCG_TEST_H

#include "prcg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- peer exchange plumbing outside a session (tests/test_distributed.py) ---------------------------------- */
/* Rank and world size WITHOUT a communicator (instead of prcg_comm_init): enough for the peer exchange's own plumbing
 * -- prcg_peer_setup / prcg_peer_connect / prcg_peer_selftest -- not for solver sessions on a block with ghost columns
 * (their set-up products and the recorders use the communicator).  Lets two PROCESSES share one GPU in the tests, which
 * RCCL refuses. */
int prcg_world_init(prcg_t* h, int rank, int nranks);
/* One round of the exchange primitives, outside any session (tests): send rows2n (this rank's n (r,s) pairs; the rows of
 * the send plan go to the neighbours' ghost areas of parity k & 1) and slot5 as this rank's slot of round k, wait for
 * every rank's slot, return their sum in rank order (sums5) and this rank's ghost area of that parity (ghost2g, 2 doubles
 * per ghost row, nullable).  Every rank calls it with the same k (increasing from call to call). */
int prcg_peer_selftest(prcg_t* h, int k, const double* rows2n, const double* slot5, double* sums5, double* ghost2g);


/* ---- measurement (bench.py: roofline.stream_ceiling_GBps; SURVEY.md 8d: "also report a measured stream ceiling") ------
 * What the memory system of the attached GPU delivers for a byte mix, with no arithmetic to speak of, over n_pairs
 * 16-byte entries per array (scratch allocated and freed inside the call): mode 0 = pure read (8 loads of 16 B in flight
 * per thread); 1 = the vector traffic of the one-launch pipelined iteration -- per row one pair read and rewritten in
 * place, one pair read from one array and written to another (2 x 16 B in, 2 x 16 B out) -- with plain stores;
 * 2 = the same with nontemporal stores (what the kernels use when the vectors exceed the Infinity Cache);
 * 3 = pure read as a streaming kernel should issue it: every wave walks contiguous 4 KB chunks with four nontemporal 16-byte
 * loads in flight, eight waves per CU (tools/readpat.hip: 6.5-7.1 TB/s where mode 0's grid-stride pattern reads 5.1-5.5).
 * gbytes_per_s: bytes moved / time between HIP events around `reps` launches on the handle's compute stream. */
int prcg_stream_ceiling(prcg_t* h, int64_t n_pairs, int mode, int reps, double* gbytes_per_s);
/* ... and for the byte MIX of a one-launch iteration on an operator that streams stream_kb_per_64_rows KB per 64 rows (read once,
 * nontemporal) beside the rows' vector traffic (two 16-byte pairs read, one rewritten in place, one written to another array with
 * nontemporal stores), no arithmetic to speak of: the ceiling of a kernel that moves this mix -- row results written beside a large
 * read stream cost the memory system far more than their bytes (tools/mixbench.hip).  gbytes_per_s counts stream + 64 B per row. */
int prcg_mix_ceiling(prcg_t* h, int64_t n_rows, int stream_kb_per_64_rows, int reps, double* gbytes_per_s);

/* ---- host-only planning helpers (no GPU needed; used by the CPU test-suite) ------ */
/* CSR-adaptive tiling of rows [0,n): consecutive rows are packed into tiles of at most
 * cap_nnz nonzeros / cap_rows rows; a row longer than cap_nnz gets a tile of its own.
 * row_class (nullable): 0/1 per row; tiles never mix classes, class-0 tiles are
 * written first.  tiles_out: pairs (row_begin,row_end), capacity in pairs;
 * returns the number of tiles (negative = error), *n_class0 = tiles of class 0. */
int64_t prcg_plan_tiles(int64_t n, const int32_t* indptr, const uint8_t* row_class,
                        int cap_nnz, int cap_rows, int32_t* tiles_out, int64_t capacity,
                        int64_t* n_class0);
/* Window tiling (row-per-lane kernels for bands and stencils): rows are cut into tiles of at most
 * rows_per_tile (64 | 128) consecutive rows of one class whose columns -- and the tile's own rows --
 * are covered by a few PAGES of 64 consecutive columns; the kernels stage those pages of the input
 * vector in LDS and stream, per nonzero, its index page*64+offset into that window (cw_out, nullable).
 * tiles_out: 20 int32 per tile {row_begin, row_end, nnz_begin, nnz_end, pages | own_row_index << 8,
 * longest row, 0, 0, first column of page 0..11}; n_cols = owned + ghost columns.  Returns the number
 * of tiles, 0 if the operator does not qualify (some tile needs too many pages or holds a row longer
 * than a tile: the CSR-adaptive kernels run instead), -needed if capacity is too small, -1 on a bad
 * argument.  What prcg_set_csr runs internally; exported for the CPU tests. */
int64_t prcg_plan_window(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                         const uint8_t* row_class, int rows_per_tile, int32_t* tiles_out, int64_t capacity,
                         uint16_t* cw_out, int64_t* n_class0, int* most_pages);
/* Stream images of the window tiles (what prcg_set_csr builds after prcg_plan_window): per tile the window
 * indices of its nonzeros and its row pointers relative to its first nonzero; with share != 0 tiles whose
 * image is byte-identical read ONE stored copy (bands and stencils repeat a few images).  out[0..6) = {tiles,
 * stored window-index images, stored row-pointer images, elements of the window-index store, elements of the
 * row-pointer store, 1 if every tile's source holds exactly its own image}.  Returns 1, 0 if the operator is no
 * window operator, -1 on a bad argument.  Exported for the CPU tests. */
int prcg_plan_window_images(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                            const uint8_t* row_class, int rows_per_tile, int share, int64_t* out);
/* Pattern tiles (what prcg_set_csr tries first for short rows; csrc/prcg_plan.h: plan_window_patterns): 64-row window
 * tiles of at most 6 pages whose rows share one sequence of <= 16 slots (window offset relative to the lane + value)
 * and differ only in which slots they have -- a constant-coefficient stencil.  tiles_out: 24 int32 per tile (the 20 of
 * prcg_plan_window, then pattern id, 0, first mask of the tile in masks_out, 1 if every row has every slot);
 * pat_out: 72 bytes per pattern {int32 slots, uint32 value selectors (2 bits per slot), int16 offset[16], double
 * value[4]}; masks_out: uint16 per row of the tiles with incomplete rows.  counts_out[0..3) = {tiles, patterns, masks}.
 * Returns 1, 0 if the operator does not qualify, -needed tiles if a capacity is too small, -1 on a bad argument.
 * The kernels read nothing else of the operator: the CPU test rebuilds the matrix from these arrays. */
int64_t prcg_plan_window_patterns(int64_t n, int64_t n_cols, const int32_t* indptr, const int32_t* indices, const double* data,
                                  const uint8_t* row_class, int32_t* tiles_out, int64_t tile_capacity, void* pat_out,
                                  int64_t pat_capacity, uint16_t* masks_out, int64_t mask_capacity, int64_t* counts_out);
/* Sweep order of the pattern tiles for a stencil on a regular grid without ghost columns (csrc/prcg_plan.h: plan_sweep_tiles;
 * what prcg_set_csr tries before prcg_plan_window_patterns): the tile TABLE is ordered so that the tiles one wave takes one
 * after the other are the same rows of consecutive grid planes, and a page the wave's previous tile left in LDS is not loaded
 * again.  Outputs as prcg_plan_window_patterns; tiles_out[6] of a tile = LDS slot of each logical page (3 bits each) |
 * carried pages << 18 | (1 << 24 if slots are addressed through the slot table); empty tiles (row_begin == row_end) pad the
 * table.  counts_out[0..6) = {tiles, patterns, masks, waves the carry bits assume, rows of a grid plane, rows per tile}.
 * max_waves: most waves the launch may run.  Returns 1, 0 if the operator does not qualify, -tiles if a capacity is too small. */
int64_t prcg_plan_sweep(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, int max_waves,
                        int32_t* tiles_out, int64_t tile_capacity, void* pat_out, int64_t pat_capacity, uint16_t* masks_out,
                        int64_t mask_capacity, int64_t* counts_out);
/* Sliced rows (lane-per-row kernels for operators with medium-length rows that are no window operators -- assembled FEM
 * matrices; csrc/prcg_plan.h: plan_sell): rows are cut into slices of up to 64 rows of one class (class-0 slices first, in
 * PROCESSING order); stored position u of the row in lane l of a slice holds val[voff + ((u/2)*64 + l)*2 + u%2] and the
 * 16-bit code col16[coff + ((u/8)*64 + l)*8 + u%8]: starting from the slice's `cbase`, every position moves the lane's
 * running column by code - 16384 and -- unless the code is 0 or 65535 (skips: -16384 / +49151, no nonzero) -- names a
 * nonzero at the column reached; shorter rows are padded (value 0, code 16384; never multiplied).
 * sigma: sorting window in rows -- 64: a slice holds consecutive rows [first row, end row); larger (SELL-C-sigma): every
 * window of sigma consecutive rows of one class is sorted by descending number of trips (ceil(length / 8), or / 24 with run
 * codes; stable) before it is cut; slices of such
 * windows, and slices with a row that needs skips, name their rows: lane l holds row rows_out[2 * (rows_off + l)] of STORED
 * length rows_out[2 * (rows_off + l) + 1] (row -1, length 0 behind the last); 0: 64 while that pads by at most 6 %, else the
 * smallest of 256, 1024, 4096, 16384 whose padding stays within 4 % or within 1 % of the least.  planes > 1: the class-0 table
 * interleaves groups of that many grid planes when the operator has a dominant far column offset (any table order is correct).
 * slices_out: 8 int32 per slice {first (smallest) row, that + rows, voff, coff, longest stored row, cbase, rows_off or -1, 0};
 * allow_runs != 0: an operator whose every row consists of aligned runs of three consecutive columns stores ONE code per run
 * (code c of a row: its positions 3c..3c+2 at the run's first column +0, +1, +2; c at col16[coff + ((c/8)*64 + l)*8 + c%8]).
 * window_granules > 0: when the rows stay consecutive (sigma 64) and EVERY slice's columns fit that many granules of 16
 * consecutive columns, WINDOW codes are stored instead of deltas: slice descriptor {.., cbase = its first entry of gran_out,
 * rows_off = -1, flags = 2 | granules << 8}; a code is 16 g + (column - gran_out[cbase + g]) for the last granule g that starts
 * at or before the (run's first) column; no skips, padding code 0.  The kernels stage the granules in LDS (at most 64).
 * stats[0..12) = {class-0 slices, elements of val, elements of col16, padded nonzeros, sigma chosen, far stride in rows
 * (0: none), planes interleaved (0: row order), elements of rows, column codes in use, positions per code (1 | 3), the most
 * granules of a slice (0: delta codes), elements of gran}.
 * Returns the number of slices, 0 if the operator does not qualify (padding beyond max_overhead x nnz), -needed if a
 * capacity is too small, -1 on a bad argument.  What prcg_set_csr runs for rows of 24 nonzeros and more. */
int64_t prcg_plan_sell(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data, const uint8_t* row_class,
                       double max_overhead, int sigma, int planes, int allow_runs, int window_granules, int32_t* slices_out, int64_t capacity,
                       double* val_out, uint16_t* col_out, int64_t array_capacity, int32_t* rows_out, int64_t rows_capacity,
                       int32_t* gran_out, int64_t gran_capacity, int64_t* stats);
/* Merged exchange of the multi-GPU pipelined loop (small halos ride on the one all-gather per
 * iteration): where in the gathered buffer do this rank's ghost rows lie?  `tables`: every rank's
 * send table, doubles_per_table doubles each: [n_peers, (peer, first row of its list, rows)...];
 * slot_doubles = 8 + 2 * (largest send list of any rank).  ghost_src[j] = index, in 16-byte pairs,
 * of ghost j (ghosts ordered as prcg_set_halo's receive segments).  Returns 0, 1+q if peer q's
 * table holds no list of the expected length for this rank, -1 on a bad argument.  What
 * prcg_solve_begin computes internally after all-gathering the tables; exported for the CPU tests. */
int prcg_plan_gather(int rank, int doubles_per_table, const double* tables, int n_peers,
                     const int32_t* peer_rank, const int64_t* recv_ptr, int64_t slot_doubles,
                     int32_t* ghost_src);
/* Diagnostic (tests): how the resident operator is laid out for the one-launch iteration, i.e. everything the summation
 * order of its inner products depends on.  out[0..8) = {1 if window operator, window geometry id, rows per window tile,
 * number of tiles T, workgroups of the last one-launch iteration (0: none yet), waves per workgroup of that launch,
 * interior tiles, (1 if XCD-chunked tile order) | (waves a sweep table assumes) << 8}, then T pairs (first row, end row) in table order.  Returns the number of int64 written, -needed
 * if capacity is too small, -1 on a bad argument.  tests/device_order.py rebuilds the launch's reduction tree from it. */
int64_t prcg_debug_layout(const prcg_t* h, int64_t* out, int64_t capacity);
/* Capacity rule of every vector a matrix-product launch reads (window pages are whole 64-column blocks, the narrow
 * column encodings decode a few out-of-tile bytes per tile): n_rows + n_ghost entries of `components` doubles plus
 * 65,536 spare entries must lie between the pointer and the end of its allocation.  The engine evaluates this at
 * every launch and returns PRCG_EINVAL instead of launching (a short source used to be a memory fault).  1 = ok. */
int prcg_window_source_ok(int64_t n_rows, int64_t n_ghost, int components, int64_t bytes_available);
/* the tile caps the device kernels were compiled for */
void prcg_tile_caps(int* cap_nnz, int* cap_rows);

#ifdef __cplusplus
}
#endif
#endif /* PRCG_TEST_H */
