"""TEST INFRASTRUCTURE: NumPy emulation of the device's inner-product reduction tree.

The fused update kernels (new_cg_variants_amd/csrc/prcg_kernels.hip: k_pipe_update,
block_reduce_store, k_reduce_final) sum a product array in a fixed order:

  thread (block b, lane t): elements (b*trips + j)*512 + t, then + 256 + t, for j = 0..trips-1
  wave   : xor butterfly 32,16,8,4,2,1          block : waves 0..3 in order
  final  : the last block (256 threads): thread t sums partials t, t+256, ...; butterfly;
           4 waves in order   (block_reduce_store_final)

Plugging ``device_dot`` into the oracle (``dot=``) makes the oracle's free-running
trajectory comparable with the device's bit for bit.

``OneLaunchTree`` is the same for the ONE-LAUNCH iteration (prcg_win.hip: k_win_tiles<2,fused>, the schedule
bench.py times and every solve uses by default): its inner products are summed

  lane l of wave (block b, wave v): rows rb + j*64 + l (j = 0..M-1) of tiles  slot, slot + W, slot + 2W, ...
                                    with slot = xcd_remap(b) * WPB + v,  W = grid * WPB  (in that order)
  wave  : xor butterfly             block : waves 0..WPB-1 in order  ->  one partial per workgroup
  (XCD-chunked order, layout['xcd_chunked']: workgroup b sweeps the share of XCD b % 8 of the table instead)
  final : the 256-thread tree over the workgroups' partials (thread t: partials t, t+256, ...; butterfly;
          4 waves in order) -- by every workgroup of the NEXT launch in its prologue (sum_prev_partials) or
          by k_reduce_final when a prcg_iterate call ends: the same tree either way

built from DeviceCSR.layout() (prcg.h: prcg_debug_layout: tile rows, grid, waves per workgroup).
"""
import numpy as np

ELEMS_PER_TRIP = 512
MAX_GRID = 2048
FINAL_THREADS = 256


def chunking(n):
    total = (n + ELEMS_PER_TRIP - 1) // ELEMS_PER_TRIP
    grid = max(1, min(total, MAX_GRID))
    trips = (total + grid - 1) // grid
    grid = max(1, (total + trips - 1) // trips) if trips > 0 else 1
    return grid, max(trips, 1)


def _butterfly(v):
    """lane-0 value of the xor butterfly over the last axis (length 64)."""
    off = 32
    while off >= 1:
        v = v[..., :off] + v[..., off:2 * off]
        off //= 2
    return v[..., 0]


def device_sum(prod):
    prod = np.asarray(prod, dtype=np.float64)
    n = prod.shape[0]
    grid, trips = chunking(n)
    padded = np.zeros(grid * trips * ELEMS_PER_TRIP)
    padded[:n] = prod
    a = padded.reshape(grid, trips, 2, 256)
    acc = np.zeros((grid, 256))
    # elements past n are never added on the device; adding +0.0 here is the same value
    for j in range(trips):
        for e in range(2):
            acc = acc + a[:, j, e, :]
    waves = _butterfly(acc.reshape(grid, 4, 64))              # (grid, 4)
    partial = waves[:, 0]
    for w in range(1, 4):
        partial = partial + waves[:, w]
    # final kernel
    rounds = (grid + FINAL_THREADS - 1) // FINAL_THREADS
    pp = np.zeros(rounds * FINAL_THREADS)
    pp[:grid] = partial
    pp = pp.reshape(rounds, FINAL_THREADS)
    t = np.zeros(FINAL_THREADS)
    for j in range(rounds):
        t = t + pp[j]
    nw = FINAL_THREADS // 64
    wf = _butterfly(t.reshape(nw, 64))
    out = wf[0]
    for w in range(1, nw):
        out = out + wf[w]
    return float(out)


def device_dot(a, b):
    return device_sum(np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64))


def xcd_remap(b, nb):
    """prcg_device.hpp: xcd_remap -- workgroup b of nb -> position in the XCD-contiguous work order."""
    xcd, idx = b & 7, b >> 3
    q, r = nb >> 3, nb & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + idx


def _final_tree(partial):
    """the 256-thread tree over one partial per workgroup (k_reduce_final / sum_prev_partials)"""
    grid = partial.shape[0]
    rounds = (grid + FINAL_THREADS - 1) // FINAL_THREADS
    pp = np.zeros(rounds * FINAL_THREADS)
    pp[:grid] = partial
    pp = pp.reshape(rounds, FINAL_THREADS)
    t = np.zeros(FINAL_THREADS)
    for j in range(rounds):
        t = t + pp[j]
    wf = _butterfly(t.reshape(FINAL_THREADS // 64, 64))
    out = wf[0]
    for w in range(1, FINAL_THREADS // 64):
        out = out + wf[w]
    return float(out)


class OneLaunchTree:
    """The summation order of the inner products of the one-launch pipelined iteration on a window operator."""

    def __init__(self, layout):
        assert layout['window'] and layout['grid'] > 0, layout
        tiles = np.asarray(layout['tiles'], dtype=np.int64)
        grid, wpb = int(layout['grid']), int(layout['waves_per_block'])
        M = int(layout['rows_per_tile']) // 64
        nt = tiles.shape[0]
        W = grid * wpb
        chunked = bool(layout.get('xcd_chunked', False))
        per_wave = (nt + W - 1) // W + (2 if chunked else 0)
        # idx[b, v, step, lane] = row summed by that lane at that step, -1: none
        idx = -np.ones((grid, wpb, per_wave * M, 64), dtype=np.int64)
        lane = np.arange(64)
        for b in range(grid):
            for v in range(wpb):
                if chunked:
                    # XCD-chunked order (k_win_tiles, A.order == 1): XCD b % 8 sweeps its contiguous share of the table
                    xcd, ix, q, r = b & 7, b >> 3, grid >> 3, grid & 7
                    nbx = q + (1 if xcd < r else 0)
                    before = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
                    lo, hi = nt * before // grid, nt * (before + nbx) // grid
                    seq = range(lo + ix * wpb + v, hi, nbx * wpb)
                else:
                    seq = range(xcd_remap(b, grid) * wpb + v, nt, W)
                for i, t in enumerate(seq):
                    rb, re = tiles[t]
                    for j in range(M):
                        rows = rb + j * 64 + lane
                        idx[b, v, i * M + j] = np.where(rows < re, rows, -1)
        self.idx = idx
        self.grid, self.wpb, self.chunked = grid, wpb, chunked
        covered = np.sort(idx[idx >= 0])
        assert np.array_equal(covered, np.arange(tiles[:, 0].min(), tiles[:, 1].max())), 'every row summed exactly once'

    def sum(self, prod):
        prod = np.asarray(prod, dtype=np.float64)
        ext = np.concatenate([prod, [0.0]])            # index -1 -> +0.0 (adding it changes nothing: acc starts at +0.0)
        acc = np.zeros(self.idx.shape[:2] + (64,))
        for step in range(self.idx.shape[2]):
            acc = acc + ext[self.idx[:, :, step, :]]
        waves = _butterfly(acc)                        # (grid, wpb)
        partial = waves[:, 0]
        for v in range(1, self.wpb):
            partial = partial + waves[:, v]
        return _final_tree(partial)

    def dot(self, a, b):
        return self.sum(np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64))
