"""TEST INFRASTRUCTURE: NumPy emulation of the device's inner-product reduction tree.

The fused update kernels (new_cg_variants_amd/csrc/prcg_kernels.hip: k_pipe_update,
block_reduce_store, k_reduce_final) sum a product array in a fixed order:

  thread (block b, lane t): elements (b*trips + j)*512 + t, then + 256 + t, for j = 0..trips-1
  wave   : xor butterfly 32,16,8,4,2,1          block : waves 0..3 in order
  final  : the last block (256 threads): thread t sums partials t, t+256, ...; butterfly;
           4 waves in order   (block_reduce_store_final)

Plugging ``device_dot`` into the oracle (``dot=``) makes the oracle's free-running
trajectory comparable with the device's bit for bit.
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
