"""CPU: the C-ABI library loads, exports every symbol include/prcg.h declares, refuses
to work without a GPU (no CPU fallback), and its host-only planning is sound."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from new_cg_variants_amd import _lib as L
from new_cg_variants_amd import problems
from new_cg_variants_amd.device import plan_tiles


def declared_symbols(headers=('prcg.h', 'prcg_test.h')):
    """include/prcg.h: the boundary a maintainer binds; include/prcg_test.h: planner / diagnostic / test hooks"""
    names = set()
    for hname in headers:
        text = open(os.path.join(ROOT, 'include', hname)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(prcg_[a-z0-9_]+)\s*\(', text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f'{name} declared in include/prcg.h but not exported'
    # and the binding covers exactly the two headers
    assert sorted(L._SIGNATURES) == names
    # the product header carries no planner / debug / test hook
    assert not [n for n in declared_symbols(('prcg.h',)) if n.startswith(('prcg_plan_', 'prcg_debug_')) or n in ('prcg_peer_selftest', 'prcg_world_init', 'prcg_tile_caps')]
    assert lib.prcg_version() == 1


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU creating a handle must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    h = C.c_void_p()
    rc = L.lib().prcg_create(C.byref(h), 0)
    assert rc == L.EHIP and not h.value
    assert b'no CPU fallback' in L.lib().prcg_last_error(None)
    from new_cg_variants_amd.cg_variants import hs_cg
    A = problems.laplace_2d(8, 8)
    with pytest.raises(L.PrcgError):
        hs_cg(A, np.ones(64), np.zeros(64), 5)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure; nothing in the package may reference it."""
    pkg = os.path.join(ROOT, 'new_cg_variants_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'ne_oracle' not in src and 'mp_oracle' not in src, f


def test_window_source_capacity_rule():
    """What the engine evaluates before every matrix-product launch (prcg_engine.cpp: check_sources): a source vector
    must hold n + g entries plus 65,536 spare ones, per component -- a deliberately short buffer is refused."""
    ok = L.lib().prcg_window_source_ok
    n, g, pad = 1000, 14, 65536
    assert ok(n, g, 1, (n + g + pad) * 8) == 1
    assert ok(n, g, 1, (n + g + pad) * 8 - 1) == 0
    assert ok(n, g, 1, n * 8) == 0                          # the round-2 fault: a source allocated without spare entries
    assert ok(n, g, 1, 16) == 0                             # an unused vector's placeholder
    assert ok(n, g, 2, (n + g + pad) * 8) == 0 and ok(n, g, 2, (n + g + pad) * 16) == 1
    assert ok(-1, 0, 1, 1 << 40) == 0 and ok(n, g, 0, 1 << 40) == 0 and ok(n, g, 1, -5) == 0


def check_tiles(indptr, tiles, n0, cap_nnz, cap_rows, row_class=None):
    n = len(indptr) - 1
    covered = np.zeros(n, dtype=np.int32)
    for t, (rb, re_) in enumerate(tiles):
        assert 0 <= rb < re_ <= n
        covered[rb:re_] += 1
        nn = indptr[re_] - indptr[rb]
        assert (re_ - rb) <= cap_rows
        assert nn <= cap_nnz or re_ == rb + 1          # only a single long row may exceed the cap
        if row_class is not None:
            cls = row_class[rb:re_]
            assert np.all(cls == cls[0])
            assert (cls[0] != 0) == (t >= n0)           # class-0 tiles come first
    assert np.all(covered == 1), 'every row in exactly one tile'


def test_tile_caps_match_kernel_constants():
    a, b = C.c_int(0), C.c_int(0)
    L.lib().prcg_tile_caps(C.byref(a), C.byref(b))
    assert a.value % 256 == 253 and b.value >= 64       # 256*steps - 3


@pytest.mark.parametrize('name', ['s1_small', 's3_small'])
def test_tiling_regular_matrices(name):
    A = problems.WORKLOADS[name]['make']()
    tiles, n0 = plan_tiles(A.indptr)
    assert n0 == len(tiles)
    check_tiles(A.indptr, tiles, n0, 509, 256)
    # packing is tight: no tile could have taken the next row
    for rb, re_ in tiles[:-1]:
        nn = A.indptr[re_] - A.indptr[rb]
        nxt = A.indptr[re_ + 1] - A.indptr[re_]
        assert nn + nxt > 509 or (re_ - rb) == 256


def test_tiling_ragged_empty_and_long_rows():
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 40, size=5000)
    lens[100] = 3000        # longer than a tile
    lens[101] = 509         # exactly a tile
    lens[102] = 510         # one more than a tile
    lens[200:700] = 0       # a long run of empty rows (row cap must cut it)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cls = (rng.random(5000) < 0.1).astype(np.uint8)
    tiles, n0 = plan_tiles(indptr, cls)
    check_tiles(indptr, tiles, n0, 509, 256, cls)
    tiles2, n02 = plan_tiles(indptr, None, cap_nnz=64, cap_rows=8)
    check_tiles(indptr, tiles2, n02, 64, 8)
    # empty matrix
    t0, _ = plan_tiles(np.zeros(1, dtype=np.int32))
    assert len(t0) == 0


def test_generators_match_their_stated_sizes():
    A = problems.laplace_2d(1000, 1000, rows=(0, 3000))
    assert A.shape == (3000, 1_000_000)
    full_nnz_s1 = 5 * 1_000_000 - 4 * 1000
    assert full_nnz_s1 == 4_996_000
    assert 7 * 216**3 - 6 * 216**2 == 70_263_936
    assert 15 * 10_000_000 - 2 * sum(range(1, 8)) == 149_999_944
    B = problems.banded_ex2b(5000, 7)
    assert B.nnz == 15 * 5000 - 56 and abs(B - B.T).max() == 0
    # ex2b.c:93: diag = 1 + (i/(n-1)) (kappa-1) rho^(n-1-i); last row has the full kappa
    assert B[4999, 4999] == 1.0 + (1e6 - 1.0)
    assert B[0, 0] == 1.0 and B[10, 11] == 1e-4


@pytest.mark.parametrize('workload,nranks', [('s3_small', 2), ('s3_small', 5), ('s1_small', 3), ('s1_small', 4)])
def test_merged_exchange_plan_moves_the_right_rows(workload, nranks):
    """The multi-GPU pipelined loop moves a small halo inside the one all-gather per iteration
    (DESIGN.md section 5).  Each rank learns from the others' send tables where its ghost rows lie
    in the gathered buffer (prcg_plan_gather = the host code prcg_solve_begin runs after
    all-gathering the tables).  Simulate all ranks in one process: pack as k_gather_pack does,
    concatenate the slots as ncclAllGather does, pick as k_gather_unpack does -- every rank must
    end up with exactly the (r,s) rows of its ghost columns."""
    from new_cg_variants_amd import partition
    A = problems.WORKLOADS[workload]['make']()
    n = A.shape[0]
    offsets, parts = partition.split_serial(A, nranks)
    rng = np.random.default_rng(4)
    RS = rng.standard_normal((n, 2))                      # global (r,s) pairs
    max_send = max(int(p[2]['send_ptr'][-1]) for p in parts)
    max_peers = max(len(p[2]['peers']) for p in parts)
    slot = 8 + 2 * max_send
    T = 1 + 3 * max_peers
    tables = np.zeros((nranks, T))
    gbuf = np.zeros((nranks, slot))
    for r, (A_loc, ghost_ids, halo) in enumerate(parts):
        tables[r, 0] = len(halo['peers'])
        for q, peer in enumerate(halo['peers']):
            tables[r, 1 + 3 * q:4 + 3 * q] = (peer, halo['send_ptr'][q], halo['send_ptr'][q + 1] - halo['send_ptr'][q])
        lo = offsets[r]
        gbuf[r, :5] = r + 1.0                                                    # stand-in for the partial sums
        rows = RS[lo + halo['send_idx']]                                         # k_gather_pack: rows[j] = rs[send_idx[j]]
        gbuf[r, 8:8 + 2 * rows.shape[0]] = rows.ravel()
    pairs = gbuf.reshape(-1, 2)                                                  # the gathered buffer, viewed as 16-byte pairs
    moved = 0
    for r, (A_loc, ghost_ids, halo) in enumerate(parts):
        g = ghost_ids.size
        src = np.full(g + 1, -1, dtype=np.int32)
        peers = np.ascontiguousarray(halo['peers'], dtype=np.int32)
        recv_ptr = np.ascontiguousarray(halo['recv_ptr'], dtype=np.int64)
        rc = L.lib().prcg_plan_gather(r, T, L.ptr(tables), len(peers), L.ptr(peers), L.ptr(recv_ptr), slot, L.ptr(src))
        assert rc == 0
        got = pairs[src[:g]]                                                     # k_gather_unpack: rs_ghost[j] = g2[ghost_src[j]]
        assert np.array_equal(got, RS[ghost_ids]), (workload, nranks, r)
        moved += g
        assert np.array_equal(gbuf[:, :5].sum(axis=0), np.full(5, nranks * (nranks + 1) / 2))
    assert moved > 0
    # a table that does not match (a peer "forgets" its list for rank 0) is reported, not mis-indexed
    A_loc, ghost_ids, halo = parts[0]
    broken = tables.copy()
    broken[int(halo['peers'][0]), 0] = 0
    src = np.zeros(ghost_ids.size + 1, dtype=np.int32)
    peers = np.ascontiguousarray(halo['peers'], dtype=np.int32)
    recv_ptr = np.ascontiguousarray(halo['recv_ptr'], dtype=np.int64)
    assert L.lib().prcg_plan_gather(0, T, L.ptr(broken), len(peers), L.ptr(peers), L.ptr(recv_ptr), slot, L.ptr(src)) == 1


def plan_window(A_local, rows_per_tile, row_class=None):
    A_local = A_local.tocsr()
    n, n_cols = A_local.shape
    indptr = np.ascontiguousarray(A_local.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A_local.indices, dtype=np.int32)
    rc = None if row_class is None else np.ascontiguousarray(row_class, dtype=np.uint8)
    cap = n + 8
    tiles = np.zeros((cap, 20), dtype=np.int32)
    cw = np.zeros(A_local.nnz + 1, dtype=np.uint16)
    n0, most = C.c_int64(0), C.c_int(0)
    got = L.lib().prcg_plan_window(n, n_cols, L.ptr(indptr), L.ptr(indices), L.ptr(rc), rows_per_tile, L.ptr(tiles), cap,
                                   L.ptr(cw), C.byref(n0), C.byref(most))
    return got, tiles[:max(got, 0)], cw[:A_local.nnz], int(n0.value), int(most.value)


@pytest.mark.parametrize('name,rows,max_pages', [('band', 64, 2), ('band', 128, 3), ('lap2d', 64, 4), ('lap2d', 128, 9),
                                                 ('lap3d', 128, 12), ('ragged', 64, 2), ('ghosts', 64, 4)])
def test_window_tiling_covers_rows_columns_and_own_rows(name, rows, max_pages):
    """Host planner of the row-per-lane kernels: every row in exactly one tile (classes apart, order kept),
    every nonzero's column = first column of its page + offset, the tile's own rows contiguous in the
    window (the fused epilogues read (r,s) of their row from it), pages inside the vector."""
    import scipy.sparse as sp
    from new_cg_variants_amd import partition
    rng = np.random.default_rng(6)
    row_class = None
    if name == 'band':
        A = problems.banded_ex2b(5000, 7)
    elif name == 'lap2d':
        A = problems.laplace_2d(90, 70)
    elif name == 'lap3d':
        A = problems.laplace_3d(24, 20, 16)
    elif name == 'ragged':
        n = 4000
        lens = rng.integers(0, 15, size=n)
        lens[100:300] = 0
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        r = np.repeat(np.arange(n), lens)
        A = sp.csr_matrix((np.ones(indptr[-1]), (r + rng.integers(-8, 9, size=r.size)).clip(0, n - 1).astype(np.int32), indptr),
                          shape=(n, n))
    else:   # a middle row block of a band: ghost columns on both sides, two classes of rows
        full = problems.banded_ex2b(6000, 7)
        A, ghost_ids = partition.localize(full[2000:4000], 2000, 4000)
        assert ghost_ids.size == 14
        row_class = (np.diff(A.indptr) > 0) & np.array([(A.indices[A.indptr[i]:A.indptr[i + 1]] >= 2000).any() for i in range(2000)])
    got, tiles, cw, n0, most = plan_window(A, rows, row_class)
    assert got > 0 and most <= max_pages, (got, most)
    n, n_cols = A.shape
    seen = np.zeros(n, dtype=int)
    for ti, t in enumerate(tiles):
        rb, re, lo, hi, geo, maxlen = (int(v) for v in t[:6])
        npages, own = geo & 255, geo >> 8
        pages = t[8:8 + npages].astype(np.int64)
        assert 0 < re - rb <= rows and hi - lo <= 1024 - 15 and (lo, hi) == (A.indptr[rb], A.indptr[re])
        # the columns a page serves never overlap: 64 from its start, but an owned-column page stops at n (ghost columns
        # may live in another buffer -- the peer-exchange schedule reads them from the rank's exchange buffer)
        served_end = np.where(pages < n, np.minimum(pages + 64, n), pages + 64)
        assert np.all(pages[1:] >= served_end[:-1]) and pages[0] >= 0 and pages[-1] + 64 <= n_cols + 64
        seen[rb:re] += 1
        if row_class is not None:
            assert np.all(row_class[rb:re] == (ti >= n0))
        assert maxlen == np.diff(A.indptr[rb:re + 1]).max()
        c = cw[lo:hi].astype(np.int64)
        assert np.all(c < npages * 64)
        assert np.array_equal(pages[c >> 6] + (c & 63), A.indices[lo:hi])
        assert np.array_equal(pages[c >> 6] < n, A.indices[lo:hi] < n), 'a ghost column is served by a ghost page, an owned one by an owned page'
        i = np.arange(re - rb) + own                       # window index of the tile's own rows
        assert np.array_equal(pages[i >> 6] + (i & 63), np.arange(rb, re))
    assert np.all(seen == 1)
    # interior tiles come first, each class in row order
    for part in (tiles[:n0], tiles[n0:]):
        assert np.all(np.diff(part[:, 0]) > 0)


def test_window_tiling_refuses_wide_and_long_rows():
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    n = 3000
    lens = rng.integers(1, 9, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    wide = sp.csr_matrix((np.ones(indptr[-1]), rng.integers(0, n, size=indptr[-1]).astype(np.int32), indptr), shape=(n, n))
    assert plan_window(wide, 128)[0] == 0                      # random columns: far too many pages per tile
    lens[7] = 1200
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    r = np.repeat(np.arange(n), lens)
    long_row = sp.csr_matrix((np.ones(indptr[-1]), (r + rng.integers(-5, 6, size=r.size)).clip(0, n - 1).astype(np.int32), indptr),
                             shape=(n, n))
    assert plan_window(long_row, 64)[0] == 0                   # one row longer than a tile


def test_operator_cache_fingerprint_sees_sum_preserving_edits():
    import scipy.sparse as sp
    from new_cg_variants_amd.cg_variants import _fingerprint
    A = sp.random(2000, 2000, density=5e-3, format='csr', random_state=0)
    f0 = _fingerprint(A)
    A.data[3], A.data[7] = A.data[3] + 1.0, A.data[7] - 1.0
    assert _fingerprint(A) != f0


@pytest.mark.parametrize('name,rows', [('band', 64), ('lap2d', 64), ('lap3d', 128), ('ragged', 64)])
def test_window_stream_images_are_shared_losslessly(name, rows):
    """Tiles whose encoded streams (window indices, relative row pointers) are byte-identical read one stored copy:
    a band keeps a handful of images for thousands of tiles, an irregular operator keeps every tile's own; either
    way what a tile reads is exactly its own image (checked inside the library), and with sharing off the store is
    the per-nonzero layout."""
    import scipy.sparse as sp
    rng = np.random.default_rng(9)
    if name == 'band':
        A = problems.banded_ex2b(40000, 7)
    elif name == 'lap2d':
        A = problems.laplace_2d(200, 160)
    elif name == 'lap3d':
        A = problems.laplace_3d(32, 32, 24)
    else:
        n = 6000
        lens = rng.integers(1, 15, size=n)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        r = np.repeat(np.arange(n), lens)
        A = sp.csr_matrix((np.ones(indptr[-1]), (r + rng.integers(-8, 9, size=r.size)).clip(0, n - 1).astype(np.int32), indptr),
                          shape=(n, n))
        A.sort_indices()
    A = A.tocsr()
    n = A.shape[0]
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    res = {}
    for share in (1, 0):
        out = np.zeros(8, dtype=np.int64)
        assert L.lib().prcg_plan_window_images(n, n, L.ptr(indptr), L.ptr(indices), None, rows, share, L.ptr(out)) == 1
        tiles, cw_images, rel_images, cw_elems, rel_elems, same = (int(v) for v in out[:6])
        assert same == 1 and tiles >= n // rows
        res[share] = (tiles, cw_images, rel_images, cw_elems, rel_elems)
    tiles, cw_images, rel_images, cw_elems, rel_elems = res[1]
    assert res[0][1] == tiles and res[0][2] == tiles                       # sharing off: one image per tile ...
    assert A.nnz <= res[0][3] <= A.nnz + 48 and res[0][4] == n + tiles + 8  # ... laid out as the per-nonzero stream
    if name == 'band':
        assert cw_images <= 8 and rel_images <= 8 and cw_elems < 16000, res[1]
    elif name in ('lap2d', 'lap3d'):
        assert cw_images < 0.5 * tiles and cw_elems < 0.6 * A.nnz, res[1]
    else:
        assert cw_images > 0.9 * tiles, res[1]
    print(f'{name}: {tiles} tiles, {cw_images} window-index images ({cw_elems} of {A.nnz} elements stored), {rel_images} row-pointer images')


def test_preconditioner_routing_diagonal_on_device_anything_else_by_callback():
    """figure_gen.py:42-44 passes a Python callable.  A diagonal scaling is recognised by an exact probe and becomes
    the device's inverse diagonal; the identity becomes "no preconditioner"; anything else is handed to the
    host-callback path as it is (never approximated, never rejected)."""
    from new_cg_variants_amd import cg_variants as cgv
    n = 50
    rng = np.random.default_rng(2)
    d = rng.random(n) + 0.5
    got, fn = cgv._diagonal_of(lambda v: d * v, n)
    assert fn is None and np.array_equal(got, d)
    assert cgv._diagonal_of(lambda v: v, n) == (None, None)
    assert cgv._diagonal_of(None, n) == (None, None)
    T = np.diag(d) + 0.1 * np.diag(np.ones(n - 1), 1) + 0.1 * np.diag(np.ones(n - 1), -1)
    tri = lambda v: np.linalg.solve(T, v)                                       # noqa: E731
    got, fn = cgv._diagonal_of(tri, n)
    assert got is None and fn is tri
    jac = cgv.Jacobi(__import__('scipy.sparse', fromlist=['diags']).diags(1 / d).tocsr())
    got, fn = cgv._diagonal_of(jac, n)
    assert fn is None and np.array_equal(got, 1 / (1 / d))
    with pytest.raises(ValueError):
        cgv._diagonal_of(lambda v: v[:-1], n)


def test_missing_x_true_is_refused_for_large_systems():
    """The reference obtains a missing x_true with a sparse direct solve inside its error callbacks
    (callbacks/error_A_norm.py:36-39); beyond n = 200,000 that never returns -- the drop-in asks for x_true instead
    (checked before any device work, so this runs without a GPU)."""
    from new_cg_variants_amd import cg_variants
    from new_cg_variants_amd.callbacks import error_A_norm
    A = problems.laplace_2d(500, 500)
    assert A.shape[0] > cg_variants._MAX_DIRECT_SOLVE
    with pytest.raises(ValueError, match='x_true'):
        cg_variants.pipe_pr_cg(A, np.ones(A.shape[0]), np.zeros(A.shape[0]), 5, callbacks=[error_A_norm])


def plan_sell(A, row_class=None, max_overhead=1.25, sigma=0, planes=8, allow_runs=True, window=0):
    A = A.tocsr()
    n = A.shape[0]
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    rc = None if row_class is None else np.ascontiguousarray(row_class, dtype=np.uint8)
    stats = np.zeros(12, dtype=np.int64)
    cap = (n // 4 if window else n // 60) + 64 + (0 if rc is None else int(np.count_nonzero(np.diff(rc.astype(np.int8)))) + 2)    # (window codes may cut slices)
    slices = np.zeros((cap, 8), dtype=np.int32)
    arr_cap = int(max_overhead * A.nnz * 1.3) + 64 * 130 * 4 + 4096
    val = np.zeros(arr_cap)
    col = np.zeros(arr_cap, dtype=np.uint16)
    rows = np.zeros((cap * 64, 2), dtype=np.int32)
    gran = np.zeros(cap * 64 + 64, dtype=np.int32)
    got = L.lib().prcg_plan_sell(n, L.ptr(indptr), L.ptr(indices), L.ptr(data), L.ptr(rc), float(max_overhead), int(sigma), int(planes), int(allow_runs),
                                 int(window), L.ptr(slices), cap, L.ptr(val), L.ptr(col), arr_cap, L.ptr(rows), rows.size, L.ptr(gran), gran.size,
                                 L.ptr(stats))
    plan_sell.gran = gran[:int(stats[11])]               # (granule starts of the last call: WINDOW codes)
    return got, slices[:max(got, 0)], val, col, stats, rows


def slice_rows(sl, rows):
    """(row, length) of every lane of a slice descriptor: consecutive rows, or the slice's entries of the row stream"""
    rb, re, _, _, _, _, rows_off, _ = sl
    if rows_off < 0:
        return [(r, None) for r in range(rb, re)]
    ent = rows[rows_off:rows_off + 64]
    k = re - rb
    assert np.all(ent[k:, 0] == -1) and np.all(ent[k:, 1] == 0)
    assert ent[:k, 0].min() == rb
    return [(int(r), int(ln)) for r, ln in ent[:k]]


@pytest.mark.parametrize('name', ['fem', 'fem_runs_off', 'ragged', 'ghosts', 'irregular', 'irregular_sigma256', 'wide_gaps', 'fem_window', 'fem_window_runs_off',
                                  'ghosts_window', 'fem_window_unsorted', 'blocks_window', 'fem_window_cut'])
def test_sliced_row_layout_holds_exactly_the_matrix(name):
    """Host planner of the lane-per-row kernels (prcg_plan.cpp: plan_sell): every row in exactly one slice (classes apart,
    class 0 first), and reading the re-laid arrays back with the kernel's index formula gives the caller's CSR rows,
    values bit for bit and columns exactly, in order; padding is zero.  With a sorting window wider than a slice
    (SELL-C-sigma: operators whose row lengths vary) the slices name their rows and lengths, rows of one window only,
    most trips first.  WINDOW codes (the `_window` cases: slices of consecutive rows whose columns fit 48 granules of 16) name a
    place in the slice's granule list instead of a delta; an operator with one slice that needs more granules keeps deltas."""
    import scipy.sparse as sp
    from new_cg_variants_amd import partition
    rng = np.random.default_rng(9)
    row_class = None
    sigma = 0
    window = 48 if '_window' in name else 0
    unsorted = name.endswith('_unsorted')
    name = name.replace('_window', '').replace('_unsorted', '')
    cut = name.endswith('_cut')
    name = name.replace('_cut', '')
    if cut:
        window = 24                          # fewer granules than this operator's slices need (up to 33): the planner cuts them
    if name == 'blocks':
        A = problems.block_band_3dof(1500, 120)      # aligned runs of three with a footprint of 49..64 granules a slice
        sigma, window = 64, 64
    elif name in ('fem', 'fem_runs_off'):
        A = problems.fem_like_3d(14 if cut else 9, 3)       # three unknowns per node, full 3 x 3 blocks: aligned runs of three columns -> one code per run
        sigma = 64
        if unsorted:                         # the runs of some rows in another order (a row's columns need not ascend)
            A = A.copy()
            for r in range(0, A.shape[0], 7):
                lo, hi = A.indptr[r], A.indptr[r + 1]
                k = (hi - lo) // 3
                o = (np.arange(k)[::-1][:, None] * 3 + np.arange(3)[None, :]).ravel()
                A.indices[lo:hi] = A.indices[lo:hi][o]; A.data[lo:hi] = A.data[lo:hi][o]
            A.has_sorted_indices = False
    elif name == 'ragged':
        n = 5000
        lens = rng.integers(40, 51, size=n)
        lens[rng.integers(0, n, size=60)] = 0
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        r = np.repeat(np.arange(n), lens)
        A = sp.csr_matrix((rng.standard_normal(r.size), (r + rng.integers(-900, 901, size=r.size)).clip(0, n - 1).astype(np.int32), indptr),
                          shape=(n, n))
    elif name.startswith('irregular'):
        A = problems.fem_irregular_3d(11)
        sigma = 256 if name.endswith('256') else 0
    elif name == 'wide_gaps':
        # rows whose consecutive columns lie up to 200,000 apart, unsorted in places: gaps beyond one 16-bit delta code cost skip positions
        n, nr = 300_000, 2000
        rows_c = []
        for i in range(nr):
            c0 = int(rng.integers(0, 40_000))
            c = np.concatenate([c0 + np.sort(rng.choice(3000, size=24, replace=False)) + off for off in (0, 120_000, 255_000)])
            if i % 97 == 0:
                c[[5, 40]] = c[[40, 5]]                    # an unsorted row: one step back by ~120,000, one forward
            rows_c.append(c)
        indptr = np.arange(nr + 1, dtype=np.int32) * 72
        A = sp.csr_matrix((rng.standard_normal(72 * nr), np.concatenate(rows_c).astype(np.int32), indptr), shape=(nr, n))
        A.has_sorted_indices = False
    else:
        full = problems.fem_like_3d(10, 3)
        A, ghost_ids = partition.localize(full[900:2100], 900, 2100)
        row_class = np.array([(A.indices[A.indptr[i]:A.indptr[i + 1]] >= 1200).any() for i in range(1200)])
        sigma = 64 if window else 0          # (WINDOW codes need consecutive rows)
    got, slices, val, col, stats, rows = plan_sell(A, row_class, sigma=sigma, allow_runs=name != 'fem_runs_off', window=window, max_overhead=6.0 if cut else 1.25)
    gran = plan_sell.gran
    assert got > 0, got
    assert (stats[10] > 0) == (window > 0) and stats[10] <= (window or 64)
    if name == 'blocks':
        assert 48 < stats[10] <= 64                      # the kernels' larger window (16 page loads a slice)
    if cut:
        assert np.count_nonzero(slices[:-1, 1] - slices[:-1, 0] < 64) > 10      # slices cut where 64 rows touch more than 24 granules
    n = A.shape[0]
    sig = int(stats[4])
    assert sig == (sigma or sig) and sig in (64, 256, 1024, 4096, 16384)
    if name == 'irregular':
        assert sig > 64                                  # consecutive rows would pad by ~30 %
        assert stats[3] <= 1.10 * A.nnz
    run = int(stats[9])
    assert run == (3 if name in ('fem', 'ghosts', 'blocks') else 1), (name, run)
    lens_all = np.diff(A.indptr)
    seen = np.zeros(n, dtype=int)
    used_v = np.zeros(int(stats[1]), dtype=bool)
    for si, sl in enumerate(slices):
        rb, re, voff, coff, width, cbase, rows_off, flags = sl
        assert 0 < re - rb <= 64
        assert (flags & 2 != 0) == (window > 0)
        lanes = slice_rows(sl, rows)
        if window:
            ng = flags >> 8
            g0 = gran[cbase:cbase + ng]
            assert 0 < ng <= window and rows_off < 0 and np.all(np.diff(g0) > 0)
        assert rows_off >= 0 or sig == 64
        rws = np.array([r for r, _ in lanes])
        seen[rws] += 1
        if row_class is not None:
            assert np.all(row_class[rws] == (si >= stats[0]))
        if sig > 64:
            assert rws.max() // sig == rws.min() // sig or row_class is not None      # rows of ONE sorting window
            assert np.all(np.diff(-(-lens_all[rws] // (8 * run))) <= 0)               # most TRIPS first (rows of one trip count keep their order)
        stored_max = 0
        for lane, (row, ln) in enumerate(lanes):
            lo, hi = A.indptr[row], A.indptr[row + 1]
            stored = (hi - lo) if ln is None else ln
            stored_max = max(stored_max, stored)
            u = np.arange(stored)
            vi = voff + ((u >> 1) * 64 + lane) * 2 + (u & 1)
            # decode as the kernel does: one code per run of `run` positions; a running column moved by code - 16384 per code (to the
            # run's first column); codes 0 / 65535 name no nonzeros
            assert stored % run == 0
            cidx = np.arange(stored // run)
            ci = coff + ((cidx >> 3) * 64 + lane) * 8 + (cidx & 7)
            code = col[ci].astype(np.int64)
            if window:
                # a code is a place in the slice's window: granule code // 16 (staged at window entries 16 g .. 16 g + 15), entry
                # code % 16; the whole run lies inside the granule
                assert np.all(code // 16 < ng) and np.all(code % 16 + run <= 16)
                start = g0[code // 16] + code % 16
                real_run = np.ones(code.size, dtype=bool)
            else:
                start = cbase + np.cumsum(code - 16384)
                real_run = (code != 0) & (code != 65535)
            running = (np.repeat(start, run) + np.tile(np.arange(run), cidx.size))
            real = np.repeat(real_run, run)
            assert real.sum() == hi - lo and (ln is not None or real.all())
            assert np.array_equal(running[real], A.indices[lo:hi])
            assert np.array_equal(val[vi][real].view(np.uint64), A.data[lo:hi].view(np.uint64))
            used_v[vi[real]] = True
            # behind the row: code 16384 (the column stays), up to the slice's width
            ncodes = ((width + run - 1) // run + 7) // 8 * 8
            tail = np.arange(stored // run, ncodes)
            assert np.all(col[coff + ((tail >> 3) * 64 + lane) * 8 + (tail & 7)] == (0 if window else 16384))
        assert width == stored_max
    assert np.all(seen == 1)
    assert np.all(val[:int(stats[1])][~used_v] == 0.0)
    assert stats[3] <= (6.0 if cut else 1.25) * max(A.nnz, 1)
    if name == 'wide_gaps':
        assert any(sl[6] >= 0 for sl in slices)          # some rows needed skips: their slices name rows and stored lengths
        assert plan_sell(A, sigma=64, max_overhead=8.0, window=48)[4][10] == 0     # 72 scattered columns per row: far beyond 48 granules a slice -> deltas
    if stats[6] == 0:
        for part in (slices[:stats[0]], slices[stats[0]:]):
            assert np.all(np.diff(part[:, 0]) > 0) or sig > 64


def test_sliced_row_table_interleaves_grid_planes():
    """Processing order of the slice table (any order is correct; this one is fast): for an operator with a dominant far
    column offset -- a 3-D discretisation in natural ordering: one grid plane -- the slices at the same place of 8
    consecutive planes are neighbours in the table, so the launch's concurrent waves (consecutive table entries) read each
    plane-neighbour row while another of them owns it.  The stride is found from the matrix alone."""
    m = 48                                                # (5184 slices: tables of fewer than 4096 stay in row order -- two rounds of a launch)
    A = problems.fem_like_3d(m, 3)
    got, slices, _, _, stats, _ = plan_sell(A, sigma=64)
    plane = 3 * m * m
    assert abs(int(stats[5]) - plane) <= 0.01 * plane, stats
    assert stats[6] == 8 and stats[4] == 64
    rb = slices[:, 0].astype(np.int64)
    assert sorted(rb.tolist()) == list(range(0, A.shape[0], 64))          # a permutation of the row-order table
    beta = int(stats[5])
    z = rb // beta
    # any 64 consecutive table entries inside a plane group hold all of its 8 planes ...
    for start in range(0, 8 * plane // 64 - 64, 17):
        assert len(set(z[start:start + 64].tolist())) == 8, start
    # ... and the plane neighbours (rows +- one plane) of an entry's rows are owned by an entry at most a few places away
    pos = {int(r): i for i, r in enumerate(rb)}
    far = []
    for i in range(0, len(rb), 7):
        for nb in (rb[i] - beta, rb[i] + beta):
            nb64 = int(nb // 64 * 64)
            if nb64 in pos and z[pos[nb64]] // 8 == z[i] // 8:
                far.append(abs(pos[nb64] - i))
    assert np.percentile(far, 95) <= 24, np.percentile(far, 95)
    # without a far stride (a band) or with planes <= 1 the table stays in row order
    got, slices, _, _, stats, _ = plan_sell(A, sigma=64, planes=1)
    assert stats[6] == 0 and np.all(np.diff(slices[:, 0]) > 0)


def test_sliced_rows_refuse_wide_and_ragged_operators():
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    n = 80_000
    lens = np.full(n, 30)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    wide = sp.csr_matrix((np.ones(indptr[-1]), rng.integers(0, n, size=indptr[-1]).astype(np.int32), indptr), shape=(n, n))
    assert plan_sell(wide)[0] == 0                       # columns all over the matrix: a position of 64 rows spans more than 16 bits, the
                                                         # slices would get a few rows each -- and pad the other lanes: refused
    lens = rng.integers(0, 100, size=4000)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    r = np.repeat(np.arange(4000), lens)
    ragged = sp.csr_matrix((np.ones(r.size), (r // 2).astype(np.int32), indptr), shape=(4000, 4000))
    assert plan_sell(ragged, sigma=64)[0] == 0           # consecutive rows: padding to each slice's longest row would double the stream
    got, _, _, _, stats, _ = plan_sell(ragged)           # a sorting window of 4096 rows brings it under 25 %
    assert got > 0 and stats[4] > 64 and stats[3] <= 1.25 * ragged.nnz


def plan_patterns(A_local, row_class=None):
    A_local = A_local.tocsr()
    n, n_cols = A_local.shape
    indptr = np.ascontiguousarray(A_local.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A_local.indices, dtype=np.int32)
    data = np.ascontiguousarray(A_local.data, dtype=np.float64)
    rc = None if row_class is None else np.ascontiguousarray(row_class, dtype=np.uint8)
    counts = np.zeros(3, dtype=np.int64)
    got = L.lib().prcg_plan_window_patterns(n, n_cols, L.ptr(indptr), L.ptr(indices), L.ptr(data), L.ptr(rc), None, 0, None, 0, None, 0,
                                            L.ptr(counts))
    if got == 0:
        return None
    assert got < 0
    tiles = np.zeros((counts[0], 24), dtype=np.int32)
    pats = np.zeros(counts[1] * 72, dtype=np.uint8)
    masks = np.zeros(counts[2], dtype=np.uint16)
    got = L.lib().prcg_plan_window_patterns(n, n_cols, L.ptr(indptr), L.ptr(indices), L.ptr(data), L.ptr(rc), L.ptr(tiles), len(tiles),
                                            L.ptr(pats), int(counts[1]), L.ptr(masks), len(masks), L.ptr(counts))
    assert got == 1
    rec = np.dtype([('nslots', '<i4'), ('vsel', '<u4'), ('cb', '<i2', 16), ('val', '<f8', 4)])
    assert rec.itemsize == 72
    return tiles, pats.view(rec), masks


def matrix_from_patterns(shape, tiles, pats, masks):
    """What the pattern kernels compute with: (pages, pattern, masks) -> every nonzero's column and value, row by row IN THE
    ORDER THE KERNEL ADDS THEM (ascending slot) -- CSR arrays to be compared with the caller's as they are."""
    per_row = {}
    for t in tiles:
        rb, re, npages, page = int(t[0]), int(t[1]), int(t[4]) & 255, t[8:20]
        p = pats[int(t[20])]
        U = int(p['nslots'])
        for lane in range(re - rb):
            mk = (1 << U) - 1 if t[23] else int(masks[int(t[22]) + lane])
            ent = []
            for u in range(U):
                if (mk >> u) & 1:
                    w = lane + int(p['cb'][u])
                    assert 0 <= w < npages * 64
                    ent.append((int(page[w // 64]) + w % 64, p['val'][(int(p['vsel']) >> (2 * u)) & 3]))
            assert rb + lane not in per_row
            per_row[rb + lane] = ent
    assert sorted(per_row) == list(range(shape[0]))
    lens = np.array([len(per_row[r]) for r in range(shape[0])])
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.array([c for r in range(shape[0]) for c, _ in per_row[r]], dtype=np.int64)
    data = np.array([v for r in range(shape[0]) for _, v in per_row[r]], dtype=np.float64)
    return indptr, indices, data


@pytest.mark.parametrize('name', ['lap3d', 'lap2d', 'lap3d_aniso', 'lap3d_block', 'nine_point'])
def test_pattern_tiles_hold_exactly_the_matrix(name):
    """Constant-coefficient stencils become PATTERN tiles (csrc/prcg_plan.h: plan_window_patterns): the matrix rebuilt from
    (pages, pattern records, slot masks) -- all the pattern kernels read of the operator -- equals the CSR, nonzero by nonzero
    in row order, value bits included; a handful of patterns and mask images serve the whole grid."""
    import scipy.sparse as sp
    from new_cg_variants_amd import partition
    row_class = None
    if name == 'lap3d':
        A = problems.laplace_3d(24, 20, 13)
    elif name == 'lap2d':
        A = problems.laplace_2d(150, 70)
    elif name == 'lap3d_aniso':
        # three different couplings + the diagonal: four value patterns, the most a pattern holds
        nx, ny, nz = 17, 11, 9
        ex, ey, ez = (np.ones(k) for k in (nx, ny, nz))
        T = lambda k, c: sp.diags([-c * np.ones(k - 1), 2 * c * np.ones(k), -c * np.ones(k - 1)], [-1, 0, 1])
        A = (sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), T(nx, 1.0))) + sp.kron(sp.eye(nz), sp.kron(T(ny, 0.25), sp.eye(nx))) +
             sp.kron(T(nz, 3.0), sp.eye(ny * nx))).tocsr()
        A.sort_indices()
    elif name == 'lap3d_block':
        # a rank's row block with ghost columns on both sides
        Afull = problems.laplace_3d(16, 16, 12)
        lo, hi = 16 * 16 * 4, 16 * 16 * 8
        A, ghosts = partition.localize(Afull[lo:hi], lo, hi)
        rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
        row_class = np.zeros(A.shape[0], dtype=np.uint8)
        row_class[np.unique(rows[A.indices >= A.shape[0]])] = 1
    else:
        n1, n2 = 90, 40
        T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
        S = sp.diags([np.ones(n2 - 1), np.ones(n2), np.ones(n2 - 1)], [-1, 0, 1])
        A = (-sp.kron(S, T) + 9.0 * sp.eye(n1 * n2)).tocsr()
        A.sort_indices()
    got = plan_patterns(A, row_class)
    assert got is not None, 'a constant-coefficient stencil must qualify'
    tiles, pats, masks = got
    indptr, indices, data = matrix_from_patterns(A.shape, tiles, pats, masks)
    assert np.array_equal(indptr, A.indptr) and np.array_equal(indices, A.indices)         # the rows' own order (ghosts: not ascending)
    assert np.array_equal(data.view(np.uint64), A.data.view(np.uint64))
    # every row in exactly one tile, classes apart
    cover = np.concatenate([np.arange(t[0], t[1]) for t in tiles])
    assert np.array_equal(np.sort(cover), np.arange(A.shape[0]))
    assert len(pats) <= 64 and len(masks) // 64 <= len(tiles)
    print(name, len(tiles), 'tiles', len(pats), 'patterns', len(masks) // 64, 'mask images', int(tiles[:, 23].sum()), 'full tiles')


def test_pattern_tiles_refuse_what_is_no_constant_stencil():
    """Varying coefficients (one value per slot is the rule), unsorted or duplicate columns, rows of more than 16 nonzeros,
    more than four distinct values: no pattern operator -- the stream geometries take it."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    A = problems.laplace_2d(80, 40).tocsr()
    B = A.copy(); B.data = B.data * (1.0 + 1e-3 * rng.standard_normal(B.nnz))
    assert plan_patterns(B) is None
    assert plan_patterns(problems.banded_ex2b(4096, 7)) is None        # ex2b: the diagonal varies near the end
    C1 = A.copy()
    r = 1000
    C1.indices[C1.indptr[r]:C1.indptr[r + 1]] = C1.indices[C1.indptr[r]:C1.indptr[r + 1]][::-1].copy()     # unsorted row
    C1.data[C1.indptr[r]:C1.indptr[r + 1]] = C1.data[C1.indptr[r]:C1.indptr[r + 1]][::-1].copy()
    assert plan_patterns(C1) is None
    D = sp.diags([np.ones(5000 - abs(k)) for k in range(-9, 10)], list(range(-9, 10))).tocsr()               # 19 per row
    assert plan_patterns(D) is None
    E = sp.diags([np.full(3000 - abs(k), 1.0 + abs(k)) for k in range(-3, 4)], list(range(-3, 4))).tocsr()    # 4 values: fine
    assert plan_patterns(E) is not None
    F = sp.diags([np.full(3000 - abs(k), 1.0 + k) for k in range(-3, 4)], list(range(-3, 4))).tocsr()         # 7 values: too many
    assert plan_patterns(F) is None


def test_sweep_tables_carry_exactly_the_pages_they_claim():
    """Sweep order of the pattern tiles (csrc/prcg_plan.h: plan_sweep_tiles), checked through prcg_plan_window_patterns' sibling
    prcg_plan_sweep: (a) the matrix rebuilt from (pages, patterns, masks) equals the CSR row by row, as for any tiling; (b) an
    LDS model of every wave -- slot s takes tiles s, s + waves, ...; a tile parks its non-carried pages in the slots its `perm`
    names -- holds, for every tile, exactly the page each logical page claims, carried or not; (c) carried pages are the
    majority for a 3-D stencil (z-, own of every plane after a sweep's first)."""
    for dims in ((72, 8, 200), (100, 9, 96), (300, 64)):
        A = (problems.laplace_3d(*dims) if len(dims) == 3 else problems.laplace_2d(*dims)).tocsr()
        n = A.shape[0]
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        data = np.ascontiguousarray(A.data, dtype=np.float64)
        counts = np.zeros(8, dtype=np.int64)
        got = L.lib().prcg_plan_sweep(n, L.ptr(indptr), L.ptr(indices), L.ptr(data), 320, None, 0, None, 0, None, 0, L.ptr(counts))
        assert got < 0, (dims, got)
        tiles = np.zeros((counts[0], 24), dtype=np.int32)
        pats = np.zeros(counts[1] * 72, dtype=np.uint8)
        masks = np.zeros(counts[2], dtype=np.uint16)
        got = L.lib().prcg_plan_sweep(n, L.ptr(indptr), L.ptr(indices), L.ptr(data), 320, L.ptr(tiles), len(tiles), L.ptr(pats), int(counts[1]),
                                      L.ptr(masks), len(masks), L.ptr(counts))
        assert got == 1
        waves, plane, rows_per_tile = int(counts[3]), int(counts[4]), int(counts[5])
        assert plane == (dims[0] * dims[1] if len(dims) == 3 else dims[0]) and waves % 4 == 0 and len(tiles) % waves == 0
        rec = np.dtype([('nslots', '<i4'), ('vsel', '<u4'), ('cb', '<i2', 16), ('val', '<f8', 4)])
        pats = pats.view(rec)
        real = tiles[tiles[:, 1] > tiles[:, 0]]
        ip2, ix2, dt2 = matrix_from_patterns(A.shape, real, pats, masks)
        assert np.array_equal(ip2, A.indptr) and np.array_equal(ix2, A.indices) and np.array_equal(dt2.view(np.uint64), A.data.view(np.uint64))
        carried = loaded = 0
        for s in range(waves):
            held = [None] * 8
            for t in tiles[s::waves]:
                if t[1] <= t[0]:
                    continue
                npages, vdf = int(t[4]) & 255, int(t[6])
                perm = [(vdf >> (3 * p)) & 7 for p in range(npages)]
                carry = (vdf >> 18) & 63
                assert len(set(perm)) == npages and max(perm) < 6
                for p in range(npages):
                    if (carry >> p) & 1:
                        assert (vdf >> 24) & 1 and held[perm[p]] == int(t[8 + p]), (s, t[:8], p)
                        carried += 1
                    else:
                        held[perm[p]] = int(t[8 + p])
                        loaded += 1
                # the row's own entry: PHYSICAL window index of row rb
                own = (int(t[4]) >> 8) & 0xffff
                lp = perm.index(own // 64)
                assert int(t[8 + lp]) + own % 64 == int(t[0])
        print(dims, 'waves', waves, 'rows per tile', rows_per_tile, 'pages carried', carried, 'loaded', loaded)
        assert carried > 0.3 * (carried + loaded)
    # no sweep for what is no stencil on full planes
    B = problems.banded_ex2b(8192, 7).tocsr()
    counts = np.zeros(8, dtype=np.int64)
    assert L.lib().prcg_plan_sweep(B.shape[0], L.ptr(np.ascontiguousarray(B.indptr, dtype=np.int32)), L.ptr(np.ascontiguousarray(B.indices, dtype=np.int32)),
                                   L.ptr(np.ascontiguousarray(B.data)), 4096, None, 0, None, 0, None, 0, L.ptr(counts)) == 0
