"""Row-block partition of a CSR operator and the halo plan between the blocks.

Pure host logic (NumPy), no GPU and no communication library: the exchange of the
"who needs what" lists is delegated to an ``allgather(obj) -> [obj_rank0, ...]``
callable (torch.distributed.all_gather_object in production, a plain list in tests).

Layout produced for rank ``r`` owning global rows ``[lo, hi)``:

* local columns ``[0, n_local)``          = owned entries (global id - lo)
* local columns ``[n_local, n_local+g)``  = ghosts, sorted by global id, hence grouped
  by owning rank in ascending order; ``recv_ptr`` delimits the groups;
* ``send_idx[send_ptr[q]:send_ptr[q+1]]`` = local rows whose entries peer ``q`` asked
  for, in the order of that peer's ghost list (= ascending global id).

The order of the nonzeros inside each row is left untouched, so a distributed SpMV sums
every row in exactly the order the single-GPU (and SciPy) product does.

This replaces the reference's dense column blocks + full-vector Allreduce
(scaling_experiments_mpi4py/scaling_tests.py:51-54, cg_variants/pipe_pr_cg.py:65-67)
with the row-block layout PETSc's MPIAIJ uses in its third experiment
(scaling_experiments_petsc/ex2b.c:67-71).
"""
import numpy as np
import scipy.sparse as sp


def even_offsets(n, nranks):
    """Contiguous row ranges of (almost) equal length: offsets[r]..offsets[r+1]."""
    base, extra = divmod(int(n), int(nranks))
    sizes = np.full(nranks, base, dtype=np.int64)
    sizes[:extra] += 1
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def nnz_balanced_offsets(indptr, nranks):
    """Split points that balance the nonzeros (prefix sum over the row pointer)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n = len(indptr) - 1
    targets = indptr[-1] * np.arange(1, nranks) / nranks
    cuts = np.searchsorted(indptr, targets, side='left')
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(off)


def localize(A_rows, lo, hi):
    """Renumber the columns of the row block ``A_rows`` (shape n_local x n_global, global
    column ids) to the local layout.  Returns (A_local, ghost_ids)."""
    A_rows = A_rows.tocsr()
    n_local = hi - lo
    assert A_rows.shape[0] == n_local
    cols = A_rows.indices.astype(np.int64, copy=False)
    owned = (cols >= lo) & (cols < hi)
    ghost_ids = np.unique(cols[~owned])
    new_cols = np.empty(cols.shape, dtype=np.int32)
    new_cols[owned] = (cols[owned] - lo).astype(np.int32)
    if ghost_ids.size:
        new_cols[~owned] = (n_local + np.searchsorted(ghost_ids, cols[~owned])).astype(np.int32)
    A_local = sp.csr_matrix((A_rows.data, new_cols, A_rows.indptr), shape=(n_local, n_local + ghost_ids.size))
    return A_local, ghost_ids


def plan_halo(ghost_ids, offsets, rank, allgather):
    """Build the halo plan of ``rank`` (dict with peers, send_ptr, send_idx, recv_ptr)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    nranks = len(offsets) - 1
    lo = offsets[rank]
    owner = np.searchsorted(offsets, ghost_ids, side='right') - 1 if ghost_ids.size else np.zeros(0, dtype=np.int64)
    assert not np.any(owner == rank)
    wants = {int(q): ghost_ids[owner == q] for q in np.unique(owner)}        # what I need, per owner
    everyone = allgather(wants)                                              # list over ranks
    gives = {q: np.asarray(everyone[q][rank], dtype=np.int64)
             for q in range(nranks) if q != rank and rank in everyone[q] and len(everyone[q][rank])}
    peers = sorted(set(wants) | set(gives))
    send_ptr, recv_ptr, send_idx = [0], [0], []
    for q in peers:
        s = gives.get(q, np.zeros(0, dtype=np.int64))
        send_idx.append((s - lo).astype(np.int32))
        send_ptr.append(send_ptr[-1] + len(s))
        recv_ptr.append(recv_ptr[-1] + len(wants.get(q, ())))
    send_idx = np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int32)
    n_local = offsets[rank + 1] - lo
    assert send_idx.size == 0 or (send_idx.min() >= 0 and send_idx.max() < n_local)
    assert recv_ptr[-1] == ghost_ids.size
    return {'peers': np.asarray(peers, dtype=np.int32), 'send_ptr': np.asarray(send_ptr, dtype=np.int64),
            'send_idx': send_idx, 'recv_ptr': np.asarray(recv_ptr, dtype=np.int64)}


def split_serial(A, nranks, offsets=None):
    """Partition a whole matrix in one process (tests, single-process drivers): returns
    (offsets, [(A_local, ghost_ids, halo)] per rank)."""
    A = A.tocsr()
    n = A.shape[0]
    offsets = even_offsets(n, nranks) if offsets is None else np.asarray(offsets, dtype=np.int64)
    local = [localize(A[offsets[r]:offsets[r + 1]], offsets[r], offsets[r + 1]) for r in range(nranks)]
    wants_all = []
    for r in range(nranks):
        g = local[r][1]
        owner = np.searchsorted(offsets, g, side='right') - 1 if g.size else np.zeros(0, dtype=np.int64)
        wants_all.append({int(q): g[owner == q] for q in np.unique(owner)})
    out = []
    for r in range(nranks):
        halo = plan_halo(local[r][1], offsets, r, lambda _w: wants_all)
        out.append((local[r][0], local[r][1], halo))
    return offsets, out


def receive_offsets(halo):
    """{peer rank: first slot of its rows in this rank's ghost area} of a halo plan -- what every OTHER rank needs to
    know to store its rows straight into this rank's exchange buffer (direct peer exchange)."""
    if halo is None:
        return {}
    return {int(q): int(halo['recv_ptr'][i]) for i, q in enumerate(halo['peers'])}


def connect_peer_exchange(dev, rank, allgather):
    """Collective over the ranks of a run (``allgather(obj) -> [obj of rank 0, ...]``): give every rank's device handle
    the direct peer exchange (prcg.h: prcg_peer_setup / prcg_peer_connect) -- exchange buffers allocated with one common
    ghost capacity, IPC handles (or, for ranks that share the process, device addresses) swapped, destination offsets
    taken from the receivers' halo plans.  Returns True if every rank connected; otherwise every rank stays on the
    RCCL schedules (the decision is the same everywhere)."""
    import os
    halo = getattr(dev, 'halo', None)
    infos = allgather({'g': int(dev.n_ghost), 'recv': receive_offsets(halo)})
    cap = max(i['g'] for i in infos)
    ok, err = True, None
    try:
        handle, ptr = dev.peer_setup(cap)
    except Exception as exc:           # noqa: BLE001 -- decided collectively below
        ok, err, handle, ptr = False, exc, b'\0' * 64, 0
    mine = os.getpid()
    everyone = allgather({'ok': ok, 'handle': handle, 'pid': mine, 'ptr': ptr})
    if all(e['ok'] for e in everyone):
        try:
            peers = [] if halo is None else [int(q) for q in halo['peers']]
            dst = [infos[q]['recv'][rank] for q in peers]
            dev.peer_connect([e['handle'] for e in everyone], [e['ptr'] if e['pid'] == mine else 0 for e in everyone], dst)
            if not dev.layout()['window']:
                # only the window kernels carry the exchange: a rank whose block is no window operator would never send
                ok, err = False, RuntimeError('this rank\'s row block is no window operator')
        except Exception as exc:       # noqa: BLE001
            ok, err = False, exc
    else:
        ok = False
    flags = allgather(bool(ok))
    if not all(flags):
        try:
            dev.peer_setup(cap)        # (resets a connected handle: all ranks or none)
        except Exception:              # noqa: BLE001
            pass
        if err is not None:
            import sys
            print(f'[prcg] rank {rank}: direct peer exchange not available ({err}); using the RCCL schedule', file=sys.stderr)
        return False
    return True


def loopback_problem(A, k, cut=None):
    """One rank's view of a row-block run, on ONE GPU: rewrite A as if it were cut at row n/2 into two
    row blocks whose halo is exchanged with ... itself.  Columns in [n/2-k, n/2) seen from rows >= n/2
    and columns in [n/2, n/2+k) seen from rows < n/2 are reached through ghost slots fed by a self
    exchange.  A_loop @ [x ; x[ghost_ids]] == A @ x with the nonzeros of every row in unchanged order.
    ``cut``: the row of the cut instead of n/2 (a grid-plane boundary of a 3-D stencil, with k = rows of a plane).
    Returns (A_loop, halo plan, nonzeros that go through a ghost slot).  Used by the tests and by
    bench.py's multi-rank-schedule leg (boundary tiles + ghost rows with a 1-rank communicator)."""
    A = A.tocsr()
    n = A.shape[0]
    h = n // 2 if cut is None else int(cut)
    assert k <= h <= n - k
    ghost_ids = np.arange(h - k, h + k)
    slot = -np.ones(n, dtype=np.int64)
    slot[ghost_ids] = n + np.arange(ghost_ids.size)
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    cols = A.indices.astype(np.int64).copy()
    via_ghost = ((cols >= h - k) & (cols < h) & (rows >= h)) | ((cols >= h) & (cols < h + k) & (rows < h))
    cols[via_ghost] = slot[cols[via_ghost]]
    A_loop = sp.csr_matrix((A.data, cols.astype(np.int32), A.indptr), shape=(n, n + ghost_ids.size))
    halo = {'peers': np.array([0], dtype=np.int32), 'send_ptr': np.array([0, ghost_ids.size], dtype=np.int64),
            'send_idx': ghost_ids.astype(np.int32), 'recv_ptr': np.array([0, ghost_ids.size], dtype=np.int64)}
    return A_loop, halo, int(via_ghost.sum())
