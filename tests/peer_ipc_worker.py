"""TEST INFRASTRUCTURE: one rank of the two-PROCESS peer-exchange test (tests/test_distributed.py).  Every rank is a
process of its own on the same GPU; exchange buffers are mapped across the processes with hipIpc handles, exactly as
between the ranks of a real multi-GPU run (RCCL refuses two ranks on one device, the peer exchange does not need it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def slot_of(rank, k):
    return np.array([rank + 1.0 + k, 10.0 * (rank + 1), 0.5 ** k, -float(rank), 7.0])


def rows_of(ids, k):
    ids = np.asarray(ids, dtype=np.float64)
    return np.stack([ids + 0.25 * k, -2.0 * ids - k], axis=1)


def main(rank, nranks, conn, rounds=6):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from new_cg_variants_amd import partition, problems
        from new_cg_variants_amd.device import DeviceCSR
        A = problems.banded_ex2b(6000, 7)
        offsets, parts = partition.split_serial(A, nranks)
        A_local, ghost_ids, halo = parts[rank]
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        dev = DeviceCSR(A_local, halo=halo, world=(rank, nranks))

        def allgather(obj):                 # lockstep with the parent: send mine, receive everybody's
            conn.send(('gather', obj))
            return conn.recv()

        ok = partition.connect_peer_exchange(dev, rank, allgather)
        bad = []
        if ok:
            for k in range(rounds):
                sums, ghost = dev.peer_selftest(k, rows_of(np.arange(lo, hi), k), slot_of(rank, k))
                want = slot_of(0, k)
                for r in range(1, nranks):
                    want = want + slot_of(r, k)             # rank order, as the device adds them
                if not np.array_equal(sums, want):
                    bad.append(('sums', k, sums.tolist(), want.tolist()))
                if not np.array_equal(ghost, rows_of(ghost_ids, k)):
                    bad.append(('ghost rows', k, int(np.sum(ghost != rows_of(ghost_ids, k)))))
        dev.close()
        conn.send(('done', {'rank': rank, 'connected': bool(ok), 'pid': os.getpid(), 'bad': bad}))
    except Exception as exc:                # noqa: BLE001 -- reported to the parent
        import traceback
        conn.send(('done', {'rank': rank, 'connected': False, 'pid': os.getpid(), 'bad': [('exception', repr(exc), traceback.format_exc()[-1500:])]}))
