// TEST INFRASTRUCTURE -- not part of the product, never loaded by it unless a test names it.
//
// A stand-in for librccl.so that connects RANKS LIVING IN THREADS OF ONE PROCESS ON ONE GPU, so that the
// N > 1 path of libprcg.so (row blocks, halo plan, merged all-gather, interior / boundary tiles, the
// one-launch schedule whose waves wait for data that depends on ANOTHER rank's kernels) can run with real
// inter-rank data on a one-GPU box, where RCCL refuses two ranks ("Duplicate GPU detected").
//
// It exports exactly the ten nccl* symbols prcg_rccl.cpp resolves (scaling_experiments_mpi4py uses the MPI
// counterparts: comm.Allreduce, scaling_tests.py:42-59) and implements them with HIP events and
// device-to-device copies on the caller's stream:
//
//   * every rank calls the same collectives in the same order (as with NCCL); call number i of one rank
//     meets call number i of the others in slot i % kRing of a shared table;
//   * a call records a "ready" event on the caller's stream, waits on the host until the ranks it reads from
//     have published call i, makes its stream wait for their ready events, enqueues its copies, records a
//     "copied" event, and before returning waits (host, then stream) for the copied events of the ranks that
//     read ITS buffer -- later work on the stream may overwrite that buffer;
//   * ncclSend / ncclRecv are collected between ncclGroupStart / ncclGroupEnd and executed as one such call.
//
// Semantics only, no performance: every call is a host rendezvous of all ranks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kRing = 64;
constexpr int kMaxRanks = 8;

struct P2P { int peer; const void* sbuf; void* rbuf; size_t bytes; bool is_send; };

struct Slot {
    long seq = -1;                       // which call this slot currently holds
    const void* sendbuf = nullptr;
    size_t bytes = 0;
    std::vector<P2P> p2p;
    bool copied = false;                 // "copied" event recorded for seq
    hipEvent_t ready = nullptr, done = nullptr;
};

struct Group {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0, joined = 0, left = 0;
    Slot slots[kMaxRanks][kRing];
    long finished[kMaxRanks];            // calls fully returned, per rank
    double* scratch[kMaxRanks];
};

struct Comm {
    Group* g;
    int rank;
    long next = 0;                       // next call number of this rank
    bool in_group = false;
    std::vector<P2P> pending;
    hipStream_t group_stream = nullptr;
};

std::mutex g_mu;
std::map<std::string, Group*> g_groups;
std::atomic<unsigned> g_ids{1};
thread_local int tl_group_depth = 0;
thread_local Comm* tl_group_comm = nullptr;

__global__ void k_reduce_ranks(const double* scratch, double* out, size_t count, int nranks, int is_max) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double v = scratch[i];
        for (int r = 1; r < nranks; ++r) {
            const double w = scratch[(size_t)r * count + i];
            v = is_max ? (w > v ? w : v) : v + w;                 // rank order: the same bits on every rank
        }
        out[i] = v;
    }
}

#define HIPOK(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)

// begin call number c->next: wait for the slot, record "ready", publish
ncclResult_t open_call(Comm* c, hipStream_t st, const void* sendbuf, size_t bytes, const std::vector<P2P>& p2p, long* seq_out) {
    Group* g = c->g;
    const long seq = c->next++;
    Slot& s = g->slots[c->rank][seq % kRing];
    {
        std::unique_lock<std::mutex> lk(g->mu);
        // the slot's previous call (seq - kRing) must have been finished by every rank
        g->cv.wait(lk, [&] {
            for (int r = 0; r < g->nranks; ++r) if (g->finished[r] < seq - kRing + 1) return false;
            return true;
        });
    }
    if (!s.ready) { HIPOK(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming)); HIPOK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming)); }
    HIPOK(hipEventRecord(s.ready, st));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        s.sendbuf = sendbuf; s.bytes = bytes; s.p2p = p2p; s.copied = false; s.seq = seq;
    }
    g->cv.notify_all();
    *seq_out = seq;
    return ncclSuccess;
}

// peer's slot for call seq, once published
Slot& peer_slot(Group* g, int peer, long seq) {
    Slot& s = g->slots[peer][seq % kRing];
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return s.seq == seq; });
    return s;
}

// end of call: my copies are enqueued -> record "copied"; then make later work on my stream wait for the
// ranks that read my buffer
ncclResult_t close_call(Comm* c, hipStream_t st, long seq, const std::vector<int>& readers) {
    Group* g = c->g;
    Slot& s = g->slots[c->rank][seq % kRing];
    HIPOK(hipEventRecord(s.done, st));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        s.copied = true;
    }
    g->cv.notify_all();
    for (int r : readers) {
        Slot& ps = g->slots[r][seq % kRing];
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return ps.seq == seq && ps.copied; });
        }
        HIPOK(hipStreamWaitEvent(st, ps.done, 0));
    }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->finished[c->rank] = seq + 1;
    }
    g->cv.notify_all();
    return ncclSuccess;
}

std::vector<int> everyone_but(Comm* c) {
    std::vector<int> v;
    for (int r = 0; r < c->g->nranks; ++r) if (r != c->rank) v.push_back(r);
    return v;
}

ncclResult_t run_p2p(Comm* c, hipStream_t st, const std::vector<P2P>& ops) {
    long seq;
    ncclResult_t rc = open_call(c, st, nullptr, 0, ops, &seq);
    if (rc != ncclSuccess) return rc;
    Group* g = c->g;
    std::vector<int> readers;
    for (const P2P& op : ops) {
        if (op.is_send) { if (op.peer != c->rank) readers.push_back(op.peer); continue; }
        // receive: find the matching send in the peer's call (k-th receive from a peer <-> its k-th send to me)
        int nth = 0;
        for (const P2P& q : ops) { if (&q == &op) break; if (!q.is_send && q.peer == op.peer) ++nth; }
        Slot& ps = peer_slot(g, op.peer, seq);
        const P2P* match = nullptr;
        int seen = 0;
        for (const P2P& q : ps.p2p)
            if (q.is_send && q.peer == c->rank) { if (seen == nth) { match = &q; break; } ++seen; }
        if (!match || match->bytes != op.bytes) return ncclInvalidUsage;
        HIPOK(hipStreamWaitEvent(st, ps.ready, 0));
        HIPOK(hipMemcpyAsync(op.rbuf, match->sbuf, op.bytes, hipMemcpyDeviceToDevice, st));
    }
    return close_call(c, st, seq, readers);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "threads-ccl-%u", g_ids.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group();
            g->nranks = nranks;
            for (int r = 0; r < kMaxRanks; ++r) { g->finished[r] = 0; g->scratch[r] = nullptr; }
            g_groups[key] = g;
        } else {
            g = it->second;
        }
    }
    if (hipMalloc(&g->scratch[rank], (size_t)kMaxRanks * (1 << 20)) != hipSuccess) return ncclUnhandledCudaError;
    Comm* c = new Comm{g, rank};
    {
        std::unique_lock<std::mutex> lk(g->mu);
        ++g->joined;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= g->nranks; });
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    (void)hipFree(c->g->scratch[c->rank]);
    delete c;                       // (groups and their events live until the process ends: test infrastructure)
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t st) {
    if (dt != ncclDouble) return ncclInvalidArgument;
    Comm* c = reinterpret_cast<Comm*>(comm);
    Group* g = c->g;
    const size_t bytes = sendcount * sizeof(double);
    long seq;
    ncclResult_t rc = open_call(c, st, sendbuff, bytes, {}, &seq);
    if (rc != ncclSuccess) return rc;
    char* out = static_cast<char*>(recvbuff);
    if (out + (size_t)c->rank * bytes != sendbuff)
        HIPOK(hipMemcpyAsync(out + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice, st));
    for (int r : everyone_but(c)) {
        Slot& ps = peer_slot(g, r, seq);
        if (ps.bytes != bytes) return ncclInvalidUsage;
        HIPOK(hipStreamWaitEvent(st, ps.ready, 0));
        HIPOK(hipMemcpyAsync(out + (size_t)r * bytes, ps.sendbuf, bytes, hipMemcpyDeviceToDevice, st));
    }
    return close_call(c, st, seq, everyone_but(c));
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || (op != ncclSum && op != ncclMax) || count * sizeof(double) > (1u << 20)) return ncclInvalidArgument;
    Comm* c = reinterpret_cast<Comm*>(comm);
    Group* g = c->g;
    const size_t bytes = count * sizeof(double);
    long seq;
    ncclResult_t rc = open_call(c, st, sendbuff, bytes, {}, &seq);
    if (rc != ncclSuccess) return rc;
    double* scr = g->scratch[c->rank];
    HIPOK(hipMemcpyAsync(scr + (size_t)c->rank * count, sendbuff, bytes, hipMemcpyDeviceToDevice, st));
    for (int r : everyone_but(c)) {
        Slot& ps = peer_slot(g, r, seq);
        if (ps.bytes != bytes) return ncclInvalidUsage;
        HIPOK(hipStreamWaitEvent(st, ps.ready, 0));
        HIPOK(hipMemcpyAsync(scr + (size_t)r * count, ps.sendbuf, bytes, hipMemcpyDeviceToDevice, st));
    }
    // (in place: the result may overwrite what the others still read -- close_call makes the stream wait for them first)
    rc = close_call(c, st, seq, everyone_but(c));
    if (rc != ncclSuccess) return rc;
    hipLaunchKernelGGL(k_reduce_ranks, dim3(1), dim3(256), 0, st, scr, static_cast<double*>(recvbuff), count, g->nranks,
                       op == ncclMax ? 1 : 0);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclGroupStart() { ++tl_group_depth; return ncclSuccess; }

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || tl_group_depth == 0) return ncclInvalidUsage;
    Comm* c = reinterpret_cast<Comm*>(comm);
    tl_group_comm = c; c->group_stream = st;
    c->pending.push_back(P2P{peer, sendbuff, nullptr, count * sizeof(double), true});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || tl_group_depth == 0) return ncclInvalidUsage;
    Comm* c = reinterpret_cast<Comm*>(comm);
    tl_group_comm = c; c->group_stream = st;
    c->pending.push_back(P2P{peer, nullptr, recvbuff, count * sizeof(double), false});
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (--tl_group_depth > 0) return ncclSuccess;
    Comm* c = tl_group_comm;
    tl_group_comm = nullptr;
    if (!c) return ncclSuccess;                                     // an empty group
    std::vector<P2P> ops;
    ops.swap(c->pending);
    return run_p2p(c, c->group_stream, ops);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "HIP call failed (threads_ccl)";
    case ncclInvalidArgument: return "invalid argument (threads_ccl)";
    case ncclInvalidUsage: return "invalid usage (threads_ccl)";
    default: return "error (threads_ccl)";
    }
}

}  // extern "C"
