// TEST INFRASTRUCTURE -- not part of the product, never loaded by it unless a test names it (PRCG_RCCL_LIB).
//
// A stand-in for librccl.so that connects RANKS LIVING IN SEPARATE PROCESSES THAT SHARE ONE GPU, so that the whole
// N > 1 path as the driver launches it -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`:
// gloo control plane, one process per rank, RowBlockOperator, exchange buffers mapped across processes with hipIpc,
// the one-launch schedule whose waves wait for ANOTHER PROCESS's kernels -- can be rehearsed on a one-GPU box, where
// RCCL refuses two ranks on one device ("Duplicate GPU detected").  tests/transport/threads_ccl.hip does the same
// for ranks in threads of one process.
//
// It exports the ten nccl* symbols prcg_rccl.cpp resolves.  Semantics only, no performance: every call is a host
// rendezvous through a POSIX shared-memory segment (named by the unique id) and device-to-device copies out of
// per-rank STAGING buffers that every rank maps with hipIpcOpenMemHandle:
//   collectives (all-reduce, all-gather): call number i of one rank meets call number i of the others -- stage my
//     contribution, publish `posted`, wait for the others', copy theirs, publish `consumed` (a rank overwrites its
//     staging area only after everybody consumed its previous call);
//   send / recv (collected between ncclGroupStart / ncclGroupEnd): one mailbox per ordered pair of ranks with its
//     own message counters, so a rank without neighbours takes no part and disturbs nobody's numbering.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kCollBytes = (size_t)16 << 20;        // staging: one collective contribution
constexpr size_t kBoxBytes = (size_t)4 << 20;          // staging: one mailbox (all sends of one group to one rank)
constexpr size_t kStageBytes = kCollBytes + kMaxRanks * kBoxBytes;
constexpr int kMaxOps = 32;
constexpr double kTimeoutSeconds = 120.0;

struct Shm {
    std::atomic<int> joined;
    std::atomic<int> handle_ready[kMaxRanks];
    hipIpcMemHandle_t handle[kMaxRanks];
    // collectives
    std::atomic<long> posted[kMaxRanks], consumed[kMaxRanks];
    unsigned long long coll_bytes[kMaxRanks];
    // mailboxes: [src][dst]
    std::atomic<long> box_posted[kMaxRanks][kMaxRanks], box_consumed[kMaxRanks][kMaxRanks];
    int box_nops[kMaxRanks][kMaxRanks];
    unsigned long long box_op_bytes[kMaxRanks][kMaxRanks][kMaxOps];
};

struct P2P { int peer; const void* sbuf; void* rbuf; size_t bytes; bool is_send; };

struct Comm {
    Shm* shm = nullptr;
    int rank = 0, nranks = 0;
    long next = 0;                        // next collective call number
    long sent[kMaxRanks] = {}, received[kMaxRanks] = {};
    char* stage = nullptr;                // my staging buffer
    char* peer_stage[kMaxRanks] = {};     // everybody's, mapped
    double* scratch = nullptr;
    std::vector<P2P> pending;
    hipStream_t group_stream = nullptr;
};

thread_local int tl_group_depth = 0;
thread_local Comm* tl_group_comm = nullptr;
std::atomic<unsigned> g_ids{1};

#define HIPOK(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)

template <typename F>
bool wait_until(F&& cond) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!cond()) {
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) return false;
    }
    return true;
}

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

// stage my contribution of collective call `seq`, publish it, wait for everybody's
ncclResult_t coll_begin(Comm* c, hipStream_t st, const void* sendbuf, size_t bytes, long* seq_out) {
    if (bytes > kCollBytes) return ncclInvalidArgument;
    Shm* s = c->shm;
    const long seq = c->next++;
    // everybody has finished reading my previous contribution
    if (!wait_until([&] { for (int r = 0; r < c->nranks; ++r) if (s->consumed[r].load(std::memory_order_acquire) < seq) return false; return true; }))
        return ncclInternalError;
    HIPOK(hipMemcpyAsync(c->stage, sendbuf, bytes, hipMemcpyDeviceToDevice, st));
    HIPOK(hipStreamSynchronize(st));
    s->coll_bytes[c->rank] = bytes;
    s->posted[c->rank].store(seq + 1, std::memory_order_release);
    if (!wait_until([&] { for (int r = 0; r < c->nranks; ++r) if (s->posted[r].load(std::memory_order_acquire) < seq + 1) return false; return true; }))
        return ncclInternalError;
    for (int r = 0; r < c->nranks; ++r) if (s->coll_bytes[r] != bytes) return ncclInvalidUsage;
    *seq_out = seq;
    return ncclSuccess;
}

ncclResult_t coll_end(Comm* c, hipStream_t st, long seq) {
    HIPOK(hipStreamSynchronize(st));                                // my copies out of the others' staging areas are done
    c->shm->consumed[c->rank].store(seq + 1, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t run_p2p(Comm* c, hipStream_t st, const std::vector<P2P>& ops) {
    Shm* s = c->shm;
    const int me = c->rank;
    // sends: all of this group's sends to one rank are ONE message in that rank's mailbox of my staging buffer
    for (int dst = 0; dst < c->nranks; ++dst) {
        std::vector<const P2P*> mine;
        for (const P2P& op : ops) if (op.is_send && op.peer == dst) mine.push_back(&op);
        if (mine.empty()) continue;
        if ((int)mine.size() > kMaxOps) return ncclInvalidUsage;
        const long k = c->sent[dst];
        if (!wait_until([&] { return s->box_consumed[me][dst].load(std::memory_order_acquire) >= k; })) return ncclInternalError;
        char* box = c->stage + kCollBytes + (size_t)dst * kBoxBytes;
        size_t off = 0;
        for (size_t i = 0; i < mine.size(); ++i) {
            if (off + mine[i]->bytes > kBoxBytes) return ncclInvalidArgument;
            HIPOK(hipMemcpyAsync(box + off, mine[i]->sbuf, mine[i]->bytes, hipMemcpyDeviceToDevice, st));
            s->box_op_bytes[me][dst][i] = mine[i]->bytes;
            off += mine[i]->bytes;
        }
        s->box_nops[me][dst] = (int)mine.size();
        HIPOK(hipStreamSynchronize(st));
        s->box_posted[me][dst].store(k + 1, std::memory_order_release);
        c->sent[dst] = k + 1;
    }
    // receives, per source rank in the order they were posted
    for (int src = 0; src < c->nranks; ++src) {
        std::vector<const P2P*> mine;
        for (const P2P& op : ops) if (!op.is_send && op.peer == src) mine.push_back(&op);
        if (mine.empty()) continue;
        const long k = c->received[src];
        if (!wait_until([&] { return s->box_posted[src][me].load(std::memory_order_acquire) >= k + 1; })) return ncclInternalError;
        if (s->box_nops[src][me] != (int)mine.size()) return ncclInvalidUsage;
        const char* box = c->peer_stage[src] + kCollBytes + (size_t)me * kBoxBytes;
        size_t off = 0;
        for (size_t i = 0; i < mine.size(); ++i) {
            if (s->box_op_bytes[src][me][i] != mine[i]->bytes) return ncclInvalidUsage;
            HIPOK(hipMemcpyAsync(mine[i]->rbuf, box + off, mine[i]->bytes, hipMemcpyDeviceToDevice, st));
            off += mine[i]->bytes;
        }
        HIPOK(hipStreamSynchronize(st));
        s->box_consumed[src][me].store(k + 1, std::memory_order_release);
        c->received[src] = k + 1;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/prcg_procs_ccl_%d_%u", (int)getpid(), g_ids.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    char name[128];
    memset(name, 0, sizeof name);
    memcpy(name, id.internal, strnlen(id.internal, sizeof id.internal < 120 ? sizeof id.internal : 120));
    if (name[0] != '/') return ncclInvalidArgument;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); return ncclSystemError; }          // (zero-filled by the first one; same size for all)
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm();
    c->shm = static_cast<Shm*>(p);
    c->rank = rank; c->nranks = nranks;
    HIPOK(hipMalloc(reinterpret_cast<void**>(&c->stage), kStageBytes));
    HIPOK(hipMalloc(reinterpret_cast<void**>(&c->scratch), (size_t)kMaxRanks * (1 << 20)));
    HIPOK(hipIpcGetMemHandle(&c->shm->handle[rank], c->stage));
    c->shm->handle_ready[rank].store(1, std::memory_order_release);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { c->peer_stage[r] = c->stage; continue; }
        if (!wait_until([&] { return c->shm->handle_ready[r].load(std::memory_order_acquire) != 0; })) return ncclInternalError;
        void* mapped = nullptr;
        HIPOK(hipIpcOpenMemHandle(&mapped, c->shm->handle[r], hipIpcMemLazyEnablePeerAccess));
        c->peer_stage[r] = static_cast<char*>(mapped);
    }
    c->shm->joined.fetch_add(1);
    if (!wait_until([&] { return c->shm->joined.load() >= nranks; })) return ncclInternalError;
    if (rank == 0) shm_unlink(name);                                 // everybody has it mapped: the name can go
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    for (int r = 0; r < c->nranks; ++r) if (r != c->rank && c->peer_stage[r]) (void)hipIpcCloseMemHandle(c->peer_stage[r]);
    (void)hipFree(c->scratch);
    // (my staging buffer may still be mapped by a rank that closes later: freed when the process ends -- test infrastructure)
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t st) {
    if (dt != ncclDouble) return ncclInvalidArgument;
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = sendcount * sizeof(double);
    long seq;
    ncclResult_t rc = coll_begin(c, st, sendbuff, bytes, &seq);
    if (rc != ncclSuccess) return rc;
    char* out = static_cast<char*>(recvbuff);
    for (int r = 0; r < c->nranks; ++r)
        HIPOK(hipMemcpyAsync(out + (size_t)r * bytes, c->peer_stage[r], bytes, hipMemcpyDeviceToDevice, st));
    return coll_end(c, st, seq);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || (op != ncclSum && op != ncclMax) || count * sizeof(double) > (1u << 20)) return ncclInvalidArgument;
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * sizeof(double);
    long seq;
    ncclResult_t rc = coll_begin(c, st, sendbuff, bytes, &seq);
    if (rc != ncclSuccess) return rc;
    for (int r = 0; r < c->nranks; ++r)
        HIPOK(hipMemcpyAsync(c->scratch + (size_t)r * count, c->peer_stage[r], bytes, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_reduce_ranks, dim3(1), dim3(256), 0, st, c->scratch, static_cast<double*>(recvbuff), count, c->nranks,
                       op == ncclMax ? 1 : 0);
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    return coll_end(c, st, seq);
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
    case ncclUnhandledCudaError: return "HIP call failed (procs_ccl)";
    case ncclInvalidArgument: return "invalid argument (procs_ccl)";
    case ncclInvalidUsage: return "invalid usage (procs_ccl)";
    case ncclInternalError: return "a rank did not arrive within the time limit (procs_ccl)";
    case ncclSystemError: return "shared-memory segment (procs_ccl)";
    default: return "error (procs_ccl)";
    }
}

}  // extern "C"
