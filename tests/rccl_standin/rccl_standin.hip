// rccl_standin.hip -- TEST INFRASTRUCTURE, never shipped or linked into libpfmi.so.
//
// An in-process stand-in for the 13 RCCL entry points csrc/comm_rccl.hip resolves with dlsym (ncclGetUniqueId, ncclCommInitAll,
// ncclCommInitRank, ncclCommDestroy, ncclCommCount, ncclAllGather, ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd,
// ncclGetErrorString, ncclGetVersion), loaded through PFMI_RCCL_LIB.  RCCL itself refuses two ranks on one GPU, so a 1-GPU box can
// only ever form a world of ONE rank -- where every shard offset is 0 and every pool column is owned.  With this library the ranks
// are contexts (streams) of one process, possibly all on the same GPU, and the collectives are done with hipMemcpyAsync / a
// small reduction kernel, ordered ACROSS the ranks' streams with events exactly as a real collective orders them (nothing is
// synchronised with the host): libpfmi's G > 1 data path -- rank offsets, G-way all-gather, owner gather with zero fill, sum
// all-reduce, the shard-size handshake -- executes for real, against the same call sequence and the same NCCL semantics:
//   * ncclCommInitAll          one thread owns all ranks; collectives must be issued for every rank inside ONE group;
//   * ncclCommInitRank         one rank per caller (threads of this process): blocks until all `nranks` callers with the same id
//                              have arrived; a collective completes when every rank has posted it (the last one enqueues the
//                              whole exchange on all streams, the others wait for that to have happened);
//   * element counts / kinds must agree across ranks (ncclInvalidArgument otherwise), in-place all-reduce and in-place all-gather
//     (sendbuff == recvbuff + rank * count) are supported;
//   * ncclSend / ncclRecv (round 6: the owners send their selected columns to rank 0): a send matches the receive the peer posts for
//     it (same count, FIFO per ordered pair); the copy is enqueued on the receiver's stream behind an event of the sender's stream,
//     the sender's stream waits for the copy; a rank whose peer never posts times out like a collective;
//   * a rank that never arrives makes the others time out after PFMI_STANDIN_TIMEOUT_S (default 20 s) with ncclInternalError
//     instead of hanging the test run.
// ncclGetVersion reports 99999 so that a test can tell which library libpfmi actually loaded.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

enum Kind { NONE = 0, ALLGATHER = 1, ALLREDUCE = 2, SEND = 3, RECV = 4 };

struct Op {
    Kind kind = NONE;
    const void *send = nullptr;
    void *recv = nullptr;
    size_t count = 0;
    ncclDataType_t dtype = ncclDouble;
    ncclRedOp_t op = ncclSum;
    hipStream_t stream = nullptr;
    int peer = -1;                         // SEND / RECV
};

struct P2P { Op op; int rank; unsigned long long ticket; };

struct World {
    int n = 0;
    int joined = 0, destroyed = 0;
    bool threaded = false;                 // formed by ncclCommInitRank (one caller per rank)
    std::vector<int> device;
    std::vector<Op> slot;
    int arrived = 0;
    unsigned long long gen = 0;            // completed collectives
    ncclResult_t last = ncclSuccess;       // result of the last collective (shared by all ranks)
    std::vector<hipEvent_t> ready, mid, done;
    std::vector<void *> tmp;               // per-rank reduction scratch
    std::vector<size_t> tmp_cap;
    std::vector<P2P> p2p;                  // posted, not yet matched sends / receives
    unsigned long long p2p_next = 1;
    std::vector<unsigned long long> p2p_matched;   // tickets matched by a peer (the poster is waiting for them)
    std::vector<hipEvent_t> p2p_events;    // freed with the world
    std::mutex mu;
    std::condition_variable cv;
};

struct Pending { Op op; World *w; int rank; };

std::mutex g_mu;
std::map<std::string, World *> g_by_id;
unsigned long long g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_pending;

double timeout_s() {
    const char *e = getenv("PFMI_STANDIN_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0 ? v : 20.0;
}

size_t dsize(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

struct Ptrs { const double *p[64]; };
__global__ void standin_reduce_kernel(Ptrs src, int n, size_t count, int op, double *dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double a = src.p[0][i];
    for (int r = 1; r < n; ++r) {                          // rank order: deterministic
        const double b = src.p[r][i];
        if (op == 0) a = a + b;
        else if (op == 1) a = (b > a || a != a) ? b : a;   // max
        else a = (b < a || a != a) ? b : a;                // min
    }
    dst[i] = a;
}

#define SI_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            fprintf(stderr, "rccl_standin: %s failed: %s\n", #call, hipGetErrorString(e__)); \
            return ncclUnhandledCudaError;                                                 \
        }                                                                                  \
    } while (0)

// all ranks have posted: enqueue the exchange on every rank's stream (called with w->mu held by the last arriver)
ncclResult_t execute(World *w) {
    const int n = w->n;
    const Op &o0 = w->slot[0];
    for (int r = 1; r < n; ++r) {
        const Op &o = w->slot[r];
        if (o.kind != o0.kind || o.count != o0.count || o.dtype != o0.dtype || (o.kind == ALLREDUCE && o.op != o0.op)) {
            fprintf(stderr, "rccl_standin: rank %d posted kind %d count %zu, rank 0 kind %d count %zu\n", r, (int)o.kind, o.count, (int)o0.kind,
                    o0.count);
            return ncclInvalidArgument;
        }
    }
    int dev0 = 0;
    SI_HIP(hipGetDevice(&dev0));
    const size_t bytes = o0.count * dsize(o0.dtype);
    for (int j = 0; j < n; ++j) {
        SI_HIP(hipSetDevice(w->device[j]));
        SI_HIP(hipEventRecord(w->ready[j], w->slot[j].stream));
    }
    if (o0.kind == ALLGATHER) {
        for (int i = 0; i < n; ++i) {
            SI_HIP(hipSetDevice(w->device[i]));
            hipStream_t s = w->slot[i].stream;
            for (int j = 0; j < n; ++j) SI_HIP(hipStreamWaitEvent(s, w->ready[j], 0));
            for (int j = 0; j < n; ++j) {
                char *dst = (char *)w->slot[i].recv + (size_t)j * bytes;
                if (dst == (const char *)w->slot[j].send) continue;        // in-place all-gather: a rank's own slot already holds its shard
                SI_HIP(hipMemcpyAsync(dst, w->slot[j].send, bytes, hipMemcpyDefault, s));
            }
            SI_HIP(hipEventRecord(w->done[i], s));
        }
    } else {
        if (o0.dtype != ncclDouble || n > 64) return ncclInvalidArgument;
        const int opc = (o0.op == ncclSum) ? 0 : (o0.op == ncclMax) ? 1 : (o0.op == ncclMin) ? 2 : -1;
        if (opc < 0) return ncclInvalidArgument;
        Ptrs src;
        for (int j = 0; j < n; ++j) src.p[j] = reinterpret_cast<const double *>(w->slot[j].send);
        for (int i = 0; i < n; ++i) {                       // phase 1: reduce into scratch (in-place buffers are still inputs)
            SI_HIP(hipSetDevice(w->device[i]));
            if (w->tmp_cap[i] < bytes) {
                if (w->tmp[i]) SI_HIP(hipFree(w->tmp[i]));
                w->tmp[i] = nullptr; w->tmp_cap[i] = 0;
                SI_HIP(hipMalloc(&w->tmp[i], bytes));
                w->tmp_cap[i] = bytes;
            }
            hipStream_t s = w->slot[i].stream;
            for (int j = 0; j < n; ++j) SI_HIP(hipStreamWaitEvent(s, w->ready[j], 0));
            hipLaunchKernelGGL(standin_reduce_kernel, dim3((unsigned)((o0.count + 255) / 256)), dim3(256), 0, s, src, n, o0.count, opc,
                               reinterpret_cast<double *>(w->tmp[i]));
            SI_HIP(hipGetLastError());
            SI_HIP(hipEventRecord(w->mid[i], s));
        }
        for (int i = 0; i < n; ++i) {                       // phase 2: everybody has read the inputs -> publish
            SI_HIP(hipSetDevice(w->device[i]));
            hipStream_t s = w->slot[i].stream;
            for (int j = 0; j < n; ++j) SI_HIP(hipStreamWaitEvent(s, w->mid[j], 0));
            SI_HIP(hipMemcpyAsync(w->slot[i].recv, w->tmp[i], bytes, hipMemcpyDeviceToDevice, s));
            SI_HIP(hipEventRecord(w->done[i], s));
        }
    }
    for (int j = 0; j < n; ++j) {                           // a send buffer may be reused only after every rank has consumed it
        SI_HIP(hipSetDevice(w->device[j]));
        for (int i = 0; i < n; ++i)
            if (i != j) SI_HIP(hipStreamWaitEvent(w->slot[j].stream, w->done[i], 0));
    }
    SI_HIP(hipSetDevice(dev0));
    return ncclSuccess;
}

// a matched send / receive pair: the copy on the receiver's stream, ordered behind the sender's stream; the sender's stream waits for it
ncclResult_t execute_p2p(World *w, const Op &snd, int srank, const Op &rcv, int rrank) {   // w->mu held
    if (snd.count != rcv.count || snd.dtype != rcv.dtype) {
        fprintf(stderr, "rccl_standin: send %d -> %d of %zu elements meets a receive of %zu\n", srank, rrank, snd.count, rcv.count);
        return ncclInvalidArgument;
    }
    int dev0 = 0;
    SI_HIP(hipGetDevice(&dev0));
    hipEvent_t e_ready = nullptr, e_done = nullptr;
    SI_HIP(hipSetDevice(w->device[(size_t)srank]));
    SI_HIP(hipEventCreateWithFlags(&e_ready, hipEventDisableTiming));
    SI_HIP(hipEventRecord(e_ready, snd.stream));
    SI_HIP(hipSetDevice(w->device[(size_t)rrank]));
    SI_HIP(hipEventCreateWithFlags(&e_done, hipEventDisableTiming));
    SI_HIP(hipStreamWaitEvent(rcv.stream, e_ready, 0));
    SI_HIP(hipMemcpyAsync(rcv.recv, snd.send, snd.count * dsize(snd.dtype), hipMemcpyDefault, rcv.stream));
    SI_HIP(hipEventRecord(e_done, rcv.stream));
    SI_HIP(hipSetDevice(w->device[(size_t)srank]));
    SI_HIP(hipStreamWaitEvent(snd.stream, e_done, 0));
    SI_HIP(hipSetDevice(dev0));
    w->p2p_events.push_back(e_ready);
    w->p2p_events.push_back(e_done);
    return ncclSuccess;
}

// post the pending ops of this thread; then wait until every collective they belong to has been enqueued
ncclResult_t flush_pending() {
    std::vector<Pending> ops;
    ops.swap(t_pending);
    std::vector<std::pair<World *, unsigned long long>> waits;
    std::vector<std::pair<World *, unsigned long long>> p2p_waits;
    ncclResult_t res = ncclSuccess;
    for (Pending &p : ops) {
        World *w = p.w;
        std::unique_lock<std::mutex> lk(w->mu);
        if (p.op.kind == SEND || p.op.kind == RECV) {
            // the oldest posted counterpart of the ordered pair (FIFO, like NCCL)
            const Kind want = p.op.kind == SEND ? RECV : SEND;
            size_t hit = w->p2p.size();
            for (size_t i = 0; i < w->p2p.size(); ++i)
                if (w->p2p[i].op.kind == want && w->p2p[i].rank == p.op.peer && w->p2p[i].op.peer == p.rank) { hit = i; break; }
            if (hit == w->p2p.size()) {
                const unsigned long long tk = w->p2p_next++;
                w->p2p.push_back(P2P{p.op, p.rank, tk});
                p2p_waits.emplace_back(w, tk);
            } else {
                const P2P other = w->p2p[hit];
                w->p2p.erase(w->p2p.begin() + (long)hit);
                const ncclResult_t r = p.op.kind == SEND ? execute_p2p(w, p.op, p.rank, other.op, other.rank) : execute_p2p(w, other.op, other.rank, p.op, p.rank);
                if (r != ncclSuccess) { res = r; w->last = r; }
                w->p2p_matched.push_back(other.ticket);
                w->cv.notify_all();
            }
            continue;
        }
        if (w->slot[p.rank].kind != NONE) {
            fprintf(stderr, "rccl_standin: rank %d posted two collectives into one exchange\n", p.rank);
            return ncclInvalidUsage;
        }
        w->slot[p.rank] = p.op;
        const unsigned long long my_gen = w->gen;
        if (++w->arrived == w->n) {
            w->last = execute(w);
            for (Op &o : w->slot) o = Op();
            w->arrived = 0;
            ++w->gen;
            if (w->last != ncclSuccess) res = w->last;
            w->cv.notify_all();
        } else {
            waits.emplace_back(w, my_gen);
        }
    }
    for (auto &pw : p2p_waits) {                            // posted first: wait until the peer has matched (and enqueued) the transfer
        World *w = pw.first;
        std::unique_lock<std::mutex> lk(w->mu);
        auto matched = [&] {
            for (size_t i = 0; i < w->p2p_matched.size(); ++i)
                if (w->p2p_matched[i] == pw.second) { w->p2p_matched.erase(w->p2p_matched.begin() + (long)i); return true; }
            return false;
        };
        const bool ok = w->cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), matched);
        if (!ok) {
            fprintf(stderr, "rccl_standin: timed out waiting for the peer of a send / receive\n");
            for (size_t i = 0; i < w->p2p.size(); ++i)
                if (w->p2p[i].ticket == pw.second) { w->p2p.erase(w->p2p.begin() + (long)i); break; }
            return ncclInternalError;
        }
        if (w->last != ncclSuccess) res = w->last;
    }
    for (auto &wg : waits) {
        World *w = wg.first;
        std::unique_lock<std::mutex> lk(w->mu);
        const bool ok = w->cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), [&] { return w->gen > wg.second; });
        if (!ok) {
            fprintf(stderr, "rccl_standin: timed out waiting for %d of %d ranks\n", w->n - w->arrived, w->n);
            for (Op &o : w->slot) o = Op();                 // give up on this exchange
            w->arrived = 0;
            return ncclInternalError;
        }
        if (w->last != ncclSuccess) res = w->last;
    }
    return res;
}

World *new_world(int n) {
    World *w = new World();
    w->n = n;
    w->device.assign((size_t)n, 0);
    w->slot.assign((size_t)n, Op());
    w->ready.assign((size_t)n, nullptr);
    w->mid.assign((size_t)n, nullptr);
    w->done.assign((size_t)n, nullptr);
    w->tmp.assign((size_t)n, nullptr);
    w->tmp_cap.assign((size_t)n, 0);
    return w;
}

ncclResult_t make_events(World *w, int r) {
    SI_HIP(hipSetDevice(w->device[(size_t)r]));
    SI_HIP(hipEventCreateWithFlags(&w->ready[(size_t)r], hipEventDisableTiming));
    SI_HIP(hipEventCreateWithFlags(&w->mid[(size_t)r], hipEventDisableTiming));
    SI_HIP(hipEventCreateWithFlags(&w->done[(size_t)r], hipEventDisableTiming));
    return ncclSuccess;
}

}  // namespace

struct ncclComm {
    World *w;
    int rank;
};

extern "C" {

ncclResult_t ncclGetVersion(int *v) {
    if (!v) return ncclInvalidArgument;
    *v = 99999;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (rccl_standin)";
        case ncclUnhandledCudaError: return "unhandled HIP error (rccl_standin)";
        case ncclInternalError: return "internal error / timeout (rccl_standin)";
        case ncclInvalidArgument: return "invalid argument: ranks disagree on the collective (rccl_standin)";
        case ncclInvalidUsage: return "invalid usage (rccl_standin)";
        default: return "error (rccl_standin)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    std::lock_guard<std::mutex> lk(g_mu);
    snprintf(id->internal, sizeof(id->internal), "pfmi-standin-%llu", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
    if (!comms || n < 1) return ncclInvalidArgument;
    World *w = new_world(n);
    for (int r = 0; r < n; ++r) {
        w->device[(size_t)r] = devs ? devs[r] : r;
        const ncclResult_t rc = make_events(w, r);
        if (rc != ncclSuccess) return rc;
        comms[r] = new ncclComm{w, r};
    }
    w->joined = n;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank) {
    if (!comm || n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
    const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    World *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_id.find(key);
        if (it == g_by_id.end()) {
            w = new_world(n);
            w->threaded = true;
            g_by_id[key] = w;
        } else {
            w = it->second;
        }
    }
    if (w->n != n) return ncclInvalidArgument;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(w->mu);
    w->device[(size_t)rank] = dev;
    const ncclResult_t rc = make_events(w, rank);
    if (rc != ncclSuccess) return rc;
    ++w->joined;
    w->cv.notify_all();
    const bool ok = w->cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), [&] { return w->joined >= n; });
    if (!ok) {
        fprintf(stderr, "rccl_standin: ncclCommInitRank: only %d of %d ranks arrived\n", w->joined, n);
        return ncclInternalError;
    }
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->w->n;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    World *w = comm->w;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        last = ++w->destroyed == w->n;
    }
    delete comm;
    if (last) {
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (auto it = g_by_id.begin(); it != g_by_id.end(); ++it)
                if (it->second == w) { g_by_id.erase(it); break; }
        }
        for (int r = 0; r < w->n; ++r) {
            (void)hipSetDevice(w->device[(size_t)r]);
            if (w->ready[(size_t)r]) (void)hipEventDestroy(w->ready[(size_t)r]);
            if (w->mid[(size_t)r]) (void)hipEventDestroy(w->mid[(size_t)r]);
            if (w->done[(size_t)r]) (void)hipEventDestroy(w->done[(size_t)r]);
            if (w->tmp[(size_t)r]) (void)hipFree(w->tmp[(size_t)r]);
        }
        for (hipEvent_t e : w->p2p_events) (void)hipEventDestroy(e);
        delete w;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return flush_pending();
}

static ncclResult_t post(const Op &op, ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    t_pending.push_back(Pending{op, comm->w, comm->rank});
    if (t_depth > 0) return ncclSuccess;
    return flush_pending();
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    Op o;
    o.kind = ALLGATHER; o.send = sendbuff; o.recv = recvbuff; o.count = sendcount; o.dtype = datatype; o.stream = stream;
    return post(o, comm);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Op o;
    o.kind = ALLREDUCE; o.send = sendbuff; o.recv = recvbuff; o.count = count; o.dtype = datatype; o.op = op; o.stream = stream;
    return post(o, comm);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->w->n || peer == comm->rank) return ncclInvalidArgument;
    Op o;
    o.kind = SEND; o.send = sendbuff; o.count = count; o.dtype = datatype; o.stream = stream; o.peer = peer;
    return post(o, comm);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->w->n || peer == comm->rank) return ncclInvalidArgument;
    Op o;
    o.kind = RECV; o.recv = recvbuff; o.count = count; o.dtype = datatype; o.stream = stream; o.peer = peer;
    return post(o, comm);
}

}  // extern "C"
