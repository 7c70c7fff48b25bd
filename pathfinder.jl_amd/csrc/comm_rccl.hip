// comm_rccl.hip -- the multi-GPU stage of the hot path behind the C ABI (include/pfmi.h, "multi-GPU" section).
//
// Reference: runs are independent until pooling (src/multipath.jl:190-208); `draws_per_component = stack(draws)`,
// `_compute_psis_result` and `_resample` (src/multipath.jl:215-225) see ALL runs.  Paths are sharded over the GPUs in
// contiguous blocks (pool order stays k-major, src/resample.jl:93); the data path has ONE exchange with real content -- an
// all-gather of the fp64 log-ratio shards (K/G * N_r doubles per GPU, 64 KB at config 4) -- after which PSIS and the index
// selection run REPLICATED and deterministically on every GPU (integer CDF => identical indices for any G).  The selected
// columns live on the GPU that owns their path and reach the caller by OWNER-ONLY transfers (round 6; until round 5 every GPU
// filled a zeroed d x ndraws buffer and a sum all-reduce assembled it on every rank: 160 MB per GPU at config 5):
//   pfmi_comm_init_all   every context writes the columns it owns STRAIGHT into the caller's host array (zero-copy stores into
//                        page-locked memory: the array itself when it came from pfmi_host_alloc, a page-locked staging block of the
//                        communicator otherwise) -- no collective, no device-side d x ndraws buffer at all;
//   pfmi_comm_init_rank  the owners SEND their columns to rank 0 (ncclSend / ncclRecv of exactly the owned columns, compact, in
//                        selection order); only rank 0 holds and returns the d x ndraws result.
// The draw pool itself is never exchanged.  The shards may be UNEQUAL (any nruns over any G, src/multipath.jl:131-146): the ranks
// learn each other's shard sizes in the handshake, the all-gather runs on shards padded to the largest one and a compaction kernel
// restores the k-major pool order (src/resample.jl:93) before the replicated PSIS -- the pooled vector, and with it every result, is
// the one a single GPU would have formed.
//
// RCCL is called directly (ncclAllGather / ncclAllReduce / ncclSend / ncclRecv on the contexts' own streams).  Two ways to form the group:
//   pfmi_comm_init_all   one host process drives G contexts, one per GPU (ncclCommInitAll) -- what a single Julia process needs;
//   pfmi_comm_init_rank  one process per GPU (ncclCommInitRank with a 128-byte id made by pfmi_comm_unique_id and shipped by
//                        the host's own launcher), e.g. under torch.distributed.run.
// librccl is opened with dlopen at the first pfmi_comm_* call that needs it, so libpfmi.so itself has no link-time dependency on
// it and a process that already carries an RCCL (PyTorch bundles one under the same SONAME) keeps a single copy.  A world of ONE
// context needs no collective at all and does not touch RCCL (PFMI_COMM_FORCE_RCCL=1 makes it, for tests of the real library on a
// 1-GPU box).
//
// Every stage is ENQUEUED on all local contexts before the first host wait (round 3): one host thread driving G GPUs keeps all of
// them busy, and the fused pfmi_comm_psis_resample synchronises exactly once.
//
// Test hooks (tests/rccl_standin): PFMI_RCCL_LIB=<path> loads that library instead of librccl (an in-process stand-in that
// implements the 11 entry points below among contexts of one process), PFMI_COMM_ALLOW_SHARED_GPU=1 lets several ranks sit on the
// same GPU -- together they execute the G > 1 data path (rank offsets, G-way gather, zero fill, reduce) on a 1-GPU box.
#include "pfmi_common.h"
#include <algorithm>
#include <chrono>

#include <dlfcn.h>
#include <math.h>
#include <mutex>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;                    // ranks may be host threads (one context each): the first ones race to load the library

int32_t rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return PFMI_OK;
    void *h = nullptr;
    const char *over = pf_debug_get("PFMI_RCCL_LIB");
    if (over && over[0]) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        PF_CHECK(h != nullptr, PFMI_ERR_UNSUPPORTED, "PFMI_RCCL_LIB=%s could not be loaded: %s", over, dlerror());
    } else {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        PF_CHECK(h != nullptr, PFMI_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
    }
#define PF_SYM(field, name)                                                                      \
    do {                                                                                         \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                 \
        PF_CHECK(g_rccl.field != nullptr, PFMI_ERR_UNSUPPORTED, "RCCL symbol %s missing", name); \
    } while (0)
    PF_SYM(GetUniqueId, "ncclGetUniqueId");
    PF_SYM(CommInitAll, "ncclCommInitAll");
    PF_SYM(CommInitRank, "ncclCommInitRank");
    PF_SYM(CommDestroy, "ncclCommDestroy");
    PF_SYM(CommCount, "ncclCommCount");
    PF_SYM(AllGather, "ncclAllGather");
    PF_SYM(AllReduce, "ncclAllReduce");
    PF_SYM(Send, "ncclSend");
    PF_SYM(Recv, "ncclRecv");
    PF_SYM(GroupStart, "ncclGroupStart");
    PF_SYM(GroupEnd, "ncclGroupEnd");
    PF_SYM(GetErrorString, "ncclGetErrorString");
    PF_SYM(GetVersion, "ncclGetVersion");
#undef PF_SYM
    g_rccl.handle = h;
    return PFMI_OK;
}

#define PF_NCCL(call)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != ncclSuccess) {                                                                         \
            pf_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
            return PFMI_ERR_COMM;                                                                         \
        }                                                                                                 \
    } while (0)

bool env_on(const char *name) {
    const char *e = pf_debug_get(name);
    return e && e[0] && e[0] != '0';
}

#define PF_MAX_WORLD 64
struct ShardTab { long long off[PF_MAX_WORLD + 1]; };
// pool order k-major (src/resample.jl:93) out of the padded all-gather: rank r's first off[r + 1] - off[r] values of slot r
__global__ void pf_compact_shards_kernel(int world, long long smax, ShardTab tab, const double *__restrict__ padded, double *__restrict__ out) {
    const int r = blockIdx.y;
    const long long n = tab.off[r + 1] - tab.off[r];
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x)
        out[tab.off[r] + j] = padded[(long long)r * smax + j];
}
__global__ void pf_fill_nan_kernel(double *out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __longlong_as_double(0x7FF8000000000000ll);
}

}  // namespace

// one communicator = the local ranks this process drives (all G under init_all, exactly one under init_rank)
struct pfmi_comm {
    int world = 0;                       // ranks in the world
    bool rccl = false;                   // collectives go through RCCL (false: a world of one context, nothing to exchange)
    std::vector<pfmi_ctx *> ctx;         // local contexts
    std::vector<ncclComm_t> comm;        // their communicators
    std::vector<int> rank;               // their world ranks
    std::vector<DevBuf> lr_all;          // [S] gathered log ratios in pool order, one per local ctx
    std::vector<DevBuf> lr_pad;          // [world * smax] padded all-gather block (unequal shards only), one per local ctx
    std::vector<DevBuf> out;             // world of one: [d * ndraws] result; init_rank: rank 0's result / a sender's compact columns
    std::vector<DevBuf> rbuf;            // init_rank, rank 0: the columns received from the other ranks, compact, rank by rank
    std::vector<DevBuf> posb;            // init_rank: selection positions (int64) of the columns gathered / scattered
    std::vector<DevBuf> hs;              // handshake / digest block, one per local ctx
    std::vector<int64_t> shards;         // [world] K_r * N_r of every rank, agreed by the last pooled stage
    std::vector<int64_t> offs;           // [world + 1] first pool column of every rank
    int64_t S = 0, N_r = 0;              // pooled size, draws per run
    bool psis_pending = false;
    int64_t rs_ndraws = 0;               // draws of the enqueued resample stage
    bool pr_pending = false, pr_importance = false;   // pfmi_comm_psis_resample_enqueue is waiting for its pfmi_comm_psis_resample_wait
    int32_t pr_rc = 0;
    char *stage = nullptr;               // init_all, world > 1: page-locked staging block of the result (caller's array is pageable)
    size_t stage_cap = 0;
    bool dead = false;                   // a member context was destroyed: the group is torn down, every call reports PFMI_ERR_STATE
};

// ---- lifetime registry: a communicator borrows its contexts.  Host languages with unordered finalisers (Julia's GC at
// exit) may destroy a context before the communicator that uses it; pfmi_destroy therefore announces the context here first and
// every live communicator that holds it is torn down THEN (streams drained, RCCL handles destroyed, buffers freed) and marked dead,
// so that a later pfmi_comm_destroy only frees the shell and never touches a dangling pfmi_ctx.
namespace {
std::mutex g_comm_mu;
std::vector<pfmi_comm *> g_comms;

void comm_teardown(pfmi_comm *c) {                                      // g_comm_mu held
    if (c->dead) return;
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        (void)hipSetDevice(c->ctx[i]->device);
        (void)hipStreamSynchronize(c->ctx[i]->stream);
    }
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        (void)hipSetDevice(c->ctx[i]->device);
        if (c->comm[i]) (void)g_rccl.CommDestroy(c->comm[i]);
        c->comm[i] = nullptr;
        c->lr_all[i].release();
        c->lr_pad[i].release();
        c->out[i].release();
        c->rbuf[i].release();
        c->posb[i].release();
        c->hs[i].release();
    }
    if (c->stage) (void)hipHostFree(c->stage);
    c->stage = nullptr; c->stage_cap = 0;
    c->ctx.clear();
    c->dead = true;
}
void comm_register(pfmi_comm *c) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    g_comms.push_back(c);
}
}  // namespace

void pf_comm_ctx_dying(pfmi_ctx *x) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (pfmi_comm *c : g_comms) {
        bool member = false;
        for (pfmi_ctx *m : c->ctx) member = member || m == x;
        if (member) comm_teardown(c);
    }
}

#define PF_COMM(c)                                                                                                          \
    do {                                                                                                                    \
        PF_CHECK((c) != nullptr, PFMI_ERR_ARG, "null pfmi_comm");                                                           \
        PF_CHECK(!(c)->dead, PFMI_ERR_STATE, "pfmi_comm: a member context was destroyed; the communicator is closed");      \
    } while (0)

namespace {

// stage timers of the collectives (pfmi_profile): an event pair around the group call on every local context's stream; the handshake of
// the process-per-GPU mode synchronises the host, so its figure is host wall-clock ("comm_handshake_host", rank's first context)
void prof_begin_all(pfmi_comm *c) {
    for (pfmi_ctx *x : c->ctx) if (x->profile) { (void)hipSetDevice(x->device); pf_kernel_begin(x); }
}
void prof_end_all(pfmi_comm *c, const char *name) {
    for (pfmi_ctx *x : c->ctx) if (x->profile) { (void)hipSetDevice(x->device); pf_kernel_end(x, name); }
}

// the staged downloads of the comm stage are the LIBRARY's (stack / member destinations): never kept across entry points, whatever the
// caller's pfmi_defer_downloads mode is while it calls
struct DeferOff {
    std::vector<pfmi_ctx *> cx; std::vector<bool> keep;
    explicit DeferOff(pfmi_comm *c) : cx(c->ctx) { for (pfmi_ctx *x : cx) { keep.push_back(x->defer); x->defer = false; } }
    ~DeferOff() { for (size_t i = 0; i < cx.size(); ++i) cx[i]->defer = keep[i]; }
};
// wait for EVERY local context, then report the first failure: no context is left unsynchronised with staged downloads that point at
// the caller's frame
int32_t sync_all(pfmi_comm *c) {
    int32_t rc = PFMI_OK;
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        int32_t r = hipSetDevice(c->ctx[i]->device) == hipSuccess ? pf_stream_sync(c->ctx[i]) : PFMI_ERR_HIP;
        if (r != PFMI_OK && rc == PFMI_OK) rc = r;
    }
    return rc;
}

int32_t group_all_reduce(pfmi_comm *c, std::vector<DevBuf> &buf, size_t count, ncclRedOp_t op, const char *stage = nullptr) {
    const size_t nl = c->ctx.size();
    if (stage) prof_begin_all(c);
    PF_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        ncclResult_t r = g_rccl.AllReduce(buf[i].p, buf[i].p, count, ncclDouble, op, c->comm[i], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclAllReduce failed on rank %d: %s", c->rank[i], g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
    }
    PF_NCCL(g_rccl.GroupEnd());
    if (stage) prof_end_all(c, stage);
    return PFMI_OK;
}

// ONE all-gather of the log-ratio shards.  Equal shards: straight into the pooled vector.  Unequal shards: every rank places its shard
// in ITS slot of a block padded to the largest shard, the all-gather runs in place on that block (ncclAllGather needs equal counts; the
// pad regions are never read), and a compaction kernel restores the pool order -- k-major over ALL runs, src/resample.jl:93.
int32_t group_all_gather(pfmi_comm *c) {
    const size_t nl = c->ctx.size();
    int64_t smax = 0;
    bool equal = true;
    for (int r = 0; r < c->world; ++r) { smax = std::max(smax, c->shards[(size_t)r]); equal = equal && c->shards[(size_t)r] == c->shards[0]; }
    prof_begin_all(c);
    if (!equal)
        for (size_t i = 0; i < nl; ++i) {
            pfmi_ctx *x = c->ctx[i];
            PF_HIP(hipSetDevice(x->device));
            PF_HIP(hipMemcpyAsync(c->lr_pad[i].as<double>() + (size_t)c->rank[i] * (size_t)smax, x->pool_lr.p,
                                  sizeof(double) * (size_t)c->shards[(size_t)c->rank[i]], hipMemcpyDeviceToDevice, x->stream));
        }
    PF_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        const void *src = equal ? x->pool_lr.p : static_cast<const void *>(c->lr_pad[i].as<double>() + (size_t)c->rank[i] * (size_t)smax);
        void *dst = equal ? c->lr_all[i].p : c->lr_pad[i].p;
        ncclResult_t r = g_rccl.AllGather(src, dst, (size_t)smax, ncclDouble, c->comm[i], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclAllGather failed on rank %d: %s", c->rank[i], g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
    }
    PF_NCCL(g_rccl.GroupEnd());
    if (!equal) {
        ShardTab tab;
        for (int r = 0; r <= c->world; ++r) tab.off[r] = c->offs[(size_t)r];
        for (size_t i = 0; i < nl; ++i) {
            pfmi_ctx *x = c->ctx[i];
            PF_HIP(hipSetDevice(x->device));
            const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (smax + 255) / 256));
            hipLaunchKernelGGL(pf_compact_shards_kernel, dim3(gx, (unsigned)c->world), dim3(256), 0, x->stream, c->world, (long long)smax, tab,
                               c->lr_pad[i].as<double>(), c->lr_all[i].as<double>());
            PF_HIP(hipGetLastError());
        }
    }
    prof_end_all(c, "comm_allgather");
    return PFMI_OK;
}

// Agree on every rank's shard size, on N_r and d, and on everybody's local status BEFORE a collective whose element counts depend on
// them.  Under pfmi_comm_init_all this process sees every rank and the check is local; under pfmi_comm_init_rank the ranks exchange one
// (6 + world)-double max all-reduce: {N_r, -N_r, error, d, -d, 0, shard_0 .. shard_{world-1}} with a rank's own shard in its slot and 0
// elsewhere.  A rank without a pool, with another N_r / d, or whose device buffers for the stage cannot be allocated makes EVERY rank
// return the same error instead of leaving the others blocked in (or corrupting) the collective.
int32_t agree_on_shards(pfmi_comm *c, bool want_lr_all) {
    const size_t nl = c->ctx.size();
    const int W = c->world;
    int local_err = 0;
    char why[256] = "";
    std::vector<double> h((size_t)(6 + W), 0.0), r((size_t)(6 + W), 0.0);
    int64_t nr = -1, dd = -1;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        if (!x->pooled) {
            local_err = 1;
            snprintf(why, sizeof(why), "rank %d has no pool (call pfmi_pool_build)", c->rank[i]);
            continue;
        }
        if ((nr >= 0 && x->N_r != nr) || (dd >= 0 && x->d != dd)) {
            local_err = 1;
            snprintf(why, sizeof(why), "draws per run / dimension differ between ranks (%lld x %d on rank %d vs %lld x %lld)", (long long)x->N_r, x->d,
                     c->rank[i], (long long)nr, (long long)dd);
        }
        nr = x->N_r; dd = x->d;
        h[(size_t)(6 + c->rank[i])] = (double)((int64_t)x->K * x->N_r);
    }
    h[0] = (double)nr; h[1] = -(double)nr; h[3] = (double)dd; h[4] = -(double)dd;
    if ((int)nl < W) {                                          // one process per GPU: the other ranks are somewhere else
        h[2] = (double)local_err;
        const auto hs_t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < nl; ++i) {
            PF_HIP(hipSetDevice(c->ctx[i]->device));
            PF_TRY(c->hs[i].ensure(sizeof(double) * (size_t)(6 + PF_MAX_WORLD)));
            PF_TRY(pf_upload(c->ctx[i], c->hs[i].p, h.data(), sizeof(double) * h.size()));
        }
        PF_TRY(group_all_reduce(c, c->hs, h.size(), ncclMax));
        PF_HIP(hipSetDevice(c->ctx[0]->device));
        PF_TRY(pf_download(c->ctx[0], r.data(), c->hs[0].p, sizeof(double) * r.size()));
        PF_TRY(pf_stream_sync(c->ctx[0]));
        if (c->ctx[0]->profile) {
            KernelStat &ks = c->ctx[0]->kstats["comm_handshake_host"];
            ks.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs_t0).count();
            ks.launches += 1;
        }
        PF_CHECK(r[2] == 0.0, PFMI_ERR_STATE, "comm: a rank of the group cannot take part in the pooled stage%s%s", local_err ? ": " : "",
                 local_err ? why : " (see that rank's error)");
        PF_CHECK(r[0] == -r[1] && r[3] == -r[4], PFMI_ERR_ARG, "comm: draws per run / dimension differ across ranks (N_r %g .. %g, d %g .. %g)", -r[1], r[0],
                 -r[4], r[3]);
    } else {
        PF_CHECK(!local_err, strstr(why, "differ") ? PFMI_ERR_ARG : PFMI_ERR_STATE, "comm: %s", why);
        r = h;
    }
    c->shards.assign((size_t)W, 0);
    c->offs.assign((size_t)W + 1, 0);
    int64_t smax = 0;
    bool equal = true;
    for (int q = 0; q < W; ++q) {
        c->shards[(size_t)q] = (int64_t)r[(size_t)(6 + q)];
        c->offs[(size_t)q + 1] = c->offs[(size_t)q] + c->shards[(size_t)q];
        smax = std::max(smax, c->shards[(size_t)q]);
        equal = equal && c->shards[(size_t)q] == c->shards[0];
    }
    c->S = c->offs[(size_t)W];
    c->N_r = (int64_t)r[0];
    // what the all-gather needs; a failure here is after the handshake, so it is agreed on through the next one (the resample stage's
    // digest exchange carries the flag) -- allocate, and let a local failure surface as this rank's error
    for (size_t i = 0; i < nl && want_lr_all && c->rccl; ++i) {
        PF_HIP(hipSetDevice(c->ctx[i]->device));
        PF_TRY(c->lr_all[i].ensure(sizeof(double) * (size_t)c->S));
        if (!equal) PF_TRY(c->lr_pad[i].ensure(sizeof(double) * (size_t)smax * (size_t)W));
    }
    return PFMI_OK;
}

// all-gather + replicated PSIS, enqueued on every local context
int32_t enqueue_pool_psis(pfmi_comm *c) {
    const size_t nl = c->ctx.size();
    PF_TRY(agree_on_shards(c, true));
    if (c->rccl) PF_TRY(group_all_gather(c));
    for (size_t i = 0; i < nl; ++i) {                       // replicated PSIS: same code, same input, fixed reduction order
        pfmi_ctx *x = c->ctx[i];
        PF_HIP(hipSetDevice(x->device));
        PF_TRY(pf_launch_psis(x, c->rccl ? c->lr_all[i].as<double>() : x->pool_lr.as<double>(), c->S));
    }
    c->psis_pending = true;
    return PFMI_OK;
}

// the pooled PSIS' scalars (k-hat, tail length, ...) of every local replica: queued for download / checked after the wait
int32_t queue_pool_psis(pfmi_comm *c, std::vector<double> &out) {
    const size_t nl = c->ctx.size();
    out.assign(4 * nl, 0.0);
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        PF_HIP(hipSetDevice(x->device));
        PF_TRY(pf_download(x, &out[4 * i], x->psis_out.p, 4 * sizeof(double)));
    }
    return PFMI_OK;
}
int32_t check_pool_psis(pfmi_comm *c, const std::vector<double> &out, double *pareto_k, int64_t *tail_len) {
    const size_t nl = c->ctx.size();
    c->psis_pending = false;
    const double k0 = out[0];
    const int64_t m0 = (int64_t)out[1];
    for (size_t i = 1; i < nl; ++i) {
        const double k = out[4 * i];
        PF_CHECK((int64_t)out[4 * i + 1] == m0 && (k == k0 || (k != k && k0 != k0)), PFMI_ERR_NUMERIC, "comm_pool_psis: replicas disagree (rank %d)",
                 c->rank[i]);
    }
    if (pareto_k) *pareto_k = k0;
    if (tail_len) *tail_len = m0;
    return PFMI_OK;
}
int32_t finish_pool_psis(pfmi_comm *c, double *pareto_k, int64_t *tail_len) {
    DeferOff off(c);
    std::vector<double> out;
    const int32_t rq = queue_pool_psis(c, out);
    const int32_t rs = sync_all(c);
    if (rq != PFMI_OK) return rq;
    if (rs != PFMI_OK) return rs;
    return check_pool_psis(c, out, pareto_k, tail_len);
}

// replicated index selection, enqueued on every local context (identical indices on every rank by construction: integer CDF, counter-based
// uniforms).  World of one: the gather into the context's result buffer follows at once.  Several ranks: the owner-only transfers are
// issued by finish_resample, which knows the destination (and, under pfmi_comm_init_rank, needs the indices on the host first).
// A local failure travels as rc (init_all: this process sees it) or in the digest block (init_rank: every rank hears about it).
int32_t enqueue_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms) {
    const size_t nl = c->ctx.size();
    const int64_t S = c->S;
    const int d = c->ctx[0]->d;
    c->rs_ndraws = ndraws;
    int32_t rc_local = PFMI_OK;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        int32_t rc = PFMI_OK;
        if (hipSetDevice(x->device) != hipSuccess) { pf_set_error("comm_resample: hipSetDevice(%d) failed", x->device); rc = PFMI_ERR_HIP; }
        if (rc == PFMI_OK && x->d != d) { pf_set_error("comm_resample: dimension differs between ranks"); rc = PFMI_ERR_ARG; }
        if (rc == PFMI_OK && importance && x->S_w != S) {
            pf_set_error("comm_resample: importance weights for S=%lld not available on rank %d (run the pooled PSIS first)", (long long)S, c->rank[i]);
            rc = PFMI_ERR_STATE;
        }
        const double *d_uni = nullptr;
        if (rc == PFMI_OK && uniforms) {
            rc = x->tailbuf.ensure(sizeof(double) * ndraws);
            if (rc == PFMI_OK) rc = pf_upload(x, x->tailbuf.p, uniforms, sizeof(double) * ndraws);
            d_uni = x->tailbuf.as<double>();
        }
        if (rc == PFMI_OK) rc = pf_enqueue_resample(x, S, ndraws, importance, replace, seed, d_uni);
        if (rc == PFMI_OK && c->world == 1) {
            rc = c->out[i].ensure(sizeof(double) * (size_t)d * (size_t)ndraws);
            if (rc == PFMI_OK) rc = pf_launch_gather(x, ndraws, x->idx.as<int64_t>(), 0, c->out[i].as<double>());
        }
        if (c->world > 1 && (int)nl < c->world) {
            // process per GPU: {local error, digest of the indices, -digest} through one 4-double max all-reduce -- every rank learns whether
            // any rank failed and whether the replicas agree BEFORE the point-to-point transfers are sized from the indices
            int32_t r2 = c->hs[i].ensure(sizeof(double) * (size_t)(6 + PF_MAX_WORLD));
            if (r2 == PFMI_OK && rc == PFMI_OK) r2 = pf_launch_idx_digest(x, ndraws, x->idx.as<int64_t>(), c->hs[i].as<double>(), 0.0);
            else if (r2 == PFMI_OK) {
                const double bad[4] = {1.0, 0.0, 0.0, 0.0};
                r2 = pf_upload(x, c->hs[i].p, bad, sizeof(bad));
            }
            if (rc == PFMI_OK) rc = r2;
        }
        if (rc != PFMI_OK) rc_local = rc;
    }
    if (c->world > 1 && (int)nl < c->world) {
        bool have = true;
        for (size_t i = 0; i < nl; ++i) have = have && c->hs[i].cap >= 4 * sizeof(double);
        if (have) PF_TRY(group_all_reduce(c, c->hs, 4, ncclMax));
    }
    return rc_local;
}

// where a kernel may write the result directly: the caller's array if it is page-locked, device-visible memory (pfmi_host_alloc), else null
double *device_view_of_host(void *host_ptr) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host_ptr) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, host_ptr, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<double *>(dp);
}

int32_t finish_resample(pfmi_comm *c, int64_t *idx, double *draws) {
    DeferOff off(c);
    const size_t nl = c->ctx.size();
    const int64_t ndraws = c->rs_ndraws;
    const int d = c->ctx[0]->d;
    const size_t n = (size_t)d * (size_t)ndraws;
    std::vector<int64_t> h_idx((size_t)ndraws * nl);
    std::vector<int> h_err(nl, 0);
    int32_t rc = PFMI_OK;
    auto keep = [&](int32_t r) { if (r != PFMI_OK && rc == PFMI_OK) rc = r; };
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        if (hipSetDevice(x->device) != hipSuccess) { keep(PFMI_ERR_HIP); continue; }
        if (x->idx.cap >= sizeof(int64_t) * (size_t)ndraws) keep(pf_download(x, &h_idx[(size_t)ndraws * i], x->idx.p, sizeof(int64_t) * ndraws));
        if (x->rs_err.p) keep(pf_download(x, &h_err[i], x->rs_err.p, sizeof(int)));
    }
    bool staged = false;
    if (c->world == 1) {
        // the small results are staged (pinned, asynchronous); the draws -- megabytes into the caller's pageable array, a blocking
        // call -- are queued LAST, so this entry point costs ONE host round trip
        pfmi_ctx *x = c->ctx[0];
        if (hipSetDevice(x->device) == hipSuccess) {
            if (c->out[0].cap < sizeof(double) * n) { pf_set_error("comm_resample: no result buffer"); keep(PFMI_ERR_STATE); }
            else if (draws) keep(pf_download(x, draws, c->out[0].p, sizeof(double) * n));
        } else keep(PFMI_ERR_HIP);
        keep(sync_all(c));
    } else if ((int)nl == c->world) {
        // ---- ONE process drives every rank: each context writes the columns it owns straight into the host result (zero-copy stores; the
        //      caller's array when it is page-locked, the communicator's page-locked staging block otherwise).  No collective.
        if (draws && rc == PFMI_OK) {
            double *dst = device_view_of_host(draws);
            if (!dst) {
                if (c->stage_cap < sizeof(double) * n) {
                    if (c->stage) (void)hipHostFree(c->stage);
                    c->stage = nullptr; c->stage_cap = 0;
                    void *pp = nullptr;
                    if (hipHostMalloc(&pp, sizeof(double) * n, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                        pf_set_error("comm_resample: cannot allocate %zu bytes of page-locked staging", sizeof(double) * n);
                        keep(PFMI_ERR_HIP);
                    } else { c->stage = static_cast<char *>(pp); c->stage_cap = sizeof(double) * n; }
                }
                if (c->stage) { dst = device_view_of_host(c->stage); staged = true; }
            }
            if (dst)
                for (size_t i = 0; i < nl; ++i) {
                    pfmi_ctx *x = c->ctx[i];
                    if (hipSetDevice(x->device) != hipSuccess) { keep(PFMI_ERR_HIP); continue; }
                    keep(pf_launch_gather(x, ndraws, x->idx.as<int64_t>(), c->offs[(size_t)c->rank[i]], dst, true));
                }
            else if (rc == PFMI_OK) { pf_set_error("comm_resample: no device-visible destination for the result"); keep(PFMI_ERR_HIP); }
        }
        keep(sync_all(c));
        if (staged && rc == PFMI_OK) memcpy(draws, c->stage, sizeof(double) * n);
    } else {
        // ---- one process per GPU: the indices first (the transfers are sized from them), then the owners SEND their columns to rank 0
        pfmi_ctx *x = c->ctx[0];
        const int me = c->rank[0];
        double blk[4] = {1.0, 0.0, 0.0, 0.0};
        if (hipSetDevice(x->device) != hipSuccess) keep(PFMI_ERR_HIP);
        if (c->hs[0].cap >= sizeof(blk)) keep(pf_download(x, blk, c->hs[0].p, sizeof(blk)));
        keep(sync_all(c));                                                   // host round trip 1 of 2: indices + the group's status
        if (rc == PFMI_OK && blk[0] != 0.0) { pf_set_error("comm_resample: a rank of the group failed its local stage"); rc = PFMI_ERR_COMM; }
        if (rc == PFMI_OK && blk[1] != -blk[2]) { pf_set_error("comm_resample: replicated index selection disagrees across ranks"); rc = PFMI_ERR_NUMERIC; }
        if (rc == PFMI_OK && h_err[0] != 0) { pf_set_error("resample: weights are all zero / not enough positive weights (rank %d)", me); rc = PFMI_ERR_NUMERIC; }
        if (rc != PFMI_OK) return rc;        // (agreed on by every rank: nobody enters the transfers)
        // owners of the selected columns, rank by rank, in selection order
        std::vector<std::vector<int64_t>> pos((size_t)c->world);
        for (int64_t t = 0; t < ndraws; ++t) {
            const int64_t g = h_idx[(size_t)t];
            int r = (int)(std::upper_bound(c->offs.begin(), c->offs.end(), g) - c->offs.begin()) - 1;
            if (r < 0 || r >= c->world) { pf_set_error("comm_resample: index %lld outside the pool", (long long)g); return PFMI_ERR_NUMERIC; }
            pos[(size_t)r].push_back(t);
        }
        const int64_t n_me = (int64_t)pos[(size_t)me].size();
        // every allocation before the group call; a failure here is fatal for the group (documented), like any PFMI_ERR_* of a comm call
        std::vector<int64_t> hp;
        if (me == 0) { for (int r = 1; r < c->world; ++r) hp.insert(hp.end(), pos[(size_t)r].begin(), pos[(size_t)r].end()); }
        else hp = pos[(size_t)me];
        const int64_t n_other = ndraws - (int64_t)pos[0].size();
        PF_TRY(c->posb[0].ensure(sizeof(int64_t) * std::max<size_t>(1, hp.size())));
        if (me == 0) {
            PF_TRY(c->out[0].ensure(sizeof(double) * n));
            PF_TRY(c->rbuf[0].ensure(sizeof(double) * (size_t)d * (size_t)std::max<int64_t>(1, n_other)));
        } else PF_TRY(c->out[0].ensure(sizeof(double) * (size_t)d * (size_t)std::max<int64_t>(1, n_me)));
        if (!hp.empty()) PF_TRY(pf_upload(x, c->posb[0].p, hp.data(), sizeof(int64_t) * hp.size()));
        if (me == 0) PF_TRY(pf_launch_gather(x, ndraws, x->idx.as<int64_t>(), c->offs[0], c->out[0].as<double>(), true));   // its own columns in place
        else PF_TRY(pf_launch_gather_pos(x, n_me, x->idx.as<int64_t>(), c->posb[0].as<int64_t>(), c->offs[(size_t)me], c->out[0].as<double>()));
        prof_begin_all(c);
        PF_NCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        if (me == 0) {
            size_t at = 0;
            for (int q = 1; q < c->world && r == ncclSuccess; ++q) {
                const size_t cnt = pos[(size_t)q].size() * (size_t)d;
                if (cnt) r = g_rccl.Recv(c->rbuf[0].as<double>() + at, cnt, ncclDouble, q, c->comm[0], x->stream);
                at += cnt;
            }
        } else if (n_me > 0) r = g_rccl.Send(c->out[0].p, (size_t)n_me * (size_t)d, ncclDouble, 0, c->comm[0], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclSend / ncclRecv failed on rank %d: %s", me, g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
        PF_NCCL(g_rccl.GroupEnd());
        prof_end_all(c, "comm_sendrecv");
        if (me == 0) {
            PF_TRY(pf_launch_scatter_cols(x, n_other, c->posb[0].as<int64_t>(), c->rbuf[0].as<double>(), c->out[0].as<double>()));
            if (draws) PF_TRY(pf_download(x, draws, c->out[0].p, sizeof(double) * n));
        }
        keep(sync_all(c));                                                   // host round trip 2 of 2
        if (me != 0 && draws && rc == PFMI_OK)                               // the result lives on rank 0: nothing stale may look like draws
            for (size_t q = 0; q < n; ++q) draws[q] = NAN;
    }
    if (rc != PFMI_OK) return rc;
    for (size_t i = 0; i < nl; ++i)
        PF_CHECK(h_err[i] == 0, PFMI_ERR_NUMERIC, "resample: weights are all zero / not enough positive weights (rank %d)", c->rank[i]);
    for (size_t i = 1; i < nl; ++i)
        PF_CHECK(memcmp(&h_idx[(size_t)ndraws * i], h_idx.data(), sizeof(int64_t) * ndraws) == 0, PFMI_ERR_NUMERIC,
                 "comm_resample: replicated index selection disagrees on rank %d", c->rank[i]);
    if (idx) memcpy(idx, h_idx.data(), sizeof(int64_t) * (size_t)ndraws);
    return PFMI_OK;
}

}  // namespace

extern "C" {

int32_t pfmi_comm_unique_id(uint8_t *id128) {
    PF_CHECK(id128 != nullptr, PFMI_ERR_ARG, "comm_unique_id: null buffer");
    PF_TRY(rccl_load());
    ncclUniqueId id;
    PF_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return PFMI_OK;
}

int32_t pfmi_comm_init_all(int32_t G, pfmi_ctx *const *ctxs, pfmi_comm **out) {
    PF_CHECK(out != nullptr, PFMI_ERR_ARG, "comm_init_all: null out");
    *out = nullptr;
    PF_CHECK(G >= 1 && G <= PF_MAX_WORLD && ctxs != nullptr, PFMI_ERR_ARG, "comm_init_all: bad arguments (1 <= ngpus <= %d)", PF_MAX_WORLD);
    const bool use_rccl = G > 1 || env_on("PFMI_COMM_FORCE_RCCL");
    if (use_rccl) PF_TRY(rccl_load());
    std::vector<int> devs((size_t)G);
    const bool shared_ok = env_on("PFMI_COMM_ALLOW_SHARED_GPU");       // test hook, see the header comment
    for (int r = 0; r < G; ++r) {
        PF_CHECK(ctxs[r] != nullptr, PFMI_ERR_ARG, "comm_init_all: null context %d", r);
        devs[(size_t)r] = ctxs[r]->device;
        for (int q = 0; q < r; ++q) {
            PF_CHECK(ctxs[q] != ctxs[r], PFMI_ERR_ARG, "comm_init_all: context %d passed twice", r);
            PF_CHECK(shared_ok || devs[(size_t)q] != devs[(size_t)r], PFMI_ERR_ARG,
                     "comm_init_all: contexts %d and %d share GPU %d (one rank per GPU)", q, r, devs[(size_t)r]);
        }
    }
    pfmi_comm *c = new pfmi_comm();
    c->world = G;
    c->rccl = use_rccl;
    c->ctx.assign(ctxs, ctxs + G);
    c->comm.assign((size_t)G, nullptr);
    c->rank.resize((size_t)G);
    for (int r = 0; r < G; ++r) c->rank[(size_t)r] = r;
    c->lr_all.resize((size_t)G);
    c->lr_pad.resize((size_t)G);
    c->out.resize((size_t)G);
    c->rbuf.resize((size_t)G);
    c->posb.resize((size_t)G);
    c->hs.resize((size_t)G);
    if (use_rccl) {
        ncclResult_t rc = g_rccl.CommInitAll(c->comm.data(), G, devs.data());
        if (rc != ncclSuccess) {
            pf_set_error("ncclCommInitAll(%d GPUs) failed: %s", G, g_rccl.GetErrorString(rc));
            delete c;
            return PFMI_ERR_COMM;
        }
    }
    comm_register(c);
    *out = c;
    return PFMI_OK;
}

int32_t pfmi_comm_init_rank(pfmi_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id128, pfmi_comm **out) {
    PF_CHECK(out != nullptr, PFMI_ERR_ARG, "comm_init_rank: null out");
    *out = nullptr;
    PF_CHECK(ctx != nullptr && id128 != nullptr && world >= 1 && world <= PF_MAX_WORLD && rank >= 0 && rank < world, PFMI_ERR_ARG, "comm_init_rank: bad arguments (1 <= world <= %d)", PF_MAX_WORLD);
    PF_TRY(rccl_load());
    PF_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    pfmi_comm *c = new pfmi_comm();
    c->world = world;
    c->rccl = true;
    c->ctx.assign(1, ctx);
    c->comm.assign(1, nullptr);
    c->rank.assign(1, rank);
    c->lr_all.resize(1);
    c->lr_pad.resize(1);
    c->out.resize(1);
    c->rbuf.resize(1);
    c->posb.resize(1);
    c->hs.resize(1);
    ncclResult_t rc = g_rccl.CommInitRank(&c->comm[0], world, id, rank);
    if (rc != ncclSuccess) {
        pf_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(rc));
        delete c;
        return PFMI_ERR_COMM;
    }
    comm_register(c);
    *out = c;
    return PFMI_OK;
}

int32_t pfmi_comm_destroy(pfmi_comm *c) {
    if (!c) return PFMI_OK;
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        for (size_t i = 0; i < g_comms.size(); ++i)
            if (g_comms[i] == c) { g_comms.erase(g_comms.begin() + (long)i); break; }
        comm_teardown(c);                                               // no-op when a dying member context already closed the group
    }
    delete c;
    return PFMI_OK;
}

int32_t pfmi_comm_info(pfmi_comm *c, int32_t *world, int32_t *nlocal, int32_t *rccl_version) {
    PF_COMM(c);
    if (world) {
        int n = c->world;
        if (c->rccl) PF_NCCL(g_rccl.CommCount(c->comm[0], &n));       // what RCCL itself says, not what the caller claimed
        *world = n;
    }
    if (nlocal) *nlocal = (int32_t)c->ctx.size();
    if (rccl_version) {
        int v = 0;                                                      // 0: a world of one context, RCCL not involved
        if (c->rccl) PF_NCCL(g_rccl.GetVersion(&v));
        *rccl_version = v;
    }
    return PFMI_OK;
}

// _compute_psis_result over the pooled runs (src/multipath.jl:221): all-gather the log-ratio shards, then PSIS on every GPU.
int32_t pfmi_comm_pool_psis(pfmi_comm *c, double *pareto_k, int64_t *tail_len) {
    if (c) for (pfmi_ctx *x : c->ctx) { pf_download_forget(x); (void)hipSetDevice(x->device); (void)pf_dl_flush(x); }
    PF_COMM(c);
    PF_TRY(enqueue_pool_psis(c));
    return finish_pool_psis(c, pareto_k, tail_len);
}

// _resample over the pooled runs (src/multipath.jl:225, src/resample.jl:58-72): replicated index selection, owner-only transfers of the
// selected columns.  idx[ndraws] (0-based, global pool columns) and draws[d * ndraws] (column-major) may be NULL.
int32_t pfmi_comm_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms,
                           int64_t *idx, double *draws) {
    if (c) for (pfmi_ctx *x : c->ctx) { pf_download_forget(x); (void)hipSetDevice(x->device); (void)pf_dl_flush(x); }
    PF_COMM(c);
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_resample: ndraws must be positive");
    PF_CHECK(c->S > 0, PFMI_ERR_STATE, "comm_resample: call pfmi_comm_pool_psis first");
    const int32_t rc = enqueue_resample(c, ndraws, importance, replace, seed, uniforms);
    if (rc != PFMI_OK && (int)c->ctx.size() == c->world) { (void)sync_all(c); return rc; }     // this process sees every rank: nothing to agree on
    const int32_t rf = finish_resample(c, idx, draws);
    return rc != PFMI_OK ? rc : rf;
}

// both stages, one synchronisation (src/multipath.jl:221-225)
int32_t pfmi_comm_psis_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms,
                                double *pareto_k, int64_t *tail_len, int64_t *idx, double *draws) {
    PF_TRY(pfmi_comm_psis_resample_enqueue(c, ndraws, importance, replace, seed, uniforms));
    return pfmi_comm_psis_resample_wait(c, pareto_k, tail_len, idx, draws);
}

// the same in two halves: everything enqueued / the one wait.  Between the two the caller may queue downloads on the member contexts
// (pfmi_defer_downloads): the wait delivers them in the same host round trip.
int32_t pfmi_comm_psis_resample_enqueue(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms) {
    if (c) for (pfmi_ctx *x : c->ctx) { pf_download_forget(x); (void)hipSetDevice(x->device); (void)pf_dl_flush(x); }
    PF_COMM(c);
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_psis_resample: ndraws must be positive");
    c->pr_pending = false;
    if (importance) PF_TRY(enqueue_pool_psis(c));
    else PF_TRY(agree_on_shards(c, false));
    c->pr_importance = importance != 0;
    c->pr_rc = enqueue_resample(c, ndraws, importance, replace, seed, uniforms);
    c->pr_pending = true;
    return PFMI_OK;
}

int32_t pfmi_comm_psis_resample_wait(pfmi_comm *c, double *pareto_k, int64_t *tail_len, int64_t *idx, double *draws) {
    PF_COMM(c);
    PF_CHECK(c->pr_pending, PFMI_ERR_STATE, "comm_psis_resample_wait: nothing enqueued");
    c->pr_pending = false;
    double k = NAN;
    int64_t m = 0;
    int32_t rp = PFMI_OK, rf = PFMI_OK;
    std::vector<double> pout;
    {
        DeferOff off(c);                                                    // the library's own staged downloads are never `kept`
        if (c->pr_importance) rp = queue_pool_psis(c, pout);
    }
    if (c->pr_rc != PFMI_OK && (int)c->ctx.size() == c->world) rf = sync_all(c);   // the local stage failed and this process is the whole group: just drain
    else rf = finish_resample(c, idx, draws);                              // ONE wait for both stages (and whatever the contexts queued)
    c->psis_pending = false;
    if (c->pr_importance && c->pr_rc == PFMI_OK && rp == PFMI_OK && rf == PFMI_OK) rp = check_pool_psis(c, pout, &k, &m);
    if (pareto_k) *pareto_k = k;
    if (tail_len) *tail_len = m;
    return c->pr_rc != PFMI_OK ? c->pr_rc : (rp != PFMI_OK ? rp : rf);
}

}  // extern "C"
