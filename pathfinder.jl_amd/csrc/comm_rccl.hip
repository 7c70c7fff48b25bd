// comm_rccl.hip -- the multi-GPU stage of the hot path behind the C ABI (include/pfmi.h, "multi-GPU" section).
//
// Reference: runs are independent until pooling (src/multipath.jl:190-208); `draws_per_component = stack(draws)`,
// `_compute_psis_result` and `_resample` (src/multipath.jl:215-225) see ALL runs.  Paths are sharded over the GPUs in
// contiguous blocks (pool order stays k-major, src/resample.jl:93); the data path has ONE exchange with real content -- an
// all-gather of the fp64 log-ratio shards (K/G * N_r doubles per GPU, 64 KB at config 4) -- after which PSIS and the index
// selection run REPLICATED and deterministically on every GPU (integer CDF => identical indices for any G).  The selected
// columns live on the GPU that owns their path: every GPU fills its own columns into a zeroed (d x ndraws) buffer and one
// sum all-reduce assembles the result (8 MB at config 4).  The draw pool itself is never exchanged.
//
// RCCL is called directly (ncclAllGather / ncclAllReduce on the contexts' own streams).  Two ways to form the group:
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

// A rank whose local stage failed still enters the all-reduce (so nobody blocks).  The result buffer carries ONE extra element behind
// the d x ndraws columns -- the failure flag: 0 from a healthy rank, 1 from a failed one -- so that after the sum every rank sees how
// many ranks failed, whatever the draws themselves contain (a NaN draw of a failed run is data, not an error).
__global__ void pf_flag_kernel(double *flag, double value) { *flag = value; }
__global__ void pf_zero_kernel(double *out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.0;
}

}  // namespace

// one communicator = the local ranks this process drives (all G under init_all, exactly one under init_rank)
struct pfmi_comm {
    int world = 0;                       // ranks in the world
    bool rccl = false;                   // collectives go through RCCL (false: a world of one context, nothing to exchange)
    std::vector<pfmi_ctx *> ctx;         // local contexts
    std::vector<ncclComm_t> comm;        // their communicators
    std::vector<int> rank;               // their world ranks
    std::vector<DevBuf> lr_all;          // [world * shard] gathered log ratios, one per local ctx
    std::vector<DevBuf> out;             // [d * ndraws] owner-filled result, one per local ctx
    std::vector<DevBuf> hs;              // [4] handshake block, one per local ctx
    int64_t shard = 0;                   // K_local * N_r agreed by the last pooled stage
    bool psis_pending = false;
    int64_t rs_ndraws = 0;               // draws of the enqueued resample stage
    bool rs_local_error = false;
    bool pr_pending = false, pr_importance = false;   // pfmi_comm_psis_resample_enqueue is waiting for its pfmi_comm_psis_resample_wait
    int32_t pr_rc = 0;
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
        c->out[i].release();
        c->hs[i].release();
    }
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

int32_t group_all_gather(pfmi_comm *c) {
    const size_t nl = c->ctx.size();
    prof_begin_all(c);
    PF_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        ncclResult_t r = g_rccl.AllGather(x->pool_lr.p, c->lr_all[i].p, (size_t)c->shard, ncclDouble, c->comm[i], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclAllGather failed on rank %d: %s", c->rank[i], g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
    }
    PF_NCCL(g_rccl.GroupEnd());
    prof_end_all(c, "comm_allgather");
    return PFMI_OK;
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

// Agree on the shard size and on everybody's local status BEFORE a collective whose element count depends on them.  Under
// pfmi_comm_init_all this process sees every rank and the check is local; under pfmi_comm_init_rank the ranks exchange
// max{shard, -shard, error} (one 4-double all-reduce): a rank without a pool, with a different K_local * N_r, or whose device buffers
// for the stage cannot be allocated makes EVERY rank return the same error instead of leaving the others blocked in (or corrupting)
// the collective.  `out_doubles` > 0: the (d x ndraws + flag) result buffers of the resample stage are allocated here too, so that
// the fused call has no fallible step left between this handshake and its collectives.
int32_t handshake(pfmi_comm *c, double v, int local_err, const char *why, double *vmax, double *vmin) {
    const size_t nl = c->ctx.size();
    *vmax = *vmin = v;
    if ((int)nl < c->world) {                                   // one process per GPU: the other ranks are somewhere else
        double h[4] = {v, -v, (double)local_err, 0.0}, r[4] = {0, 0, 0, 0};
        const auto hs_t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < nl; ++i) {
            PF_HIP(hipSetDevice(c->ctx[i]->device));
            PF_TRY(c->hs[i].ensure(sizeof(h)));                 // 256 bytes, allocated by the first handshake of the communicator
            PF_TRY(pf_upload(c->ctx[i], c->hs[i].p, h, sizeof(h)));
        }
        PF_TRY(group_all_reduce(c, c->hs, 4, ncclMax));
        PF_HIP(hipSetDevice(c->ctx[0]->device));
        PF_TRY(pf_download(c->ctx[0], r, c->hs[0].p, sizeof(r)));
        PF_TRY(pf_stream_sync(c->ctx[0]));
        if (c->ctx[0]->profile) {
            KernelStat &ks = c->ctx[0]->kstats["comm_handshake_host"];
            ks.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs_t0).count();
            ks.launches += 1;
        }
        PF_CHECK(r[2] == 0.0, PFMI_ERR_STATE, "comm: a rank of the group cannot take part in the pooled stage%s%s", local_err ? ": " : "",
                 local_err ? why : " (see that rank's error)");
        *vmax = r[0];
        *vmin = -r[1];
    } else {
        PF_CHECK(!local_err, strstr(why, "differ") ? PFMI_ERR_ARG : PFMI_ERR_STATE, "comm: %s%s", why,
                 strstr(why, "differ") ? ": equal paths per GPU keep the result independent of G" : "");
    }
    return PFMI_OK;
}

int32_t agree_on_shard(pfmi_comm *c, int64_t *shard_out, bool want_lr_all, int64_t out_doubles) {
    const size_t nl = c->ctx.size();
    int64_t shard = -1;
    int local_err = 0;
    char why[256] = "";
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        if (!x->pooled) {
            local_err = 1;
            snprintf(why, sizeof(why), "rank %d has no pool (call pfmi_pool_build)", c->rank[i]);
            continue;
        }
        const int64_t s = (int64_t)x->K * x->N_r;
        if (shard >= 0 && s != shard) {
            local_err = 1;
            snprintf(why, sizeof(why), "log-ratio shards differ in size (%lld vs %lld)", (long long)s, (long long)shard);
        }
        shard = s;
    }
    // everything the stage allocates, BEFORE the handshake: a failure here is a local error every rank hears about
    for (size_t i = 0; i < nl && !local_err; ++i) {
        int32_t rc = hipSetDevice(c->ctx[i]->device) == hipSuccess ? PFMI_OK : PFMI_ERR_HIP;
        if (rc == PFMI_OK && want_lr_all && c->rccl) rc = c->lr_all[i].ensure(sizeof(double) * (size_t)shard * (size_t)c->world);
        if (rc == PFMI_OK && out_doubles > 0) rc = c->out[i].ensure(sizeof(double) * (size_t)out_doubles);
        if (rc != PFMI_OK) {
            local_err = 1;
            snprintf(why, sizeof(why), "rank %d could not allocate the buffers of the pooled stage (%s)", c->rank[i], pfmi_last_error());
        }
    }
    double smax = 0.0, smin = 0.0;
    PF_TRY(handshake(c, (double)shard, local_err, why, &smax, &smin));
    PF_CHECK(smax == smin, PFMI_ERR_ARG,
             "comm: log-ratio shards differ in size across ranks (%lld .. %lld): equal paths per GPU keep the result independent of G",
             (long long)smin, (long long)smax);
    *shard_out = shard;
    return PFMI_OK;
}

// all-gather + replicated PSIS, enqueued on every local context
int32_t enqueue_pool_psis(pfmi_comm *c, int64_t out_doubles) {
    const size_t nl = c->ctx.size();
    int64_t shard = 0;
    PF_TRY(agree_on_shard(c, &shard, true, out_doubles));
    c->shard = shard;
    const int64_t S = shard * c->world;
    if (c->rccl) PF_TRY(group_all_gather(c));                  // lr_all was allocated before the handshake
    for (size_t i = 0; i < nl; ++i) {                       // replicated PSIS: same code, same input, fixed reduction order
        pfmi_ctx *x = c->ctx[i];
        PF_HIP(hipSetDevice(x->device));
        PF_TRY(pf_launch_psis(x, c->rccl ? c->lr_all[i].as<double>() : x->pool_lr.as<double>(), S));
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
    std::vector<double> out;
    PF_TRY(queue_pool_psis(c, out));
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        PF_HIP(hipSetDevice(c->ctx[i]->device));
        PF_TRY(pf_stream_sync(c->ctx[i]));
    }
    return check_pool_psis(c, out, pareto_k, tail_len);
}

// replicated index selection -> owner gather (zeros elsewhere) -> sum all-reduce, enqueued on every local context.  The result buffers
// (d x ndraws columns + the failure flag) were allocated before the handshake of the stage (agree_on_shard / resample_handshake): no
// step between here and the all-reduce returns early, a local failure travels as flag = 1.
int32_t enqueue_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms) {
    const size_t nl = c->ctx.size();
    const int64_t S = c->shard * c->world;
    const int d = c->ctx[0]->d;
    const long long n = (long long)d * ndraws;
    c->rs_ndraws = ndraws;
    c->rs_local_error = false;
    int32_t rc_local = PFMI_OK;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        int32_t rc = PFMI_OK;
        if (hipSetDevice(x->device) != hipSuccess) { pf_set_error("comm_resample: hipSetDevice(%d) failed", x->device); rc = PFMI_ERR_HIP; }
        if (rc == PFMI_OK && c->out[i].cap < sizeof(double) * (size_t)(n + 1)) {
            pf_set_error("comm_resample: result buffer of rank %d was not allocated by the handshake", c->rank[i]);
            rc = PFMI_ERR_STATE;
        }
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
        if (rc == PFMI_OK) rc = pf_launch_gather(x, ndraws, x->idx.as<int64_t>(), (int64_t)c->rank[i] * c->shard, c->out[i].as<double>());
        if (c->out[i].cap >= sizeof(double) * (size_t)(n + 1)) {
            if (rc != PFMI_OK)                                  // a defined contribution: zeros + flag 1
                hipLaunchKernelGGL(pf_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, x->stream, c->out[i].as<double>(), n);
            hipLaunchKernelGGL(pf_flag_kernel, dim3(1), dim3(1), 0, x->stream, c->out[i].as<double>() + n, rc != PFMI_OK ? 1.0 : 0.0);
        }
        if (rc != PFMI_OK) {
            rc_local = rc;
            c->rs_local_error = true;
        }
    }
    if (c->rccl) {
        bool all_buffers = true;
        for (size_t i = 0; i < nl; ++i) all_buffers = all_buffers && c->out[i].cap >= sizeof(double) * (size_t)(n + 1);
        // (without its buffer a rank cannot enter: only reachable when the caller skipped the handshake, which the entry points never do)
        if (all_buffers) PF_TRY(group_all_reduce(c, c->out, (size_t)n + 1, ncclSum, "comm_allreduce"));
    }
    return rc_local;
}

int32_t finish_resample(pfmi_comm *c, int64_t *idx, double *draws) {
    const size_t nl = c->ctx.size();
    const int64_t ndraws = c->rs_ndraws;
    const int d = c->ctx[0]->d;
    const size_t n = (size_t)d * (size_t)ndraws;
    std::vector<int64_t> h_idx((size_t)ndraws * nl);
    std::vector<int> h_err(nl, 0);
    double flag = 0.0;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        PF_HIP(hipSetDevice(x->device));
        if (x->idx.cap >= sizeof(int64_t) * (size_t)ndraws)
            PF_TRY(pf_download(x, &h_idx[(size_t)ndraws * i], x->idx.p, sizeof(int64_t) * ndraws));
        if (x->rs_err.p) PF_TRY(pf_download(x, &h_err[i], x->rs_err.p, sizeof(int)));
    }
    {
        // the small results are staged (pinned, asynchronous); the draws -- megabytes into the caller's pageable array, a blocking
        // call -- are queued LAST, so this entry point costs ONE host round trip
        pfmi_ctx *x = c->ctx[0];
        PF_HIP(hipSetDevice(x->device));
        PF_CHECK(c->out[0].cap >= sizeof(double) * (n + 1), PFMI_ERR_STATE, "comm_resample: no result buffer");
        PF_TRY(pf_download(x, &flag, c->out[0].as<double>() + n, sizeof(double)));
        if (draws) PF_TRY(pf_download(x, draws, c->out[0].p, sizeof(double) * n));
    }
    for (size_t i = 0; i < nl; ++i) {
        PF_HIP(hipSetDevice(c->ctx[i]->device));
        PF_TRY(pf_stream_sync(c->ctx[i]));
    }
    for (size_t i = 0; i < nl; ++i)
        PF_CHECK(h_err[i] == 0, PFMI_ERR_NUMERIC, "resample: weights are all zero / not enough positive weights (rank %d)", c->rank[i]);
    // the summed failure flag: how many ranks of the group failed their local stage
    PF_CHECK(flag == 0.0, PFMI_ERR_COMM, "comm_resample: %d rank(s) of the group failed their local stage", (int)flag);
    for (size_t i = 1; i < nl; ++i)
        PF_CHECK(memcmp(&h_idx[(size_t)ndraws * i], h_idx.data(), sizeof(int64_t) * ndraws) == 0, PFMI_ERR_NUMERIC,
                 "comm_resample: replicated index selection disagrees on rank %d", c->rank[i]);
    if (idx) memcpy(idx, h_idx.data(), sizeof(int64_t) * (size_t)ndraws);
    return PFMI_OK;
}

// the unfused pfmi_comm_resample has no pooled-stage handshake in front of it: allocate the result buffers and let every rank hear
// about a failure (one 4-double all-reduce under pfmi_comm_init_rank, nothing under pfmi_comm_init_all)
int32_t resample_handshake(pfmi_comm *c, int64_t out_doubles) {
    int local_err = 0;
    char why[256] = "";
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        int32_t rc = hipSetDevice(c->ctx[i]->device) == hipSuccess ? PFMI_OK : PFMI_ERR_HIP;
        if (rc == PFMI_OK) rc = c->out[i].ensure(sizeof(double) * (size_t)out_doubles);
        if (rc != PFMI_OK) {
            local_err = 1;
            snprintf(why, sizeof(why), "rank %d could not allocate the result buffer of the resample stage (%s)", c->rank[i], pfmi_last_error());
        }
    }
    double a, b;
    return handshake(c, (double)out_doubles, local_err, why, &a, &b);
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
    PF_CHECK(G >= 1 && ctxs != nullptr, PFMI_ERR_ARG, "comm_init_all: bad arguments");
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
    c->out.resize((size_t)G);
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
    PF_CHECK(ctx != nullptr && id128 != nullptr && world >= 1 && rank >= 0 && rank < world, PFMI_ERR_ARG, "comm_init_rank: bad arguments");
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
    c->out.resize(1);
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
    if (c) for (pfmi_ctx *x : c->ctx) pf_download_forget(x);      // staged downloads of an entry point that failed half-way
    PF_COMM(c);
    PF_TRY(enqueue_pool_psis(c, 0));
    return finish_pool_psis(c, pareto_k, tail_len);
}

// _resample over the pooled runs (src/multipath.jl:225, src/resample.jl:58-72): replicated index selection, owner gather,
// sum all-reduce.  idx[ndraws] (0-based, global pool columns) and draws[d * ndraws] (column-major) may be NULL.
int32_t pfmi_comm_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms,
                           int64_t *idx, double *draws) {
    if (c) for (pfmi_ctx *x : c->ctx) pf_download_forget(x);      // staged downloads of an entry point that failed half-way
    PF_COMM(c);
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_resample: ndraws must be positive");
    PF_CHECK(c->shard > 0, PFMI_ERR_STATE, "comm_resample: call pfmi_comm_pool_psis first");
    PF_TRY(resample_handshake(c, (int64_t)c->ctx[0]->d * ndraws + 1));
    const int32_t rc = enqueue_resample(c, ndraws, importance, replace, seed, uniforms);
    const int32_t rf = finish_resample(c, idx, draws);
    return rc != PFMI_OK ? rc : rf;
}

// both stages, one synchronisation (src/multipath.jl:221-225)
int32_t pfmi_comm_psis_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms,
                                double *pareto_k, int64_t *tail_len, int64_t *idx, double *draws) {
    if (c) for (pfmi_ctx *x : c->ctx) pf_download_forget(x);      // staged downloads of an entry point that failed half-way
    PF_COMM(c);
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_psis_resample: ndraws must be positive");
    const int64_t out_doubles = (int64_t)c->ctx[0]->d * ndraws + 1;       // d x ndraws columns + the failure flag
    if (importance) PF_TRY(enqueue_pool_psis(c, out_doubles));
    else {
        int64_t shard = 0;
        PF_TRY(agree_on_shard(c, &shard, false, out_doubles));
        c->shard = shard;
    }
    const int32_t rc = enqueue_resample(c, ndraws, importance, replace, seed, uniforms);
    double k = NAN;
    int64_t m = 0;
    int32_t rp = PFMI_OK;
    std::vector<double> pout;
    if (importance) rp = queue_pool_psis(c, pout);                      // the PSIS scalars ride in front of the resample stage's results:
    const int32_t rf = finish_resample(c, idx, draws);                  // ONE wait for both stages
    c->psis_pending = false;
    if (importance && rp == PFMI_OK && rf == PFMI_OK) rp = check_pool_psis(c, pout, &k, &m);     // (a failed wait delivered nothing)
    if (pareto_k) *pareto_k = k;
    if (tail_len) *tail_len = m;
    return rc != PFMI_OK ? rc : (rp != PFMI_OK ? rp : rf);
}

// the same in two halves: everything enqueued / the one wait.  Between the two the caller may queue downloads on the member contexts
// (pfmi_defer_downloads): the wait delivers them in the same host round trip.
int32_t pfmi_comm_psis_resample_enqueue(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms) {
    if (c) for (pfmi_ctx *x : c->ctx) pf_download_forget(x);
    PF_COMM(c);
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_psis_resample: ndraws must be positive");
    const int64_t out_doubles = (int64_t)c->ctx[0]->d * ndraws + 1;
    if (importance) PF_TRY(enqueue_pool_psis(c, out_doubles));
    else {
        int64_t shard = 0;
        PF_TRY(agree_on_shard(c, &shard, false, out_doubles));
        c->shard = shard;
    }
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
    int32_t rp = PFMI_OK;
    std::vector<double> pout;
    if (c->pr_importance) rp = queue_pool_psis(c, pout);
    const int32_t rf = finish_resample(c, idx, draws);                  // ONE wait for both stages (and whatever the contexts queued)
    c->psis_pending = false;
    if (c->pr_importance && rp == PFMI_OK && rf == PFMI_OK) rp = check_pool_psis(c, pout, &k, &m);
    if (pareto_k) *pareto_k = k;
    if (tail_len) *tail_len = m;
    return c->pr_rc != PFMI_OK ? c->pr_rc : (rp != PFMI_OK ? rp : rf);
}

}  // extern "C"
