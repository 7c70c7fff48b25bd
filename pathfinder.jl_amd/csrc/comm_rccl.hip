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
// librccl is opened with dlopen at the first pfmi_comm_* call, so libpfmi.so itself has no link-time dependency on it and a
// process that already carries an RCCL (PyTorch bundles one under the same SONAME) keeps a single copy.
#include "pfmi_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
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

int32_t rccl_load() {
    if (g_rccl.handle) return PFMI_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    PF_CHECK(h != nullptr, PFMI_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
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

}  // namespace

// one communicator = the local ranks this process drives (all G under init_all, exactly one under init_rank)
struct pfmi_comm {
    int world = 0;                       // ranks in the RCCL world
    std::vector<pfmi_ctx *> ctx;         // local contexts
    std::vector<ncclComm_t> comm;        // their communicators
    std::vector<int> rank;               // their world ranks
    std::vector<DevBuf> lr_all;          // [world * shard] gathered log ratios, one per local ctx
    std::vector<DevBuf> out;             // [d * ndraws] owner-filled result, one per local ctx
    int64_t shard = 0;                   // K_local * N_r of the last all-gather
};

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
    PF_TRY(rccl_load());
    std::vector<int> devs((size_t)G);
    for (int r = 0; r < G; ++r) {
        PF_CHECK(ctxs[r] != nullptr, PFMI_ERR_ARG, "comm_init_all: null context %d", r);
        devs[(size_t)r] = ctxs[r]->device;
        for (int q = 0; q < r; ++q)
            PF_CHECK(devs[(size_t)q] != devs[(size_t)r], PFMI_ERR_ARG, "comm_init_all: contexts %d and %d share GPU %d (one rank per GPU)", q, r,
                     devs[(size_t)r]);
    }
    pfmi_comm *c = new pfmi_comm();
    c->world = G;
    c->ctx.assign(ctxs, ctxs + G);
    c->comm.assign((size_t)G, nullptr);
    c->rank.resize((size_t)G);
    for (int r = 0; r < G; ++r) c->rank[(size_t)r] = r;
    c->lr_all.resize((size_t)G);
    c->out.resize((size_t)G);
    ncclResult_t rc = g_rccl.CommInitAll(c->comm.data(), G, devs.data());
    if (rc != ncclSuccess) {
        pf_set_error("ncclCommInitAll(%d GPUs) failed: %s", G, g_rccl.GetErrorString(rc));
        delete c;
        return PFMI_ERR_COMM;
    }
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
    c->ctx.assign(1, ctx);
    c->comm.assign(1, nullptr);
    c->rank.assign(1, rank);
    c->lr_all.resize(1);
    c->out.resize(1);
    ncclResult_t rc = g_rccl.CommInitRank(&c->comm[0], world, id, rank);
    if (rc != ncclSuccess) {
        pf_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(rc));
        delete c;
        return PFMI_ERR_COMM;
    }
    *out = c;
    return PFMI_OK;
}

int32_t pfmi_comm_destroy(pfmi_comm *c) {
    if (!c) return PFMI_OK;
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        (void)hipSetDevice(c->ctx[i]->device);
        (void)hipStreamSynchronize(c->ctx[i]->stream);
        if (c->comm[i]) (void)g_rccl.CommDestroy(c->comm[i]);
        c->lr_all[i].release();
        c->out[i].release();
    }
    delete c;
    return PFMI_OK;
}

int32_t pfmi_comm_info(pfmi_comm *c, int32_t *world, int32_t *nlocal, int32_t *rccl_version) {
    PF_CHECK(c != nullptr, PFMI_ERR_ARG, "null pfmi_comm");
    if (world) {
        int n = 0;
        PF_NCCL(g_rccl.CommCount(c->comm[0], &n));         // what RCCL itself says, not what the caller claimed
        *world = n;
    }
    if (nlocal) *nlocal = (int32_t)c->ctx.size();
    if (rccl_version) {
        int v = 0;
        PF_NCCL(g_rccl.GetVersion(&v));
        *rccl_version = v;
    }
    return PFMI_OK;
}

// _compute_psis_result over the pooled runs (src/multipath.jl:221): all-gather the log-ratio shards, then PSIS on every GPU.
int32_t pfmi_comm_pool_psis(pfmi_comm *c, double *pareto_k, int64_t *tail_len) {
    PF_CHECK(c != nullptr, PFMI_ERR_ARG, "null pfmi_comm");
    const size_t nl = c->ctx.size();
    int64_t shard = -1;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        PF_CHECK(x->pooled, PFMI_ERR_STATE, "comm_pool_psis: rank %d has no pool (call pfmi_pool_build)", c->rank[i]);
        const int64_t s = (int64_t)x->K * x->N_r;
        PF_CHECK(shard < 0 || s == shard, PFMI_ERR_ARG,
                 "comm_pool_psis: log-ratio shards differ in size (%lld vs %lld): equal paths per GPU keep the result independent of G",
                 (long long)s, (long long)shard);
        shard = s;
    }
    c->shard = shard;
    const int64_t S = shard * c->world;
    for (size_t i = 0; i < nl; ++i) {
        PF_HIP(hipSetDevice(c->ctx[i]->device));
        PF_TRY(c->lr_all[i].ensure(sizeof(double) * (size_t)S));
    }
    PF_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        ncclResult_t r = g_rccl.AllGather(x->pool_lr.p, c->lr_all[i].p, (size_t)shard, ncclDouble, c->comm[i], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclAllGather failed on rank %d: %s", c->rank[i], g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
    }
    PF_NCCL(g_rccl.GroupEnd());
    double k0 = 0.0;
    int64_t m0 = 0;
    for (size_t i = 0; i < nl; ++i) {                       // replicated PSIS: same code, same input, fixed reduction order
        double k;
        int64_t m;
        PF_TRY(pfmi_psis_dev(c->ctx[i], c->lr_all[i].p, S, nullptr, nullptr, &k, &m));
        if (i == 0) { k0 = k; m0 = m; }
        else PF_CHECK(m == m0 && (k == k0 || (k != k && k0 != k0)), PFMI_ERR_NUMERIC, "comm_pool_psis: replicas disagree (rank %d)", c->rank[i]);
    }
    if (pareto_k) *pareto_k = k0;
    if (tail_len) *tail_len = m0;
    return PFMI_OK;
}

// _resample over the pooled runs (src/multipath.jl:225, src/resample.jl:58-72): replicated index selection, owner gather,
// sum all-reduce.  idx[ndraws] (0-based, global pool columns) and draws[d * ndraws] (column-major) may be NULL.
int32_t pfmi_comm_resample(pfmi_comm *c, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms,
                           int64_t *idx, double *draws) {
    PF_CHECK(c != nullptr, PFMI_ERR_ARG, "null pfmi_comm");
    PF_CHECK(ndraws >= 1, PFMI_ERR_ARG, "comm_resample: ndraws must be positive");
    PF_CHECK(c->shard > 0, PFMI_ERR_STATE, "comm_resample: call pfmi_comm_pool_psis first");
    const size_t nl = c->ctx.size();
    const int64_t S = c->shard * c->world;
    const int d = c->ctx[0]->d;
    std::vector<int64_t> h_idx((size_t)ndraws), h_idx0;
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        PF_CHECK(x->d == d, PFMI_ERR_ARG, "comm_resample: dimension differs between ranks");
        PF_TRY(pfmi_resample_indices(x, S, ndraws, importance, replace, seed, uniforms, h_idx.data()));
        if (i == 0) h_idx0 = h_idx;
        else PF_CHECK(h_idx == h_idx0, PFMI_ERR_NUMERIC, "comm_resample: replicated index selection disagrees on rank %d", c->rank[i]);
        PF_TRY(c->out[i].ensure(sizeof(double) * (size_t)d * ndraws));
        PF_TRY(pfmi_pool_gather_dev(x, ndraws, h_idx0.data(), (int64_t)c->rank[i] * c->shard, c->out[i].p));   // zeros where not owned
    }
    PF_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        pfmi_ctx *x = c->ctx[i];
        ncclResult_t r = g_rccl.AllReduce(c->out[i].p, c->out[i].p, (size_t)d * ndraws, ncclDouble, ncclSum, c->comm[i], x->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pf_set_error("ncclAllReduce failed on rank %d: %s", c->rank[i], g_rccl.GetErrorString(r));
            return PFMI_ERR_COMM;
        }
    }
    PF_NCCL(g_rccl.GroupEnd());
    for (size_t i = 0; i < nl; ++i) {
        PF_HIP(hipSetDevice(c->ctx[i]->device));
        PF_HIP(hipStreamSynchronize(c->ctx[i]->stream));
    }
    if (idx) memcpy(idx, h_idx0.data(), sizeof(int64_t) * (size_t)ndraws);
    if (draws) {
        pfmi_ctx *x = c->ctx[0];
        PF_HIP(hipSetDevice(x->device));
        PF_HIP(hipMemcpyAsync(draws, c->out[0].p, sizeof(double) * (size_t)d * ndraws, hipMemcpyDeviceToHost, x->stream));
        PF_HIP(hipStreamSynchronize(x->stream));
    }
    return PFMI_OK;
}

}  // extern "C"
