// pfmi_api.hip -- the extern "C" boundary declared in include/pfmi.h (host side of libpfmi.so).
#include "pfmi_common.h"
#include "fit_args.h"
#include <limits.h>

#include <array>
#include <chrono>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string>
#include <thread>
#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

static thread_local char g_err[1024] = "";

void pf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Stage timers.  Mode 1 (pfmi_profile(ctx, 1)) synchronises the host after every stage: each stage then starts on an idle GPU
// and its figure includes the host's launch latency (~1 ms for the scan's launch sequence).  Mode 2 leaves the event pairs in the
// stream and reads them when pfmi_kernel_time asks: the pipeline runs as it does unprofiled, the first event of a stage gets its
// time stamp when the previous stage's last kernel retires, so a stage's figure is its kernels' time.
static hipEvent_t pf_kev_get(pfmi_ctx *c) {
    if (!c->kev_pool.empty()) { hipEvent_t e = c->kev_pool.back(); c->kev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void pf_kernel_resolve(pfmi_ctx *c, bool keep) {
    for (auto &p : c->kpending) {
        float ms = 0.f;
        if (keep && p.e0 && p.e1 && hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            KernelStat &s = c->kstats[p.name];
            s.ms += ms;
            s.launches += 1;
        }
        if (p.e0) c->kev_pool.push_back(p.e0);
        if (p.e1) c->kev_pool.push_back(p.e1);
    }
    c->kpending.clear();
}
void pf_kernel_begin(pfmi_ctx *c) {
    if (c->profile == 1) (void)hipEventRecord(c->kev0, c->stream);
    else if (c->profile == 2) {
        if (c->kpending.size() >= 4096) pf_kernel_resolve(c, true);
        if (c->kcur) c->kev_pool.push_back(c->kcur);                  // a begin that was never ended (no kernel took the launch): reuse its event
        c->kcur = pf_kev_get(c);
        if (c->kcur) (void)hipEventRecord(c->kcur, c->stream);
    }
}
void pf_kernel_end(pfmi_ctx *c, const char *name) {
    if (c->profile == 2) {
        hipEvent_t e1 = pf_kev_get(c);
        if (e1) (void)hipEventRecord(e1, c->stream);
        c->kpending.push_back({name, c->kcur, e1});                   // (name: a string literal of the caller)
        c->kcur = nullptr;
        return;
    }
    if (c->profile != 1) return;
    (void)hipEventRecord(c->kev1, c->stream);
    (void)hipEventSynchronize(c->kev1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->kev0, c->kev1);
    KernelStat &s = c->kstats[name];
    s.ms += ms;
    s.launches += 1;
}

// ---- host <-> device copies ---------------------------------------------------------------------------------------
// Uploads up to PF_ARENA_MAX bytes are staged in the ctx's pinned arena and copied asynchronously: the caller's buffer is free on
// return and the stream is NOT synchronised, so a chain of entry points (fit -> scan -> pool -> PSIS -> resample) is enqueued
// without a single host round trip.  Larger blocks (traces, parity-mode normals) take the plain synchronous path.
#define PF_ARENA_BYTES (16u << 20)
#define PF_ARENA_MAX (4u << 20)
void pf_arena_reset(pfmi_ctx *c) { c->arena.off = 0; }
int32_t pf_upload(pfmi_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return PFMI_OK;
    if (bytes > PF_ARENA_MAX) {
        PF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        PF_HIP(hipStreamSynchronize(c->stream));
        pf_arena_reset(c);
        return PFMI_OK;
    }
    PinArena &a = c->arena;
    if (!a.base) {
        void *pp = nullptr;
        PF_HIP(hipHostMalloc(&pp, PF_ARENA_BYTES, hipHostMallocDefault));
        a.base = reinterpret_cast<char *>(pp); a.cap = PF_ARENA_BYTES; a.off = 0;
    }
    if (a.off + bytes > a.cap) {                 // full: everything staged so far must have been consumed before the rewind
        PF_HIP(hipStreamSynchronize(c->stream));
        a.off = 0;
    }
    memcpy(a.base + a.off, src, bytes);
    PF_HIP(hipMemcpyAsync(dst, a.base + a.off, bytes, hipMemcpyHostToDevice, c->stream));
    a.off += (bytes + 255) & ~(size_t)255;
    return PFMI_OK;
}
static int32_t h2d(pfmi_ctx *c, void *dst, const void *src, size_t bytes) { return pf_upload(c, dst, src, bytes); }
int32_t pf_dl_flush(pfmi_ctx *c);
#define PF_DL_BYTES (4u << 20)
#define PF_DL_MAX (1u << 20)
int32_t pf_download(pfmi_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return PFMI_OK;
    PinArena &a = c->dl;
    if (bytes <= PF_DL_MAX) {
        if (!a.base) {
            void *pp = nullptr;
            PF_HIP(hipHostMalloc(&pp, PF_DL_BYTES, hipHostMallocMapped));          // (the gather kernel of pf_dl_flush writes into it)
            a.base = reinterpret_cast<char *>(pp); a.cap = PF_DL_BYTES; a.off = 0;
        }
        if (a.off + bytes <= a.cap) {
            // (round 5) the copy itself is NOT issued here: every small result of an entry point used to be its own blit kernel in the stream
            // (eight of them, 5 us each, behind the last kernel of a step); pf_dl_flush -- called by the wait -- moves all of them with ONE
            // kernel that writes straight into the page-locked arena.  The sources are result buffers nothing rewrites before the wait.
            static const bool immediate = [] { const char *e = getenv("PFMI_DL_IMMEDIATE"); return e && e[0] == '1'; }();      // A/B: one blit copy per result, at once
            if (immediate) PF_HIP(hipMemcpyAsync(a.base + a.off, src, bytes, hipMemcpyDeviceToHost, c->stream));
            c->dl_pending.push_back(DlPending{dst, a.base + a.off, bytes, c->defer, immediate ? nullptr : static_cast<const char *>(src), immediate});
            a.off += (bytes + 255) & ~(size_t)255;
            return PFMI_OK;
        }
    }
    // a large result: its own copy (pageable destination: blocks until the copy is done).  The staged small results go out FIRST, as their one
    // gather kernel, so that the wait behind this copy finds them delivered instead of paying a launch + a second round trip
    PF_TRY(pf_dl_flush(c));
    PF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return PFMI_OK;
}
static void pf_deferred_drop(pfmi_ctx *c) { c->dl_pending.clear(); c->post_sync.clear(); c->dl.off = 0; }
// ---- one kernel for all staged downloads of a wait: block (x, y) copies slice x of segment y into the page-locked arena (zero copy)
#define PF_DL_NSEG 24
struct DlSegs { const char *src[PF_DL_NSEG]; char *dst[PF_DL_NSEG]; uint32_t bytes[PF_DL_NSEG]; };
__global__ __launch_bounds__(256) void pf_dl_gather_kernel(DlSegs S) {
    const int y = blockIdx.y;
    const char *src = S.src[y];
    char *dst = S.dst[y];
    const uint32_t n = S.bytes[y];
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | n) & 7u) == 0) {
        const uint32_t n8 = n >> 3;
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n8; i += gridDim.x * 256)
            reinterpret_cast<uint64_t *>(dst)[i] = reinterpret_cast<const uint64_t *>(src)[i];
    } else {
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
    }
}
int32_t pf_dl_flush(pfmi_ctx *c) {
    DlSegs S;
    int n = 0;
    uint32_t big = 0;
    auto launch = [&]() -> int32_t {
        if (n == 0) return PFMI_OK;
        const unsigned gx = big > (64u << 10) ? 32 : big > (4u << 10) ? 4 : 1;
        hipLaunchKernelGGL(pf_dl_gather_kernel, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, S);
        PF_HIP(hipGetLastError());
        n = 0; big = 0;
        return PFMI_OK;
    };
    for (DlPending &q : c->dl_pending) {
        if (q.issued || q.src == nullptr) continue;
        S.src[n] = q.src; S.dst[n] = const_cast<char *>(q.slot); S.bytes[n] = (uint32_t)q.bytes;
        if ((uint32_t)q.bytes > big) big = (uint32_t)q.bytes;
        q.issued = true;
        if (++n == PF_DL_NSEG) PF_TRY(launch());
    }
    return launch();
}
int32_t pf_stream_sync(pfmi_ctx *c) {
    {
        const int32_t rf = pf_dl_flush(c);
        if (rf != PFMI_OK) { pf_deferred_drop(c); return rf; }
    }
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { pf_deferred_drop(c); PF_HIP(e); }        // nothing is delivered after a failed wait
    for (const DlPending &q : c->dl_pending) memcpy(q.dst, q.slot, q.bytes);
    c->dl_pending.clear();
    c->dl.off = 0;
    pf_arena_reset(c);
    // what deferred entry points would have done behind their own wait (pfmi_defer_downloads): the first failure is this wait's result
    int32_t rc = PFMI_OK;
    std::vector<std::function<int32_t()>> hooks;
    hooks.swap(c->post_sync);
    for (auto &h : hooks) { const int32_t r = h(); if (rc == PFMI_OK) rc = r; }
    return rc;
}
// Staged downloads are delivered by the pf_stream_sync of the entry point that queued them.  An entry point that FAILED between queuing and
// waiting leaves entries whose destinations (stack variables, a caller's array) may be gone: every public entry point forgets them first.
// (entries queued under pfmi_defer_downloads stay: their destinations are the caller's until the next wait)
void pf_download_forget(pfmi_ctx *c) {
    size_t w = 0;
    for (size_t i = 0; i < c->dl_pending.size(); ++i) if (c->dl_pending[i].keep) c->dl_pending[w++] = c->dl_pending[i];
    c->dl_pending.resize(w);
    if (w == 0) c->dl.off = 0;
}
static int32_t d2h_async(pfmi_ctx *c, void *dst, const void *src, size_t bytes) { return pf_download(c, dst, src, bytes); }
static int32_t stream_sync(pfmi_ctx *c) { return pf_stream_sync(c); }
static int32_t d2h(pfmi_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return PFMI_OK;
    PF_TRY(d2h_async(c, dst, src, bytes));
    return stream_sync(c);
}
// pinned host staging of the callback path (grown on demand, freed in pfmi_destroy)
static int32_t ensure_pinned(pfmi_ctx *c, size_t x_bytes, size_t lp_bytes) {
    for (int b = 0; b < 2; ++b) {
        if (x_bytes > c->pin_x_cap) {
            if (c->pin_x[b]) (void)hipHostFree(c->pin_x[b]);
            c->pin_x[b] = nullptr;
            PF_HIP(hipHostMalloc(&c->pin_x[b], x_bytes, hipHostMallocDefault));
        }
        if (lp_bytes > c->pin_lp_cap) {
            if (c->pin_lp[b]) (void)hipHostFree(c->pin_lp[b]);
            c->pin_lp[b] = nullptr;
            PF_HIP(hipHostMalloc(&c->pin_lp[b], lp_bytes, hipHostMallocDefault));
        }
        if (!c->cb_ev[b]) PF_HIP(hipEventCreateWithFlags(&c->cb_ev[b], hipEventDisableTiming));
    }
    if (x_bytes > c->pin_x_cap) c->pin_x_cap = x_bytes;
    if (lp_bytes > c->pin_lp_cap) c->pin_lp_cap = lp_bytes;
    return PFMI_OK;
}
#define PF_CTX(c)                                                          \
    do {                                                                   \
        PF_CHECK((c) != nullptr, PFMI_ERR_ARG, "null pfmi_ctx");           \
        PF_HIP(hipSetDevice((c)->device));                                 \
        pf_download_forget(c);                                             \
    } while (0)
// Entry points that enqueue work which REWRITES (or may reallocate) result buffers: the staged downloads still pending -- entries queued under
// pfmi_defer_downloads survive several entry points, and their copy is only issued by the gather kernel of the next wait -- are issued NOW, in
// stream order in front of the new work, so that they deliver the values of the call they were queued for (a source must be a
// snapshot at queueing time, as it was when every download was its own stream-ordered copy).  No pending entry: no launch.
#define PF_CTX_MUT(c)                                                      \
    do {                                                                   \
        PF_CTX(c);                                                         \
        PF_TRY(pf_dl_flush(c));                                            \
    } while (0)

// ---- test / tuning hooks (pfmi_common.h: pf_debug_get) ---------------------------------------------------------------------
namespace {
std::mutex g_dbg_mu;
// key -> value.  A value is an immutable heap string that is NEVER freed or modified once published (a later pfmi_debug_set of the same key
// installs a new string and leaks the old one: a few bytes per call of a test hook), so the pointer pf_debug_get hands out stays valid after
// the mutex is released, whatever other threads set meanwhile.  An unset key keeps a nullptr tombstone.
std::map<std::string, const char *> g_dbg;
}
const char *pf_debug_get(const char *name) {
    {
        std::lock_guard<std::mutex> lk(g_dbg_mu);
        auto it = g_dbg.find(name);
        if (it != g_dbg.end()) return it->second;
    }
    static const bool env_hooks = [] { const char *e = getenv("PFMI_DEBUG_HOOKS"); return e && e[0] == '1'; }();
    return env_hooks ? getenv(name) : nullptr;
}

extern "C" {

const char *pfmi_last_error(void) { return g_err; }

int32_t pfmi_debug_set(const char *key, const char *value) {
    PF_CHECK(key != nullptr && strncmp(key, "PFMI_", 5) == 0, PFMI_ERR_ARG, "debug_set: keys are the PFMI_* hook names");
    const char *copy = (value && value[0]) ? strdup(value) : nullptr;
    PF_CHECK(copy != nullptr || !(value && value[0]), PFMI_ERR_ARG, "debug_set: out of memory");
    std::lock_guard<std::mutex> lk(g_dbg_mu);
    g_dbg[key] = copy;                                   // the previous string (if any) is deliberately leaked: readers may still hold it
    return PFMI_OK;
}
int32_t pfmi_version(void) { return 100; }

int32_t pfmi_device_count(int32_t *count) {
    PF_CHECK(count != nullptr, PFMI_ERR_ARG, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; pf_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return PFMI_ERR_HIP; }
    *count = n;
    return PFMI_OK;
}

int32_t pfmi_create(int32_t device, pfmi_ctx **out) {
    PF_CHECK(out != nullptr, PFMI_ERR_ARG, "null out");
    *out = nullptr;
    int n = 0;
    PF_HIP(hipGetDeviceCount(&n));
    PF_CHECK(n > 0, PFMI_ERR_HIP, "no HIP device visible: libpfmi has no CPU fallback");
    PF_CHECK(device >= 0 && device < n, PFMI_ERR_ARG, "device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, device));
    PF_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, PFMI_ERR_UNSUPPORTED,
             "libpfmi is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    PF_HIP(hipSetDevice(device));
    pfmi_ctx *c = new pfmi_ctx();
    c->device = device;
    c->ncu = prop.multiProcessorCount;
    PF_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    PF_HIP(hipEventCreate(&c->ev0));
    PF_HIP(hipEventCreate(&c->ev1));
    PF_HIP(hipEventCreate(&c->kev0));
    PF_HIP(hipEventCreate(&c->kev1));
    *out = c;
    return PFMI_OK;
}

// a streaming call that was never waited for (pfmi_stream_wait): let what is in flight on the side streams finish -- the optimiser still writes the
// staging trace -- and forget the call.  Called by whatever is about to reuse or free that memory.
static void stream_abandon(pfmi_ctx *c) {
    for (hipStream_t s : {c->s_opt, c->s_fit, c->s_scan1}) if (s) (void)hipStreamSynchronize(s);
    for (hipStream_t s : c->s_cb) if (s) (void)hipStreamSynchronize(s);
    c->sr.active = false;
    c->stream_pending = false;
}

int32_t pfmi_destroy(pfmi_ctx *c) {
    if (!c) return PFMI_OK;
    pf_comm_ctx_dying(c);                   // communicators that borrow this context close themselves first (any finaliser order is safe)
    (void)hipSetDevice(c->device);
    stream_abandon(c);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->theta, &c->grad, &c->d_off, &c->d_path_of, &c->target.mean, &c->target.a, &c->target.wd,
                      &c->target.g, &c->target.wd16, &c->alpha_all, &c->hist_len, &c->hist_src, &c->hist_acc, &c->n_rej, &c->vh, &c->tmat, &c->vchol,
                      &c->rq, &c->dmat, &c->sqrt_alpha, &c->mu, &c->logdet, &c->status, &c->seeds, &c->logp, &c->logq,
                      &c->elbo, &c->se, &c->best_iter, &c->fit_list, &c->ubuf, &c->xbuf, &c->scratch, &c->qf_share_s[0], &c->qf_share_s[1], &c->fit_scratch, &c->pool,
                      &c->pool_lr, &c->pool_lp, &c->pool_lq, &c->pool_points, &c->pool_seeds, &c->lw, &c->w,
                      &c->psis_out, &c->psis_aux, &c->tailbuf, &c->cdf, &c->idx, &c->gbuf, &c->trace_lp, &c->st_theta, &c->st_grad,
                      &c->st_lp, &c->st_npts, &c->lb_hs, &c->lb_hy, &c->lb_x0, &c->sortk, &c->sorti,
                      &c->pool_ok, &c->fail_seeds, &c->rs_err, &c->hs_ial, &c->hs_nacc};
    for (DevBuf *b : bufs) b->release();
    for (int b = 0; b < 2; ++b) {
        c->cb_x[b].release(); c->cb_lp[b].release();
        if (c->pin_x[b]) (void)hipHostFree(c->pin_x[b]);
        if (c->pin_lp[b]) (void)hipHostFree(c->pin_lp[b]);
        if (c->cb_ev[b]) (void)hipEventDestroy(c->cb_ev[b]);
    }
    if (c->arena.base) (void)hipHostFree(c->arena.base);
    if (c->dl.base) (void)hipHostFree(c->dl.base);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    (void)hipEventDestroy(c->kev0); (void)hipEventDestroy(c->kev1);
    pf_kernel_resolve(c, false);
    for (hipEvent_t e : c->kev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c->sg_fit, c->sg_opt, c->sg_scan[0], c->sg_scan[1], c->sg_start}) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : {c->s_opt, c->s_fit, c->s_scan1}) if (s) (void)hipStreamDestroy(s);
    for (hipStream_t s : c->s_cb) if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : c->sg_cb) if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < PF_DCB_NB; ++b) { c->dcb_x[b].release(); c->dcb_lp[b].release(); }
    if (c->h_prog) (void)hipHostFree(c->h_prog);
    if (c->h_list) (void)hipHostFree(c->h_list);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PFMI_OK;
}

int32_t pfmi_sync(pfmi_ctx *c) {
    PF_CTX(c);
    return stream_sync(c);
}

int32_t pfmi_timer_start(pfmi_ctx *c) {
    PF_CTX(c);
    PF_HIP(hipEventRecord(c->ev0, c->stream));
    return PFMI_OK;
}
int32_t pfmi_timer_stop(pfmi_ctx *c, double *ms) {
    PF_CTX(c);
    PF_HIP(hipEventRecord(c->ev1, c->stream));
    PF_HIP(hipEventSynchronize(c->ev1));
    float f = 0.f;
    PF_HIP(hipEventElapsedTime(&f, c->ev0, c->ev1));
    if (ms) *ms = f;
    return PFMI_OK;
}
int32_t pfmi_profile(pfmi_ctx *c, int32_t enable) {
    PF_CTX(c);
    PF_CHECK(enable >= 0 && enable <= 2, PFMI_ERR_ARG, "profile: mode %d outside 0..2", enable);
    pf_kernel_resolve(c, false);
    c->profile = enable;
    c->kstats.clear();
    return PFMI_OK;
}
int32_t pfmi_kernel_time(pfmi_ctx *c, const char *name, double *ms, int64_t *launches) {
    PF_CTX(c);
    PF_CHECK(name != nullptr, PFMI_ERR_ARG, "null name");
    pf_kernel_resolve(c, true);                                       // mode 2: waits for the recorded stages
    if (strcmp(name, "qf_handover_lost") == 0) {                      // not a stage: pieces of the scan that ever gave up waiting on this ctx
        if (ms) *ms = 0.0;
        if (launches) *launches = c->qf_lost_total;
        return PFMI_OK;
    }
    auto it = c->kstats.find(name);
    if (ms) *ms = (it == c->kstats.end()) ? 0.0 : it->second.ms;
    if (launches) *launches = (it == c->kstats.end()) ? 0 : it->second.launches;
    return PFMI_OK;
}

// ---- inputs ------------------------------------------------------------------------------------------
int32_t pfmi_set_target(pfmi_ctx *c, const pfmi_target *t) {
    PF_CTX_MUT(c);
    PF_CHECK(t != nullptr, PFMI_ERR_ARG, "null target");
    PF_CHECK(t->d > 0, PFMI_ERR_ARG, "target dimension must be positive");
    TargetDev &T = c->target;
    T.kind = t->kind; T.d = t->d; T.r = 0; T.rpad = 0; T.offset = 0.0; T.fn = nullptr; T.dev_fn = nullptr; T.user = nullptr;
    if (t->kind == PFMI_TARGET_GAUSS) {
        PF_CHECK(t->mean && t->a, PFMI_ERR_ARG, "GAUSS target needs mean and a");
        PF_CHECK(t->r >= 0 && t->r <= 16, PFMI_ERR_UNSUPPORTED, "GAUSS target rank %d > 16 unsupported", t->r);
        PF_CHECK(t->r == 0 || (t->Wd && t->G), PFMI_ERR_ARG, "GAUSS target with r > 0 needs Wd and G");
        const int d = t->d, r = t->r, rpad = (r == 0) ? 0 : (r <= 8 ? 8 : 16);
        T.r = r; T.rpad = rpad; T.offset = t->offset;
        PF_TRY(T.mean.ensure(sizeof(double) * d));
        PF_TRY(T.a.ensure(sizeof(double) * d));
        PF_TRY(h2d(c, T.mean.p, t->mean, sizeof(double) * d));
        PF_TRY(h2d(c, T.a.p, t->a, sizeof(double) * d));
        if (r > 0) {
            std::vector<double> wd((size_t)d * rpad, 0.0), g((size_t)rpad * rpad, 0.0);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < r; ++j) wd[(size_t)i * rpad + j] = t->Wd[i + (size_t)d * j];
            for (int j = 0; j < r; ++j)
                for (int l = 0; l <= j; ++l) g[(size_t)j * rpad + l] = t->G[j + (size_t)r * l];
            PF_TRY(T.wd.ensure(sizeof(double) * wd.size()));
            PF_TRY(T.g.ensure(sizeof(double) * g.size()));
            PF_TRY(h2d(c, T.wd.p, wd.data(), sizeof(double) * wd.size()));
            PF_TRY(h2d(c, T.g.p, g.data(), sizeof(double) * g.size()));
            const size_t rows16 = ((size_t)d + 15) / 16 * 16;
            std::vector<double> w16(rows16 * 16, 0.0);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < r; ++j) w16[(size_t)i * 16 + j] = t->Wd[i + (size_t)d * j];
            PF_TRY(T.wd16.ensure(sizeof(double) * w16.size()));
            PF_TRY(h2d(c, T.wd16.p, w16.data(), sizeof(double) * w16.size()));
        }
    } else if (t->kind == PFMI_TARGET_FUNNEL) {
        /* no parameters */
    } else if (t->kind == PFMI_TARGET_HOST_CALLBACK) {
        PF_CHECK(t->fn != nullptr, PFMI_ERR_ARG, "HOST_CALLBACK target needs fn");
        T.fn = t->fn; T.user = t->user;
    } else if (t->kind == PFMI_TARGET_DEVICE_CALLBACK) {
        PF_CHECK(t->dev_fn != nullptr, PFMI_ERR_ARG, "DEVICE_CALLBACK target needs dev_fn");
        T.dev_fn = t->dev_fn; T.user = t->user;
    } else {
        T.kind = -1;
        PF_CHECK(false, PFMI_ERR_ARG, "unknown target kind %d", t->kind);
    }
    return PFMI_OK;
}

int32_t pfmi_set_traces(pfmi_ctx *c, int32_t K, const int64_t *npoints, int32_t d, const double *theta,
                        const double *grad) {
    PF_CTX_MUT(c);
    if (c->sr.active) stream_abandon(c);
    PF_CHECK(K > 0 && d > 0 && npoints && theta && grad, PFMI_ERR_ARG, "set_traces: bad arguments");
    c->off.assign((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) {
        PF_CHECK(npoints[k] >= 1, PFMI_ERR_ARG, "path %d has no points", k);
        c->off[k + 1] = c->off[k] + npoints[k];
    }
    const int64_t P = c->off[K];
    PF_CHECK(P < (1ll << 31), PFMI_ERR_UNSUPPORTED, "too many trace points");
    c->path_of.resize((size_t)P);
    for (int k = 0; k < K; ++k)
        for (int64_t p = c->off[k]; p < c->off[k + 1]; ++p) c->path_of[(size_t)p] = k;
    c->K = K; c->d = d; c->P = P; c->virt = false;
    c->fitted = false; c->elbo_done = false; c->pooled = false; c->have_trace_lp = false;
    const size_t bytes = sizeof(double) * (size_t)P * d;
    PF_TRY(c->theta.ensure(bytes));
    PF_TRY(c->grad.ensure(bytes));
    PF_TRY(c->d_off.ensure(sizeof(int64_t) * (K + 1)));
    PF_TRY(c->d_path_of.ensure(sizeof(int32_t) * P));
    PF_TRY(h2d(c, c->theta.p, theta, bytes));
    PF_TRY(h2d(c, c->grad.p, grad, bytes));
    PF_TRY(h2d(c, c->d_off.p, c->off.data(), sizeof(int64_t) * (K + 1)));
    PF_TRY(h2d(c, c->d_path_of.p, c->path_of.data(), sizeof(int32_t) * P));
    return PFMI_OK;
}

// ---- device trajectory generation ------------------------------------------------------------------------
int32_t pfmi_optimize_batch_enqueue(pfmi_ctx *c, int32_t K, const double *x0, int32_t J, int32_t maxiters, double g_tol) {
    PF_CTX_MUT(c);
    if (c->sr.active) stream_abandon(c);
    const TargetDev &T = c->target;
    PF_CHECK(T.kind == PFMI_TARGET_GAUSS || T.kind == PFMI_TARGET_FUNNEL, PFMI_ERR_UNSUPPORTED,
             "optimize_batch: needs a built-in target (optimise callback targets on the host, then pfmi_set_traces)");
    PF_CHECK(K > 0 && x0 && maxiters >= 0, PFMI_ERR_ARG, "optimize_batch: bad arguments");
    PF_CHECK(J >= 1 && J <= 16, PFMI_ERR_UNSUPPORTED, "optimize_batch: history_length %d outside 1..16", J);
    const int d = T.d;
    const size_t cap = (size_t)maxiters + 1;
    PF_TRY(c->st_theta.ensure(sizeof(double) * K * cap * d));
    PF_TRY(c->st_grad.ensure(sizeof(double) * K * cap * d));
    PF_TRY(c->st_lp.ensure(sizeof(double) * K * cap));
    PF_TRY(c->st_npts.ensure(sizeof(int32_t) * K));
    PF_TRY(c->lb_x0.ensure(sizeof(double) * (size_t)K * d));
    PF_TRY(h2d(c, c->lb_x0.p, x0, sizeof(double) * (size_t)K * d));
    pf_kernel_begin(c);
    PF_TRY(pf_launch_lbfgs(c, K, J, maxiters, g_tol, c->lb_x0.as<double>()));
    pf_kernel_end(c, "optimize");
    c->opt_pending = true; c->opt_K = K; c->opt_cap = (int32_t)cap;
    c->fitted = false; c->elbo_done = false; c->pooled = false; c->have_trace_lp = false; c->P = 0;
    return PFMI_OK;
}

int32_t pfmi_optimize_batch_wait(pfmi_ctx *c, int64_t *npoints) {
    PF_CTX_MUT(c);
    PF_CHECK(c->opt_pending, PFMI_ERR_STATE, "optimize_batch_wait: no pfmi_optimize_batch_enqueue outstanding");
    PF_CHECK(npoints != nullptr, PFMI_ERR_ARG, "optimize_batch_wait: null npoints");
    c->opt_pending = false;
    const int K = c->opt_K, d = c->target.d;
    const size_t cap = (size_t)c->opt_cap;
    std::vector<int32_t> np32((size_t)K);
    PF_TRY(d2h(c, np32.data(), c->st_npts.p, sizeof(int32_t) * K));
    c->off.assign((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) {
        PF_CHECK(np32[(size_t)k] >= 1 && (size_t)np32[(size_t)k] <= cap, PFMI_ERR_NUMERIC, "optimize_batch: path %d produced %d points", k,
                 np32[(size_t)k]);
        npoints[k] = np32[(size_t)k];
        c->off[k + 1] = c->off[k] + np32[(size_t)k];
    }
    const int64_t P = c->off[K];
    PF_CHECK(P < (1ll << 31), PFMI_ERR_UNSUPPORTED, "too many trace points");
    c->path_of.resize((size_t)P);
    for (int k = 0; k < K; ++k)
        for (int64_t p = c->off[k]; p < c->off[k + 1]; ++p) c->path_of[(size_t)p] = k;
    c->K = K; c->d = d; c->P = P; c->virt = false;
    const size_t bytes = sizeof(double) * (size_t)P * d;
    PF_TRY(c->theta.ensure(bytes));
    PF_TRY(c->grad.ensure(bytes));
    PF_TRY(c->trace_lp.ensure(sizeof(double) * P));
    PF_TRY(c->d_off.ensure(sizeof(int64_t) * (K + 1)));
    PF_TRY(c->d_path_of.ensure(sizeof(int32_t) * P));
    PF_TRY(h2d(c, c->d_off.p, c->off.data(), sizeof(int64_t) * (K + 1)));
    PF_TRY(h2d(c, c->d_path_of.p, c->path_of.data(), sizeof(int32_t) * P));
    pf_kernel_begin(c);
    PF_TRY(pf_launch_trace_pack(c, (int64_t)cap));
    pf_kernel_end(c, "trace_pack");
    c->have_trace_lp = true;
    return PFMI_OK;
}

int32_t pfmi_optimize_batch(pfmi_ctx *c, int32_t K, const double *x0, int32_t J, int32_t maxiters, double g_tol,
                            int64_t *npoints) {
    PF_CHECK(npoints != nullptr, PFMI_ERR_ARG, "optimize_batch: bad arguments");
    PF_TRY(pfmi_optimize_batch_enqueue(c, K, x0, J, maxiters, g_tol));
    return pfmi_optimize_batch_wait(c, npoints);
}

int32_t pfmi_get_trace(pfmi_ctx *c, int32_t k, double *theta, double *logp, double *grad) {
    PF_CTX(c);
    PF_CHECK(c->P > 0 && k >= 0 && k < c->K, PFMI_ERR_ARG, "get_trace: bad path index");
    const int64_t p0 = c->off[(size_t)k], n = c->path_npts(k);
    const size_t bytes = sizeof(double) * (size_t)n * c->d;
    if (theta) PF_TRY(d2h(c, theta, c->th() + (size_t)p0 * c->d, bytes));
    if (grad) PF_TRY(d2h(c, grad, c->gr() + (size_t)p0 * c->d, bytes));
    if (logp) {
        PF_CHECK(c->have_trace_lp, PFMI_ERR_STATE, "get_trace: log densities exist only for pfmi_optimize_batch traces");
        PF_TRY(d2h(c, logp, c->tlp() + p0, sizeof(double) * n));
    }
    return PFMI_OK;
}

// ---- streaming pipeline: optimise, fit and scan as ONE dataflow the calling thread schedules -----------------------------------------
// (reference: src/multipath.jl:190-208 runs optimisation, fit and ELBO of a run back to back inside one task; here the K optimisations
// run as one persistent kernel and the fits / scans of the trace points they have ALREADY produced run on the other CUs meanwhile)
//
//   stream s_opt   pf_lbfgs_kernel: one workgroup per path; publishes its point count every PF_STREAM_PUB points into page-locked HOST memory
//   host           pfmi_stream_pump (called by pfmi_stream_wait until the pipeline is drained): reads the counts; once every path that is still
//                  running has recorded l1 points, the segment [l0, l1) of trace positions is complete and its work is launched:
//   stream s_fit   history walk of the segment (state carried from the previous segment) -> fits of the segment's points -> event
//   ctx stream / s_scan1 (alternating)   upload of the segment's work list -> wait for the fits -> ELBO scan of the segment's fits
//   ctx stream     at the end: waits for all of them -> mean / SE / argmax; then whatever the caller enqueues (pool, PSIS, resample)
//
// No kernel waits for another kernel: a consumer is only launched when its inputs are complete (the host saw the count; a kernel launch
// acquires what was released before it), so nothing depends on dispatch order, residency or the number of CUs.  The trace points stay in
// the optimiser's fixed-stride staging buffers (point l of path k = slot k * (maxiters + 1) + l): nothing is packed, no offset depends on
// another path's length.  Same kernels, same arithmetic as fit_batch + elbo_batch_enqueue on the packed trace: bit-identical results
// (tests/test_gpu_stream.py).  Four streams in all -- the default number of hardware queues, so none of them shares a queue.
#define PF_STREAM_PUB 16                   // points between two publications of the optimiser's progress (a power of two; hook PFMI_STREAM_PUB)
struct StreamSwap {                       // the launch helpers enqueue on c->stream: point it at a side stream for the scope
    pfmi_ctx *c; hipStream_t keep; int slot; bool seg;
    StreamSwap(pfmi_ctx *c_, hipStream_t s, int qf_slot = 0, bool seg_mode = false) : c(c_), keep(c_->stream), slot(c_->qf_slot), seg(c_->qf_seg_mode) {
        c->stream = s; c->qf_slot = qf_slot; c->qf_seg_mode = seg_mode;
    }
    ~StreamSwap() { c->stream = keep; c->qf_slot = slot; c->qf_seg_mode = seg; }
};
static int32_t qf_share_reserve(pfmi_ctx *c, int slot, size_t bytes) {
    DevBuf &b = c->qf_share_s[slot];
    if (b.cap >= bytes) return PFMI_OK;
    PF_TRY(b.ensure(bytes));
    PF_HIP(hipMemsetAsync(b.p, 0, b.cap, c->stream));
    c->qf_epoch_s[slot] = 0;
    return PFMI_OK;
}

// the runs' predrawn seed streams [K][cap]: host copy of the per-point seeds (the scan's work lists are cut from it) + ONE upload of the
// per-point seeds (slot k * cap + l holds stream value l - 1 of run k) and the raw streams (pfmi_pool_build_best: a failed run's draw seed)
static int32_t stream_set_seeds(pfmi_ctx *c, const uint64_t *seeds) {
    const size_t K = (size_t)c->K, cap = (size_t)c->vcap, Pz = K * cap;
    std::vector<uint64_t> stage(2 * Pz);
    for (size_t k = 0; k < K; ++k) {
        stage[k * cap] = 0;
        if (cap > 1) memcpy(stage.data() + k * cap + 1, seeds + k * cap, sizeof(uint64_t) * (cap - 1));
    }
    memcpy(stage.data() + Pz, seeds, sizeof(uint64_t) * Pz);
    PF_TRY(h2d(c, c->seeds.p, stage.data(), sizeof(uint64_t) * 2 * Pz));     // on the ctx stream: in front of everything that reads them
    stage.resize(Pz);
    c->sr.seeds_pt.swap(stage);
    c->sr.have_seeds = true;
    return PFMI_OK;
}

static int32_t stream_enqueue_impl(pfmi_ctx *c, int32_t K, const double *x0, int32_t J, int32_t maxiters, double g_tol, double eps, int64_t N,
                                   const uint64_t *seeds);
int32_t pfmi_stream_enqueue(pfmi_ctx *c, int32_t K, const double *x0, int32_t J, int32_t maxiters, double g_tol, double eps, int64_t N,
                            const uint64_t *seeds) {
    PF_CTX_MUT(c);
    const int32_t rc = stream_enqueue_impl(c, K, x0, J, maxiters, g_tol, eps, N, seeds);
    if (rc != PFMI_OK && rc != PFMI_ERR_ARG && rc != PFMI_ERR_UNSUPPORTED) {
        // failed half-way (an allocation, an upload, the producer's launch): nothing of the call survives -- the side streams are drained and the
        // context forgets the half-built layout, so that later calls see "no traces" instead of a layout whose producer never ran
        stream_abandon(c);
        c->virt = false; c->P = 0; c->K = 0; c->fitted = false; c->elbo_done = false; c->pooled = false; c->opt_pending = false;
    }
    return rc;
}
static int32_t stream_enqueue_impl(pfmi_ctx *c, int32_t K, const double *x0, int32_t J, int32_t maxiters, double g_tol, double eps, int64_t N,
                                   const uint64_t *seeds) {
    const TargetDev &T = c->target;
    PF_CHECK(T.kind == PFMI_TARGET_GAUSS || T.kind == PFMI_TARGET_FUNNEL, PFMI_ERR_UNSUPPORTED,
             "stream_enqueue: needs a built-in target (the optimiser runs on the device)");
    PF_CHECK(K > 0 && x0 && maxiters >= 0 && N >= 1, PFMI_ERR_ARG, "stream_enqueue: bad arguments");
    PF_CHECK(J >= 1 && J <= 16, PFMI_ERR_UNSUPPORTED, "stream_enqueue: history_length %d outside 1..16", J);
    // a previous streaming call that was never waited for (a host exception between enqueue and wait): drain and forget it, like
    // pfmi_set_traces / pfmi_optimize_batch_enqueue do -- a persistent context must stay usable
    if (c->sr.active) stream_abandon(c);
    if (c->s_opt) PF_TRY(stream_sync(c));      // the page-locked progress words and work lists are reused: the previous call's copies must have landed
    const int ncu = c->ncu > 0 ? c->ncu : 256;
    const int d = T.d;
    PF_CHECK(d <= 16384, PFMI_ERR_UNSUPPORTED, "stream_enqueue: the device optimiser takes d <= 16384");
    const int64_t cap = (int64_t)maxiters + 1, P = (int64_t)K * cap;
    PF_CHECK(P < (1ll << 31), PFMI_ERR_UNSUPPORTED, "stream_enqueue: too many trace slots");
    int kpad = 0;
    for (int o : {4, 8, 12, 16, 20, 32}) if (2 * J <= o) { kpad = o; break; }
    {   // the fixed-stride layout sizes every per-point buffer for the LONGEST possible path: refuse what the device cannot hold
        size_t free_b = 0, total_b = 0;
        const double need = 8.0 * (double)P * ((double)d * (kpad + 5) + 2.0 * (double)N + 4.0 * kpad * kpad);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            double have = (double)free_b;
            for (const DevBuf *b : {&c->vh, &c->alpha_all, &c->sqrt_alpha, &c->mu, &c->logp, &c->logq, &c->st_theta, &c->st_grad}) have += (double)b->cap;
            PF_CHECK(need < 0.8 * have, PFMI_ERR_UNSUPPORTED, "stream_enqueue: the fixed-stride layout needs %.1f GB for %d paths x %lld slots "
                     "(%.1f GB available): use the packed route", need / 1e9, K, (long long)cap, have / 1e9);
        }
    }
    // ---- every allocation BEFORE the first launch (a hipFree behind a grown buffer would synchronise the device in mid-flight)
    const size_t Pz = (size_t)P, dz = (size_t)d, kk = (size_t)kpad * kpad;
    PF_TRY(c->st_theta.ensure(sizeof(double) * Pz * dz));
    PF_TRY(c->st_grad.ensure(sizeof(double) * Pz * dz));
    PF_TRY(c->st_lp.ensure(sizeof(double) * Pz));
    PF_TRY(c->st_npts.ensure(sizeof(int32_t) * K));
    PF_TRY(c->lb_x0.ensure(sizeof(double) * (size_t)K * dz));
    PF_TRY(c->hs_ial.ensure(sizeof(double) * (size_t)K * dz));
    PF_TRY(c->hs_nacc.ensure(sizeof(int32_t) * K));
    PF_TRY(c->alpha_all.ensure(sizeof(double) * Pz * dz));
    PF_TRY(c->hist_len.ensure(sizeof(int32_t) * Pz));
    PF_TRY(c->hist_src.ensure(sizeof(int32_t) * Pz * J));
    PF_TRY(c->hist_acc.ensure(sizeof(int32_t) * Pz));
    PF_TRY(c->n_rej.ensure(sizeof(int32_t) * K));
    PF_TRY(c->vh.ensure(sizeof(double) * Pz * dz * kpad));
    PF_TRY(c->tmat.ensure(sizeof(double) * Pz * kk));
    PF_TRY(c->vchol.ensure(sizeof(double) * Pz * kk));
    PF_TRY(c->rq.ensure(sizeof(double) * Pz * kk));
    PF_TRY(c->dmat.ensure(sizeof(double) * Pz * kk));
    PF_TRY(c->sqrt_alpha.ensure(sizeof(double) * Pz * dz));
    PF_TRY(c->mu.ensure(sizeof(double) * Pz * dz));
    PF_TRY(c->logdet.ensure(sizeof(double) * Pz));
    PF_TRY(c->status.ensure(sizeof(int32_t) * Pz));
    PF_TRY(c->logp.ensure(sizeof(double) * Pz * N));
    PF_TRY(c->logq.ensure(sizeof(double) * Pz * N));
    PF_TRY(c->elbo.ensure(sizeof(double) * Pz));
    PF_TRY(c->se.ensure(sizeof(double) * Pz));
    PF_TRY(c->best_iter.ensure(sizeof(int64_t) * K));
    PF_TRY(c->d_off.ensure(sizeof(int64_t) * (K + 1)));
    PF_TRY(c->d_path_of.ensure(sizeof(int32_t) * Pz));
    for (int q = 0; q < 2; ++q) PF_TRY(qf_share_reserve(c, q, (size_t)ncu * 1700 * sizeof(double) + ((size_t)ncu + 1) * sizeof(unsigned) + (64 << 10)));
    if (!c->s_opt) {
        int lo = 0, hi = 0;
        PF_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));              // (hi is the numerically smaller = more urgent one)
        PF_HIP(hipStreamCreateWithPriority(&c->s_opt, hipStreamNonBlocking, hi));     // the producer and the fits go in front of queued scan workgroups
        PF_HIP(hipStreamCreateWithPriority(&c->s_fit, hipStreamNonBlocking, hi));
        PF_HIP(hipStreamCreateWithFlags(&c->s_scan1, hipStreamNonBlocking));
        for (hipEvent_t *e : {&c->sg_fit, &c->sg_opt, &c->sg_scan[0], &c->sg_scan[1], &c->sg_start}) PF_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (c->h_prog_cap < 2 * K) {
        if (c->h_prog) (void)hipHostFree(c->h_prog);
        c->h_prog = nullptr; c->h_prog_cap = 0;
        PF_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_prog), sizeof(int32_t) * 2 * (size_t)K, hipHostMallocCoherent | hipHostMallocMapped));
        c->h_prog_cap = 2 * K;
    }
    const int64_t nf = (int64_t)K * (cap - 1);
    const size_t list_bytes = (size_t)(nf > 0 ? nf : 1) * (sizeof(uint64_t) + sizeof(int32_t));
    if (c->h_list_cap < list_bytes) {
        if (c->h_list) (void)hipHostFree(c->h_list);
        c->h_list = nullptr; c->h_list_cap = 0;
        PF_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_list), list_bytes, hipHostMallocDefault));
        c->h_list_cap = list_bytes;
    }
    // ---- host-side layout: slot offsets
    const bool same_layout = c->virt && c->K == K && c->vcap == cap && c->d == d && (int64_t)c->path_of.size() == P;
    c->off.assign((size_t)K + 1, 0);
    for (int k = 0; k <= K; ++k) c->off[(size_t)k] = (int64_t)k * cap;
    if (!same_layout) {
        c->path_of.resize(Pz);
        for (int k = 0; k < K; ++k)
            for (int64_t l = 0; l < cap; ++l) c->path_of[(size_t)(k * cap + l)] = k;
    }
    c->K = K; c->d = d; c->P = P; c->J = J; c->kpad = kpad; c->N_e = N; c->virt = true; c->vcap = cap;
    c->npts_h.assign((size_t)K, 0);
    c->fitted = false; c->elbo_done = false; c->pooled = false; c->have_trace_lp = false; c->opt_pending = false;
    // ONE upload: the per-point seeds (slot k * cap + l holds stream value l - 1 of run k), then the raw streams; behind them the room of the
    // scan's work lists (filled segment by segment from h_list)
    const size_t off_tab = sizeof(uint64_t) * Pz, off_ls = off_tab + sizeof(uint64_t) * Pz;
    const size_t off_li = off_ls + sizeof(uint64_t) * (size_t)(nf > 0 ? nf : 1);
    const size_t all_bytes = off_li + sizeof(int32_t) * (size_t)(nf > 0 ? nf : 1);
    PF_TRY(c->seeds.ensure(all_bytes + 16));
    c->d_stream_tab = reinterpret_cast<const uint64_t *>(c->seeds.as<char>() + off_tab);
    PF_TRY(h2d(c, c->lb_x0.p, x0, sizeof(double) * (size_t)K * dz));
    PF_TRY(h2d(c, c->d_off.p, c->off.data(), sizeof(int64_t) * (K + 1)));
    if (!same_layout) PF_TRY(h2d(c, c->d_path_of.p, c->path_of.data(), sizeof(int32_t) * Pz));
    // slots no path reaches: PFMI_FIT_ABSENT, j_eff 0, NaN ELBO (0xFF.. is a NaN) -- nothing is ever launched for them
    PF_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->status.p), PFMI_FIT_ABSENT, Pz, c->stream));
    PF_HIP(hipMemsetAsync(c->hist_len.p, 0, sizeof(int32_t) * Pz, c->stream));
    PF_HIP(hipMemsetAsync(c->elbo.p, 0xFF, sizeof(double) * Pz, c->stream));
    PF_HIP(hipMemsetAsync(c->se.p, 0xFF, sizeof(double) * Pz, c->stream));
    PF_HIP(hipMemsetAsync(c->st_npts.p, 0, sizeof(int32_t) * K, c->stream));
    PF_HIP(hipMemsetAsync(c->hist_src.p, 0, sizeof(int32_t) * Pz * J, c->stream));
    for (int i = 0; i < 2 * K; ++i) __atomic_store_n(c->h_prog + i, 0, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    pfmi_ctx::StreamRun &R = c->sr;
    {
        std::vector<uint64_t> keep;
        keep.swap(R.seeds_pt);
        R = pfmi_ctx::StreamRun();
        R.seeds_pt.swap(keep);
    }
    R.K = K; R.J = J; R.cap = (int)cap; R.N = N; R.eps = eps;
    R.d_lseeds = reinterpret_cast<uint64_t *>(c->seeds.as<char>() + off_ls);
    R.d_list = reinterpret_cast<int32_t *>(c->seeds.as<char>() + off_li);
    // a segment should fill at least half the CUs with one workgroup per fit
    R.pub = PF_STREAM_PUB;
    // when are the segments cut?  Few paths (one publication step = at most one round of CUs): at FIXED, geometrically growing positions, as soon
    // as a segment's inputs are complete.  Many paths: whenever a scan stream is free, with everything that has arrived meanwhile
    // (profiles/r05_experiments.md section 4: 8 / 16 paths 3.94 / 6.84 against 4.01 / 6.93 ms, and far less sensitive to what else
    // shares the hardware queues; 32 / 64 paths 12.66 / 24.29 against 12.46 / 24.03 ms)
    R.policy = (K * PF_STREAM_PUB <= ncu) ? 1 : 0;
    { const char *po = pf_debug_get("PFMI_STREAM_POLICY"); if (po) R.policy = atoi(po); }
    { const char *fe = pf_debug_get("PFMI_STREAM_FIT_EAGER"); R.fit_eager = fe ? atoi(fe) : 1; }
    { const char *pb = pf_debug_get("PFMI_STREAM_PUB"); if (pb && (atoi(pb) == 4 || atoi(pb) == 8 || atoi(pb) == 32)) R.pub = atoi(pb); }
    R.minlen = ((ncu / 2 + K - 1) / K + R.pub - 1) / R.pub * R.pub;
    if (R.minlen < R.pub) R.minlen = R.pub;
    { const char *ml = pf_debug_get("PFMI_STREAM_MINLEN"); if (ml && atoi(ml) > 0) R.minlen = (atoi(ml) + R.pub - 1) / R.pub * R.pub; }
    if (seeds) PF_TRY(stream_set_seeds(c, seeds));
    c->stream_pending = true;
    PF_HIP(hipEventRecord(c->sg_start, c->stream));
    for (hipStream_t s : {c->s_opt, c->s_fit, c->s_scan1}) PF_HIP(hipStreamWaitEvent(s, c->sg_start, 0));
    {   // ---- the producer
        StreamSwap sw(c, c->s_opt);
        pf_kernel_begin(c);
        PF_TRY(pf_launch_lbfgs(c, K, J, maxiters, g_tol, c->lb_x0.as<double>(), R.pub - 1, c->h_prog));
        pf_kernel_end(c, "optimize");
        PF_HIP(hipEventRecord(c->sg_opt, c->s_opt));
    }
    R.active = true;
    R.t_progress = R.t_start = std::chrono::steady_clock::now();
    return PFMI_OK;
}

// Give up an outstanding streaming call (a host that failed between pfmi_stream_enqueue and pfmi_stream_wait): what is in flight is drained, the
// half-made results are forgotten, the context is usable again.  No call outstanding: no-op.
int32_t pfmi_stream_cancel(pfmi_ctx *c) {
    PF_CTX(c);
    if (!c->sr.active && !c->stream_pending) return PFMI_OK;
    stream_abandon(c);
    (void)hipStreamSynchronize(c->stream);
    c->fitted = false; c->elbo_done = false; c->elbo_pending = false; c->pooled = false;
    return PFMI_OK;
}

int32_t pfmi_stream_seeds(pfmi_ctx *c, const uint64_t *seeds) {
    PF_CTX(c);
    PF_CHECK(c->sr.active && !c->sr.have_seeds, PFMI_ERR_STATE, "stream_seeds: no pfmi_stream_enqueue(..., seeds = NULL) outstanding");
    PF_CHECK(seeds != nullptr, PFMI_ERR_ARG, "stream_seeds: null seeds");
    return stream_set_seeds(c, seeds);
}

// One scheduling pass: launches the next segment when its inputs are complete (and a scan stream is free), finishes the call when the last
// one is out.  *finished = 1: everything is enqueued (pfmi_stream_wait then returns at once).  Never blocks.
static int32_t stream_pump_pass(pfmi_ctx *c, int32_t *finished);
int32_t pfmi_stream_pump(pfmi_ctx *c, int32_t *finished) {
    PF_CTX_MUT(c);
    const auto tp0 = std::chrono::steady_clock::now();
    const int seg0 = c->sr.l_next;
    const bool was_active = c->sr.active;
    const int32_t rc = stream_pump_pass(c, finished);
    if (was_active && (c->sr.l_next != seg0 || !c->sr.active)) {     // this pass launched a segment (or the reduction): the scheduler's real cost
        c->sr.host_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
        c->sr.host_n += 1;
        if (!c->sr.active && c->profile) {                            // "stream_host_schedule": host milliseconds per call spent launching segments
            KernelStat &ks = c->kstats["stream_host_schedule"];
            ks.ms += c->sr.host_s * 1e3; ks.launches += c->sr.host_n;
        }
    }
    if (rc != PFMI_OK && c->sr.active) stream_abandon(c);          // a failed pass ends the call: what is in flight is drained, the context is usable again
    return rc;
}
static int32_t stream_pump_pass(pfmi_ctx *c, int32_t *finished) {
    pfmi_ctx::StreamRun &R = c->sr;
    if (finished) *finished = R.active ? 0 : 1;
    if (!R.active) return PFMI_OK;
    const int K = R.K;
    bool all_done = true;
    int avail = INT_MAX, lmax = 0;
    for (int k = 0; k < K; ++k) {
        const int dn = __atomic_load_n(c->h_prog + K + k, __ATOMIC_ACQUIRE);         // flag first: a raised flag means the count is final
        const int n = __atomic_load_n(c->h_prog + k, __ATOMIC_ACQUIRE);
        c->npts_h[(size_t)k] = dn ? n : -n;                                           // (negative: still running, at least that many points)
        if (!dn) { all_done = false; if (n < avail) avail = n; }
        if (n > lmax) lmax = n;
    }
    if (!R.have_seeds) return PFMI_OK;                                              // pfmi_stream_seeds has not been called yet
    const auto now = std::chrono::steady_clock::now();
    if (!all_done) {
        // Watchdog.  The producer is a plain kernel: it runs, or its stream carries an error.  An ERROR ends the call at once; no
        // progress alone ends it only after PFMI_STREAM_TIMEOUT_S seconds (default 60: a host that pumps rarely or a GPU shared with another
        // process must not trip it), and never with a device-wide synchronise -- that would block for ever behind a hung kernel and stall
        // every other context of the device.  The context's own side streams are drained by the next entry point that reuses its memory.
        const hipError_t q = hipStreamQuery(c->s_opt);
        if (q != hipSuccess && q != hipErrorNotReady) {
            R.active = false; c->stream_pending = false;
            PF_CHECK(false, PFMI_ERR_HIP, "stream_pump: the optimiser's stream reports %s", hipGetErrorString(q));
        }
        if (avail != R.last_min) { R.last_min = avail; R.t_progress = now; }
        else {
            double limit = 60.0;
            if (const char *tl = pf_debug_get("PFMI_STREAM_TIMEOUT_S")) { const double v = atof(tl); if (v > 0) limit = v; }
            if (std::chrono::duration<double>(now - R.t_progress).count() > limit) {
                R.active = false; c->stream_pending = false;
                PF_CHECK(false, PFMI_ERR_HIP, "stream_pump: the optimiser made no progress for %.0f s (stream state: %s)", limit, hipGetErrorString(q));
            }
        }
    }
    int l1 = all_done ? lmax : avail / R.pub * R.pub;
    if (l1 > R.cap) l1 = R.cap;
    const int l0 = R.l_next;
    // at most one scan launch queued per scan stream: the next segment then takes everything that has arrived meanwhile
    int q_free = -1;
    for (int q = 0; q < 2 && q_free < 0; ++q) {
        const int qq = (R.nseg + q) & 1;
        if (!R.scan_used[qq] || hipEventQuery(c->sg_scan[qq]) == hipSuccess) q_free = qq;
    }
    if (R.policy == 1) {
        // fixed boundaries (geometric): a segment goes out as soon as its positions are complete, whatever the scan streams are doing
        static const int step[] = {1, 1, 2, 2, 2, 4, 4, 8, 8, 16};               // in publication steps: 16, 32, 64, 96, 128, 192, 256, 384, ... at 16 points
        int bnd = 0, i = 0;
        while (bnd <= l0) { bnd += R.pub * step[i < 10 ? i : 9]; ++i; }
        if (!all_done) {
            if (l1 < bnd) return PFMI_OK;
            l1 = bnd;
        }
        q_free = R.nseg & 1;
    } else {
        if (!all_done && (l1 - l0 < R.minlen || q_free < 0)) return PFMI_OK;
        if (R.policy == 2 && !all_done) {
            // not behind a much larger launch that has just started: a small segment's fits would wait a round for CUs and its scan would
            // block its stream for the work that arrives meanwhile
            const int other = q_free ^ 1;
            if (R.scan_used[other] && hipEventQuery(c->sg_scan[other]) != hipSuccess && (int64_t)K * (l1 - l0) * 2 < R.scan_ns[other]) return PFMI_OK;
        }
        if (q_free < 0) q_free = R.nseg & 1;
    }
    if (l1 > l0) {
        // ---- the walk and the fits: of EVERYTHING that has arrived, not only of this segment.  Once the chip is full of scan workgroups
        //      (0.5 ms each, not preemptible) a fit launch waits for the next round boundary whatever its size; one launch that covers all
        //      the positions available then spares the later segments that wait (measured: 8 paths 3.99 -> 3.97 ms, 64 paths 24.23 -> 24.10 ms end to end;
        //      PFMI_STREAM_FIT_EAGER = 0 restores per-segment fits)
        if (R.l_fit < l1) {
            int lf = all_done ? lmax : avail / R.pub * R.pub;
            if (lf > R.cap) lf = R.cap;
            if (lf < l1) lf = l1;
            if (R.fit_eager == 0) lf = l1;
            StreamSwap sw(c, c->s_fit);
            const HistSeg sg{c->st_npts.as<int32_t>(), R.l_fit, lf, c->hs_ial.as<double>(), c->hs_nacc.as<int32_t>(), c->hinit};
            PF_TRY(pf_launch_history(c, R.eps, &sg));
            PF_TRY(pf_launch_fit(c, R.l_fit, lf - R.l_fit));
            PF_HIP(hipEventRecord(c->sg_fit, c->s_fit));
            R.l_fit = lf;
        }
        // ---- its scan: the fits that exist, position-major
        uint64_t *hs = reinterpret_cast<uint64_t *>(c->h_list);
        int32_t *hl = reinterpret_cast<int32_t *>(c->h_list + sizeof(uint64_t) * (size_t)((int64_t)K * (R.cap - 1) > 0 ? (int64_t)K * (R.cap - 1) : 1));
        int64_t t = R.s0;
        for (int l = l0 > 1 ? l0 : 1; l < l1; ++l)
            for (int k = 0; k < K; ++k) {
                const int n = c->npts_h[(size_t)k];
                if (n >= 0 && l >= n) continue;                                       // the path ended before this position
                const int64_t pp = (int64_t)k * R.cap + l;
                hl[t] = (int32_t)pp; hs[t] = R.seeds_pt[(size_t)pp]; ++t;
            }
        const int64_t ns = t - R.s0;
        if (ns > 0) {
            const int qq = q_free;
            hipStream_t ss = qq == 0 ? c->stream : c->s_scan1;
            StreamSwap sw(c, ss, qq, !all_done);
            PF_HIP(hipMemcpyAsync(R.d_lseeds + R.s0, hs + R.s0, sizeof(uint64_t) * (size_t)ns, hipMemcpyHostToDevice, ss));
            PF_HIP(hipMemcpyAsync(R.d_list + R.s0, hl + R.s0, sizeof(int32_t) * (size_t)ns, hipMemcpyHostToDevice, ss));
            PF_HIP(hipStreamWaitEvent(ss, c->sg_fit, 0));
            PF_TRY(pf_launch_elbo_draws(c, R.d_list + R.s0, R.d_lseeds + R.s0, ns, 0, R.N, nullptr, 0, nullptr, 0, c->logp.as<double>(),
                                        c->logq.as<double>(), R.N, true, true));
            PF_HIP(hipEventRecord(c->sg_scan[qq], ss));
            R.scan_used[qq] = true;
            R.scan_ns[qq] = ns;
            ++R.nseg;
        }
        R.trace.insert(R.trace.end(), {std::chrono::duration<double>(now - R.t_start).count() * 1e6, (double)l0, (double)l1, (double)ns, (double)q_free});
        R.s0 = t;
        R.l_next = l1;
    }
    if (!all_done) return PFMI_OK;
    // ---- drained: join the side streams, reduce
    for (int k = 0; k < K; ++k) {
        PF_CHECK(c->npts_h[(size_t)k] >= 1 && c->npts_h[(size_t)k] <= R.cap, PFMI_ERR_NUMERIC, "stream: path %d produced %d points", k, c->npts_h[(size_t)k]);
    }
    PF_HIP(hipEventRecord(c->sg_fit, c->s_fit));
    PF_HIP(hipStreamWaitEvent(c->stream, c->sg_opt, 0));
    PF_HIP(hipStreamWaitEvent(c->stream, c->sg_fit, 0));
    if (R.scan_used[1]) PF_HIP(hipStreamWaitEvent(c->stream, c->sg_scan[1], 0));
    R.active = false;
    c->stream_pending = false;
    c->fitted = true;
    if (const char *tr = pf_debug_get("PFMI_STREAM_TRACE")) {         // test / tuning hook: what was launched when
        if (tr[0] == '1') {
            fprintf(stderr, "stream: %d segments, drained at %.0f us:", R.nseg, std::chrono::duration<double>(std::chrono::steady_clock::now() - R.t_start).count() * 1e6);
            for (size_t i = 0; i + 4 < R.trace.size() + 1; i += 5)
                fprintf(stderr, "  [%.0f us: %d..%d, %d fits, q%d]", R.trace[i], (int)R.trace[i + 1], (int)R.trace[i + 2], (int)R.trace[i + 3], (int)R.trace[i + 4]);
            fprintf(stderr, "\n");
        }
    }
    PF_TRY(pf_launch_elbo_reduce(c));
    c->elbo_done = true; c->elbo_pending = true; c->have_trace_lp = true;
    if (finished) *finished = 1;
    return PFMI_OK;
}

int32_t pfmi_stream_wait(pfmi_ctx *c, int64_t *npoints) {
    PF_CTX(c);
    PF_CHECK(c->virt && (c->sr.active || c->elbo_done), PFMI_ERR_STATE, "stream_wait: no pfmi_stream_enqueue outstanding");
    PF_CHECK(npoints != nullptr, PFMI_ERR_ARG, "stream_wait: null npoints");
    PF_CHECK(!c->sr.active || c->sr.have_seeds, PFMI_ERR_STATE, "stream_wait: pfmi_stream_enqueue was given no seeds: call pfmi_stream_seeds first");
    int32_t fin = 0;
    while (true) {
        PF_TRY(pfmi_stream_pump(c, &fin));
        if (fin) break;
        for (int i = 0; i < 64; ++i) __builtin_ia32_pause();
    }
    for (int k = 0; k < c->K; ++k) npoints[k] = c->npts_h[(size_t)k];
    return PFMI_OK;
}

// ---- fit ----------------------------------------------------------------------------------------------
static int32_t fit_batch_impl(pfmi_ctx *c, int32_t J, double eps);
int32_t pfmi_set_hinit(pfmi_ctx *c, int32_t hinit) {
    PF_CTX(c);
    PF_CHECK(hinit == PFMI_HINIT_GILBERT || hinit == PFMI_HINIT_SCALAR_YS_OVER_YY, PFMI_ERR_ARG, "set_hinit: unknown Hinit %d", hinit);
    c->hinit = hinit;
    return PFMI_OK;
}
int32_t pfmi_fit_batch(pfmi_ctx *c, int32_t J, double eps) {
    PF_CTX_MUT(c);
    return fit_batch_impl(c, J, eps);
}
int32_t pfmi_fit_batch_ex(pfmi_ctx *c, int32_t J, double eps, int32_t hinit) {
    PF_CTX_MUT(c);
    PF_CHECK(hinit == PFMI_HINIT_GILBERT || hinit == PFMI_HINIT_SCALAR_YS_OVER_YY, PFMI_ERR_ARG, "fit_batch_ex: unknown Hinit %d", hinit);
    const int keep = c->hinit;
    c->hinit = hinit;                                       // for THIS call (pfmi_set_hinit is the persistent setting)
    const int32_t rc = fit_batch_impl(c, J, eps);
    c->hinit = keep;
    return rc;
}
static int32_t fit_batch_impl(pfmi_ctx *c, int32_t J, double eps) {
    PF_CHECK(c->P > 0, PFMI_ERR_STATE, "fit_batch: no traces set");
    PF_CHECK(J >= 1, PFMI_ERR_ARG, "history_length must be >= 1");
    const int m = 2 * J;
    int kpad = 0;
    // (column padding 64 = history_length 17 .. 32: the slow-but-correct route -- memory-resident fit kernel with its small matrices in
    //  global memory, lane-per-draw kernel for every draw / scan; the tuned kernels stop at 32 columns)
    const int opts[] = {4, 8, 12, 16, 20, 32, 64};
    for (int o : opts) if (m <= o) { kpad = o; break; }
    PF_CHECK(kpad != 0, PFMI_ERR_UNSUPPORTED, "history_length %d > 32 unsupported", J);
    c->J = J; c->kpad = kpad;
    const size_t P = (size_t)c->P, d = (size_t)c->d, kk = (size_t)kpad * kpad;
    PF_TRY(c->alpha_all.ensure(sizeof(double) * P * d));
    PF_TRY(c->hist_len.ensure(sizeof(int32_t) * P));
    PF_TRY(c->hist_src.ensure(sizeof(int32_t) * P * J));
    PF_TRY(c->hist_acc.ensure(sizeof(int32_t) * P));
    PF_TRY(c->n_rej.ensure(sizeof(int32_t) * c->K));
    PF_TRY(c->vh.ensure(sizeof(double) * P * d * kpad));
    PF_TRY(c->tmat.ensure(sizeof(double) * P * kk));
    PF_TRY(c->vchol.ensure(sizeof(double) * P * kk));
    PF_TRY(c->rq.ensure(sizeof(double) * P * kk));
    PF_TRY(c->dmat.ensure(sizeof(double) * P * kk));
    PF_TRY(c->sqrt_alpha.ensure(sizeof(double) * P * d));
    PF_TRY(c->mu.ensure(sizeof(double) * P * d));
    PF_TRY(c->logdet.ensure(sizeof(double) * P));
    PF_TRY(c->status.ensure(sizeof(int32_t) * P));
    PF_HIP(hipMemsetAsync(c->hist_src.p, 0, sizeof(int32_t) * P * J, c->stream));
    if (c->virt) {                                            // streaming layout: every path's slots, the absent ones marked as such
        const HistSeg sg{c->st_npts.as<int32_t>(), 0, INT_MAX, nullptr, nullptr, c->hinit};
        PF_TRY(pf_launch_history(c, eps, &sg));
        PF_TRY(pf_launch_fit(c, 0, (int)c->vcap));
    } else {
        PF_TRY(pf_launch_history(c, eps));
        PF_TRY(pf_launch_fit(c));
    }
    c->fitted = true; c->elbo_done = false; c->pooled = false;
    return PFMI_OK;
}

int32_t pfmi_get_fit_status(pfmi_ctx *c, int32_t *status, int32_t *j_eff, double *logdet, int64_t *n_rejected) {
    PF_CTX(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "get_fit_status: call pfmi_fit_batch first");
    auto r = std::make_shared<std::vector<int32_t>>((size_t)c->K);
    if (status) PF_TRY(d2h_async(c, status, c->status.p, sizeof(int32_t) * c->P));
    if (j_eff) PF_TRY(d2h_async(c, j_eff, c->hist_len.p, sizeof(int32_t) * c->P));
    if (logdet) PF_TRY(d2h_async(c, logdet, c->logdet.p, sizeof(double) * c->P));
    if (n_rejected) PF_TRY(d2h_async(c, r->data(), c->n_rej.p, sizeof(int32_t) * c->K));
    const int K = c->K;
    auto widen = [r, n_rejected, K]() -> int32_t {
        if (n_rejected)
            for (int k = 0; k < K; ++k) n_rejected[k] = (*r)[(size_t)k];
        return PFMI_OK;
    };
    if (c->defer) { c->post_sync.push_back(widen); return PFMI_OK; }
    PF_TRY(stream_sync(c));
    return widen();
}

int32_t pfmi_get_fit(pfmi_ctx *c, int64_t p, double *alpha, double *B, double *D, double *qr_factors, double *T,
                     double *V, double *mu, double *logdet) {
    PF_CTX(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "get_fit: call pfmi_fit_batch first");
    PF_CHECK(p >= 0 && p < c->P, PFMI_ERR_ARG, "get_fit: point %lld out of range", (long long)p);
    const int d = c->d, J = c->J, kp = c->kpad;
    int32_t j = 0;
    PF_TRY(d2h(c, &j, c->hist_len.as<int32_t>() + p, sizeof(int32_t)));
    const int m = 2 * j, k = d < m ? d : m;
    std::vector<double> al((size_t)d);
    PF_TRY(d2h(c, al.data(), c->alpha_all.as<double>() + (size_t)p * d, sizeof(double) * d));
    if (alpha) memcpy(alpha, al.data(), sizeof(double) * d);
    if (mu) PF_TRY(d2h(c, mu, c->mu.as<double>() + (size_t)p * d, sizeof(double) * d));
    if (logdet) PF_TRY(d2h(c, logdet, c->logdet.as<double>() + p, sizeof(double)));
    const size_t kk = (size_t)kp * kp;
    std::vector<double> small(kk);
    if (B && j > 0) {   // B = [alpha .* Y  S], columns oldest -> newest  (src/inverse_hessian.jl:105-118)
        std::vector<int32_t> src((size_t)J);
        PF_TRY(d2h(c, src.data(), c->hist_src.as<int32_t>() + (size_t)p * J, sizeof(int32_t) * J));
        const int64_t p0 = c->off[(size_t)c->path_of[(size_t)p]];
        std::vector<double> t0((size_t)d), t1((size_t)d), g0((size_t)d), g1((size_t)d);
        for (int cidx = 0; cidx < j; ++cidx) {
            const size_t q0 = (size_t)(p0 + src[(size_t)cidx]) * d, q1 = q0 + d;
            PF_TRY(d2h(c, t0.data(), c->th() + q0, sizeof(double) * d));
            PF_TRY(d2h(c, t1.data(), c->th() + q1, sizeof(double) * d));
            PF_TRY(d2h(c, g0.data(), c->gr() + q0, sizeof(double) * d));
            PF_TRY(d2h(c, g1.data(), c->gr() + q1, sizeof(double) * d));
            for (int i = 0; i < d; ++i) {
                B[i + (size_t)d * cidx] = al[(size_t)i] * (g0[(size_t)i] - g1[(size_t)i]);
                B[i + (size_t)d * (j + cidx)] = t1[(size_t)i] - t0[(size_t)i];
            }
        }
    }
    if (D && m > 0) {
        PF_TRY(d2h(c, small.data(), c->dmat.as<double>() + (size_t)p * kk, sizeof(double) * kk));
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) D[a + (size_t)m * b] = small[(size_t)a * kp + b];
    }
    if (T && k > 0) {
        PF_TRY(d2h(c, small.data(), c->tmat.as<double>() + (size_t)p * kk, sizeof(double) * kk));
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) T[a + (size_t)k * b] = small[(size_t)a * kp + b];
    }
    if (V && k > 0) {
        PF_TRY(d2h(c, small.data(), c->vchol.as<double>() + (size_t)p * kk, sizeof(double) * kk));
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) V[a + (size_t)k * b] = small[(size_t)a * kp + b];
    }
    if (qr_factors && m > 0) {
        std::vector<double> vh((size_t)d * kp);
        PF_TRY(d2h(c, vh.data(), c->vh.as<double>() + (size_t)p * d * kp, sizeof(double) * d * kp));
        PF_TRY(d2h(c, small.data(), c->rq.as<double>() + (size_t)p * kk, sizeof(double) * kk));
        for (int b = 0; b < m; ++b)
            for (int i = 0; i < d; ++i) {
                double v;
                if (i <= b && i < k) v = small[(size_t)i * kp + b];          // R (upper trapezoid)
                else if (b < k) v = vh[(size_t)i * kp + b];                  // Householder vector
                else v = 0.0;
                qr_factors[i + (size_t)d * b] = v;
            }
    }
    return PFMI_OK;
}

// ---- ELBO ---------------------------------------------------------------------------------------------
// The user's host closure on a staged block of `n` columns.  pfmi_set_callback_threads(ctx, t) with t > 1 -- the reference's `ntasks`
// (src/elbo.jl:3-6, src/resample.jl:85-92 through src/utils.jl:33-49; the caller asserts thread safety, src/multipath.jl:104-108) --
// cuts the block into t contiguous column ranges and calls `fn` on each from its own thread (the calling thread takes the first
// range): same columns, same per-column results, whatever t is.  Threads live for one block: a block is milliseconds of closure
// time, their creation tens of microseconds.
static void host_callback_block(pfmi_ctx *c, const double *X, int d, int64_t n, double *out) {
    const TargetDev &T = c->target;
    int t = c->cb_threads;
    if (t > n) t = (int)n;
    if (t <= 1) { T.fn(X, d, n, out, T.user); return; }
    const int64_t per = (n + t - 1) / t;
    std::vector<std::thread> th;
    th.reserve((size_t)t - 1);
    for (int i = 1; i < t; ++i) {
        const int64_t j0 = (int64_t)i * per, j1 = std::min<int64_t>(n, j0 + per);
        if (j0 >= j1) break;
        try {
            th.emplace_back([=, &T] { T.fn(X + (size_t)j0 * (size_t)d, d, j1 - j0, out + j0, T.user); });
        } catch (...) {                                        // no thread to be had (resource limits): this range runs on the calling thread
            T.fn(X + (size_t)j0 * (size_t)d, d, j1 - j0, out + j0, T.user);
        }
    }
    T.fn(X, d, std::min<int64_t>(n, per), out, T.user);
    for (std::thread &x : th) x.join();
}

int32_t pfmi_set_callback_threads(pfmi_ctx *c, int32_t nthreads) {
    PF_CTX(c);
    PF_CHECK(nthreads >= 1 && nthreads <= 1024, PFMI_ERR_ARG, "set_callback_threads: 1 <= nthreads <= 1024");
    c->cb_threads = nthreads;
    return PFMI_OK;
}

// evaluate the host-callback target on `n` columns stored at device pointer d_x; results to d_lp (pinned staging, 64 MB blocks)
static int32_t callback_logp(pfmi_ctx *c, const double *d_x, int64_t n, double *d_lp) {
    const int d = c->d;
    int64_t chunk = (int64_t)((64ll << 20) / (sizeof(double) * (size_t)d));
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    PF_TRY(ensure_pinned(c, sizeof(double) * (size_t)chunk * d, sizeof(double) * (size_t)chunk));
    for (int64_t s0 = 0; s0 < n; s0 += chunk) {
        const int64_t ns = (n - s0 < chunk) ? n - s0 : chunk;
        PF_TRY(d2h(c, c->pin_x[0], d_x + (size_t)s0 * d, sizeof(double) * (size_t)ns * d));
        host_callback_block(c, reinterpret_cast<const double *>(c->pin_x[0]), d, ns, reinterpret_cast<double *>(c->pin_lp[0]));
        PF_TRY(h2d(c, d_lp + s0, c->pin_lp[0], sizeof(double) * (size_t)ns));
    }
    return PFMI_OK;
}

// chunk of fits whose draws are materialised in HBM at a time for a DEVICE_CALLBACK target (PFMI_DEVCB_CHUNK_MB, default 2048 MB)
static int64_t devcb_chunk_fits(const pfmi_ctx *c, int64_t per_fit_doubles, int64_t nf, int64_t N) {
    // the draws of a block of fits live in one scratch buffer: 2 GB by default (config 3: 256 fits of 8 MB).  When a fit's draws are
    // large (config 5: 160 MB) such a block holds a dozen fits and the writer has to cut every fit into workgroups of a few draw groups,
    // each of which streams the whole factor block (857 GB/s written at one GPU's share of config 5); the block then grows until a
    // launch has 16 draw groups per workgroup on two workgroups per CU -- bounded by a quarter of the free device memory
    double mb = 2048.0;
    const char *e = pf_debug_get("PFMI_DEVCB_CHUNK_MB");
    if (e && atof(e) > 0) mb = atof(e);
    else {
        const int64_t ngroups = (N + 15) / 16, ncu = c->ncu > 0 ? c->ncu : 256;
        const int64_t split = ngroups / 16 > 0 ? ngroups / 16 : 1;              // workgroups a fit can be cut into
        const double want_mb = (double)((2 * ncu + split - 1) / split) * sizeof(double) * (double)per_fit_doubles / 1048576.0;
        size_t free_b = 0, total_b = 0;
        if (want_mb > mb && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double cap_mb = (double)free_b / 4.0 / 1048576.0;
            mb = want_mb < cap_mb ? want_mb : (cap_mb > mb ? cap_mb : mb);
        }
    }
    int64_t chunk = (int64_t)(mb * 1048576.0 / (sizeof(double) * (double)per_fit_doubles));
    if (chunk < 1) chunk = 1;
    if (chunk > nf) chunk = nf > 0 ? nf : 1;
    return chunk;
}

// DEVICE_CALLBACK: the user's kernel on `n` columns at d_x -> d_lp, on the ctx stream.  Failed fits leave their draws unwritten (the
// closure then sees stale memory); their log densities are NaN, like an exception in the reference: `npts` slots of `N` values each
// are patched by status afterwards.
static int32_t devcb_logp(pfmi_ctx *c, const double *d_x, int64_t n, double *d_lp, const int32_t *d_points, int64_t npts, int64_t N) {
    pf_kernel_begin(c);
    c->target.dev_fn(d_x, c->d, n, d_lp, (void *)c->stream, c->target.user);
    pf_kernel_end(c, "device_callback");
    PF_HIP(hipGetLastError());
    return pf_launch_nan_failed(c, npts, N, d_points, d_lp);
}

int32_t pfmi_elbo_batch_enqueue(pfmi_ctx *c, int64_t N, const uint64_t *seeds, const double *u_host) {
    PF_CTX_MUT(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "elbo_batch: call pfmi_fit_batch first");
    PF_CHECK(c->target.kind >= 0, PFMI_ERR_STATE, "elbo_batch: call pfmi_set_target first");
    PF_CHECK(c->target.d == c->d, PFMI_ERR_ARG, "target dimension %d != trace dimension %d", c->target.d, c->d);
    PF_CHECK(N >= 1 && seeds, PFMI_ERR_ARG, "elbo_batch: bad arguments");
    const int64_t P = c->P;
    const int d = c->d;
    c->N_e = N;
    PF_TRY(c->logp.ensure(sizeof(double) * P * N));
    PF_TRY(c->logq.ensure(sizeof(double) * P * N));
    PF_TRY(c->elbo.ensure(sizeof(double) * P));
    PF_TRY(c->se.ensure(sizeof(double) * P));
    PF_TRY(c->best_iter.ensure(sizeof(int64_t) * c->K));
    // ONE upload (round 4; three copies in the stream cost three 12-us hand-overs between the fit and the scan): the per-point seeds, then
    // the list of fits = every point that is not the first of its path (fit_distributions[2:end]) -- their seeds, then their indices
    int64_t nf = 0;
    for (int k = 0; k < c->K; ++k) nf += c->path_npts(k) - 1;
    const size_t off_ls = sizeof(uint64_t) * (size_t)P, off_li = off_ls + sizeof(uint64_t) * (size_t)(nf > 0 ? nf : 1);
    const size_t up_bytes = off_li + sizeof(int32_t) * (size_t)(nf > 0 ? nf : 1);
    std::vector<char> stage(up_bytes);
    memcpy(stage.data(), seeds, sizeof(uint64_t) * (size_t)P);
    {
        uint64_t *ls = reinterpret_cast<uint64_t *>(stage.data() + off_ls);
        int32_t *li = reinterpret_cast<int32_t *>(stage.data() + off_li);
        int64_t t = 0;
        for (int k = 0; k < c->K; ++k)
            for (int64_t p = c->off[k] + 1; p < c->off[k] + c->path_npts(k); ++p, ++t) { li[t] = (int32_t)p; ls[t] = seeds[p]; }
    }
    PF_TRY(c->seeds.ensure(up_bytes + 16));
    c->d_stream_tab = nullptr;                                  // (the streams of a streaming call lived in this buffer)
    PF_TRY(h2d(c, c->seeds.p, stage.data(), up_bytes));
    uint64_t *d_lseeds = reinterpret_cast<uint64_t *>(c->seeds.as<char>() + off_ls);
    int32_t *d_list = reinterpret_cast<int32_t *>(c->seeds.as<char>() + off_li);
    const double *d_u = nullptr;
    if (u_host) {
        PF_TRY(c->ubuf.ensure(sizeof(double) * (size_t)P * d * N));
        PF_TRY(h2d(c, c->ubuf.p, u_host, sizeof(double) * (size_t)P * d * N));
        d_u = c->ubuf.as<double>();
    }
    const int64_t ustride = (int64_t)d * N;
    if (c->target.kind == PFMI_TARGET_DEVICE_CALLBACK) {
        // device closure (the reference's general logp kept on the GPU): the draws of a block of fits are materialised in HBM
        // (8 d bytes written per draw), the user's kernel reads them on the same stream (8 d bytes read per draw) and its log
        // densities are scattered into the point-indexed table.  Nothing crosses PCIe, nothing synchronises.
        const int64_t per = (int64_t)d * N;
        int64_t chunk = devcb_chunk_fits(c, per, nf, N);
        // Round 6: the blocks alternate between PF_DCB_NB = 2 streams (two buffers), so that the closure of block i runs beside the writer of block
        // i + 1: the writer is bound by no single resource (SIMD issue ~70 %, LDS pipe ~50 %, HBM writes at half the link) and the closure
        // wants what it leaves idle, HBM reads -- each kernel fills the other's launch tails and memory pipe.  7.90 -> 6.87 ms for 1 403 fits x 1000
        // draws at d = 1000 (2.85 -> 3.27 TB/s moved); a third stream adds nothing.  Same kernels on the same data: the same bits.
        // PFMI_DEVCB_OVERLAP = 0 | 2 (debug hook): one stream (kernel timing) / two.
        int nb = nf > 1 ? PF_DCB_NB : 1;
        if (const char *ov = pf_debug_get("PFMI_DEVCB_OVERLAP")) { const int v = atoi(ov); nb = (v <= 0 || nf <= 1) ? 1 : (v < PF_DCB_NB ? (v < 2 ? 2 : v) : PF_DCB_NB); }
        if (nb > 1) {
            size_t free_b = 0, total_b = 0;
            double have = 0.0;
            for (int b = 0; b < PF_DCB_NB; ++b) have += (double)c->dcb_x[b].cap;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) have += (double)free_b;
            // full-size buffers only when memory is plentiful and there are blocks to alternate; else the same footprint cut into nb blocks
            if ((double)nb * (double)sizeof(double) * (double)chunk * (double)(per + N) > 0.5 * have || chunk * nb > nf) chunk = (chunk + nb - 1) / nb;
            if (chunk < 1) chunk = 1;
        }
        for (int b = 0; b < nb; ++b) {
            PF_TRY(c->dcb_x[b].ensure(sizeof(double) * (size_t)chunk * per));
            PF_TRY(c->dcb_lp[b].ensure(sizeof(double) * (size_t)chunk * N));
        }
        for (int b = 1; b < nb; ++b)
            if (!c->s_cb[b - 1]) PF_HIP(hipStreamCreateWithFlags(&c->s_cb[b - 1], hipStreamNonBlocking));
        for (int b = 0; b < nb; ++b)
            if (!c->sg_cb[b]) PF_HIP(hipEventCreateWithFlags(&c->sg_cb[b], hipEventDisableTiming));
        if (nb > 1) {                                                // the side streams start behind the uploads and the fits
            PF_HIP(hipEventRecord(c->sg_cb[0], c->stream));
            for (int b = 1; b < nb; ++b) PF_HIP(hipStreamWaitEvent(c->s_cb[b - 1], c->sg_cb[0], 0));
        }
        c->cb_bytes_dev = 0.0;
        int64_t blk = 0;
        for (int64_t s0 = 0; s0 < nf; s0 += chunk, ++blk) {
            const int64_t ns = (nf - s0 < chunk) ? nf - s0 : chunk;
            const int b = (int)(blk % nb);
            hipStream_t ss = b ? c->s_cb[b - 1] : c->stream;
            StreamSwap sw(c, ss);
            PF_TRY(pf_launch_elbo_draws(c, d_list + s0, d_lseeds + s0, ns, 0, N, d_u, ustride, c->dcb_x[b].as<double>(), per,
                                        c->logp.as<double>(), c->logq.as<double>(), N, false, true));
            pf_kernel_begin(c);
            c->target.dev_fn(c->dcb_x[b].as<double>(), d, ns * N, c->dcb_lp[b].as<double>(), (void *)ss, c->target.user);
            pf_kernel_end(c, "device_callback");
            PF_TRY(pf_launch_scatter_rows(c, ns, N, d_list + s0, c->dcb_lp[b].as<double>(), c->logp.as<double>()));
            c->cb_bytes_dev += (double)sizeof(double) * (double)ns * (double)per;
        }
        for (int b = 1; b < nb && b < blk; ++b) {                    // the reduction waits for the side streams' blocks
            PF_HIP(hipEventRecord(c->sg_cb[b], c->s_cb[b - 1]));
            PF_HIP(hipStreamWaitEvent(c->stream, c->sg_cb[b], 0));
        }
    } else if (c->target.kind != PFMI_TARGET_HOST_CALLBACK) {
        PF_TRY(pf_launch_elbo_draws(c, d_list, d_lseeds, nf, 0, N, d_u, ustride, nullptr, 0, c->logp.as<double>(),
                                    c->logq.as<double>(), N, true, true));
    } else {
        // host closure (the reference's general logp, src/elbo.jl:15): the draws of a block of fits are materialised on the device,
        // handed to the callback through PINNED staging and the log densities scattered back by one kernel.  Double buffered:
        // while the host evaluates block i, the device already generates and downloads block i + 1.
        const int64_t per = (int64_t)d * N;
        int64_t chunk = (int64_t)((64ll << 20) / (sizeof(double) * (size_t)per));
        if (chunk < 1) chunk = 1;
        if (chunk > nf) chunk = nf > 0 ? nf : 1;
        const size_t xb = sizeof(double) * (size_t)chunk * per, lb = sizeof(double) * (size_t)chunk * N;
        PF_TRY(ensure_pinned(c, xb, lb));
        for (int b = 0; b < 2; ++b) { PF_TRY(c->cb_x[b].ensure(xb)); PF_TRY(c->cb_lp[b].ensure(lb)); }
        c->cb_seconds = 0.0; c->cb_bytes_d2h = 0.0;
        const int64_t nblocks = (nf + chunk - 1) / chunk;
        auto enqueue = [&](int64_t i) -> int32_t {                  // generate block i and start its download
            const int b = (int)(i & 1);
            const int64_t s0 = i * chunk, ns = (nf - s0 < chunk) ? nf - s0 : chunk;
            PF_TRY(pf_launch_elbo_draws(c, d_list + s0, d_lseeds + s0, ns, 0, N, d_u, ustride, c->cb_x[b].as<double>(), per,
                                        c->logp.as<double>(), c->logq.as<double>(), N, false, true));
            PF_HIP(hipMemcpyAsync(c->pin_x[b], c->cb_x[b].p, sizeof(double) * (size_t)ns * per, hipMemcpyDeviceToHost, c->stream));
            PF_HIP(hipEventRecord(c->cb_ev[b], c->stream));
            return PFMI_OK;
        };
        if (nblocks > 0) PF_TRY(enqueue(0));
        for (int64_t i = 0; i < nblocks; ++i) {
            const int b = (int)(i & 1);
            const int64_t s0 = i * chunk, ns = (nf - s0 < chunk) ? nf - s0 : chunk;
            if (i + 1 < nblocks) PF_TRY(enqueue(i + 1));
            PF_HIP(hipEventSynchronize(c->cb_ev[b]));
            const auto t0 = std::chrono::steady_clock::now();
            host_callback_block(c, reinterpret_cast<const double *>(c->pin_x[b]), d, ns * N, reinterpret_cast<double *>(c->pin_lp[b]));
            c->cb_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            c->cb_bytes_d2h += (double)sizeof(double) * (double)ns * (double)per;
            PF_HIP(hipMemcpyAsync(c->cb_lp[b].p, c->pin_lp[b], sizeof(double) * (size_t)ns * N, hipMemcpyHostToDevice, c->stream));
            PF_TRY(pf_launch_scatter_rows(c, ns, N, d_list + s0, c->cb_lp[b].as<double>(), c->logp.as<double>()));
        }
    }
    PF_TRY(pf_launch_elbo_reduce(c));
    c->elbo_done = true;
    c->elbo_pending = true;
    return PFMI_OK;
}

int32_t pfmi_elbo_batch_wait(pfmi_ctx *c, double *elbo, double *se, int64_t *best_iter) {
    PF_CTX(c);
    PF_CHECK(c->elbo_done, PFMI_ERR_STATE, "elbo_batch_wait: call pfmi_elbo_batch_enqueue first");
    c->elbo_pending = false;
    if (elbo) PF_TRY(d2h_async(c, elbo, c->elbo.p, sizeof(double) * c->P));
    if (se) PF_TRY(d2h_async(c, se, c->se.p, sizeof(double) * c->P));
    if (best_iter) PF_TRY(d2h_async(c, best_iter, c->best_iter.p, sizeof(int64_t) * c->K));
    // pieces of the scan that gave up waiting for their fit's constants (they poison their draws; elbo_qf_kernel.hip): the counter is
    // the last word of the hand-over buffer
    auto lost2 = std::make_shared<std::array<uint32_t, 2>>();
    (*lost2)[0] = (*lost2)[1] = 0;
    for (int q = 0; q < 2; ++q)
        if (c->qf_share_s[q].cap >= sizeof(uint32_t))
            PF_TRY(d2h_async(c, &(*lost2)[(size_t)q], c->qf_share_s[q].as<char>() + c->qf_share_s[q].cap - sizeof(uint32_t), sizeof(uint32_t)));
    auto after = [c, lost2]() -> int32_t {
        uint32_t lost = (*lost2)[0] + (*lost2)[1];
        for (int q = 0; q < 2; ++q)
            if ((*lost2)[(size_t)q] != 0) (void)hipMemsetAsync(c->qf_share_s[q].as<char>() + c->qf_share_s[q].cap - sizeof(uint32_t), 0, sizeof(uint32_t), c->stream);
        {
            const char *fake = pf_debug_get("PFMI_QF_FAKE_LOST");       // test hook: the first wait of a ctx reports one lost piece (exercises the retry path)
            if (fake && fake[0] == '1' && !c->qf_no_share) lost = 1;
        }
        if (lost != 0) {
            // plain contention (another process on the GPU, CU masking, a profiler serialising dispatch) must not turn into wrong
            // results or a hard failure.  The poisoned scan (and whatever was enqueued behind it) is void; this ctx takes the two-launch cut --
            // no in-kernel wait -- from now on, and the caller is told to enqueue the step again.
            c->qf_no_share = true;
            c->qf_lost_total += lost;
            c->elbo_done = false; c->pooled = false;
            pf_set_error("ELBO scan: %u workgroup(s) gave up waiting for their fit's constants (GPU shared or dispatch serialised?); results discarded, "
                         "later scans on this context use the two-launch cut: enqueue the step again", lost);
            return PFMI_ERR_RETRY;
        }
        return PFMI_OK;
    };
    if (c->defer) { c->post_sync.push_back(after); return PFMI_OK; }
    PF_TRY(stream_sync(c));
    return after();
}

int32_t pfmi_elbo_batch(pfmi_ctx *c, int64_t N, const uint64_t *seeds, const double *u_host, double *elbo,
                        double *se, int64_t *best_iter) {
    PF_TRY(pfmi_elbo_batch_enqueue(c, N, seeds, u_host));
    return pfmi_elbo_batch_wait(c, elbo, se, best_iter);
}

int32_t pfmi_callback_stats(pfmi_ctx *c, double *callback_seconds, double *bytes_to_host) {
    PF_CTX(c);
    if (callback_seconds) *callback_seconds = c->cb_seconds;
    if (bytes_to_host) *bytes_to_host = c->cb_bytes_d2h;
    return PFMI_OK;
}

int32_t pfmi_callback_stats_dev(pfmi_ctx *c, double *bytes_in_hbm) {
    PF_CTX(c);
    if (bytes_in_hbm) *bytes_in_hbm = c->cb_bytes_dev;
    return PFMI_OK;
}

int32_t pfmi_get_elbo_logs(pfmi_ctx *c, int64_t p, double *logp, double *logq) {
    PF_CTX(c);
    PF_CHECK(c->elbo_done, PFMI_ERR_STATE, "get_elbo_logs: call pfmi_elbo_batch first");
    PF_CHECK(p >= 0 && p < c->P, PFMI_ERR_ARG, "point out of range");
    if (logp) PF_TRY(d2h(c, logp, c->logp.as<double>() + (size_t)p * c->N_e, sizeof(double) * c->N_e));
    if (logq) PF_TRY(d2h(c, logq, c->logq.as<double>() + (size_t)p * c->N_e, sizeof(double) * c->N_e));
    return PFMI_OK;
}

int32_t pfmi_draws(pfmi_ctx *c, int64_t p, uint64_t seed, int64_t n0, int64_t N, const double *u_host, double *X,
                   double *logp, double *logq) {
    PF_CTX_MUT(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "draws: call pfmi_fit_batch first");
    PF_CHECK(p >= 0 && p < c->P && N >= 1 && n0 >= 0, PFMI_ERR_ARG, "draws: bad arguments");
    const int d = c->d;
    const bool have_t = c->target.kind >= 0 && c->target.d == d;
    PF_TRY(c->xbuf.ensure(sizeof(double) * (size_t)d * N));
    PF_TRY(c->scratch.ensure(sizeof(double) * 2 * N + 64));
    double *d_lp = c->scratch.as<double>(), *d_lq = d_lp + N;
    int32_t *d_pt = reinterpret_cast<int32_t *>(d_lq + N);
    uint64_t *d_sd = reinterpret_cast<uint64_t *>(d_lq + N) + 1;
    const int32_t pt = (int32_t)p;
    PF_TRY(h2d(c, d_pt, &pt, sizeof(int32_t)));
    PF_TRY(h2d(c, d_sd, &seed, sizeof(uint64_t)));
    const double *d_u = nullptr;
    if (u_host) {
        PF_TRY(c->ubuf.ensure(sizeof(double) * (size_t)d * N));
        PF_TRY(h2d(c, c->ubuf.p, u_host, sizeof(double) * (size_t)d * N));
        d_u = c->ubuf.as<double>();
    }
    const bool cb = have_t && c->target.kind == PFMI_TARGET_HOST_CALLBACK;
    const bool dcb = have_t && c->target.kind == PFMI_TARGET_DEVICE_CALLBACK;
    PF_TRY(pf_launch_elbo_draws(c, d_pt, d_sd, 1, n0, N, d_u, 0, c->xbuf.as<double>(), 0, d_lp, d_lq, N,
                                have_t && !cb && !dcb, false));
    if (cb) PF_TRY(callback_logp(c, c->xbuf.as<double>(), N, d_lp));
    if (dcb) PF_TRY(devcb_logp(c, c->xbuf.as<double>(), N, d_lp, d_pt, 1, N));
    if (X) PF_TRY(d2h(c, X, c->xbuf.p, sizeof(double) * (size_t)d * N));
    if (logp) PF_TRY(d2h(c, logp, d_lp, sizeof(double) * N));
    if (logq) PF_TRY(d2h(c, logq, d_lq, sizeof(double) * N));
    return PFMI_OK;
}

int32_t pfmi_logpdf(pfmi_ctx *c, int64_t p, int64_t N, const double *X, double *out) {
    PF_CTX(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "logpdf: call pfmi_fit_batch first");
    PF_CHECK(p >= 0 && p < c->P && N >= 1 && X && out, PFMI_ERR_ARG, "logpdf: bad arguments");
    PF_TRY(c->xbuf.ensure(sizeof(double) * (size_t)c->d * N));
    PF_TRY(c->scratch.ensure(sizeof(double) * N));
    PF_TRY(h2d(c, c->xbuf.p, X, sizeof(double) * (size_t)c->d * N));
    PF_TRY(pf_launch_logpdf(c, p, N, c->xbuf.as<double>(), c->scratch.as<double>()));
    PF_TRY(d2h(c, out, c->scratch.p, sizeof(double) * N));
    return PFMI_OK;
}

// ---- remaining Woodbury operator surface ------------------------------------------------------------------
int32_t pfmi_woodbury_apply(pfmi_ctx *c, int64_t p, int32_t op, int64_t N, const double *X, double *out) {
    PF_CTX(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "woodbury_apply: call pfmi_fit_batch first");
    PF_CHECK(p >= 0 && p < c->P && N >= 1 && X && out, PFMI_ERR_ARG, "woodbury_apply: bad arguments");
    PF_CHECK(op >= PFMI_OP_UNWHITEN && op <= PFMI_OP_INVQUAD, PFMI_ERR_ARG, "woodbury_apply: unknown op %d", op);
    const size_t bytes = sizeof(double) * (size_t)c->d * N;
    PF_TRY(c->xbuf.ensure(bytes));
    PF_TRY(c->gbuf.ensure(bytes));
    PF_TRY(c->scratch.ensure(bytes));
    PF_TRY(h2d(c, c->xbuf.p, X, bytes));
    double *in = c->xbuf.as<double>(), *o1 = c->gbuf.as<double>(), *o2 = c->scratch.as<double>();
    const double *res = o1;
    switch (op) {
        case PFMI_OP_UNWHITEN: PF_TRY(pf_launch_woodbury_prim(c, 0, p, N, in, o1)); break;
        case PFMI_OP_WHITEN: PF_TRY(pf_launch_woodbury_prim(c, 1, p, N, in, o1)); break;
        case PFMI_OP_RMUL: PF_TRY(pf_launch_woodbury_prim(c, 2, p, N, in, o1)); break;
        case PFMI_OP_INVUNWHITEN: PF_TRY(pf_launch_woodbury_prim(c, 3, p, N, in, o1)); break;
        case PFMI_OP_MUL:                                            // lmul!(F.L, lmul!(F.R, x))
            PF_TRY(pf_launch_woodbury_prim(c, 2, p, N, in, o1));
            PF_TRY(pf_launch_woodbury_prim(c, 0, p, N, o1, o2));
            res = o2; break;
        case PFMI_OP_SOLVE:                                          // ldiv!(F.R, ldiv!(F.L, x))
            PF_TRY(pf_launch_woodbury_prim(c, 1, p, N, in, o1));
            PF_TRY(pf_launch_woodbury_prim(c, 3, p, N, o1, o2));
            res = o2; break;
        case PFMI_OP_QUAD:
            PF_TRY(pf_launch_woodbury_prim(c, 2, p, N, in, o1));
            PF_TRY(pf_launch_colsumsq(c, N, o1, o2));
            return d2h(c, out, o2, sizeof(double) * N);
        case PFMI_OP_INVQUAD:
            PF_TRY(pf_launch_woodbury_prim(c, 1, p, N, in, o1));
            PF_TRY(pf_launch_colsumsq(c, N, o1, o2));
            return d2h(c, out, o2, sizeof(double) * N);
    }
    return d2h(c, out, res, bytes);
}

int32_t pfmi_woodbury_diag(pfmi_ctx *c, int64_t p, double *diag) {
    PF_CTX(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "woodbury_diag: call pfmi_fit_batch first");
    PF_CHECK(p >= 0 && p < c->P && diag, PFMI_ERR_ARG, "woodbury_diag: bad arguments");
    PF_TRY(c->gbuf.ensure(sizeof(double) * c->d));
    PF_TRY(pf_launch_woodbury_diag(c, p, c->gbuf.as<double>()));
    return d2h(c, diag, c->gbuf.p, sizeof(double) * c->d);
}

// ---- pool / PSIS / resample ---------------------------------------------------------------------------
static int32_t pool_fill(pfmi_ctx *c);
static int32_t pool_alloc(pfmi_ctx *c, int64_t N_r) {
    const int K = c->K, d = c->d;
    c->N_r = N_r;
    const size_t S = (size_t)K * N_r;
    PF_TRY(c->pool.ensure(sizeof(double) * S * d));
    PF_TRY(c->pool_lr.ensure(sizeof(double) * S));
    PF_TRY(c->pool_lp.ensure(sizeof(double) * S));
    PF_TRY(c->pool_lq.ensure(sizeof(double) * S));
    PF_TRY(c->pool_points.ensure(sizeof(int32_t) * K));
    PF_TRY(c->pool_seeds.ensure(sizeof(uint64_t) * K));
    return PFMI_OK;
}

int32_t pfmi_pool_build(pfmi_ctx *c, int64_t N_r, const int64_t *points, const uint64_t *seeds) {
    PF_CTX_MUT(c);
    PF_CHECK(c->fitted, PFMI_ERR_STATE, "pool_build: call pfmi_fit_batch first");
    PF_CHECK(c->target.kind >= 0, PFMI_ERR_STATE, "pool_build: call pfmi_set_target first");
    PF_CHECK(N_r >= 1 && points && seeds, PFMI_ERR_ARG, "pool_build: bad arguments");
    const int K = c->K;
    std::vector<int32_t> pts((size_t)K);
    for (int k = 0; k < K; ++k) {
        PF_CHECK(points[k] >= c->off[k] && points[k] < c->off[k] + c->path_npts(k), PFMI_ERR_ARG,
                 "pool_build: point %lld does not belong to path %d", (long long)points[k], k);
        pts[(size_t)k] = (int32_t)points[k];
    }
    PF_TRY(pool_alloc(c, N_r));
    PF_TRY(h2d(c, c->pool_points.p, pts.data(), sizeof(int32_t) * K));
    PF_TRY(h2d(c, c->pool_seeds.p, seeds, sizeof(uint64_t) * K));
    c->pool_from_best = false;
    return pool_fill(c);
}

// draws of the K fits in pool_points / pool_seeds (device) -> pool, log ratios
static int32_t pool_fill(pfmi_ctx *c) {
    const int K = c->K, d = c->d;
    const int64_t N_r = c->N_r;
    const size_t S = (size_t)K * N_r;
    const bool cb = c->target.kind == PFMI_TARGET_HOST_CALLBACK, dcb = c->target.kind == PFMI_TARGET_DEVICE_CALLBACK;
    PF_TRY(pf_launch_elbo_draws(c, c->pool_points.as<int32_t>(), c->pool_seeds.as<uint64_t>(), K, 0, N_r, nullptr, 0,
                                c->pool.as<double>(), (int64_t)N_r * d, c->pool_lp.as<double>(),
                                c->pool_lq.as<double>(), N_r, !cb && !dcb, false));
    if (cb) PF_TRY(callback_logp(c, c->pool.as<double>(), (int64_t)S, c->pool_lp.as<double>()));
    if (dcb) PF_TRY(devcb_logp(c, c->pool.as<double>(), (int64_t)S, c->pool_lp.as<double>(), c->pool_points.as<int32_t>(), K, N_r));
    PF_TRY(pf_launch_logratio(c, (int64_t)S));
    c->pooled = true;
    return PFMI_OK;
}

int32_t pfmi_pool_build_best(pfmi_ctx *c, int64_t N_r, const uint64_t *fail_seeds) {
    PF_CTX_MUT(c);
    PF_CHECK(c->elbo_done, PFMI_ERR_STATE, "pool_build_best: call pfmi_elbo_batch[_enqueue] first");
    PF_CHECK(N_r >= 1, PFMI_ERR_ARG, "pool_build_best: bad arguments");
    PF_TRY(pool_alloc(c, N_r));
    PF_TRY(c->pool_ok.ensure(sizeof(int32_t) * c->K));
    if (fail_seeds) {
        PF_TRY(c->fail_seeds.ensure(sizeof(uint64_t) * c->K));
        PF_TRY(h2d(c, c->fail_seeds.p, fail_seeds, sizeof(uint64_t) * c->K));
    }
    PF_TRY(pf_launch_pool_pick(c, fail_seeds != nullptr));
    c->pool_from_best = true;
    return pool_fill(c);
}

int32_t pfmi_pool_winners(pfmi_ctx *c, int64_t *points, uint64_t *seeds, int32_t *success) {
    PF_CTX(c);
    PF_CHECK(c->pooled, PFMI_ERR_STATE, "pool_winners: call pfmi_pool_build[_best] first");
    std::vector<int32_t> pts((size_t)c->K);
    if (points) PF_TRY(d2h_async(c, pts.data(), c->pool_points.p, sizeof(int32_t) * c->K));
    if (seeds) PF_TRY(d2h_async(c, seeds, c->pool_seeds.p, sizeof(uint64_t) * c->K));
    if (success) {
        PF_CHECK(c->pool_from_best, PFMI_ERR_STATE, "pool_winners: success flags exist only after pfmi_pool_build_best");
        PF_TRY(d2h_async(c, success, c->pool_ok.p, sizeof(int32_t) * c->K));
    }
    PF_TRY(stream_sync(c));
    if (points)
        for (int k = 0; k < c->K; ++k) points[k] = pts[(size_t)k];
    return PFMI_OK;
}

int32_t pfmi_pool_get(pfmi_ctx *c, double *draws, double *log_ratios) {
    PF_CTX(c);
    PF_CHECK(c->pooled, PFMI_ERR_STATE, "pool_get: call pfmi_pool_build first");
    const size_t S = (size_t)c->K * c->N_r;
    if (draws) PF_TRY(d2h(c, draws, c->pool.p, sizeof(double) * S * c->d));
    if (log_ratios) PF_TRY(d2h(c, log_ratios, c->pool_lr.p, sizeof(double) * S));
    return PFMI_OK;
}

int32_t pfmi_pool_log_ratios_dev(pfmi_ctx *c, void **dev_ptr, int64_t *count) {
    PF_CTX(c);
    PF_CHECK(c->pooled, PFMI_ERR_STATE, "pool_log_ratios_dev: call pfmi_pool_build first");
    PF_HIP(hipStreamSynchronize(c->stream));
    if (dev_ptr) *dev_ptr = c->pool_lr.p;
    if (count) *count = (int64_t)c->K * c->N_r;
    return PFMI_OK;
}

int32_t pfmi_psis_dev(pfmi_ctx *c, const void *lr_dev, int64_t S, double *weights, double *log_weights,
                      double *pareto_k, int64_t *tail_len) {
    PF_CTX_MUT(c);
    PF_CHECK(lr_dev != nullptr && S > 0, PFMI_ERR_ARG, "psis: bad arguments");
    PF_TRY(pf_launch_psis(c, reinterpret_cast<const double *>(lr_dev), S));
    double out[4];
    PF_TRY(d2h(c, out, c->psis_out.p, sizeof(out)));
    if (pareto_k) *pareto_k = out[0];
    if (tail_len) *tail_len = (int64_t)out[1];
    if (weights) PF_TRY(d2h(c, weights, c->w.p, sizeof(double) * S));
    if (log_weights) PF_TRY(d2h(c, log_weights, c->lw.p, sizeof(double) * S));
    return PFMI_OK;
}

int32_t pfmi_psis_weights(pfmi_ctx *c, int64_t S, double *weights, double *log_weights) {
    PF_CTX(c);
    PF_CHECK(S > 0 && c->S_w == S, PFMI_ERR_STATE, "psis_weights: no PSIS result for S=%lld on this ctx", (long long)S);
    if (weights) PF_TRY(d2h_async(c, weights, c->w.p, sizeof(double) * S));
    if (log_weights) PF_TRY(d2h_async(c, log_weights, c->lw.p, sizeof(double) * S));
    if (c->defer) return PFMI_OK;
    return stream_sync(c);
}

int32_t pfmi_defer_downloads(pfmi_ctx *c, int32_t mode) {
    PF_CHECK(c != nullptr, PFMI_ERR_ARG, "null pfmi_ctx");
    if (mode < 0) { pf_deferred_drop(c); c->defer = false; return PFMI_OK; }       // cancel: nothing queued is ever delivered
    c->defer = mode != 0;
    return PFMI_OK;
}

int32_t pfmi_psis(pfmi_ctx *c, const double *lr, int64_t S, double *weights, double *log_weights, double *pareto_k,
                  int64_t *tail_len) {
    PF_CTX_MUT(c);
    PF_CHECK(lr != nullptr && S > 0, PFMI_ERR_ARG, "psis: bad arguments");
    PF_TRY(c->gbuf.ensure(sizeof(double) * S));
    PF_TRY(h2d(c, c->gbuf.p, lr, sizeof(double) * S));
    return pfmi_psis_dev(c, c->gbuf.p, S, weights, log_weights, pareto_k, tail_len);
}

int32_t pfmi_resample_indices(pfmi_ctx *c, int64_t S, int64_t ndraws, int32_t importance, int32_t replace,
                              uint64_t seed, const double *uniforms, int64_t *idx) {
    PF_CTX_MUT(c);
    PF_CHECK(S > 0 && ndraws >= 0, PFMI_ERR_ARG, "resample: bad arguments");
    PF_CHECK(!importance || c->S_w == S, PFMI_ERR_STATE,
             "resample: importance weights for S=%lld not available (run pfmi_psis first)", (long long)S);
    const double *d_uni = nullptr;
    if (uniforms && ndraws > 0) {
        PF_TRY(c->tailbuf.ensure(sizeof(double) * ndraws));
        PF_TRY(h2d(c, c->tailbuf.p, uniforms, sizeof(double) * ndraws));
        d_uni = c->tailbuf.as<double>();
    }
    PF_TRY(pf_launch_resample(c, S, ndraws, importance, replace, seed, d_uni));
    if (idx && ndraws > 0) PF_TRY(d2h(c, idx, c->idx.p, sizeof(int64_t) * ndraws));
    return PFMI_OK;
}

int32_t pfmi_resample_indices_direct(pfmi_ctx *c, int64_t S, int64_t ndraws, const double *uniforms, int64_t *idx) {
    PF_CTX_MUT(c);
    PF_CHECK(S > 0 && ndraws >= 0 && (uniforms != nullptr || ndraws == 0), PFMI_ERR_ARG, "resample_direct: bad arguments");
    PF_CHECK(c->S_w == S, PFMI_ERR_STATE, "resample_direct: importance weights for S=%lld not available (run pfmi_psis first)",
             (long long)S);
    if (ndraws == 0) return PFMI_OK;
    for (int64_t t = 0; t < ndraws; ++t)
        PF_CHECK(uniforms[t] >= 0.0 && uniforms[t] < 1.0, PFMI_ERR_ARG, "resample_direct: uniforms[%lld] = %g is not in [0, 1)", (long long)t,
                 uniforms[t]);
    PF_TRY(c->tailbuf.ensure(sizeof(double) * ndraws));
    PF_TRY(h2d(c, c->tailbuf.p, uniforms, sizeof(double) * ndraws));
    PF_TRY(pf_launch_resample_direct(c, S, ndraws, c->tailbuf.as<double>()));
    if (idx) PF_TRY(d2h(c, idx, c->idx.p, sizeof(int64_t) * ndraws));
    return PFMI_OK;
}

int32_t pfmi_pool_gather_dev(pfmi_ctx *c, int64_t ndraws, const int64_t *idx, int64_t col_offset, void *draws_dev) {
    PF_CTX(c);
    PF_CHECK(c->pooled, PFMI_ERR_STATE, "pool_gather: call pfmi_pool_build first");
    PF_CHECK(ndraws >= 0 && idx && draws_dev, PFMI_ERR_ARG, "pool_gather: bad arguments");
    PF_TRY(c->idx.ensure(sizeof(int64_t) * (ndraws > 0 ? ndraws : 1)));
    PF_TRY(h2d(c, c->idx.p, idx, sizeof(int64_t) * ndraws));
    PF_TRY(pf_launch_gather(c, ndraws, c->idx.as<int64_t>(), col_offset, reinterpret_cast<double *>(draws_dev)));
    PF_HIP(hipStreamSynchronize(c->stream));
    return PFMI_OK;
}

int32_t pfmi_pool_gather(pfmi_ctx *c, int64_t ndraws, const int64_t *idx, int64_t col_offset, double *draws) {
    PF_CTX(c);
    PF_CHECK(draws != nullptr, PFMI_ERR_ARG, "pool_gather: null output");
    PF_CHECK(c->pooled, PFMI_ERR_STATE, "pool_gather: call pfmi_pool_build first");
    PF_CHECK(ndraws >= 0 && (idx != nullptr || ndraws == 0), PFMI_ERR_ARG, "pool_gather: bad arguments");
    const int64_t owned = (int64_t)c->K * c->N_r;
    for (int64_t t = 0; t < ndraws; ++t)   // the host variant hands back real columns only: no silent zero fill
        PF_CHECK(idx[t] >= col_offset && idx[t] < col_offset + owned, PFMI_ERR_ARG,
                 "pool_gather: index %lld (position %lld) is outside this pool's columns [%lld, %lld)", (long long)idx[t],
                 (long long)t, (long long)col_offset, (long long)(col_offset + owned));
    PF_TRY(c->gbuf.ensure(sizeof(double) * (size_t)(ndraws > 0 ? ndraws : 1) * c->d));
    PF_TRY(pfmi_pool_gather_dev(c, ndraws, idx, col_offset, c->gbuf.p));
    PF_TRY(d2h(c, draws, c->gbuf.p, sizeof(double) * (size_t)ndraws * c->d));
    return PFMI_OK;
}

// ---- host utility: the seed-hierarchy generator of the Python / C mirrors ------------------------------------------
// out[i] = low 64 bits of Philox4x32-10(counter = (t0 + i, stream), key = seed) -- bit-identical to pf_rand_u64 on the device
// and to pfmi/hostrng.py (which falls back to NumPy when the library predates this entry).  Pure host code, no ctx.
int32_t pfmi_host_rand_u64(uint64_t seed, uint64_t t0, int64_t n, uint32_t stream, uint64_t *out) {
    PF_CHECK(n >= 0 && (out != nullptr || n == 0), PFMI_ERR_ARG, "host_rand_u64: bad arguments");
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t t = t0 + (uint64_t)i;
        uint32_t c0 = (uint32_t)t, c1 = (uint32_t)(t >> 32), c2 = stream, c3 = 0u, k0 = key0, k1 = key1;
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[i] = (uint64_t)c0 | ((uint64_t)c1 << 32);
    }
    return PFMI_OK;
}
// the same for m generators in one call (out is [m][n] row-major): a host that seeds every run of a batch pays one call, not m
int32_t pfmi_host_rand_u64_multi(int32_t m, const uint64_t *seeds, const uint64_t *t0, int64_t n, uint32_t stream, uint64_t *out) {
    PF_CHECK(m >= 0 && n >= 0 && (m == 0 || (seeds && t0)) && (out != nullptr || (int64_t)m * n == 0), PFMI_ERR_ARG, "host_rand_u64_multi: bad arguments");
    for (int32_t j = 0; j < m; ++j) PF_TRY(pfmi_host_rand_u64(seeds[j], t0[j], n, stream, out + (size_t)j * (size_t)n));
    return PFMI_OK;
}

// ---- device utilities ----------------------------------------------------------------------------------
int32_t pfmi_malloc_dev(pfmi_ctx *c, int64_t bytes, void **dev_ptr) {
    PF_CTX(c);
    PF_CHECK(dev_ptr != nullptr && bytes > 0, PFMI_ERR_ARG, "malloc_dev: bad arguments");
    PF_HIP(hipMalloc(dev_ptr, (size_t)bytes));
    return PFMI_OK;
}
int32_t pfmi_free_dev(pfmi_ctx *c, void *dev_ptr) {
    PF_CTX(c);
    if (dev_ptr) PF_HIP(hipFree(dev_ptr));
    return PFMI_OK;
}
int32_t pfmi_host_alloc(int64_t bytes, void **host_ptr) {
    PF_CHECK(bytes > 0 && host_ptr, PFMI_ERR_ARG, "host_alloc: bad arguments");
    *host_ptr = nullptr;
    PF_HIP(hipHostMalloc(host_ptr, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped));   // (mapped: the owner-only assembly of a multi-GPU result writes into it from every GPU)
    return PFMI_OK;
}
int32_t pfmi_host_free(void *host_ptr) {
    if (host_ptr) PF_HIP(hipHostFree(host_ptr));
    return PFMI_OK;
}
int32_t pfmi_memcpy_h2d(pfmi_ctx *c, void *dst, const void *src, int64_t bytes) {
    PF_CTX(c);
    return h2d(c, dst, src, (size_t)bytes);
}
int32_t pfmi_memcpy_d2h(pfmi_ctx *c, void *dst, const void *src, int64_t bytes) {
    PF_CTX(c);
    return d2h(c, dst, src, (size_t)bytes);
}

}  // extern "C"
