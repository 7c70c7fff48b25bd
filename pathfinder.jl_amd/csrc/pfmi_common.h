// pfmi_common.h -- shared declarations of the HIP implementation behind include/pfmi.h (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <map>
#include <set>
#include <vector>

#include "../../include/pfmi.h"
#include "pfmi_icdftab.h"

#define PF_LOG2PI 1.8378770664093454835606594728112352797227949472755668

// ---- error handling ------------------------------------------------------------------------------
void pf_set_error(const char *fmt, ...);
#define PF_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            pf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PFMI_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)
#define PF_CHECK(cond, code, ...)                                                                 \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            pf_set_error(__VA_ARGS__);                                                            \
            return (code);                                                                        \
        }                                                                                         \
    } while (0)
#define PF_TRY(expr)                                                                              \
    do {                                                                                          \
        int32_t rc__ = (expr);                                                                    \
        if (rc__ != PFMI_OK) return rc__;                                                         \
    } while (0)

// ---- growable device buffer ------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int32_t ensure(size_t bytes) {
        if (bytes <= cap) return PFMI_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        PF_HIP(hipMalloc(&p, want));
        cap = want;
        return PFMI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct KernelStat { double ms = 0.0; int64_t launches = 0; };

// pinned host arena of a ctx: small host -> device uploads are staged here and copied asynchronously on the ctx stream, so that an
// upload never forces a stream synchronisation in the middle of an enqueued pipeline (pf_upload in pfmi_api.hip).  Bump allocated;
// rewound whenever the stream is known to be idle (pf_arena_reset after a full synchronisation).
struct PinArena { char *base = nullptr; size_t cap = 0, off = 0; };
// a small device -> host download staged in the ctx's pinned download arena: delivered to `dst` by pf_stream_sync
struct DlPending { void *dst; const char *slot; size_t bytes; bool keep; const char *src; bool issued; };   // keep: queued under pfmi_defer_downloads (survives the next entry points)

// device-resident target description (Gaussian family rows are stored row-major for scalar loads)
struct TargetDev {
    int32_t kind = -1, d = 0, r = 0, rpad = 0;
    double offset = 0.0;
    DevBuf mean, a, wd /* [d][rpad] row-major */, g /* [rpad][rpad] row-major, lower */;
    DevBuf wd16;   // [ceil(d/16)*16][16] row-major, zero padded: MFMA A-operand source of the low-rank target part
    pfmi_logp_fn fn = nullptr;
    pfmi_logp_dev_fn dev_fn = nullptr;   // DEVICE_CALLBACK: launches the user's kernel(s) on the ctx stream
    void *user = nullptr;
};

struct pfmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, kev0 = nullptr, kev1 = nullptr;
    int profile = 0;                        // pfmi_profile: 0 off, 1 = every stage host-synchronised, 2 = event pairs left in the stream
    std::map<std::string, KernelStat> kstats;
    struct PendingStat { const char *name; hipEvent_t e0, e1; };
    std::vector<PendingStat> kpending;      // mode 2: pairs recorded but not read yet
    std::vector<hipEvent_t> kev_pool;
    hipEvent_t kcur = nullptr;
    std::set<const void *> lds_attr_done;   // kernels whose dynamic-LDS limit was raised on THIS ctx's device (see pf_raise_lds_limit)
    PinArena arena;                         // staging of small uploads (pf_upload)
    int cb_threads = 1;                     // host-closure fan-out (pfmi_set_callback_threads: the reference's ntasks)
    int hinit = 0;                          // Hinit of the history walk (PFMI_HINIT_*; pfmi_set_hinit / pfmi_fit_batch_ex)
    PinArena dl;                            // staging of small downloads (pf_download)
    std::vector<DlPending> dl_pending;
    // pfmi_defer_downloads: the download-only entry points queue their copies (and what they would do after their wait) and return at once;
    // the next wait on this ctx delivers everything in one host round trip
    bool defer = false;
    std::vector<std::function<int32_t()>> post_sync;
    int ncu = 0;                            // compute units of the device

    // traces
    int32_t K = 0, d = 0;
    int64_t P = 0;
    std::vector<int64_t> off;      // K+1 point offsets (host)
    std::vector<int32_t> path_of;  // P
    DevBuf theta, grad, d_off /* int64 K+1 */, d_path_of /* int32 P */;
    TargetDev target;
    // device trajectory generation
    bool have_trace_lp = false;
    DevBuf trace_lp;                          // [P]
    DevBuf st_theta, st_grad, st_lp, st_npts; // staging [K][maxiters+1][d]
    // streaming layout (pfmi_stream_enqueue): the trace points stay where the optimiser records them -- point l of path k is
    // p = k * vcap + l, vcap = maxiters + 1 (theta / grad ARE st_theta / st_grad, nothing is packed), P = K * vcap slots of which path k fills
    // st_npts[k]; off[k] = k * vcap.  Every per-point buffer is indexed by the slot.
    bool virt = false;
    int64_t vcap = 0;
    std::vector<int32_t> npts_h;              // host copy of st_npts (valid after pfmi_stream_wait)
    const uint64_t *d_stream_tab = nullptr;   // device copy of the runs' predrawn seed streams [K][vcap] (inside `seeds`)
    DevBuf hs_ial, hs_nacc;                   // state of the segmented history walk: carried 1 / alpha [K][d], accepted count [K]
    // The HOST schedules the pipeline (pfmi_stream_pump): the optimiser publishes its progress into page-locked memory, the calling thread
    // launches the walk / fits / scan of every segment of trace positions whose inputs are complete.  No kernel ever waits for another.
    int32_t *h_prog = nullptr;                // page-locked, coherent: [2 K] point counts, then "ended" flags -- written by pf_lbfgs_kernel
    int h_prog_cap = 0;
    char *h_list = nullptr;                   // page-locked staging of the scan's work lists (seeds, then points, of every fit slot)
    size_t h_list_cap = 0;
    hipStream_t s_opt = nullptr, s_fit = nullptr, s_scan1 = nullptr;    // producer; walk + fits; the second scan stream (the first is `stream`)
    hipEvent_t sg_fit = nullptr, sg_opt = nullptr, sg_scan[2] = {nullptr, nullptr}, sg_start = nullptr;
    struct StreamRun {
        bool active = false;                  // a pfmi_stream_enqueue is being pumped
        int K = 0, J = 0, cap = 0, l_next = 0, nseg = 0, minlen = 16, last_min = -1, pub = 16;
        int64_t N = 0, s0 = 0;                // s0: fit slots handed to the scan so far (offset into the work lists)
        double eps = 0.0;
        uint64_t *d_lseeds = nullptr; int32_t *d_list = nullptr;
        bool scan_used[2] = {false, false};
        int64_t scan_ns[2] = {0, 0};          // fits of the launch in flight on each scan stream
        int policy = 0, fit_eager = 1;
        int l_fit = 0;                        // positions whose walk + fits have been launched (>= l_next: the fits run ahead of the scans)
        bool have_seeds = false;              // the runs' seed streams have arrived (pfmi_stream_enqueue or, later, pfmi_stream_seeds)
        std::vector<uint64_t> seeds_pt;       // host copy of the per-point seeds (the scan's work lists are cut from it)
        std::chrono::steady_clock::time_point t_progress, t_start;
        std::vector<double> trace;            // PFMI_STREAM_TRACE: (t_us, l0, l1, fits, scan stream) per segment
        double host_s = 0.0; int host_n = 0;   // host time of the scheduling passes that LAUNCHED something (not the idle polls), and their number
    } sr;
    bool stream_pending = false;
    bool qf_seg_mode = false;                 // scan launches of a segment that is not the last: one workgroup per fit, no tail cut (they overlap)
    // the trace as the kernels see it: the packed buffers, or -- streaming layout -- the optimiser's staging buffers themselves
    double *th() const { return virt ? st_theta.as<double>() : theta.as<double>(); }
    double *gr() const { return virt ? st_grad.as<double>() : grad.as<double>(); }
    double *tlp() const { return virt ? st_lp.as<double>() : trace_lp.as<double>(); }
    int64_t path_npts(int k) const { return virt ? (int64_t)npts_h[(size_t)k] : off[(size_t)k + 1] - off[(size_t)k]; }
    int ncu_eff = 0;                          // CUs the scan's launch geometry assumes (0: all) -- the streaming segments leave the producers' out
    DevBuf lb_hs, lb_hy, lb_x0;               // (s, y) ring scratch [K][J][d], x0 [K][d]

    // fit state
    bool fitted = false;
    int32_t J = 0, kpad = 0;
    DevBuf alpha_all;   // [P][d]
    DevBuf hist_len;    // int32 [P]
    DevBuf hist_src;    // int32 [P][J]
    DevBuf hist_acc;    // int32 [P]: per path, the accepted updates in order (scratch of the history walk)
    DevBuf n_rej;       // int32 [K]
    DevBuf vh;          // [P][d][kpad] row-major Householder vectors (explicit unit diagonal)
    DevBuf tmat;        // [P][kpad][kpad] compact-WY T (row-major, upper)
    DevBuf vchol;       // [P][kpad][kpad] upper Cholesky factor V (row-major), identity padded
    DevBuf rq;          // [P][kpad][kpad] R factor of the QR (row-major, rows < k)
    DevBuf dmat;        // [P][kpad][kpad] Byrd D (row-major, m x m block)
    DevBuf sqrt_alpha;  // [P][d]
    DevBuf mu;          // [P][d]
    DevBuf logdet;      // [P]
    DevBuf status;      // int32 [P]
    DevBuf fit_scratch; // per-workgroup column-major working block of the panel fit kernel (d > 1024) + its work counter

    // ELBO state
    bool elbo_done = false;
    int64_t N_e = 0;
    DevBuf seeds;       // u64 [P]
    DevBuf logp, logq;  // [P][N_e]
    DevBuf elbo, se;    // [P]
    DevBuf best_iter;   // int64 [K]
    DevBuf fit_list;    // int32 list of points to process
    DevBuf ubuf;        // parity-mode normals
    DevBuf xbuf;        // scratch draws (callback path / pfmi_draws)
    DevBuf scratch;     // misc
    DevBuf qf_share_s[2];   // scan: per-fit constants handed from a tail fit's first piece to its other pieces, + one flag per tail fit
    uint32_t qf_epoch_s[2] = {0, 0};   // launch counter of the shared-constants scan: a flag equal to it means "published in THIS launch"
                                       // (one hand-over buffer + counter per scan stream: launches on different streams overlap)
    int qf_slot = 0;        // which of the two the next scan launch uses (0 outside the streaming pipeline)
    bool qf_no_share = false; // a hand-over of the shared-constants scan timed out on this ctx: later scans take the two-launch cut (no in-kernel wait)
    int64_t qf_lost_total = 0; // pieces that ever gave up waiting (pfmi_kernel_time("qf_handover_lost") reports it as `launches`)

    // pool / PSIS / resample state
    bool pooled = false;
    int64_t N_r = 0;
    DevBuf pool;        // [K][N_r][d]  == column-major (d, N_r, K)
    DevBuf pool_lr;     // [K*N_r]
    DevBuf pool_lp, pool_lq;
    DevBuf pool_points; // int32 [K]
    DevBuf pool_seeds;  // u64 [K]
    int64_t S_w = 0;
    DevBuf lw;          // [S] smoothed, normalised log weights
    DevBuf w;           // [S] weights
    DevBuf psis_out;    // [4] pareto_k, tail_len, ...
    DevBuf psis_aux;    // PsisAux: histogram, candidate list, smoothed tail and normalisation constants of the multi-workgroup PSIS
    DevBuf tailbuf;     // tail keys / idx
    DevBuf cdf;         // u64 [S]
    DevBuf idx;         // int64 [ndraws]
    DevBuf gbuf;        // gather output
    DevBuf sortk, sorti; // (key, index) arrays of the large replace = false request
    // host-callback targets: double-buffered device blocks, pinned host staging, one event per buffer
    DevBuf cb_x[2], cb_lp[2];
    void *pin_x[2] = {nullptr, nullptr}, *pin_lp[2] = {nullptr, nullptr};
    size_t pin_x_cap = 0, pin_lp_cap = 0;
    hipEvent_t cb_ev[2] = {nullptr, nullptr};
    // DEVICE_CALLBACK scans: PF_DCB_NB alternating block buffers / streams (stream 0 is `stream`)
#define PF_DCB_NB 2
    DevBuf dcb_x[PF_DCB_NB], dcb_lp[PF_DCB_NB];
    hipStream_t s_cb[PF_DCB_NB - 1] = {};
    hipEvent_t sg_cb[PF_DCB_NB] = {};
    double cb_seconds = 0.0;      // wall time spent inside the user's callback during the last elbo_batch
    double cb_bytes_d2h = 0.0;    // bytes of draws handed to the callback
    double cb_bytes_dev = 0.0;    // DEVICE_CALLBACK: bytes of draws materialised in HBM for the callback
    // enqueue / wait split of the blocking entry points
    bool opt_pending = false;     // pfmi_optimize_batch_enqueue issued, _wait not yet called
    int32_t opt_K = 0, opt_cap = 0;
    bool elbo_pending = false;    // pfmi_elbo_batch_enqueue issued
    bool pool_from_best = false;  // the pool was filled by pfmi_pool_build_best (pool_ok is valid)
    DevBuf pool_ok;               // int32 [K]: 1 = the path's winning fit was a success (pfmi_pool_build_best)
    DevBuf fail_seeds;            // u64 [K]
    DevBuf rs_err;                // int32: error flag of the last enqueued index selection
};

// small host -> device upload on the ctx stream WITHOUT synchronising it (pinned arena); large blocks take the synchronous path
int32_t pf_upload(pfmi_ctx *c, void *dst, const void *src, size_t bytes);
// the stream is idle: the arena may be reused from its start
void pf_arena_reset(pfmi_ctx *c);
// device -> host download on the ctx stream that does NOT block the host: up to 1 MB goes through a pinned arena and reaches `dst` in
// pf_stream_sync (a hipMemcpyAsync into pageable memory is a blocking call: each of an entry point's result downloads used to cost a
// host round trip of 20 - 70 us, ~0.3 ms per step; round 4).  Larger blocks are copied directly (blocking), so queue them last.
int32_t pf_download(pfmi_ctx *c, void *dst, const void *src, size_t bytes);
// hipStreamSynchronize + delivery of the staged downloads + arena rewind
int32_t pf_stream_sync(pfmi_ctx *c);
int32_t pf_dl_flush(pfmi_ctx *c);         // issue the staged downloads still pending as ONE gather kernel on the ctx stream (no wait)
// drops staged downloads that were never delivered (an entry point failed between queuing and waiting); called on entry by every public call
void pf_download_forget(pfmi_ctx *c);

// ---- launch helpers (implemented in the .hip files) ----------------------------------------------
struct HistSeg;
int32_t pf_launch_history(pfmi_ctx *c, double eps, const HistSeg *seg = nullptr);
int32_t pf_launch_fit(pfmi_ctx *c, int seg_l0 = 0, int seg_len = 0);
int32_t pf_launch_elbo_draws(pfmi_ctx *c, const int32_t *d_points, const uint64_t *d_seeds, int64_t nfits,
                             int64_t n0, int64_t N, const double *d_u, int64_t u_stride,
                             double *d_x, int64_t x_stride, double *d_logp, double *d_logq,
                             int64_t log_stride, bool with_target, bool by_point);
int32_t pf_launch_elbo_reduce(pfmi_ctx *c);
int32_t pf_launch_logpdf(pfmi_ctx *c, int64_t point, int64_t N, const double *d_x, double *d_out);
int32_t pf_launch_psis(pfmi_ctx *c, const double *d_lr, int64_t S);
// Test / tuning hooks (kernel selection, the RCCL stand-in, ...): a value set with pfmi_debug_set(), or -- ONLY when the process was
// started with PFMI_DEBUG_HOOKS=1 -- the environment variable of that name.  A production process therefore never has its numerics or
// its dlopen path changed by a stray environment variable.  NULL when unset.
const char *pf_debug_get(const char *name);
// comm_rccl.hip: pfmi_destroy announces a dying context; every live communicator that holds it is torn down first
void pf_comm_ctx_dying(pfmi_ctx *c);
int32_t pf_launch_resample(pfmi_ctx *c, int64_t S, int64_t ndraws, int importance, int replace,
                           uint64_t seed, const double *d_uniforms);
// the same without the final synchronisation: the error flag stays in c->rs_err until pf_resample_check reads it
int32_t pf_enqueue_resample(pfmi_ctx *c, int64_t S, int64_t ndraws, int importance, int replace,
                            uint64_t seed, const double *d_uniforms);
int32_t pf_resample_check(pfmi_ctx *c);
int32_t pf_launch_pool_pick(pfmi_ctx *c, int have_fail_seeds);
// d_lp[s * N + n] = NaN for every slot s whose fit d_points[s] failed
int32_t pf_launch_nan_failed(pfmi_ctx *c, int64_t npts, int64_t N, const int32_t *d_points, double *d_lp);
int32_t pf_launch_resample_direct(pfmi_ctx *c, int64_t S, int64_t ndraws, const double *d_uniforms);
int32_t pf_launch_gather(pfmi_ctx *c, int64_t ndraws, const int64_t *d_idx, int64_t col_offset, double *d_out, bool skip_unowned = false);
int32_t pf_launch_gather_pos(pfmi_ctx *c, int64_t n, const int64_t *d_idx, const int64_t *d_pos, int64_t col_offset, double *d_out);
int32_t pf_launch_scatter_cols(pfmi_ctx *c, int64_t n, const int64_t *d_pos, const double *d_in, double *d_out);
int32_t pf_launch_idx_digest(pfmi_ctx *c, int64_t ndraws, const int64_t *d_idx, double *d_out4, double local_err);
int32_t pf_launch_logratio(pfmi_ctx *c, int64_t n);
int32_t pf_launch_scatter_rows(pfmi_ctx *c, int64_t ns, int64_t N, const int32_t *d_points, const double *d_src, double *d_dst);
int32_t pf_launch_lbfgs(pfmi_ctx *c, int K, int J, int maxiters, double g_tol, const double *d_x0, int pub_mask = -1, int32_t *h_prog = nullptr);
int32_t pf_launch_trace_pack(pfmi_ctx *c, int64_t cap);
int32_t pf_launch_woodbury_prim(pfmi_ctx *c, int mode, int64_t p, int64_t N, const double *d_in, double *d_out);
int32_t pf_launch_colsumsq(pfmi_ctx *c, int64_t N, const double *d_x, double *d_out);
int32_t pf_launch_woodbury_diag(pfmi_ctx *c, int64_t p, double *d_out);

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (device, kernel) pair: remembered per ctx (one ctx = one device, one
// host thread), never in a process-wide static -- a second ctx on another GPU must raise the limit again.
inline int32_t pf_raise_lds_limit(pfmi_ctx *c, const void *kern, int bytes) {
    if (c->lds_attr_done.count(kern)) return PFMI_OK;
    PF_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    c->lds_attr_done.insert(kern);
    return PFMI_OK;
}

void pf_kernel_begin(pfmi_ctx *c);
void pf_kernel_end(pfmi_ctx *c, const char *name);

// ---- device helpers ----------------------------------------------------------------------------------
#ifdef __HIPCC__

// Philox4x32-10 (Salmon et al. 2011); identical bit stream to oracle/pf_oracle.c:pfo_philox4x32_10
// a ^ b ^ k with the (wave-uniform) round key as the one scalar operand VOP3 allows: LLVM splits this into two v_xor_b32 when k
// lives in an SGPR, which costs 20 extra instructions per Philox call
__device__ __forceinline__ uint32_t pf_xor3_key(uint32_t a, uint32_t b, uint32_t k) {
    if (__builtin_constant_p(k)) return a ^ b ^ k;
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(k));   // gfx950: three-input bit op, truth table 0x96 = a ^ b ^ c
    return r;
}
// Rounds of the NORMAL-generation stream.  Philox4x32 is crush-resistant from 7 rounds (Salmon et al. 2011, table 2: 7 is the
// minimum that passes BigCrush, 10 the default with a safety margin); the 1.1e10 normals of a step are the kernel's second
// largest issue stream and each round costs 24 SIMD cycles (two 64-bit multiplies), so the normals use 7.  Seeds, resampling
// uniforms and everything a host sees keep the 10-round function with its published known answers.
#define PF_NORMAL_ROUNDS 7
template <int ROUNDS>
__device__ __forceinline__ void pf_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = pf_xor3_key((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = pf_xor3_key((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// one Philox4x32 round as a separate step, for kernels that spread a call over several scheduling phases (elbo_mfma_kernel.hip)
__device__ __forceinline__ void pf_philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                                uint32_t &k0, uint32_t &k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
}
__device__ __forceinline__ void pf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                 uint32_t (&out)[4]) {
    pf_philox4x32<10>(c0, c1, c2, c3, k0, k1, out);
}
__device__ __forceinline__ void pf_philox_normals(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                  uint32_t (&out)[4]) {
    pf_philox4x32<PF_NORMAL_ROUNDS>(c0, c1, c2, c3, k0, k1, out);
}

// ---- the standard-normal generator ------------------------------------------------------------------------------
// One 32-bit Philox word -> one normal through the piecewise-cubic inverse normal CDF of pfmi_icdftab.h (generated by
// tools/gen_icdf_table.py, which documents the construction): p = (mag + 1/2) 2^-32; exponent / top-5 mantissa bits of v = (double) mag
// select a table entry, z = +-(c0 + v (c1 + v (c2 + v c3))) -- the cubic of every interval in GLOBAL monomial form in v (the 1/2 folded
// into the coefficients).  Only exactly rounded IEEE operations (cvt, fma) => bit-identical to the CPU checker (pfo_randn4).
// 8 VALU instructions + two 16-byte LDS reads per normal in the scan's form (pf_icdf_issue_adj), round 1's Box-Muller: ~27 + Philox.  Words with mag < 2^PF_ICDF_TAILBITS (probability 2^-19) are refined with a second Philox word
// (counter word 3 = 1) so that the tails reach |z| = 9.1 with >= 12 bits of resolution.
static __device__ const double PF_ICDF_TAB_DEV[PF_ICDF_ENTRIES][4] = { PF_ICDF_TABLE_ROWS };

// copy the first NB binades of the table (default 19: 2^-2 .. 2^-20 = everything a word with mag >= 2^12 can touch, 608 entries,
// 19 KB) into LDS; all threads, then a barrier.  A kernel short of LDS may keep fewer binades (NB = 12: 12 KB): words below
// 2^(31-NB) then take the slow path through the full table in global memory -- same values, probability 2^-NB per normal.
// LDS layout: two arrays of 16-byte entries, {c0, c1}[32 NB] then {c2, c3}[32 NB] -- with the random per-lane index a 16-lane LDS
// group then spreads over 16 bank quads instead of 8.
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ void pf_icdf_load(double2 *tab) {
    constexpr int NENT = NB << PF_ICDF_B;
    const double2 *src = reinterpret_cast<const double2 *>(&PF_ICDF_TAB_DEV[0][0]);
    // LDS order is ASCENDING in p (entry NENT - 1 - i of the generated table at slot i): the slot is then (hi32(P) >> 15) minus a
    // constant, one shift + one shift-add per look-up
    // (four independent loads per trip: as a plain loop every entry was its own global round trip -- load, s_waitcnt vmcnt(0), ds_write)
    for (int i0 = threadIdx.x; i0 < 2 * NENT; i0 += 4 * blockDim.x) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * (int)blockDim.x; v[u] = src[i < 2 * NENT ? i : 2 * NENT - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * (int)blockDim.x; if (i < 2 * NENT) tab[(i & 1) * NENT + (NENT - 1 - (i >> 1))] = v[u]; }
    }
}
// any word, full table in global memory (slow path / kernels without an LDS copy)
__device__ __forceinline__ double pf_icdf_any(uint32_t x, uint32_t x2) {
    const uint32_t mag = x & 0x7FFFFFFFu;
    double v;                                                       // the polynomial variable (tools/gen_icdf_table.py)
    if (mag >= (1u << PF_ICDF_TAILBITS)) v = (double)mag;
    else v = ((double)(((uint64_t)mag << 32) | x2) + 0.5) * 0x1p-32;
    const unsigned hi = (unsigned)__double2hiint(v);
    const int idx = PF_ICDF_IDX0 - (int)(hi >> (20 - PF_ICDF_B));
    const double *c = PF_ICDF_TAB_DEV[idx];
    const double q = fma(fma(fma(c[3], v, c[2]), v, c[1]), v, c[0]);
    return (x >> 31) ? -q : q;
}
// the common case in two halves (software-pipelined callers put work between them): `issue` converts the word and starts the two
// table reads from the LDS copy, `finish` evaluates the cubic once the coefficients have landed
// CLAMP = false: words beyond the LDS copy (fixed up by the caller anyway) read whatever lies up to 2^(PF_ICDF_B) * (32 - NB) * 16
// bytes BELOW the table -- the caller guarantees that much LDS in front of it (the ELBO scan: >= 40 KB of staged factor block)
template <int NB = PF_ICDF_NB_LDS, bool CLAMP = true>
__device__ __forceinline__ void pf_icdf_issue(uint32_t x, const double2 *lds_tab, double &v, double2 &c01, double2 &c23) {
    constexpr int NENT = NB << PF_ICDF_B;
    constexpr int SLOT0 = PF_ICDF_IDX0 - (NENT - 1);                // (hi32(v) >> 15) of the lowest interval kept in LDS
    v = (double)(x & 0x7FFFFFFFu);                                  // polynomial variable of the common case: mag
    const unsigned hi = (unsigned)__double2hiint(v);
    int slot = (int)(hi >> (20 - PF_ICDF_B)) - SLOT0;
    if (CLAMP) slot = slot > 0 ? slot : 0;                          // words beyond the LDS copy are fixed up by the caller
    c01 = lds_tab[slot]; c23 = lds_tab[NENT + slot];
}
// The same look-up with the address arithmetic folded into one v_lshl_add_u32: `adj` is the LDS byte address the interval index
// (hi32(P) >> 15) = 0 would have (loop-invariant, kept opaque so that the compiler does not split the constant off again -- it
// emitted three v_add_u32 per normal), the second coefficient pair sits NENT entries further (immediate ds_read offset).  Only for
// callers that guarantee readable LDS below the table (CLAMP = false above).
#ifndef PF_ICDF_CLAMP_ADJ
#define PF_ICDF_CLAMP_ADJ 1
#endif
typedef double pf_v2d __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) pf_v2d *pf_lds_d2;
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ uint32_t pf_icdf_adj(const double2 *lds_tab) {
    constexpr int NENT = NB << PF_ICDF_B;
    constexpr int SLOT0 = PF_ICDF_IDX0 - (NENT - 1);
    uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)(pf_lds_d2)lds_tab - (uint32_t)SLOT0 * 16u));   // wave-uniform
    asm volatile("" : "+s"(off));
    return off;
}
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ void pf_icdf_issue_adj(uint32_t x, uint32_t adj, double &v, double2 &c01, double2 &c23) {
    constexpr int NENT = NB << PF_ICDF_B;
    v = (double)(x & 0x7FFFFFFFu);
    const unsigned hi = (unsigned)__double2hiint(v);
    unsigned idx = hi >> (20 - PF_ICDF_B);
#if PF_ICDF_CLAMP_ADJ                  // never form an LDS address in front of the caller's guard region (mag = 0: idx = 0)
    idx = max(idx, (unsigned)(PF_ICDF_IDX0 - ((32 << PF_ICDF_B) - 1)));
#endif
    asm("" : "+v"(idx));                                            // keeps (idx << 4) + adj one v_lshl_add_u32 (else: shift, mask, add)
    const pf_lds_d2 e = (pf_lds_d2)(uintptr_t)(adj + (idx << 4));
    const pf_v2d a = e[0], b = e[NENT];
    c01 = make_double2(a.x, a.y); c23 = make_double2(b.x, b.y);
}
// the table holds magnitudes (> 0 on every interval), so "apply the sign of the word" is a bit-field insert (one v_bfi_b32)
__device__ __forceinline__ double pf_icdf_finish(uint32_t x, double v, const double2 &c01, const double2 &c23) {
    const double q = fma(fma(fma(c23.y, v, c23.x), v, c01.y), v, c01.x);
    unsigned hq;                                                    // (mask & hi(q)) | (~mask & x); spelled out because the compiler
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hq) : "s"(0x7FFFFFFFu), "v"((unsigned)__double2hiint(q)), "v"(x));   // only matched half of the sites
    return __hiloint2double((int)hq, __double2loint(q));
}
// true if one of the four words falls outside the LDS copy
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ bool pf_icdf_miss4(const uint32_t (&x)[4]) {
    const uint32_t m01 = min(x[0] & 0x7FFFFFFFu, x[1] & 0x7FFFFFFFu), m23 = min(x[2] & 0x7FFFFFFFu, x[3] & 0x7FFFFFFFu);
    return min(m01, m23) < (1u << (31 - NB));
}
// slow path: replace the normals of words outside the LDS copy by their values from the full table; words below
// 2^PF_ICDF_TAILBITS are first refined with the second Philox call (counter word 3 = 1)
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ void pf_icdf4_fix(const uint32_t (&x)[4], uint32_t n, uint32_t g, uint32_t stream, uint32_t k0, uint32_t k1,
                                             double (&z)[4]) {
    uint32_t x2[4] = {0u, 0u, 0u, 0u};
    if (__any(pf_icdf_miss4<PF_ICDF_NB_LDS>(x))) pf_philox_normals(n, g, stream, 1u, k0, k1, x2);
#pragma unroll
    for (int t = 0; t < 4; ++t) if ((x[t] & 0x7FFFFFFFu) < (1u << (31 - NB))) z[t] = pf_icdf_any(x[t], x2[t]);
}
// the four normals of Philox words x[0..3] (call (n, g, stream)), LDS table for the common case
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ void pf_icdf4(const uint32_t (&x)[4], uint32_t n, uint32_t g, uint32_t stream, uint32_t k0, uint32_t k1,
                                         const double2 *lds_tab, double (&z)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        double dp;
        double2 c01, c23;
        pf_icdf_issue<NB>(x[t], lds_tab, dp, c01, c23);
        z[t] = pf_icdf_finish(x[t], dp, c01, c23);
    }
    if (__builtin_expect(__any(pf_icdf_miss4<NB>(x)), 0)) pf_icdf4_fix<NB>(x, n, g, stream, k0, k1, z);
}
// four standard normals for rows 4g..4g+3 of draw n
template <int NB = PF_ICDF_NB_LDS>
__device__ __forceinline__ void pf_randn4(uint64_t seed, uint32_t g, uint32_t n, uint32_t stream, const double2 *lds_tab,
                                          double (&z)[4]) {
    uint32_t x[4];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    pf_philox_normals(n, g, stream, 0u, k0, k1, x);
    pf_icdf4<NB>(x, n, g, stream, k0, k1, lds_tab, z);
}

__device__ __forceinline__ uint64_t pf_rand_u64(uint64_t seed, uint64_t t, uint32_t stream) {
    uint32_t x[4];
    pf_philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    return (uint64_t)x[0] | ((uint64_t)x[1] << 32);
}

// wave64 sum on the DPP data path (no LDS traffic): row_shr 1/2/4/8 scan inside each row of 16 lanes, row_bcast15 /
// row_bcast31 across rows, total read from lane 63 into SGPRs -> every lane gets the same value.  Fixed order ->
// deterministic.  (fp64 adds have no DPP form: each step moves the two halves with v_mov_b32_dpp and adds.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double pf_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
// lane `lane`'s value of x in every lane (two v_readlane_b32 through SGPRs; `lane` must be wave-uniform)
__device__ __forceinline__ double pf_readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pf_wave_sum(double v) {
    v = pf_dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = pf_dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = pf_dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = pf_dpp_add<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of each row holds the row sum
    v = pf_dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = pf_dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// s + (s of lane ^ 16) and s + (s of lane ^ 32) on the VALU (gfx950: v_permlane16_swap / v_permlane32_swap) instead of
// ds_bpermute round trips through the LDS crossbar.  swap(A, A') with A = A' = s leaves A = [r0 r0 r2 r2], A' = [r1 r1 r3 r3] by
// rows of 16 (resp. [lo lo], [hi hi] by halves), so A + A' is the pairwise sum in every lane -- bit-identical to the shuffle form.
__device__ __forceinline__ double pf_add_xor16(double s) {
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double pf_add_xor32(double s) {
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// sum over the four lanes {c, c + 16, c + 32, c + 48}
__device__ __forceinline__ double pf_sum_q(double s) { return pf_add_xor32(pf_add_xor16(s)); }
__device__ __forceinline__ double pf_wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum of NV values per thread.  `red` is LDS scratch of (blockDim.x/64)*NV doubles.
// On return every thread holds the totals in v[].  Waves are combined in wave-index order.
template <int NV>
__device__ __forceinline__ void pf_block_sum(double (&v)[NV], double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = pf_wave_sum(v[i]);
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i];
#pragma unroll 1
    for (int w = 1; w < nw; ++w) {                 // wave loop outermost and rolled: NV reads per LDS round trip (see pf_block_sum_mv)
        const double *bw = red + w * NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += bw[i];
    }
}
// Same with ONE barrier per reduction: consecutive reductions alternate between two scratch halves (`flip`), so the barrier
// that publishes reduction n + 1 also guarantees that everybody finished reading reduction n's half before reduction n + 2
// overwrites it.  `red` holds 2 * (blockDim.x / 64) * NVMAX doubles; every thread must make the same sequence of calls.
// NW = number of waves of the workgroup when it is a compile-time constant (0: read blockDim): with a run-time wave count the
// cross-wave sum is a loop of dependent LDS reads with an s_waitcnt each; a static count lets all reads fly at once.
template <int NV, int NVMAX, int NW = 0>
__device__ __forceinline__ void pf_block_sum_pp(double (&v)[NV], double *red, int &flip) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = NW ? NW : (int)(blockDim.x >> 6);
    double *buf = red + flip * (nw * NVMAX);
    flip ^= 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = pf_wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) buf[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (NW) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = buf[i];
#pragma unroll
            for (int w = 1; w < (NW ? NW : 1); ++w) s += buf[w * NV + i];
            v[i] = s;
        }
    } else {                                   // wave loop outermost and rolled: see pf_block_sum_mv
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = buf[i];
#pragma unroll 1
        for (int w = 1; w < nw; ++w) {
            const double *bw = buf + w * NV;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += bw[i];
        }
    }
}
// ---- multi-value butterfly (NV % 4 == 0): the wave sums of NV values in NV/2 + NV/4 permlane-swap steps plus 4 DPP steps on NV/4
// values, instead of 6 DPP steps on each of the NV values (12 values: 63 instead of 240 instructions).
//   step 1  v_permlane32_swap on (v[j], v[j + NV/2]): lanes < 32 keep the pair sum of v[j], lanes >= 32 that of v[j + NV/2]
//   step 2  v_permlane16_swap on (r[j], r[j + NV/4]): row rho of 16 lanes now owns values rho NV/4 + i, i < NV/4, each lane holding
//           the sum over its 4 lanes {c, c + 16, c + 32, c + 48}
//   step 3  row_shr 1/2/4/8 inside each row: lane 15 of row rho holds the wave totals of its NV/4 values
// pf_block_sum_mv has the interface of pf_block_sum_pp (one barrier, ping-pong scratch of 2 * nwaves * NVMAX doubles); the totals
// are read back from LDS, so every thread gets bit-identical values (the summation order differs from pf_block_sum_pp's).
__device__ __forceinline__ double pf_swap32_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double pf_swap16_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int NV, int NVMAX, int NW = 0>
__device__ __forceinline__ void pf_block_sum_mv(double (&v)[NV], double *red, int &flip) {
    static_assert(NV % 4 == 0, "pf_block_sum_mv: NV must be a multiple of 4");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = NW ? NW : (int)(blockDim.x >> 6);
    double *buf = red + flip * (nw * NVMAX);
    flip ^= 1;
    double r[NV / 2], q[NV / 4];
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) r[j] = pf_swap32_add(v[j], v[j + NV / 2]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) q[j] = pf_swap16_add(r[j], r[j + NV / 4]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        q[j] = pf_dpp_add<0x111, 0xf>(q[j]);   // row_shr:1
        q[j] = pf_dpp_add<0x112, 0xf>(q[j]);   // row_shr:2
        q[j] = pf_dpp_add<0x114, 0xf>(q[j]);   // row_shr:4
        q[j] = pf_dpp_add<0x118, 0xf>(q[j]);   // row_shr:8  -> lane 15 of each row
    }
    if ((lane & 15) == 15) {
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) buf[wave * NV + (lane >> 4) * (NV / 4) + j] = q[j];
    }
    __syncthreads();
    if (NW) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = buf[i];
#pragma unroll
            for (int w = 1; w < (NW ? NW : 1); ++w) s += buf[w * NV + i];      // same order as the run-time loop
            v[i] = s;
        }
    } else {
        // run-time wave count: the WAVE loop is the outer, rolled one -- NV independent reads per trip and one wait, nw - 1 round trips
        // per reduction.  (Until round 4 the value loop was the outer one: every partial was its own LDS round trip with an s_waitcnt,
        // 48 dependent trips per 12-value reduction of the fit kernel, ~4 000 cycles of a fit's critical path 15 times per fit.  The
        // fully static form makes the register allocator spill ~600 dwords there: the rolled loop also fences the scheduler.)
        // Same order ((w0 + w1) + w2) + ...: the same bits.
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = buf[i];
#pragma unroll 1
        for (int w = 1; w < nw; ++w) {
            const double *bw = buf + w * NV;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += bw[i];
        }
    }
}
template <int NVMAX, int NW = 0>
__device__ __forceinline__ double pf_block_sum1_pp(double x, double *red, int &flip) {
    double v[1] = {x};
    pf_block_sum_pp<1, NVMAX, NW>(v, red, flip);
    return v[0];
}
template <int NVMAX, int NW = 0>
__device__ __forceinline__ double pf_block_max1_pp(double x, double *red, int &flip) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = NW ? NW : (int)(blockDim.x >> 6);
    double *buf = red + flip * (nw * NVMAX);
    flip ^= 1;
    x = pf_wave_max(x);
    if (lane == 0) buf[wave] = x;
    __syncthreads();
    if (NW) {
        double t[NW ? NW : 1];
#pragma unroll
        for (int w = 0; w < (NW ? NW : 1); ++w) t[w] = buf[w];
        double s = t[0];
#pragma unroll
        for (int w = 1; w < (NW ? NW : 1); ++w) s = fmax(s, t[w]);
        return s;
    }
    double s = buf[0];
    for (int w = 1; w < nw; ++w) s = fmax(s, buf[w]);
    return s;
}
__device__ __forceinline__ double pf_block_sum1(double x, double *red) {
    double v[1] = {x};
    pf_block_sum<1>(v, red);
    return v[0];
}
__device__ __forceinline__ double pf_block_max1(double x, double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    x = pf_wave_max(x);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < nw; ++w) s = fmax(s, red[w]);
    return s;
}
#endif  // __HIPCC__
