// psis_kernels.hip -- Pareto-smoothed importance sampling + resampling over the pooled draws (gfx950).
//
//   pf_psis_kernel     : PSIS.psis(log_ratios) (called at reference src/resample.jl:78; algorithm of
//                        Vehtari et al. 2024 with the Zhang-Stephens GPD fit, as in PSIS.jl 0.9).
//                        One 1024-thread workgroup: radix-select of the (M+1) largest log ratios
//                        ((value, index) composite order -> deterministic under ties), LDS bitonic
//                        sort of the tail, GPD profile-likelihood grid (one wave per theta), tail
//                        replacement by GPD quantiles, logsumexp normalisation.
//   pf_cdf_kernel      : exact fixed-point CDF (u64 prefix sums of floor(w 2^62)); integer arithmetic
//                        makes the table independent of summation order, launch geometry and GPU count.
//   pf_sample_kernel   : inverse-CDF index draw (stands in for StatsBase.sample, src/resample.jl:61-66).
//   pf_norep_kernel    : Efraimidis-Spirakis sampling without replacement (replace = false).
//   pf_gather_kernel   : draws = draws_all[:, inds] (src/resample.jl:68).
#include "pfmi_common.h"
#include <stdlib.h>
#include <stddef.h>

#define PSIS_THREADS 1024
#ifndef PSIS_PROF
#define PSIS_PROF 0                     // 1: thread 0 of pf_psis_kernel prints its section times (10 ns ticks); experiment builds only
#endif
#if PSIS_PROF
#define PS_STAMP(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); ps_t[k] = t_ - ps_last; ps_last = t_; } } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif
#define TAILCAP 4096

__device__ __forceinline__ uint64_t pf_key_of(double x) {   // order-preserving map double -> u64
    uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double pf_val_of(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// for (i = tid; i < S; i += nt) f(i, vals[i]) with PF_SB loads in flight per thread: a single workgroup walks S = 64 000 values
// a dozen times, and with one load per trip every trip cost an L2 round trip (0.30 ms for the whole kernel)
#define PF_SB 8
template <typename F>
__device__ __forceinline__ void pf_foreach_batched(const double *vals, long long S, F f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (long long i0 = tid; i0 < S; i0 += (long long)PF_SB * nt) {
        double v[PF_SB];
#pragma unroll
        for (int u = 0; u < PF_SB; ++u) { const long long i = i0 + (long long)u * nt; v[u] = vals[i < S ? i : S - 1]; }
#pragma unroll
        for (int u = 0; u < PF_SB; ++u) { const long long i = i0 + (long long)u * nt; if (i < S) f(i, v[u]); }
    }
}

struct SelectState {
    unsigned hist[256];
    unsigned long long prefix;
    long long remaining;
    unsigned count, eq_count;
    unsigned long long wmin[16], wmax[16];
    unsigned wtot[4];
};

// From the 256-bin histogram pick the bin (scanning from the top) in which the `remaining`-th largest element falls;
// updates st->prefix / st->remaining (and eq_count = population of the picked bin).  Parallel suffix scan by the first 256
// threads (thread t owns bin 255 - t) instead of a 256-step serial walk by one thread.  All threads must call it.
__device__ __forceinline__ void pf_hist_pick(SelectState *st, unsigned long long prefix, int pass) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long long rem = st->remaining;
    unsigned h = 0u, incl = 0u;
    if (tid < 256) {
        h = st->hist[255 - tid];
        incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) st->wtot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        unsigned base = 0u;
        for (int w = 0; w < (tid >> 6); ++w) base += st->wtot[w];
        incl += base;
        const unsigned excl = incl - h;
        if ((long long)excl < rem && (long long)incl >= rem) {          // exactly one bin qualifies
            st->remaining = rem - (long long)excl;
            st->prefix = prefix | ((unsigned long long)(255 - tid) << (8 * pass));
            st->eq_count = h;
        }
    }
    __syncthreads();
}

// Select the R largest elements of vals[0..S) in (value, index) order.  On return (all threads):
// tkeys/tidx [0..R) hold them sorted ASCENDING (LDS).  Requires R <= TAILCAP, blockDim = PSIS_THREADS.
__device__ void pf_select_top_sorted(const double *__restrict__ vals, long long S, int R, uint64_t *tkeys,
                                     uint32_t *tidx, SelectState *st) {
    const int tid = threadIdx.x, nt = blockDim.x;
    // ---- radix select of the R-th largest key.  Bytes shared by ALL keys are skipped: log ratios of one pool have the same
    //      sign / exponent, so the top 1-2 passes would serialise every LDS atomic on a single histogram bin.
    int top_pass = 7;
    {
        uint64_t kmin = ~0ull, kmax = 0ull;
        pf_foreach_batched(vals, S, [&](long long, double x) { const uint64_t k = pf_key_of(x); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; });
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
            kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
        }
        if ((tid & 63) == 0) { st->wmin[tid >> 6] = kmin; st->wmax[tid >> 6] = kmax; }
        __syncthreads();
        kmin = st->wmin[0]; kmax = st->wmax[0];
        for (int w = 1; w < (nt >> 6); ++w) { kmin = st->wmin[w] < kmin ? st->wmin[w] : kmin; kmax = st->wmax[w] > kmax ? st->wmax[w] : kmax; }
        const uint64_t diff = kmin ^ kmax;
        top_pass = diff ? (63 - __clzll((long long)diff)) / 8 : 0;
        __syncthreads();
        if (tid == 0) { st->prefix = (top_pass == 7) ? 0ull : (kmax & (~0ull << (8 * (top_pass + 1)))); st->remaining = R; st->eq_count = 0u; }
    }
    __syncthreads();
    for (int pass = top_pass; pass >= 0; --pass) {
        for (int b = tid; b < 256; b += nt) st->hist[b] = 0u;
        __syncthreads();
        const unsigned long long prefix = st->prefix;
        const unsigned long long himask = (pass == 7) ? 0ull : (~0ull << (8 * (pass + 1)));
        pf_foreach_batched(vals, S, [&](long long, double x) {
            const uint64_t k = pf_key_of(x);
            if ((k & himask) == prefix) atomicAdd(&st->hist[(unsigned)((k >> (8 * pass)) & 0xFF)], 1u);
        });
        __syncthreads();
        pf_hist_pick(st, prefix, pass);
    }
    const uint64_t tk = st->prefix;          // threshold key; st->remaining of the equal-key elements are needed
    const long long need_eq = st->remaining;
    const bool all_eq = (long long)st->eq_count == need_eq;     // no tie to break: every equal-key element is selected
    __syncthreads();
    // ---- among key == tk take the `need_eq` LARGEST indices: radix select on the index (only when there is a tie)
    if (tid == 0) { st->prefix = 0ull; st->remaining = need_eq; }
    __syncthreads();
    for (int pass = all_eq ? -1 : 3; pass >= 0; --pass) {
        for (int b = tid; b < 256; b += nt) st->hist[b] = 0u;
        __syncthreads();
        const unsigned long long prefix = st->prefix;
        const unsigned long long himask = (pass == 3) ? 0ull : ((~0ull << (8 * (pass + 1))) & 0xFFFFFFFFull);
        pf_foreach_batched(vals, S, [&](long long i, double x) {
            if (pf_key_of(x) == tk && (((unsigned long long)i) & himask) == prefix)
                atomicAdd(&st->hist[(unsigned)((i >> (8 * pass)) & 0xFF)], 1u);
        });
        __syncthreads();
        pf_hist_pick(st, prefix, pass);
    }
    const unsigned long long ti = st->prefix;   // index threshold
    // ---- compact the selected elements into LDS, pad with sentinels, bitonic sort ascending
    int npow = 1;
    while (npow < R) npow <<= 1;
    for (int t = tid; t < npow; t += nt) { tkeys[t] = ~0ull; tidx[t] = 0xFFFFFFFFu; }
    if (tid == 0) st->count = 0u;
    __syncthreads();
    pf_foreach_batched(vals, S, [&](long long i, double x) {
        const uint64_t k = pf_key_of(x);
        if (k > tk || (k == tk && (unsigned long long)i >= ti)) {
            const unsigned slot = atomicAdd(&st->count, 1u);
            if (slot < (unsigned)TAILCAP) { tkeys[slot] = k; tidx[slot] = (uint32_t)i; }
        }
    });
    __syncthreads();
    for (int size = 2; size <= npow; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < npow / 2; t += nt) {
                const int lo = 2 * t - (t & (stride - 1));   // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const uint64_t ka = tkeys[lo], kb = tkeys[hi];
                const uint32_t ia = tidx[lo], ib = tidx[hi];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) { tkeys[lo] = kb; tkeys[hi] = ka; tidx[lo] = ib; tidx[hi] = ia; }
            }
            __syncthreads();
        }
    }
}

// ---- multi-workgroup PSIS (S >= PSIS_MULTI_MIN) ------------------------------------------------------------------------------
// The single-workgroup kernel walks the S log ratios ~10 times (select) + 2 x exp (normalisation): 0.26 ms at S = 64 000, none of
// which shrinks with the number of GPUs.  The passes over S move to short multi-workgroup kernels; what is left for the single
// workgroup is the sort of <= 4096 candidates and the GPD fit.
//   pf_psis_minmax_kernel   key range (atomicMax on the key and on its complement: one memset initialises both)
//   pf_psis_hist_kernel     4096-bin histogram of the 12 key bits below the highest differing bit (LDS histogram per workgroup)
//   pf_psis_compact_kernel  every workgroup scans the histogram from the top for the bin that holds the (M+1)-th largest key; if that
//                           bin and everything above are <= TAILCAP elements they are appended to the candidate list (any order:
//                           the tail kernel sorts by (key, index)); otherwise `done` stays 0 and the tail kernel selects by itself
//   pf_psis_kernel<true>    candidates -> sorted tail, Zhang-Stephens fit, smoothed tail + normalisation constants to `aux`
//   pf_psis_sum_kernel      partial sums of exp(lr - max) over the untouched elements (fixed slice per workgroup)
//   pf_psis_norm_kernel     lse from the partials in a fixed order; lw, w
#define PSIS_MULTI_MIN 8192
#define PSIS_MW 64                      // workgroups of the multi-workgroup passes
#define PSIS_MT 256
struct PsisAux {
    unsigned long long kmax, kmin_inv;  // max key, max ~key
    unsigned hist[4096];
    unsigned cand_count, done, replaced, ti0;
    unsigned long long tk0;
    double mx, se_tail, pareto_k, sigma;
    int M, pad;
    double partial[PSIS_MW];
    unsigned long long cand_key[TAILCAP];
    unsigned cand_idx[TAILCAP];
    double tail_val[TAILCAP];
    unsigned tail_idx[TAILCAP];
};
__device__ __forceinline__ int pf_psis_shift0(unsigned long long kmin, unsigned long long kmax) {
    const unsigned long long diff = kmin ^ kmax;
    const int top = diff ? 63 - __clzll((long long)diff) : 0;
    return top >= 11 ? top - 11 : 0;
}
__global__ __launch_bounds__(PSIS_MT) void pf_psis_minmax_kernel(long long S, const double *__restrict__ lr, PsisAux *aux) {
    unsigned long long kmax = 0ull, kinv = 0ull;
    for (long long i = (long long)blockIdx.x * PSIS_MT + threadIdx.x; i < S; i += (long long)PSIS_MW * PSIS_MT) {
        const unsigned long long k = pf_key_of(lr[i]);
        kmax = k > kmax ? k : kmax; kinv = ~k > kinv ? ~k : kinv;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(kmax, off, 64), b = __shfl_xor(kinv, off, 64);
        kmax = a > kmax ? a : kmax; kinv = b > kinv ? b : kinv;
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&aux->kmax, kmax); atomicMax(&aux->kmin_inv, kinv); }
}
__global__ __launch_bounds__(PSIS_MT) void pf_psis_hist_kernel(long long S, const double *__restrict__ lr, PsisAux *aux) {
    __shared__ unsigned h[4096];
    for (int b = threadIdx.x; b < 4096; b += PSIS_MT) h[b] = 0u;
    __syncthreads();
    const int sh = pf_psis_shift0(~aux->kmin_inv, aux->kmax);
    for (long long i = (long long)blockIdx.x * PSIS_MT + threadIdx.x; i < S; i += (long long)PSIS_MW * PSIS_MT)
        atomicAdd(&h[(unsigned)((pf_key_of(lr[i]) >> sh) & 0xFFFull)], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < 4096; b += PSIS_MT) if (h[b]) atomicAdd(&aux->hist[b], h[b]);
}
__global__ __launch_bounds__(PSIS_MT) void pf_psis_compact_kernel(long long S, const double *__restrict__ lr, PsisAux *aux, int R) {
    __shared__ unsigned part[PSIS_MT];
    __shared__ int s_bin, s_ok;
    const int tid = threadIdx.x;
    // bins 4095 .. 0, 16 per thread from the top: thread t owns bins 4095 - 16 t .. 4080 - 16 t
    unsigned loc[16], sum = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) { loc[j] = aux->hist[4095 - 16 * tid - j]; sum += loc[j]; }
    part[tid] = sum;
    if (tid == 0) { s_bin = -1; s_ok = 0; }
    __syncthreads();
    unsigned above = 0u;
    for (int t = 0; t < tid; ++t) above += part[t];              // elements in the bins above this thread's
    if (above < (unsigned)R && above + sum >= (unsigned)R) {      // exactly one thread
        unsigned cum = above;
        for (int j = 0; j < 16; ++j) {
            if (cum < (unsigned)R && cum + loc[j] >= (unsigned)R) { s_bin = 4095 - 16 * tid - j; s_ok = (cum + loc[j] <= (unsigned)TAILCAP); break; }
            cum += loc[j];
        }
    }
    __syncthreads();
    const int bin = s_bin;
    if (blockIdx.x == 0 && tid == 0) aux->done = (bin >= 0 && s_ok) ? 1u : 0u;
    if (bin < 0 || !s_ok) return;
    const int sh = pf_psis_shift0(~aux->kmin_inv, aux->kmax);
    for (long long i = (long long)blockIdx.x * PSIS_MT + tid; i < S; i += (long long)PSIS_MW * PSIS_MT) {
        const unsigned long long k = pf_key_of(lr[i]);
        if ((int)((k >> sh) & 0xFFFull) >= bin) {
            const unsigned slot = atomicAdd(&aux->cand_count, 1u);
            if (slot < (unsigned)TAILCAP) { aux->cand_key[slot] = k; aux->cand_idx[slot] = (unsigned)i; }
        }
    }
}
__device__ __forceinline__ bool pf_psis_is_tail(const PsisAux *aux, long long i, double x) {
    const unsigned long long k = pf_key_of(x);
    return aux->replaced && (k > aux->tk0 || (k == aux->tk0 && (unsigned long long)i > (unsigned long long)aux->ti0));
}
__global__ __launch_bounds__(PSIS_MT) void pf_psis_sum_kernel(long long S, const double *__restrict__ lr, PsisAux *aux) {
    __shared__ double red[PSIS_MT / 64];
    const double mx = aux->mx;
    double se = 0.0;
    if (isfinite(mx))
        for (long long i = (long long)blockIdx.x * PSIS_MT + threadIdx.x; i < S; i += (long long)PSIS_MW * PSIS_MT) {
            const double x = lr[i];
            if (!pf_psis_is_tail(aux, i, x)) se += exp(x - mx);
        }
    se = pf_block_sum1(se, red);
    if (threadIdx.x == 0) aux->partial[blockIdx.x] = se;
}
__global__ __launch_bounds__(PSIS_MT) void pf_psis_norm_kernel(long long S, const double *__restrict__ lr, const PsisAux *aux,
                                                              double *__restrict__ lw, double *__restrict__ wout,
                                                              double *__restrict__ out, const double *__restrict__ tail_val,
                                                              const unsigned *__restrict__ tail_idx) {
    if (tail_val == nullptr) { tail_val = aux->tail_val; tail_idx = aux->tail_idx; }     // (the large-tail route keeps them in its own buffers)
    const double mx = aux->mx;
    double se = aux->se_tail;
    for (int b = 0; b < PSIS_MW; ++b) se += aux->partial[b];      // fixed order: identical in every thread
    const double lse = isfinite(mx) ? mx + log(se) : mx;
    for (long long i = (long long)blockIdx.x * PSIS_MT + threadIdx.x; i < S; i += (long long)PSIS_MW * PSIS_MT) {
        const double x = lr[i];
        if (!pf_psis_is_tail(aux, i, x)) { const double v = x - lse; lw[i] = v; wout[i] = exp(v); }
    }
    if (blockIdx.x == 0) {
        if (aux->replaced)
            for (int t = threadIdx.x; t < aux->M; t += PSIS_MT) {
                const double v = tail_val[t] - lse;
                lw[tail_idx[t]] = v; wout[tail_idx[t]] = exp(v);
            }
        if (threadIdx.x == 0) { out[0] = aux->pareto_k; out[1] = (double)aux->M; out[2] = aux->sigma; out[3] = lse; }
    }
}

// out[0] = pareto_k, out[1] = tail length M, out[2] = sigma (scaled), out[3] = logsumexp
// MULTI: the passes over S run in the multi-workgroup kernels above; this workgroup sorts the candidates (or selects by itself when
// the candidate list overflowed), fits the tail and leaves the smoothed tail + normalisation constants in `aux`.
template <bool MULTI>
__global__ __launch_bounds__(PSIS_THREADS) void pf_psis_kernel(long long S, const double *__restrict__ lr,
                                                               double *__restrict__ lw, double *__restrict__ wout,
                                                               double *__restrict__ out, int M, PsisAux *aux) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    __shared__ uint64_t tkeys[TAILCAP];   // reused as double w[] after the sort
    __shared__ uint32_t tidx[TAILCAP];
    __shared__ SelectState st;
    __shared__ double red[PSIS_THREADS / 64];
    __shared__ double s_theta[128], s_ll[128];
    __shared__ double s_sigma, s_mu;

#if PSIS_PROF
    long long ps_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps_last = wall_clock64();
#endif
    double pareto_k = NAN;
    // lw is written once, at the end: everything but the M tail entries is lr - logsumexp, and the smoothed tail sits in LDS until then
    bool have_sel = false, replaced = false;
    uint64_t tk0 = 0; uint32_t ti0 = 0; double lmax_all = NAN, logu_all = NAN;
    if (tid == 0) s_sigma = NAN;
    __syncthreads();
    if (M >= 5 && M + 1 <= TAILCAP && (long long)(M + 1) <= S) {
        if (MULTI && aux->done) {
            // candidates (everything from the threshold bin upwards, <= TAILCAP, any order) -> ascending by (key, index); the M + 1
            // largest move to the front
            const int R = M + 1, cnt = (int)aux->cand_count;
            int npow = 1;
            while (npow < cnt) npow <<= 1;
            for (int t = tid; t < npow; t += nt) {
                tkeys[t] = t < cnt ? aux->cand_key[t] : ~0ull;
                tidx[t] = t < cnt ? aux->cand_idx[t] : 0xFFFFFFFFu;
            }
            __syncthreads();
            for (int size = 2; size <= npow; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = tid; t < npow / 2; t += nt) {
                        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const uint64_t ka = tkeys[lo], kb = tkeys[hi];
                        const uint32_t ia = tidx[lo], ib = tidx[hi];
                        const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                        if (a_gt_b == up) { tkeys[lo] = kb; tkeys[hi] = ka; tidx[lo] = ib; tidx[hi] = ia; }
                    }
                    __syncthreads();
                }
            uint64_t mk[(TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS];
            uint32_t mi[(TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS];
#pragma unroll
            for (int q = 0; q < (TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS; ++q) {
                const int t = tid + q * nt;
                if (t < R) { mk[q] = tkeys[cnt - R + t]; mi[q] = tidx[cnt - R + t]; }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS; ++q) {
                const int t = tid + q * nt;
                if (t < R) { tkeys[t] = mk[q]; tidx[t] = mi[q]; }
            }
            __syncthreads();
        } else pf_select_top_sorted(lr, S, M + 1, tkeys, tidx, &st);
        PS_STAMP(0);
        // tkeys[0] = cutoff, tkeys[1..M] = the M largest, ascending
        double *w = reinterpret_cast<double *>(tkeys);
        const double logu = pf_val_of(tkeys[0]);
        const double lmax = pf_val_of(tkeys[M]);
        have_sel = true; tk0 = tkeys[0]; ti0 = tidx[0]; lmax_all = lmax; logu_all = logu;
        double bad = 0.0;
        for (int t = 1 + tid; t <= M; t += nt) if (!isfinite(pf_val_of(tkeys[t]))) bad = 1.0;
        bad = pf_block_sum1(bad, red);
        __syncthreads();
        if (bad == 0.0) {
            const double mu_s = exp(logu - lmax);
            double vals[(TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS];
            double nz = 0.0;
#pragma unroll
            for (int q = 0; q < (TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS; ++q) {
                const int t = tid + q * nt;          // tail element t (0-based) lives at tkeys[t + 1]
                vals[q] = (t < M) ? (exp(pf_val_of(tkeys[t + 1]) - lmax) - mu_s) : 0.0;
                if (t < M && vals[q] != 0.0) nz = 1.0;
            }
            nz = pf_block_sum1(nz, red);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (TAILCAP + PSIS_THREADS - 1) / PSIS_THREADS; ++q) {
                const int t = tid + q * nt;
                if (t < M) w[t] = vals[q];           // tail idx t is tidx[t + 1]
            }
            __syncthreads();
            PS_STAMP(1);
            if (nz > 0.0) {
                // ---- Zhang & Stephens (2009) profile likelihood on a grid of m theta values
                const int mest = 30 + (int)floor(sqrt((double)M));
                const double xstar = w[(M + 2) / 4 - 1], xmax = w[M - 1];
                for (int i = wave; i < mest; i += nw) {
                    const double p = ((double)(i + 1) - 0.5) / (double)mest;
                    const double theta = 1.0 / xmax + (1.0 - sqrt(1.0 / p)) / (3.0 * xstar);
                    double kk = 0.0;
                    for (int t = lane; t < M; t += 64) kk += log1p(-theta * w[t]);
                    kk = pf_wave_sum(kk) / (double)M;
                    if (lane == 0) {
                        s_theta[i] = theta;
                        s_ll[i] = (double)M * (log(-theta / kk) - kk - 1.0);
                    }
                }
                __syncthreads();
                PS_STAMP(2);
                if (tid < 64) {                                    // posterior-mean theta: wave 0, lanes over the grid
                    double lmx = -INFINITY;
                    for (int i = lane; i < mest; i += 64) lmx = fmax(lmx, s_ll[i]);
                    lmx = pf_wave_max(lmx);
                    double ws = 0.0, ts = 0.0;
                    for (int i = lane; i < mest; i += 64) { const double e = exp(s_ll[i] - lmx); ws += e; ts += e * s_theta[i]; }
                    ws = pf_wave_sum(ws); ts = pf_wave_sum(ts);
                    if (lane == 0) s_mu = ts / ws;
                }
                __syncthreads();
                const double th = s_mu;
                double kk = 0.0;
                for (int t = tid; t < M; t += nt) kk += log1p(-th * w[t]);
                kk = pf_block_sum1(kk, red) / (double)M;
                const double sigma = -kk / th;
                double kadj = kk;
                if (isfinite(kk)) kadj = (kk * (double)M + 5.0) / ((double)M + 10.0);   // prior adjustment
                pareto_k = kadj;
                PS_STAMP(3);
                if (isfinite(kadj) && isfinite(sigma)) {
                    __syncthreads();                               // everybody is done reading w[] (the kk sum above)
                    for (int t = tid; t < M; t += nt) {
                        const double p = ((double)(t + 1) - 0.5) / (double)M;
                        const double nl = -log1p(-p);
                        const double z = (kadj == 0.0) ? nl : expm1(kadj * nl) / kadj;
                        double v = log(sigma * z + mu_s);
                        if (v > 0.0) v = 0.0;
                        w[t] = v + lmax;                           // smoothed log weight of element tidx[t + 1]
                    }
                    replaced = true;
                }
                if (tid == 0) s_sigma = sigma;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    PS_STAMP(4);
    // ---- log-normalise: lw = (smoothed) log ratios - logsumexp; weights = exp(lw).  Two passes over lr: the maximum needs none --
    //      the largest untouched element is the cutoff (or the overall maximum when nothing was replaced).
    const double *wt = reinterpret_cast<const double *>(tkeys);
    auto is_tail = [&](long long i, double x) {
        const uint64_t k = pf_key_of(x);
        return replaced && (k > tk0 || (k == tk0 && (uint64_t)i > (uint64_t)ti0));
    };
    double mx = -INFINITY;
    if (replaced) {
        for (int t = tid; t < M; t += nt) mx = fmax(mx, wt[t]);
        mx = fmax(pf_block_max1(mx, red), logu_all);
    } else if (have_sel) mx = lmax_all;
    else {
        pf_foreach_batched(lr, S, [&](long long, double x) { mx = fmax(mx, x); });
        mx = pf_block_max1(mx, red);
    }
    __syncthreads();
    if (MULTI) {                                                   // hand over to pf_psis_sum_kernel / pf_psis_norm_kernel
        double se = 0.0;
        if (replaced && isfinite(mx)) for (int t = tid; t < M; t += nt) se += exp(wt[t] - mx);
        se = pf_block_sum1(se, red);
        if (replaced) for (int t = tid; t < M; t += nt) { aux->tail_val[t] = wt[t]; aux->tail_idx[t] = tidx[t + 1]; }
        if (tid == 0) {
            aux->mx = mx; aux->se_tail = se; aux->replaced = replaced ? 1u : 0u; aux->tk0 = tk0; aux->ti0 = ti0;
            aux->pareto_k = pareto_k; aux->sigma = s_sigma; aux->M = M;
        }
#if PSIS_PROF
        PS_STAMP(5);
        if (tid == 0) printf("PSIS_PROF multi S %lld M %d (10 ns): select+sort %lld vals %lld grid %lld mean+k %lld smooth %lld hand-over %lld\n", S, M, ps_t[0], ps_t[1], ps_t[2], ps_t[3], ps_t[4], ps_t[5]);
#endif
        return;
    }
    double se = 0.0;
    if (isfinite(mx)) {
        pf_foreach_batched(lr, S, [&](long long i, double x) { if (!is_tail(i, x)) se += exp(x - mx); });
        if (replaced) for (int t = tid; t < M; t += nt) se += exp(wt[t] - mx);
    }
    se = pf_block_sum1(se, red);
    const double lse = isfinite(mx) ? mx + log(se) : mx;
    pf_foreach_batched(lr, S, [&](long long i, double x) {
        if (!is_tail(i, x)) {
            const double v = x - lse;
            lw[i] = v;
            wout[i] = exp(v);
        }
    });
    if (replaced) for (int t = tid; t < M; t += nt) {
        const double v = wt[t] - lse;
        lw[tidx[t + 1]] = v;
        wout[tidx[t + 1]] = exp(v);
    }
    if (tid == 0) { out[0] = pareto_k; out[1] = (double)M; out[2] = s_sigma; out[3] = lse; }
#if PSIS_PROF
    PS_STAMP(5);
    if (tid == 0) printf("PSIS_PROF single S %lld M %d (10 ns): select+sort %lld vals %lld grid %lld mean+k %lld smooth %lld normalise %lld\n", S, M, ps_t[0], ps_t[1], ps_t[2], ps_t[3], ps_t[4], ps_t[5]);
#endif
}

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t pf_weight_to_fixed(double w) {
    if (!(w > 0.0)) return 0ull;
    if (w >= 1.0) return 1ull << 62;
    return (uint64_t)floor(w * 4611686018427387904.0);   // 2^62
}
__device__ __forceinline__ uint64_t pf_uniform_to_bits(double u) {
    return ((uint64_t)floor(u * 9007199254740992.0)) << 11;
}
// inclusive scan of a u64 across the wave (lane i gets the sum of lanes 0..i)
__device__ __forceinline__ uint64_t pf_wave_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned lo = __shfl_up((unsigned)v, off, 64), hi = __shfl_up((unsigned)(v >> 32), off, 64);
        if (lane >= off) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
// single workgroup.  Wave w owns a contiguous run of 64-element tiles: coalesced loads, a wave-level scan per tile and a running
// carry (integer sums: the table does not depend on the order of these partial sums); pass 1 = run totals, pass 2 = scan + store.
// (Round 1 gave every thread a contiguous chunk: 64 different cache lines per load instruction, 0.15 ms at S = 64 000.)
__global__ __launch_bounds__(PSIS_THREADS) void pf_cdf_kernel(long long S, const double *__restrict__ w,
                                                              uint64_t *__restrict__ cdf) {
    // (round 4: a lane owns FOUR consecutive elements of a 256-element tile -- a quarter of the wave scans and of the dependent load
    //  round trips of the 64-element tiles; 0.066 -> 0.02 ms at S = 64 000.  Integer sums: the same table.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    __shared__ uint64_t part[PSIS_THREADS / 64];
    const long long ntile = (S + 255) >> 8, per = (ntile + nw - 1) / nw;
    const long long t0 = (long long)wave * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    auto load4 = [&](const long long t, uint64_t (&x)[4]) {
        const long long i = (t << 8) + 4 * lane;
        if (i + 3 < S) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = pf_weight_to_fixed(w[i + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (i + e < S) ? pf_weight_to_fixed(w[i + e]) : 0ull;
        }
    };
    uint64_t s = 0;
    for (long long t = t0; t < t1; ++t) {
        uint64_t x[4];
        load4(t, x);
        s += (x[0] + x[1]) + (x[2] + x[3]);
    }
    s = pf_wave_scan_u64(s, lane);
    if (lane == 63) part[wave] = s;
    __syncthreads();
    uint64_t carry = 0;
    for (int v = 0; v < wave; ++v) carry += part[v];
    for (long long t = t0; t < t1; ++t) {
        uint64_t x[4];
        load4(t, x);
        x[1] += x[0]; x[2] += x[1]; x[3] += x[2];                      // inclusive prefix inside the lane
        const uint64_t incl = pf_wave_scan_u64(x[3], lane);           // ... of the lane totals
        const uint64_t base = carry + (incl - x[3]);
        const long long i = (t << 8) + 4 * lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i + e < S) cdf[i + e] = base + x[e];
        carry += ((uint64_t)__shfl((unsigned)(incl >> 32), 63, 64) << 32) | (uint64_t)__shfl((unsigned)incl, 63, 64);
    }
}
__global__ void pf_sample_kernel(long long S, long long ndraws, int weighted, uint64_t seed,
                                 const double *__restrict__ uniforms, const uint64_t *__restrict__ cdf,
                                 int64_t *__restrict__ idx, int *__restrict__ err) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ndraws) return;
    const uint64_t R = uniforms ? pf_uniform_to_bits(uniforms[t]) : pf_rand_u64(seed, (uint64_t)t, 1u);
    if (!weighted) { idx[t] = (int64_t)__umul64hi(R, (uint64_t)S); return; }
    const uint64_t Q = cdf[S - 1];
    if (Q == 0) { if (t == 0) *err = 1; idx[t] = 0; return; }
    const uint64_t r = __umul64hi(R, Q);
    long long lo = 0, hi = S - 1;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1; }
    idx[t] = lo;
}

// Efraimidis-Spirakis keys (negated so that the selection of the LARGEST picks the smallest keys)
__global__ void pf_norep_keys_kernel(long long S, int weighted, uint64_t seed, const double *__restrict__ w,
                                     double *__restrict__ negkey) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t R = pf_rand_u64(seed, (uint64_t)i, 2u);
    const double u = ((double)(R >> 11) + 0.5) * 1.1102230246251565404e-16;
    const double wi = weighted ? w[i] : 1.0;
    negkey[i] = (wi > 0.0) ? -(-log(u) / wi) : -INFINITY;
}
__global__ __launch_bounds__(PSIS_THREADS) void pf_norep_select_kernel(long long S, int ndraws,
                                                                       const double *__restrict__ negkey,
                                                                       int64_t *__restrict__ idx, int *__restrict__ err) {
    __shared__ uint64_t tkeys[TAILCAP];
    __shared__ uint32_t tidx[TAILCAP];
    __shared__ SelectState st;
    pf_select_top_sorted(negkey, S, ndraws, tkeys, tidx, &st);
    // ascending in -key  ==  descending key; emit in ascending key order (smallest key first)
    for (int t = threadIdx.x; t < ndraws; t += blockDim.x) {
        idx[t] = (int64_t)tidx[ndraws - 1 - t];
        if (!isfinite(pf_val_of(tkeys[ndraws - 1 - t]))) *err = 1;   // not enough positive weights
    }
}

// ---- replace = false beyond TAILCAP draws: full (key, index) bitonic sort in global memory ------------------------------------
// keys[i] = order-preserving u64 of the Efraimidis-Spirakis key (-negkey), padded with ~0; ascending (key, index) order equals
// the oracle's sort, so the first ndraws entries are the sample.  log2(n)(log2(n)+1)/2 launches: a rare, large-request path.
__global__ void pf_sortkeys_init_kernel(long long S, long long n2, const double *__restrict__ negkey, uint64_t *__restrict__ keys,
                                        uint32_t *__restrict__ ids) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    keys[i] = (i < S) ? pf_key_of(-negkey[i]) : ~0ull;
    ids[i] = (uint32_t)i;
}
__global__ void pf_bitonic_step_kernel(long long n2, long long j, long long k, uint64_t *__restrict__ keys, uint32_t *__restrict__ ids) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const long long l = i ^ j;
    if (l <= i) return;
    const uint64_t ka = keys[i], kb = keys[l];
    const uint32_t ia = ids[i], ib = ids[l];
    const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
    const bool up = (i & k) == 0;
    if (a_gt_b == up) { keys[i] = kb; keys[l] = ka; ids[i] = ib; ids[l] = ia; }
}
__global__ void pf_sorted_take_kernel(long long ndraws, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ ids,
                                      int64_t *__restrict__ idx, int *__restrict__ err) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ndraws) return;
    idx[t] = (int64_t)ids[t];
    if (!isfinite(pf_val_of(keys[t]))) *err = 1;                   // not enough positive weights
}

// ---- StatsBase.direct_sample! compatible index selection (src/resample.jl:61-66 -> StatsBase.sample(rng, wv)) ------------------
//   t = rand(rng) * sum(wv);  i = 1; cw = wv[1];  while cw < t && i < n:  i += 1; cw += wv[i]
// ProbabilityWeights(w, 1) fixes sum(wv) = 1.  The running sum is SEQUENTIAL fp64 (left to right), so the prefix sums are
// produced by one thread in exactly that order (chunks staged through LDS by the whole workgroup); with cw non-decreasing the
// loop's answer is the first i with cw_i >= t, found by binary search.  Given the uniforms a Julia host drew with its own
// rng, the indices equal StatsBase's bit for bit.
__global__ __launch_bounds__(PSIS_THREADS) void pf_seqcdf_kernel(long long S, const double *__restrict__ w, double *__restrict__ cw) {
    __shared__ double buf[4096];
    __shared__ double carry;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) carry = 0.0;
    for (long long c0 = 0; c0 < S; c0 += 4096) {
        const int n = (int)((S - c0 < 4096) ? S - c0 : 4096);
        __syncthreads();
        for (int i = tid; i < n; i += nt) buf[i] = w[c0 + i];
        __syncthreads();
        if (tid == 0) {
            double acc = carry;
            if (c0 == 0) { acc = buf[0]; buf[0] = acc; for (int i = 1; i < n; ++i) { acc += buf[i]; buf[i] = acc; } }   // cw = wv[1]
            else for (int i = 0; i < n; ++i) { acc += buf[i]; buf[i] = acc; }
            carry = acc;
        }
        __syncthreads();
        for (int i = tid; i < n; i += nt) cw[c0 + i] = buf[i];
    }
}
__global__ void pf_direct_sample_kernel(long long S, long long ndraws, const double *__restrict__ uniforms, const double *__restrict__ cw,
                                        int64_t *__restrict__ idx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ndraws) return;
    const double thr = uniforms[t] * 1.0;                          // rand(rng) * wv.sum
    long long lo = 0, hi = S - 1;                                  // first i with cw[i] >= thr, else the last index
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (cw[mid] < thr) lo = mid + 1; else hi = mid; }
    idx[t] = lo;
}

// out[:, t] = owned(idx[t]) ? pool[:, idx[t] - col_offset] : 0     (skip_unowned: columns of other ranks are left untouched -- the owner-only
// assembly of the multi-GPU result, csrc/comm_rccl.hip: every rank writes ITS columns of one shared, host-resident result)
__global__ void pf_gather_kernel(int d, long long ndraws, long long ncols_local, long long col_offset,
                                 const int64_t *__restrict__ idx, const double *__restrict__ pool,
                                 double *__restrict__ out, int skip_unowned) {
    const long long t = blockIdx.x;
    const long long g = idx[t] - col_offset;
    const bool own = (g >= 0 && g < ncols_local);
    if (!own && skip_unowned) return;
    for (int i = threadIdx.x; i < d; i += blockDim.x)
        out[(size_t)t * d + i] = own ? pool[(size_t)g * d + i] : 0.0;
}
// compact owner gather: out[:, j] = pool[:, idx[pos[j]] - col_offset], j < n   (the columns a rank SENDS to the root, in selection order)
__global__ void pf_gather_pos_kernel(int d, long long ncols_local, long long col_offset, const int64_t *__restrict__ idx,
                                     const int64_t *__restrict__ pos, const double *__restrict__ pool, double *__restrict__ out) {
    const long long j = blockIdx.x;
    const long long g = idx[pos[j]] - col_offset;
    const bool own = (g >= 0 && g < ncols_local);                      // (always true by construction; a foreign column would be NaN, never stale)
    for (int i = threadIdx.x; i < d; i += blockDim.x)
        out[(size_t)j * d + i] = own ? pool[(size_t)g * d + i] : __longlong_as_double(0x7FF8000000000000ll);
}
// out[:, pos[j]] = in[:, j]   (the root places the columns it received, rank by rank, at their selection positions)
__global__ void pf_scatter_cols_kernel(int d, const int64_t *__restrict__ pos, const double *__restrict__ in, double *__restrict__ out) {
    const long long j = blockIdx.x;
    const long long t = pos[j];
    for (int i = threadIdx.x; i < d; i += blockDim.x) out[(size_t)t * d + i] = in[(size_t)j * d + i];
}
// order-independent 52-bit digest of the selected indices (exact in a double): the ranks of a process-per-GPU group compare it
__global__ void pf_idx_digest_kernel(long long ndraws, const int64_t *__restrict__ idx, double *__restrict__ out4, double local_err) {
    __shared__ unsigned long long sh[256];
    unsigned long long h = 0;
    for (long long t = threadIdx.x; t < ndraws; t += blockDim.x) {
        unsigned long long v = (unsigned long long)idx[t] * 0x9E3779B97F4A7C15ull + (unsigned long long)t * 0xC2B2AE3D27D4EB4Full;
        v ^= v >> 29; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 32;
        h += v;
    }
    sh[threadIdx.x] = h;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double hv = (double)(sh[0] >> 12);                        // 52 bits: exactly representable
        out4[0] = local_err; out4[1] = hv; out4[2] = -hv; out4[3] = 0.0;
    }
}

// ---- PSIS with a tail beyond the LDS capacity (M + 1 > TAILCAP, i.e. pools of more than 1 863 225 draws; round 4) -----------------
// The (key, index) pairs of ALL log ratios are sorted in global memory (the bitonic kernels above; padding keys 0 sort to the front),
// the M + 1 largest are then the last entries, ascending by (key, index) like the LDS route's tail.  One workgroup fits the tail with
// the arithmetic of pf_psis_kernel (same loops, same summation order; w[] lives in a global buffer instead of LDS), the passes over
// S are the multi-workgroup kernels of the regular route.
__global__ void pf_sortkeys_init_lr_kernel(long long S, long long n2, const double *__restrict__ lr, uint64_t *__restrict__ keys,
                                           uint32_t *__restrict__ ids) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    keys[i] = (i < S) ? pf_key_of(lr[i]) : 0ull;
    ids[i] = (i < S) ? (uint32_t)i : 0u;
}
__global__ __launch_bounds__(PSIS_THREADS) void pf_psis_bigtail_kernel(int M, const uint64_t *__restrict__ tkeys,
                                                                       const uint32_t *__restrict__ tidx, double *__restrict__ w,
                                                                       PsisAux *aux) {
    // tkeys[0] = cutoff, tkeys[1..M] = the M largest, ascending
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    __shared__ double red[PSIS_THREADS / 64];
    __shared__ double s_theta[640], s_ll[640];                     // 30 + sqrt(M) grid points: M <= 196 608 at S < 2^32
    __shared__ double s_sigma, s_mu;
    double pareto_k = NAN;
    bool replaced = false;
    const double logu = pf_val_of(tkeys[0]), lmax = pf_val_of(tkeys[M]);
    if (tid == 0) s_sigma = NAN;
    __syncthreads();
    double bad = 0.0;
    for (int t = 1 + tid; t <= M; t += nt) if (!isfinite(pf_val_of(tkeys[t]))) bad = 1.0;
    bad = pf_block_sum1(bad, red);
    __syncthreads();
    if (bad == 0.0) {
        const double mu_s = exp(logu - lmax);
        double nz = 0.0;
        for (int t = tid; t < M; t += nt) {
            const double v = exp(pf_val_of(tkeys[t + 1]) - lmax) - mu_s;
            w[t] = v;
            if (v != 0.0) nz = 1.0;
        }
        nz = pf_block_sum1(nz, red);
        __threadfence_block();
        __syncthreads();
        if (nz > 0.0) {
            const int mest = 30 + (int)floor(sqrt((double)M));
            const double xstar = w[(M + 2) / 4 - 1], xmax = w[M - 1];
            for (int i = wave; i < mest; i += nw) {
                const double p = ((double)(i + 1) - 0.5) / (double)mest;
                const double theta = 1.0 / xmax + (1.0 - sqrt(1.0 / p)) / (3.0 * xstar);
                double kk = 0.0;
                for (int t = lane; t < M; t += 64) kk += log1p(-theta * w[t]);
                kk = pf_wave_sum(kk) / (double)M;
                if (lane == 0) {
                    s_theta[i] = theta;
                    s_ll[i] = (double)M * (log(-theta / kk) - kk - 1.0);
                }
            }
            __syncthreads();
            if (tid < 64) {
                double lmx = -INFINITY;
                for (int i = lane; i < mest; i += 64) lmx = fmax(lmx, s_ll[i]);
                lmx = pf_wave_max(lmx);
                double ws = 0.0, ts = 0.0;
                for (int i = lane; i < mest; i += 64) { const double e = exp(s_ll[i] - lmx); ws += e; ts += e * s_theta[i]; }
                ws = pf_wave_sum(ws); ts = pf_wave_sum(ts);
                if (lane == 0) s_mu = ts / ws;
            }
            __syncthreads();
            const double th = s_mu;
            double kk = 0.0;
            for (int t = tid; t < M; t += nt) kk += log1p(-th * w[t]);
            kk = pf_block_sum1(kk, red) / (double)M;
            const double sigma = -kk / th;
            double kadj = kk;
            if (isfinite(kk)) kadj = (kk * (double)M + 5.0) / ((double)M + 10.0);   // prior adjustment
            pareto_k = kadj;
            if (isfinite(kadj) && isfinite(sigma)) {
                __syncthreads();
                for (int t = tid; t < M; t += nt) {
                    const double p = ((double)(t + 1) - 0.5) / (double)M;
                    const double nl = -log1p(-p);
                    const double z = (kadj == 0.0) ? nl : expm1(kadj * nl) / kadj;
                    double v = log(sigma * z + mu_s);
                    if (v > 0.0) v = 0.0;
                    w[t] = v + lmax;
                }
                replaced = true;
            }
            if (tid == 0) s_sigma = sigma;
        }
    }
    __threadfence_block();
    __syncthreads();
    double mx = lmax;
    if (replaced) {
        mx = -INFINITY;
        for (int t = tid; t < M; t += nt) mx = fmax(mx, w[t]);
        mx = fmax(pf_block_max1(mx, red), logu);
    }
    __syncthreads();
    double se = 0.0;
    if (replaced && isfinite(mx)) for (int t = tid; t < M; t += nt) se += exp(w[t] - mx);
    se = pf_block_sum1(se, red);
    if (tid == 0) {
        aux->mx = mx; aux->se_tail = se; aux->replaced = replaced ? 1u : 0u; aux->tk0 = tkeys[0]; aux->ti0 = tidx[0];
        aux->pareto_k = pareto_k; aux->sigma = s_sigma; aux->M = M;
    }
}

// ---------------------------------------------------------------------------------------------------
static long long psis_tail_length(long long S) {
    long long a = (S + 4) / 5;
    long long b = (long long)ceil(3.0 * sqrt((double)S));
    return a < b ? a : b;
}

int32_t pf_launch_psis(pfmi_ctx *c, const double *d_lr, int64_t S) {
    PF_CHECK(S > 0, PFMI_ERR_ARG, "psis: empty log-ratio vector");
    PF_CHECK(S < (1ll << 32), PFMI_ERR_UNSUPPORTED, "psis: S too large");
    const long long M = psis_tail_length(S);
    PF_TRY(c->lw.ensure(sizeof(double) * S));
    PF_TRY(c->w.ensure(sizeof(double) * S));
    PF_TRY(c->psis_out.ensure(sizeof(double) * 4));
    pf_kernel_begin(c);
    const char *force = pf_debug_get("PFMI_PSIS_KERNEL");              // "single": the one-workgroup kernel for every S (tests); "big": the large-tail route
    if (M + 1 > TAILCAP || (force && force[0] == 'b' && M >= 5 && M + 1 <= S)) {
        // tail beyond the LDS capacity: global sort of every (key, index) pair, tail fit on the sorted run
        long long n2 = 1;
        while (n2 < S) n2 <<= 1;
        PF_TRY(c->sortk.ensure(sizeof(uint64_t) * (size_t)n2));
        PF_TRY(c->sorti.ensure(sizeof(uint32_t) * (size_t)n2));
        PF_TRY(c->psis_aux.ensure(sizeof(PsisAux)));
        PF_TRY(c->scratch.ensure(sizeof(double) * (size_t)(M + 1)));
        PsisAux *aux = c->psis_aux.as<PsisAux>();
        PF_HIP(hipMemsetAsync(aux, 0, offsetof(PsisAux, cand_key), c->stream));
        const dim3 grid((unsigned)((n2 + 255) / 256));
        hipLaunchKernelGGL(pf_sortkeys_init_lr_kernel, grid, dim3(256), 0, c->stream, (long long)S, n2, d_lr, c->sortk.as<uint64_t>(),
                           c->sorti.as<uint32_t>());
        for (long long k = 2; k <= n2; k <<= 1)
            for (long long j = k >> 1; j > 0; j >>= 1)
                hipLaunchKernelGGL(pf_bitonic_step_kernel, grid, dim3(256), 0, c->stream, n2, j, k, c->sortk.as<uint64_t>(),
                                   c->sorti.as<uint32_t>());
        const uint64_t *tk = c->sortk.as<uint64_t>() + (n2 - (M + 1));
        const uint32_t *ti = c->sorti.as<uint32_t>() + (n2 - (M + 1));
        hipLaunchKernelGGL(pf_psis_bigtail_kernel, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (int)M, tk, ti, c->scratch.as<double>(), aux);
        hipLaunchKernelGGL(pf_psis_sum_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux);
        hipLaunchKernelGGL(pf_psis_norm_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux,
                           c->lw.as<double>(), c->w.as<double>(), c->psis_out.as<double>(), (const double *)c->scratch.as<double>(),
                           (const unsigned *)(ti + 1));
    } else if (S >= PSIS_MULTI_MIN && M >= 5 && !(force && force[0] == 's')) {
        PF_TRY(c->psis_aux.ensure(sizeof(PsisAux)));
        PsisAux *aux = c->psis_aux.as<PsisAux>();
        PF_HIP(hipMemsetAsync(aux, 0, offsetof(PsisAux, cand_key), c->stream));
        hipLaunchKernelGGL(pf_psis_minmax_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux);
        hipLaunchKernelGGL(pf_psis_hist_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux);
        hipLaunchKernelGGL(pf_psis_compact_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux, (int)M + 1);
        hipLaunchKernelGGL(pf_psis_kernel<true>, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (long long)S, d_lr,
                           c->lw.as<double>(), c->w.as<double>(), c->psis_out.as<double>(), (int)M, aux);
        hipLaunchKernelGGL(pf_psis_sum_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux);
        hipLaunchKernelGGL(pf_psis_norm_kernel, dim3(PSIS_MW), dim3(PSIS_MT), 0, c->stream, (long long)S, d_lr, aux,
                           c->lw.as<double>(), c->w.as<double>(), c->psis_out.as<double>(), (const double *)nullptr, (const unsigned *)nullptr);
    } else
        hipLaunchKernelGGL(pf_psis_kernel<false>, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (long long)S, d_lr,
                           c->lw.as<double>(), c->w.as<double>(), c->psis_out.as<double>(), (int)M, (PsisAux *)nullptr);
    pf_kernel_end(c, "psis");
    PF_HIP(hipGetLastError());
    c->S_w = S;
    return PFMI_OK;
}

int32_t pf_enqueue_resample(pfmi_ctx *c, int64_t S, int64_t ndraws, int importance, int replace, uint64_t seed,
                            const double *d_uniforms) {
    PF_TRY(c->idx.ensure(sizeof(int64_t) * (ndraws > 0 ? ndraws : 1)));
    PF_TRY(c->scratch.ensure(sizeof(double) * (S > 0 ? S : 1) + 64));
    PF_TRY(c->rs_err.ensure(sizeof(int)));
    int *d_err = c->rs_err.as<int>();
    PF_HIP(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    pf_kernel_begin(c);
    if (replace) {
        if (importance) {
            PF_TRY(c->cdf.ensure(sizeof(uint64_t) * S));
            hipLaunchKernelGGL(pf_cdf_kernel, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (long long)S,
                               c->w.as<double>(), c->cdf.as<uint64_t>());
        }
        hipLaunchKernelGGL(pf_sample_kernel, dim3((unsigned)((ndraws + 255) / 256)), dim3(256), 0, c->stream,
                           (long long)S, (long long)ndraws, importance, seed, d_uniforms, c->cdf.as<uint64_t>(),
                           c->idx.as<int64_t>(), d_err);
    } else {
        PF_CHECK(ndraws <= S, PFMI_ERR_ARG, "cannot draw %lld without replacement from %lld", (long long)ndraws,
                 (long long)S);
        hipLaunchKernelGGL(pf_norep_keys_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream,
                           (long long)S, importance, seed, c->w.as<double>(), c->scratch.as<double>());
        if (ndraws <= TAILCAP) {                                   // radix select + LDS sort of the sample (one workgroup)
            hipLaunchKernelGGL(pf_norep_select_kernel, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (long long)S,
                               (int)ndraws, c->scratch.as<double>(), c->idx.as<int64_t>(), d_err);
        } else {                                                   // large request: sort every (key, index) pair
            long long n2 = 1;
            while (n2 < S) n2 <<= 1;
            PF_TRY(c->sortk.ensure(sizeof(uint64_t) * (size_t)n2));
            PF_TRY(c->sorti.ensure(sizeof(uint32_t) * (size_t)n2));
            const dim3 grid((unsigned)((n2 + 255) / 256));
            hipLaunchKernelGGL(pf_sortkeys_init_kernel, grid, dim3(256), 0, c->stream, (long long)S, n2, c->scratch.as<double>(),
                               c->sortk.as<uint64_t>(), c->sorti.as<uint32_t>());
            for (long long k = 2; k <= n2; k <<= 1)
                for (long long j = k >> 1; j > 0; j >>= 1)
                    hipLaunchKernelGGL(pf_bitonic_step_kernel, grid, dim3(256), 0, c->stream, n2, j, k, c->sortk.as<uint64_t>(),
                                       c->sorti.as<uint32_t>());
            hipLaunchKernelGGL(pf_sorted_take_kernel, dim3((unsigned)((ndraws + 255) / 256)), dim3(256), 0, c->stream,
                               (long long)ndraws, c->sortk.as<uint64_t>(), c->sorti.as<uint32_t>(), c->idx.as<int64_t>(), d_err);
        }
    }
    pf_kernel_end(c, "resample");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

// blocks: reads the error flag of the last pf_enqueue_resample
int32_t pf_resample_check(pfmi_ctx *c) {
    int err = 0;
    PF_TRY(pf_download(c, &err, c->rs_err.p, sizeof(int)));
    PF_TRY(pf_stream_sync(c));
    PF_CHECK(err == 0, PFMI_ERR_NUMERIC, "resample: weights are all zero / not enough positive weights");
    return PFMI_OK;
}

int32_t pf_launch_resample(pfmi_ctx *c, int64_t S, int64_t ndraws, int importance, int replace, uint64_t seed,
                           const double *d_uniforms) {
    PF_TRY(pf_enqueue_resample(c, S, ndraws, importance, replace, seed, d_uniforms));
    return pf_resample_check(c);
}

int32_t pf_launch_resample_direct(pfmi_ctx *c, int64_t S, int64_t ndraws, const double *d_uniforms) {
    PF_TRY(c->idx.ensure(sizeof(int64_t) * (ndraws > 0 ? ndraws : 1)));
    PF_TRY(c->scratch.ensure(sizeof(double) * (size_t)S));
    pf_kernel_begin(c);
    hipLaunchKernelGGL(pf_seqcdf_kernel, dim3(1), dim3(PSIS_THREADS), 0, c->stream, (long long)S, c->w.as<double>(), c->scratch.as<double>());
    hipLaunchKernelGGL(pf_direct_sample_kernel, dim3((unsigned)((ndraws + 255) / 256)), dim3(256), 0, c->stream, (long long)S,
                       (long long)ndraws, d_uniforms, c->scratch.as<double>(), c->idx.as<int64_t>());
    pf_kernel_end(c, "resample");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_gather(pfmi_ctx *c, int64_t ndraws, const int64_t *d_idx, int64_t col_offset, double *d_out, bool skip_unowned) {
    if (ndraws <= 0) return PFMI_OK;
    pf_kernel_begin(c);
    hipLaunchKernelGGL(pf_gather_kernel, dim3((unsigned)ndraws), dim3(256), 0, c->stream, c->d, (long long)ndraws,
                       (long long)(c->K * c->N_r), (long long)col_offset, d_idx, c->pool.as<double>(), d_out, skip_unowned ? 1 : 0);
    pf_kernel_end(c, "resample");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
int32_t pf_launch_gather_pos(pfmi_ctx *c, int64_t n, const int64_t *d_idx, const int64_t *d_pos, int64_t col_offset, double *d_out) {
    if (n <= 0) return PFMI_OK;
    pf_kernel_begin(c);
    hipLaunchKernelGGL(pf_gather_pos_kernel, dim3((unsigned)n), dim3(256), 0, c->stream, c->d, (long long)(c->K * c->N_r), (long long)col_offset,
                       d_idx, d_pos, c->pool.as<double>(), d_out);
    pf_kernel_end(c, "resample");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
int32_t pf_launch_scatter_cols(pfmi_ctx *c, int64_t n, const int64_t *d_pos, const double *d_in, double *d_out) {
    if (n <= 0) return PFMI_OK;
    pf_kernel_begin(c);
    hipLaunchKernelGGL(pf_scatter_cols_kernel, dim3((unsigned)n), dim3(256), 0, c->stream, c->d, d_pos, d_in, d_out);
    pf_kernel_end(c, "resample");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
int32_t pf_launch_idx_digest(pfmi_ctx *c, int64_t ndraws, const int64_t *d_idx, double *d_out4, double local_err) {
    hipLaunchKernelGGL(pf_idx_digest_kernel, dim3(1), dim3(256), 0, c->stream, (long long)ndraws, d_idx, d_out4, local_err);
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
