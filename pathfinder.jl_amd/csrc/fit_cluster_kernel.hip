// fit_cluster_kernel.hip -- fit_mvnormals for LARGE d (d > 1024): lbfgs_inverse_hessian + pdfactorize + mean
// (reference src/inverse_hessian.jl:98-133, src/woodbury.jl:201-207, src/mvnormal.jl:14-21), same operations and the same
// LAPACK reflector convention as pf_fit_reg_kernel (fit_kernels.hip), with the d x KC block held in the REGISTERS of a
// CLUSTER of cooperating workgroups instead of being swept through L2/HBM once (twice) per Householder column.
//
// Why: the memory-resident kernel re-reads and re-writes the whole d x KC block for every column: 3 x 1.6 MB x 20 columns
// = 96 MB of traffic per fit at d = 1e4, J = 10 against 3.6 MB of algorithmic bytes -- it ran at 1.2 % of the HBM roofline
// (36 us per fit, 1.16 s for the 32 000 fits of one GPU's share of config 5).  Here member w of a cluster owns rows
// [w R, (w+1) R), R = NT * RPT, in registers for the whole factorisation; what used to be a block reduction becomes a block
// reduction + one small exchange through global memory:
//     partial sums -> cl_buf[parity][member][:],  release fence,  arrival counter += 1,  spin until all members arrived,
//     every member adds the partials in member order (=> bit-identical totals in all members, so the replicated O(m^3) work --
//     D, T, C = I + R D R', Cholesky -- stays consistent without further communication).
// The kernel is launched COOPERATIVELY (hipLaunchCooperativeKernel: all workgroups co-resident, so the spin waits cannot
// deadlock) with a persistent grid; cluster c walks over fits c, c + nclusters, ...  Members of a cluster are 8 workgroup ids
// apart, i.e. on the same XCD under the round-robin workgroup dispatch, so the exchange stays inside one L2.
#include "pfmi_common.h"
#include "fit_args.h"

#define CL_NVMAX 72
#define CL_MAXWG 32                    // members per cluster (d <= 32 * NT * RPT)

struct ClusterCtx {
    int nwg, member;
    unsigned long long *counter;
    double *buf;                 // [2][nwg][CL_NVMAX]
    unsigned long long phase;
};

// cluster-wide sum of NV values per thread; every thread of every member gets the same totals
template <int NV>
__device__ __forceinline__ void cl_sum(double (&v)[NV], double *red, double *xch, double *xall, ClusterCtx &cl) {
    static_assert(NV <= CL_NVMAX, "payload too large");
    pf_block_sum<NV>(v, red);
    if (cl.nwg == 1) return;
    const int tid = threadIdx.x;
    const int par = (int)(cl.phase & 1ull);
    double *mine = cl.buf + ((size_t)par * cl.nwg + cl.member) * CL_NVMAX;
    if (tid < NV) {
        double val = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) val = (tid == i) ? v[i] : val;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(mine) + tid, (unsigned long long)__double_as_longlong(val),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // All cross-workgroup traffic uses relaxed AGENT-scope atomics, which are performed at the device coherence point
    // (they bypass the per-CU L1 and the per-XCD L2), so no cache write-back / invalidate is needed: an agent-scope
    // release/acquire pair would flush and invalidate the whole L2 on this multi-XCD part (measured: ~30 us per exchange).
    // What remains is ordering: the partials must have been performed before the arrival counter moves (workgroup-scope
    // release = wait for the outstanding stores), and nothing below may be hoisted above the spin loop.
    // (a workgroup-scope release emits no wait for global stores -- within a workgroup they share the L1 --, so the wait for
    //  the store acknowledgements is spelled out: without it the arrival counter overtook the partials about once in 10 runs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long target = (cl.phase + 1ull) * (unsigned long long)cl.nwg;
        __hip_atomic_fetch_add(cl.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cl.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    {   // all members' partials: the (long-latency, coherence-point) loads are spread over the whole workgroup and issued
        // back to back, then summed from LDS in member order (fixed order => identical totals in every member)
        const unsigned long long *all = reinterpret_cast<const unsigned long long *>(cl.buf + (size_t)par * cl.nwg * CL_NVMAX);
        const int tot = cl.nwg * NV;
        constexpr int NT_ = 256;
        unsigned long long tmp[(CL_MAXWG * NV + NT_ - 1) / NT_];
#pragma unroll
        for (int e = 0; e < (CL_MAXWG * NV + NT_ - 1) / NT_; ++e) {
            const int idx = tid + e * (int)blockDim.x;
            tmp[e] = 0ull;
            if (idx < tot) tmp[e] = __hip_atomic_load(all + (size_t)(idx / NV) * CL_NVMAX + (idx % NV), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int e = 0; e < (CL_MAXWG * NV + NT_ - 1) / NT_; ++e) {
            const int idx = tid + e * (int)blockDim.x;
            if (idx < tot) xall[idx] = __longlong_as_double((long long)tmp[e]);
        }
    }
    __syncthreads();
    if (tid < NV) {
        double s = 0.0;
        for (int w = 0; w < cl.nwg; ++w) s += xall[w * NV + tid];
        xch[tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xch[i];
    cl.phase += 1ull;
}

template <int KPAD, int RPT, int NT>
__global__ __launch_bounds__(NT) void pf_fit_cl_kernel(FitArgs A) {
    const int tid = threadIdx.x;
    const int d = A.d, J = A.J;
    constexpr int NP = KPAD * (KPAD + 1) / 2;            // Gram entries (upper triangle)
    constexpr int CH = NP < 26 ? NP : 26;                // entries per (cluster) reduction
    constexpr int NCH = (NP + CH - 1) / CH;
    constexpr int RED = CH > 2 * KPAD ? CH : 2 * KPAD;
    __shared__ double red[(NT / 64) * RED];
    __shared__ double xch[CL_NVMAX];
    __shared__ double xall[CL_MAXWG * RED];
    __shared__ double sHead[KPAD], sTmp[KPAD];
    __shared__ double sD[KPAD * KPAD], sR[KPAD * KPAD], sT[KPAD * KPAD], sV[KPAD * KPAD], sG[KPAD * KPAD];
    __shared__ double sLogdetV;
    __shared__ int sStatus;

    // ---- cluster geometry: members of a cluster are 8 workgroup ids apart (same XCD)
    ClusterCtx cl;
    cl.nwg = A.cl_nwg;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        const int cluster = (slot / cl.nwg) * 8 + xcd;
        cl.member = slot % cl.nwg;
        cl.counter = A.cl_counter + cluster;
        cl.buf = A.cl_buf + (size_t)cluster * 2 * cl.nwg * CL_NVMAX;
        cl.phase = 0ull;
        const int rbase = cl.member * NT * RPT;
        const bool lead = cl.member == 0;

        for (int64_t p = cluster; p < A.P; p += A.cl_nclusters) {
            __syncthreads();                                    // LDS of the previous fit is free
            const int path = A.path_of[p];
            const int64_t p0 = A.off[path];
            const int j = A.hist_len[p], m = 2 * j, k = d < m ? d : m;
            const double *alpha = A.alpha_all + (size_t)p * d;
            double *Vh = A.vh + (size_t)p * d * KPAD;
            double *sqa = A.sqrt_alpha + (size_t)p * d;
            double *mu = A.mu + (size_t)p * d;
            const double *theta_p = A.theta + (size_t)p * d, *grad_p = A.grad + (size_t)p * d;

            double a[RPT][KPAD];       // this thread's rows of B~ (later: Householder vectors)
            double sq[RPT], ag[RPT];   // sqrt(alpha_i), sqrt(alpha_i) * grad_i
            double bad = 0.0, ldu = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rbase + tid + NT * i;
                sq[i] = 1.0; ag[i] = 0.0;
                if (row < d) {
                    const double al = alpha[row];
                    if (!(al > 0.0) || !isfinite(al)) bad = 1.0;
                    const double s = sqrt(al);
                    sq[i] = s;
                    sqa[row] = s;
                    ldu += log(s);
                    ag[i] = s * grad_p[row];
                }
            }
            {
                double v[2] = {bad, ldu};
                cl_sum<2>(v, red, xch, xall, cl);
                bad = v[0]; ldu = v[1];
            }
            for (int t = tid; t < KPAD * KPAD; t += NT) { sD[t] = 0.0; sR[t] = 0.0; sT[t] = 0.0; sV[t] = 0.0; sG[t] = 0.0; }
            const size_t sm = (size_t)p * KPAD * KPAD;
            if (bad > 0.0) {                                           // A not positive definite (src/woodbury.jl:202)
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int row = rbase + tid + NT * i;
                    if (row < d) {
                        mu[row] = NAN;
                        for (int c = 0; c < KPAD; ++c) Vh[(size_t)row * KPAD + c] = 0.0;
                    }
                }
                if (lead) {
                    for (int t = tid; t < KPAD * KPAD; t += NT) { A.tmat[sm + t] = 0.0; A.vchol[sm + t] = 0.0; A.rq[sm + t] = 0.0; A.dmat[sm + t] = 0.0; }
                    if (tid == 0) { A.status[p] = PFMI_FIT_A_NOT_PD; A.logdet[p] = NAN; }
                }
                continue;
            }
            // ---- rows of B~ = U' \ [alpha.Y  S]   (src/inverse_hessian.jl:117-118, src/woodbury.jl:204)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rbase + tid + NT * i;
#pragma unroll
                for (int c = 0; c < KPAD; ++c) a[i][c] = 0.0;
                if (row < d) {
                    const double al = alpha[row], sa = sq[i];
#pragma unroll
                    for (int c = 0; c < KPAD / 2; ++c) {
                        if (c < j) {
                            const int src = A.hist_src[(size_t)p * J + c];
                            const size_t q0 = (size_t)(p0 + src) * d + row, q1 = (size_t)(p0 + src + 1) * d + row;
                            const double y = A.grad[q0] - A.grad[q1];
                            const double s = A.theta[q1] - A.theta[q0];
                            const double by = (al * y) / sa, bs = s / sa;
#pragma unroll
                            for (int cc = 0; cc < KPAD; ++cc) {
                                if (cc == c) a[i][cc] = by;
                                if (cc == j + c) a[i][cc] = bs;
                            }
                        }
                    }
                }
            }
            // ---- Gram matrix G = B~'B~ (compile-time column pairs), CH entries per cluster reduction
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                double g2[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) g2[e] = 0.0;
                {
                    int e = 0;
#pragma unroll
                    for (int ca = 0; ca < KPAD; ++ca)
#pragma unroll
                        for (int cb = ca; cb < KPAD; ++cb) {
                            if (e / CH == ch) {
#pragma unroll
                                for (int i = 0; i < RPT; ++i) g2[e % CH] += a[i][ca] * a[i][cb];
                            }
                            ++e;
                        }
                }
                cl_sum<CH>(g2, red, xch, xall, cl);
                if (tid == 0) {
                    int e = 0;
#pragma unroll
                    for (int ca = 0; ca < KPAD; ++ca)
#pragma unroll
                        for (int cb = ca; cb < KPAD; ++cb) {
                            if (e / CH == ch) { sG[ca * KPAD + cb] = g2[e % CH]; sG[cb * KPAD + ca] = g2[e % CH]; }
                            ++e;
                        }
                }
            }
            __syncthreads();
            // ---- D (m x m)   (src/inverse_hessian.jl:119-130), replicated in every member
            if (j > 0) {
                double *R = sT, *nRinv = sV;
                for (int t = tid; t < j * j; t += NT) {
                    const int aa = t / j, b = t % j;
                    R[aa * KPAD + b] = (b >= aa) ? sG[(j + aa) * KPAD + b] : 0.0;      // triu(S'Y)   :119-121
                    nRinv[aa * KPAD + b] = 0.0;
                }
            }
            __syncthreads();
            if (tid < j) {                                   // -R^{-1}: lane c solves column c by back substitution :122-124
                const double *R = sT;
                double *nRinv = sV;
                const int c = tid;
                for (int r = c; r >= 0; --r) {
                    double rhs = (r == c) ? -1.0 : 0.0;
                    for (int t = r + 1; t <= c; ++t) rhs -= R[r * KPAD + t] * nRinv[t * KPAD + c];
                    nRinv[r * KPAD + c] = rhs / R[r * KPAD + r];
                }
            }
            __syncthreads();
            if (j > 0) {   // M = Y'alpha Y + diag(R); D12, D21
                for (int t = tid; t < j * j; t += NT) {
                    const int aa = t / j, b = t % j;
                    sD[aa * KPAD + (j + b)] = sV[aa * KPAD + b];
                    sD[(j + aa) * KPAD + b] = sV[b * KPAD + aa];
                    double v = (aa <= b) ? sG[aa * KPAD + b] : sG[b * KPAD + aa];
                    if (aa == b) v += sT[aa * KPAD + aa];
                    sR[aa * KPAD + b] = v;                                   // M
                }
            }
            __syncthreads();
            if (j > 0) {   // T1 = M nRinv  -> sG
                for (int t = tid; t < j * j; t += NT) {
                    const int aa = t / j, b = t % j;
                    double v = 0.0;
                    for (int u = 0; u <= b; ++u) v += sR[aa * KPAD + u] * sV[u * KPAD + b];
                    sG[aa * KPAD + b] = v;
                }
            }
            __syncthreads();
            if (j > 0) {   // D22 = nRinv' T1
                for (int t = tid; t < j * j; t += NT) {
                    const int aa = t / j, b = t % j;
                    double v = 0.0;
                    for (int u = 0; u <= aa; ++u) v += sV[u * KPAD + aa] * sG[u * KPAD + b];
                    sD[(j + aa) * KPAD + (j + b)] = v;
                }
            }
            __syncthreads();
            for (int t = tid; t < KPAD * KPAD; t += NT) { sT[t] = 0.0; sV[t] = 0.0; sR[t] = 0.0; }
            __syncthreads();

            // ---- Householder QR, one cluster reduction per column; the lead member owns the head rows and ships row c
            //      with the same exchange.  Thread aa < KPAD of EVERY member keeps row aa of the compact-WY T in registers.
            double trow[KPAD];
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) trow[cc] = 0.0;
            for (int c = 0; c < k; ++c) {
                double acc2[2 * KPAD];                       // [0, KPAD): sum_{row > c} x_c x_cc ; [KPAD, 2 KPAD): row c itself
#pragma unroll
                for (int cc = 0; cc < 2 * KPAD; ++cc) acc2[cc] = 0.0;
                if (lead && tid == c) {
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) acc2[KPAD + cc] = a[0][cc];
                }
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int row = rbase + tid + NT * i;
                    if (row > c) {
                        double xc = 0.0;
#pragma unroll
                        for (int cc = 0; cc < KPAD; ++cc) if (cc == c) xc = a[i][cc];
#pragma unroll
                        for (int cc = 0; cc < KPAD; ++cc) acc2[cc] += xc * a[i][cc];
                    }
                }
                cl_sum<2 * KPAD>(acc2, red, xch, xall, cl);
                double xn2 = 0.0, alpha_c = 0.0;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc == c) { xn2 = acc2[cc]; alpha_c = acc2[KPAD + cc]; }
                const double xnorm = sqrt(xn2);
                double tau, scal, beta;
                if (xnorm == 0.0) { tau = 0.0; scal = 0.0; beta = alpha_c; }
                else {
                    beta = -copysign(sqrt(fma(alpha_c, alpha_c, xn2)), alpha_c);
                    tau = (beta - alpha_c) / beta;
                    scal = 1.0 / (alpha_c - beta);
                }
                double wv[KPAD];                     // cc > c: tau * (v_c . column cc); cc < c: v_c . v_cc
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) {
                    const double vdot = acc2[KPAD + cc] + scal * acc2[cc];
                    wv[cc] = (cc > c) ? tau * vdot : vdot;
                }
                if (tid <= c && tid < KPAD) {        // T column c, one row per thread
                    double v = 0.0;
#pragma unroll
                    for (int b = 0; b < KPAD; ++b) if (b >= tid && b < c) v += trow[b] * wv[b];
                    const double tnew = (tid == c) ? tau : -tau * v;
#pragma unroll
                    for (int b = 0; b < KPAD; ++b) if (b == c) trow[b] = tnew;
                }
                if (tid == 0) {                      // R row c (every member: same numbers as the lead's registers)
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) {
                        if (cc == c) sR[c * KPAD + cc] = beta;
                        else if (cc > c) sR[c * KPAD + cc] = (cc < m) ? acc2[KPAD + cc] - wv[cc] : 0.0;
                    }
                }
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int row = rbase + tid + NT * i;
                    if (row > c) {
                        double v = 0.0;
#pragma unroll
                        for (int cc = 0; cc < KPAD; ++cc) if (cc == c) v = a[i][cc] * scal;
#pragma unroll
                        for (int cc = 0; cc < KPAD; ++cc) {
                            if (cc == c) a[i][cc] = v;
                            else if (cc > c) a[i][cc] -= wv[cc] * v;
                        }
                    } else if (row == c) {
#pragma unroll
                        for (int cc = 0; cc < KPAD; ++cc) {
                            if (cc == c) a[i][cc] = beta;
                            else if (cc > c) a[i][cc] -= wv[cc];
                        }
                    }
                }
            }
            if (tid < KPAD) {
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) sT[tid * KPAD + cc] = trow[cc];
            }
            // ---- Householder vectors get an explicit unit diagonal (head rows live in the lead member)
            if (lead && tid < k) {
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc >= tid) a[0][cc] = (cc == tid) ? 1.0 : 0.0;
            }
            __syncthreads();
            // ---- C = I + R D R' (k x k): RD -> sG, then C -> sV (upper), all threads; Cholesky on one wave
            for (int t = tid; t < k * m; t += NT) {
                const int aa = t / m, b = t % m;
                double v = 0.0;
                for (int u = aa; u < m; ++u) v += sR[aa * KPAD + u] * sD[u * KPAD + b];
                sG[aa * KPAD + b] = v;
            }
            __syncthreads();
            for (int t = tid; t < k * k; t += NT) {
                const int aa = t / k, b = t % k;
                if (b >= aa) {
                    double v = (aa == b) ? 1.0 : 0.0;
                    for (int u = b; u < m; ++u) v += sG[aa * KPAD + u] * sR[b * KPAD + u];
                    sV[aa * KPAD + b] = v;
                }
            }
            __syncthreads();
            if (tid < 64) {                          // wave 0: left-looking Cholesky, lane b owns column b
                const int b = tid;
                volatile double *Vv = sV;
                volatile int *vst = &sStatus;
                volatile double *vld = &sLogdetV;
                if (b == 0) { *vst = PFMI_FIT_OK; *vld = 0.0; }
                for (int c = 0; c < k; ++c) {
                    if (*vst != PFMI_FIT_OK) break;
                    if (b == c) {
                        double diag = Vv[c * KPAD + c];
                        for (int t = 0; t < c; ++t) { const double x = Vv[t * KPAD + c]; diag -= x * x; }
                        if (!(diag > 0.0) || !isfinite(diag)) *vst = PFMI_FIT_C_NOT_PD;    // src/woodbury.jl:205
                        else { diag = sqrt(diag); Vv[c * KPAD + c] = diag; *vld = *vld + log(diag); }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (*vst != PFMI_FIT_OK) break;
                    if (b > c && b < k) {
                        double v = Vv[c * KPAD + b];
                        for (int t = 0; t < c; ++t) v -= Vv[t * KPAD + c] * Vv[t * KPAD + b];
                        Vv[c * KPAD + b] = v / Vv[c * KPAD + c];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (b >= k && b < KPAD) Vv[b * KPAD + b] = 1.0;                             // identity padding
            }
            __syncthreads();
            if (lead) {
                for (int t = tid; t < KPAD * KPAD; t += NT) {
                    A.tmat[sm + t] = sT[t]; A.vchol[sm + t] = sV[t]; A.rq[sm + t] = sR[t]; A.dmat[sm + t] = sD[t];
                }
            }
            // Householder block out (row-major [d][KPAD])
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rbase + tid + NT * i;
                if (row < d) {
                    double *o = Vh + (size_t)row * KPAD;
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) o[cc] = a[i][cc];
                }
            }
            if (sStatus != PFMI_FIT_OK) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) { const int row = rbase + tid + NT * i; if (row < d) mu[row] = NAN; }
                if (lead && tid == 0) { A.status[p] = sStatus; A.logdet[p] = NAN; }
                continue;
            }
            // ---- mu = theta + U' Q [V'V 0;0 I] Q' U g
            double acc[KPAD];
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i)
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) acc[cc] += ag[i] * a[i][cc];
            cl_sum<KPAD>(acc, red, xch, xall, cl);
            double t1[KPAD];                          // t1 = T' w1
#pragma unroll
            for (int aa = 0; aa < KPAD; ++aa) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < KPAD; ++b) if (b <= aa) v += sT[b * KPAD + aa] * acc[b];
                t1[aa] = v;
            }
            double bv[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                double v = ag[i];
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) v -= a[i][cc] * t1[cc];
                bv[i] = v;
            }
            if (lead) {                                  // head <- V'(V head): the head rows live in the lead member
                if (tid < k) sHead[tid] = bv[0];
                __syncthreads();
                if (tid < 64) {
                    const int aa = tid;
                    volatile double *tmp = sTmp;
                    volatile double *hd = sHead;
                    double v = 0.0;
                    if (aa < k) for (int b = aa; b < k; ++b) v += sV[aa * KPAD + b] * hd[b];
                    if (aa < k) tmp[aa] = v;
                    __builtin_amdgcn_wave_barrier();
                    double v2 = 0.0;
                    if (aa < k) for (int b = 0; b <= aa; ++b) v2 += sV[b * KPAD + aa] * tmp[b];
                    __builtin_amdgcn_wave_barrier();
                    if (aa < k) hd[aa] = v2;
                }
                __syncthreads();
                if (tid < k) bv[0] = sHead[tid];
            }
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rbase + tid + NT * i;
                if (row < d) {
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) acc[cc] += bv[i] * a[i][cc];
                }
            }
            cl_sum<KPAD>(acc, red, xch, xall, cl);
#pragma unroll
            for (int aa = 0; aa < KPAD; ++aa) {       // t2 = T w2
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < KPAD; ++b) if (b >= aa) v += sT[aa * KPAD + b] * acc[b];
                t1[aa] = v;
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rbase + tid + NT * i;
                if (row < d) {
                    double v = bv[i];
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) v -= a[i][cc] * t1[cc];
                    mu[row] = theta_p[row] + sq[i] * v;
                }
            }
            if (lead && tid == 0) {
                A.status[p] = PFMI_FIT_OK;
                A.logdet[p] = 2.0 * (ldu + sLogdetV);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
template <int KPAD, int RPT, int NT>
static int32_t launch_cl(pfmi_ctx *c, FitArgs a, bool *handled) {
    auto kern = pf_fit_cl_kernel<KPAD, RPT, NT>;
    const int rows_per_wg = NT * RPT;
    const int nwg = (a.d + rows_per_wg - 1) / rows_per_wg;
    int per_cu = 0;
    PF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, 0));
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, c->device));
    const int capacity = per_cu * prop.multiProcessorCount;
    if (nwg > CL_MAXWG) return PFMI_OK;
    int ncl8 = capacity / (8 * nwg);                         // clusters per XCD
    if (ncl8 < 1 || !prop.cooperativeLaunch) return PFMI_OK; // cannot hold one cluster per XCD: fall back
    const int64_t need8 = (a.P + 7) / 8;
    if (ncl8 > need8) ncl8 = (int)need8;
    const int ncl = ncl8 * 8, grid = ncl * nwg;
    PF_TRY(c->cl_counter.ensure(sizeof(unsigned long long) * ncl));
    PF_TRY(c->cl_buf.ensure(sizeof(double) * (size_t)ncl * 2 * nwg * CL_NVMAX));
    PF_HIP(hipMemsetAsync(c->cl_counter.p, 0, sizeof(unsigned long long) * ncl, c->stream));
    a.P = c->P; a.cl_counter = c->cl_counter.as<unsigned long long>(); a.cl_buf = c->cl_buf.as<double>();
    a.cl_nwg = nwg; a.cl_nclusters = ncl;
    void *args[] = {&a};
    PF_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3((unsigned)grid), dim3(NT), args, 0, c->stream));
    *handled = true;
    return PFMI_OK;
}

int32_t pf_launch_fit_cluster(pfmi_ctx *c, FitArgs a, bool *handled) {
    *handled = false;
    // Opt-in (PFMI_FIT_KERNEL=cluster).  Measured on MI355X (tests/fitcl_probe-style runs, round 1): one exchange through the
    // device coherence point costs ~15 us, ~30 exchanges per fit => ~0.4-0.6 ms per fit.  That halves the LATENCY of a small batch
    // of large-d fits (d = 3000, 62 fits: 0.41 ms vs 0.77 ms) but at d = 1e4, J = 10 only 8 clusters of 20 workgroups are
    // resident and the THROUGHPUT is below the memory-resident kernel's (117 ms vs 55 ms for 1600 fits), so it is not the default.
    const char *force = getenv("PFMI_FIT_KERNEL");
    if (!(force && force[0] == 'c')) return PFMI_OK;
    switch (c->kpad) {
        case 4: return launch_cl<4, 4, 256>(c, a, handled);
        case 8: return launch_cl<8, 4, 256>(c, a, handled);
        case 12: return launch_cl<12, 4, 256>(c, a, handled);
        case 16: return launch_cl<16, 2, 256>(c, a, handled);
        case 20: return launch_cl<20, 2, 256>(c, a, handled);
        default: return PFMI_OK;                              // KC = 32: register budget too small, memory-resident kernel
    }
}
