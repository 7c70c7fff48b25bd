/*
 * pf_oracle.c -- CPU restatement (scalar fp64 C99) of the ELBO / sampling / inverse-Hessian /
 * PSIS-resampling hot path of mlcolab/Pathfinder.jl v0.10.7.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (libpfmi.so, HIP) never links,
 * imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * The reference is 100 % Julia and cannot be executed in this environment (no julia binary), so
 * the restatement is pinned by (tests/test_oracle_*.py):
 *   - the reference's literal S0/Y0 fixture + explicit dense Byrd formula (test/inverse_hessian.jl:8-44)
 *   - dense-algebra identities for every Woodbury operation incl. n < m (test/woodbury.jl:18-404)
 *   - the analytic ELBO known answers (test/elbo.jl:7-54), _findmax_skipnan cases (test/utils.jl:8-12)
 *   - iso-normal exactness after one iteration (test/singlepath.jl:13-41), log-ratio ordering
 *     (test/resample.jl:62-89), degenerate-weight resampling (test/resample.jl:36-49)
 *   - SciPy's LAPACK dgeqrf (same Householder convention as Julia's qr) and scipy.stats.genpareto.
 * PARITY UNPINNED for two third-party pieces whose source is not under /root/reference:
 *   PSIS.jl (compat 0.2-0.9; restated here from Vehtari et al. 2024 / Zhang & Stephens 2009) and
 *   StatsBase.sample (0.33.17-0.34; replaced by this repo's own deterministic fixed-point
 *   inverse-CDF sampler -- see pfo_sample_weighted).  The reference's own tests only check
 *   sum(weights) ~ 1 and membership for those (test/resample.jl:91-109, 8-60).
 *
 * Layout convention: every matrix is column-major Float64 (Julia default); A[i + ld*j].
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define PFO_LOG2PI 1.8378770664093454835606594728112352797227949472755668

/* status codes shared with include/pfmi.h */
#define PFO_OK 0
#define PFO_A_NOT_PD 1
#define PFO_C_NOT_PD 2
#define PFO_NONFINITE 3

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* Julia Base.mapreduce_impl pairwise summation (blksize 1024) -- used by Statistics.mean/var
 * (src/elbo.jl:17-18 call them). */
static double sum_pairwise(const double *x, long n) {
    if (n <= 0) return 0.0;
    if (n < 1024) {
        double s = x[0];
        for (long i = 1; i < n; ++i) s += x[i];
        return s;
    }
    long mid = n >> 1;
    return sum_pairwise(x, mid) + sum_pairwise(x + mid, n - mid);
}
static double sumsq_centered_pairwise(const double *x, long n, double m) {
    if (n <= 0) return 0.0;
    if (n < 1024) {
        double s = (x[0] - m) * (x[0] - m);
        for (long i = 1; i < n; ++i) s += (x[i] - m) * (x[i] - m);
        return s;
    }
    long mid = n >> 1;
    return sumsq_centered_pairwise(x, mid, m) + sumsq_centered_pairwise(x + mid, n - mid, m);
}

static double dotp(const double *a, const double *b, long n) {
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* gilbert_init -- src/inverse_hessian.jl:5-10                                                 */
/* ------------------------------------------------------------------------------------------ */
void pfo_gilbert_init(int d, const double *alpha, const double *s, const double *y, double *out) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = 0; i < d; ++i) {
        a += y[i] * alpha[i] * y[i];   /* dot(y, Diagonal(α), y)        :6 */
        b += y[i] * s[i];              /* dot(y, s)                      :7 */
        c += s[i] * (1.0 / alpha[i]) * s[i]; /* dot(s, inv(Diagonal(α)), s) :8 */
    }
    for (int i = 0; i < d; ++i) {
        double sa = s[i] / alpha[i];
        out[i] = b / (a / alpha[i] + y[i] * y[i] - (a / c) * sa * sa);   /* :9 */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* lbfgs_inverse_hessian (Byrd compact form) -- src/inverse_hessian.jl:98-133                  */
/* S, Y already ordered oldest -> newest (hist_inds at :105), d x j column-major.              */
/* B: d x 2j, D: 2j x 2j (both column-major, fully overwritten).                               */
/* ------------------------------------------------------------------------------------------ */
void pfo_lbfgs_inverse_hessian(int d, int j, const double *alpha, const double *S, const double *Y,
                               double *B, double *D) {
    int m = 2 * j;
    if (j == 0) return;                                              /* :103 */
    for (int i = 0; i < m * m; ++i) D[i] = 0.0;                      /* :102 */
    /* B1 = diag(α) Y, B2 = S                                           :117-118 */
    for (int c = 0; c < j; ++c)
        for (int i = 0; i < d; ++i) {
            B[i + (long)d * c] = alpha[i] * Y[i + (long)d * c];
            B[i + (long)d * (j + c)] = S[i + (long)d * c];
        }
    /* R = triu(S'Y)                                                    :119-121 */
    double *R = (double *)calloc((size_t)j * j, sizeof(double));
    for (int a = 0; a < j; ++a)
        for (int b = a; b < j; ++b) R[a + j * b] = dotp(S + (long)d * a, Y + (long)d * b, d);
    /* nRinv = -R^{-1} (upper triangular): solve R X = -I by back substitution  :122-124 */
    double *nRinv = (double *)calloc((size_t)j * j, sizeof(double));
    for (int c = 0; c < j; ++c) {
        for (int r = c; r >= 0; --r) {
            double rhs = (r == c) ? -1.0 : 0.0;
            for (int t = r + 1; t <= c; ++t) rhs -= R[r + j * t] * nRinv[t + j * c];
            nRinv[r + j * c] = rhs / R[r + j * r];
        }
    }
    /* D12 = nRinv, D21 = nRinv'                                        :122-125 */
    for (int a = 0; a < j; ++a)
        for (int b = 0; b < j; ++b) {
            D[a + m * (j + b)] = nRinv[a + j * b];
            D[(j + a) + m * b] = nRinv[b + j * a];
        }
    /* M = diag(R) + Y' diag(α) Y  (symmetric)                          :126-128 */
    double *M = (double *)calloc((size_t)j * j, sizeof(double));
    for (int a = 0; a < j; ++a)
        for (int b = a; b < j; ++b) {
            double v = dotp(Y + (long)d * a, B + (long)d * b, d);
            if (a == b) v += R[a + j * a];
            M[a + j * b] = v;
            M[b + j * a] = v;
        }
    /* D22 = nRinv' * (M * nRinv)                                       :129-130 */
    double *T = (double *)calloc((size_t)j * j, sizeof(double));
    for (int a = 0; a < j; ++a)
        for (int b = 0; b < j; ++b) {
            double v = 0.0;
            for (int t = 0; t <= b; ++t) v += M[a + j * t] * nRinv[t + j * b];
            T[a + j * b] = v;
        }
    for (int a = 0; a < j; ++a)
        for (int b = 0; b < j; ++b) {
            double v = 0.0;
            for (int t = 0; t <= a; ++t) v += nRinv[t + j * a] * T[t + j * b];
            D[(j + a) + m * (j + b)] = v;
        }
    free(R); free(nRinv); free(M); free(T);
}

/* ------------------------------------------------------------------------------------------ */
/* lbfgs_inverse_hessians trace walk -- src/inverse_hessian.jl:25-66                           */
/* theta, grad: (L+1) points of d (point l at offset l*d).                                     */
/* Outputs, for every point l = 0..L (Hs[l+1] in Julia):                                       */
/*   alpha_all[l*d .. ]     the diagonal H0 used by fit l                                       */
/*   hist_len[l]            j (effective history length)                                        */
/*   hist_src[l*J + c]      trace iteration (0-based l' of s = theta[l'+1]-theta[l']) of the    */
/*                          c-th oldest history column, c < hist_len[l]  (order of :105)       */
/* returns num_bfgs_updates_rejected.                                                          */
/* ------------------------------------------------------------------------------------------ */
/* Hinit (src/inverse_hessian.jl:25 keyword): 0 = gilbert_init (the default), 1 = (alpha, s, y) -> fill(y's / y'y), the Nocedal-Wright
 * scaling the reference's own test passes (test/inverse_hessian.jl:49).  A process-wide setting of this TEST oracle (pfo_set_hinit). */
static int g_pfo_hinit = 0;
void pfo_set_hinit(int hinit) { g_pfo_hinit = hinit; }
int pfo_lbfgs_history(int d, int L, const double *theta, const double *grad, int J, double eps,
                      double *alpha_all, int *hist_len, int *hist_src) {
    int history_ind = 0;      /* 1-based slot of last set entry, 0 = none   :32 */
    int history_len_eff = 0;  /*                                            :33 */
    int Jcap = J < L ? J : L; /* ring width min(history_length, L)           :36-37 */
    int *slot_src = (int *)calloc((size_t)(Jcap > 0 ? Jcap : 1), sizeof(int));
    double *alpha = (double *)malloc(sizeof(double) * d);
    double *anew = (double *)malloc(sizeof(double) * d);
    double *s = (double *)malloc(sizeof(double) * d);
    double *y = (double *)malloc(sizeof(double) * d);
    for (int i = 0; i < d; ++i) alpha[i] = 1.0;                     /* :38 */
    memcpy(alpha_all, alpha, sizeof(double) * d);                   /* H0 = I, empty history :39 */
    hist_len[0] = 0;
    int rejected = 0;
    for (int l = 1; l <= L; ++l) {                                  /* :43 */
        const double *th0 = theta + (long)(l - 1) * d, *th1 = theta + (long)l * d;
        const double *g0 = grad + (long)(l - 1) * d, *g1 = grad + (long)l * d;
        double ys = 0.0, yy = 0.0;
        for (int i = 0; i < d; ++i) {
            s[i] = th1[i] - th0[i];                                 /* :45 */
            y[i] = g0[i] - g1[i];                                   /* :46 */
        }
        for (int i = 0; i < d; ++i) { ys += y[i] * s[i]; yy += y[i] * y[i]; }
        if (ys > eps * yy) {                                        /* :47 */
            history_ind = (history_ind % J) + 1;                    /* mod1(ind+1, J) :49 */
            if (history_ind > history_len_eff) history_len_eff = history_ind; /* :50 */
            slot_src[history_ind - 1] = l - 1;                      /* :51-52 */
            if (g_pfo_hinit == 1) { for (int i = 0; i < d; ++i) anew[i] = ys / yy; }   /* fill!(similar(α), dot(y, s) / sum(abs2, y)), test/inverse_hessian.jl:49 */
            else pfo_gilbert_init(d, alpha, s, y, anew);             /* :55 */
            memcpy(alpha, anew, sizeof(double) * d);
        } else {
            rejected += 1;                                          /* :57 */
        }
        memcpy(alpha_all + (long)l * d, alpha, sizeof(double) * d);
        hist_len[l] = history_len_eff;
        /* hist_inds = [ind+1 : len ; 1 : ind]                         :105 */
        int c = 0;
        for (int t = history_ind + 1; t <= history_len_eff; ++t) hist_src[(long)l * J + c++] = slot_src[t - 1];
        for (int t = 1; t <= history_ind; ++t) hist_src[(long)l * J + c++] = slot_src[t - 1];
    }
    free(slot_src); free(alpha); free(anew); free(s); free(y);
    return rejected;
}

/* ------------------------------------------------------------------------------------------ */
/* Householder QR, LAPACK dgeqr2/dlarfg convention (what Julia's qr -> geqrt! produces):       */
/*   beta = -sign(alpha)*||x||, tau = (beta-alpha)/beta, v = [1; x_rest/(alpha-beta)].          */
/* A (n x m, column-major, ld = n) is overwritten: R in the upper triangle, v below.           */
/* tau has k = min(n, m) entries.                     used by src/woodbury.jl:204               */
/* ------------------------------------------------------------------------------------------ */
void pfo_householder_qr(int n, int m, double *A, double *tau) {
    int k = n < m ? n : m;
    for (int c = 0; c < k; ++c) {
        double *col = A + (long)n * c;
        double alpha = col[c];
        double xnorm = 0.0;
        for (int i = c + 1; i < n; ++i) xnorm += col[i] * col[i];
        xnorm = sqrt(xnorm);
        if (xnorm == 0.0) { tau[c] = 0.0; continue; }           /* dlarfg: H = I */
        double beta = -copysign(hypot(alpha, xnorm), alpha);
        tau[c] = (beta - alpha) / beta;
        double scal = 1.0 / (alpha - beta);
        for (int i = c + 1; i < n; ++i) col[i] *= scal;
        col[c] = beta;
        /* apply H_c to the trailing columns (dlarf, left) */
        for (int cc = c + 1; cc < m; ++cc) {
            double *o = A + (long)n * cc;
            double w = o[c];
            for (int i = c + 1; i < n; ++i) w += col[i] * o[i];
            w *= tau[c];
            o[c] -= w;
            for (int i = c + 1; i < n; ++i) o[i] -= w * col[i];
        }
    }
}

/* x <- Q x (trans = 0) or Q' x (trans = 1); Q = H_0 H_1 ... H_{k-1}; x is n x N column-major.
 * LAPACK dorm2r semantics == Julia lmul!(Q, x) / lmul!(Q', x) (src/woodbury.jl:132,140,154,162) */
void pfo_apply_q(int n, int k, const double *QR, const double *tau, int trans, double *X, long N) {
    for (long col = 0; col < N; ++col) {
        double *x = X + (long)n * col;
        for (int t = 0; t < k; ++t) {
            int c = trans ? t : (k - 1 - t);
            if (tau[c] == 0.0) continue;
            const double *v = QR + (long)n * c;
            double w = x[c];
            for (int i = c + 1; i < n; ++i) w += v[i] * x[i];
            w *= tau[c];
            x[c] -= w;
            for (int i = c + 1; i < n; ++i) x[i] -= w * v[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* pdfactorize(A::Diagonal, B, D) -- src/woodbury.jl:201-207                                   */
/* in : alpha (d), B (d x m), D (m x m)                                                        */
/* out: sqrt_alpha (d) = U ; QR (d x m) + tau (k) = qr(U' \ B) ; V (k x k upper, col-major)    */
/* returns PFO_OK / PFO_A_NOT_PD / PFO_C_NOT_PD (Julia throws PosDefException :202,:205)       */
/* ------------------------------------------------------------------------------------------ */
int pfo_pdfactorize(int d, int m, const double *alpha, const double *B, const double *D,
                    double *sqrt_alpha, double *QR, double *tau, double *V) {
    int k = d < m ? d : m;
    for (int i = 0; i < d; ++i) {
        if (!(alpha[i] > 0.0) || !isfinite(alpha[i])) return PFO_A_NOT_PD;      /* :202 */
        sqrt_alpha[i] = sqrt(alpha[i]);                                          /* :203 */
    }
    for (int c = 0; c < m; ++c)
        for (int i = 0; i < d; ++i) QR[i + (long)d * c] = B[i + (long)d * c] / sqrt_alpha[i]; /* U' \ B :204 */
    pfo_householder_qr(d, m, QR, tau);                                           /* :204 */
    if (k == 0) return PFO_OK;
    /* C = I + R D R'   with R = k x m upper trapezoidal                            :205 */
    double *RD = (double *)calloc((size_t)k * m, sizeof(double));
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < m; ++b) {
            double v = 0.0;
            for (int t = a; t < m; ++t) v += QR[a + (long)d * t] * D[t + m * b];
            RD[a + k * b] = v;
        }
    for (int a = 0; a < k; ++a)
        for (int b = a; b < k; ++b) {   /* Symmetric(...) reads the upper triangle */
            double v = (a == b) ? 1.0 : 0.0;
            for (int t = b; t < m; ++t) v += RD[a + k * t] * QR[b + (long)d * t];
            V[a + k * b] = v;
        }
    free(RD);
    /* upper Cholesky C = V'V (dpotrf 'U')                                          :205 */
    for (int c = 0; c < k; ++c) {
        double diag = V[c + k * c];
        for (int t = 0; t < c; ++t) diag -= V[t + k * c] * V[t + k * c];
        if (!(diag > 0.0) || !isfinite(diag)) return PFO_C_NOT_PD;
        diag = sqrt(diag);
        V[c + k * c] = diag;
        for (int b = c + 1; b < k; ++b) {
            double v = V[c + k * b];
            for (int t = 0; t < c; ++t) v -= V[t + k * c] * V[t + k * b];
            V[c + k * b] = v / diag;
        }
        for (int r = c + 1; r < k; ++r) V[r + k * c] = 0.0;
    }
    return PFO_OK;
}

/* logdet(W) = 2 (logdet U + logdet V) -- src/woodbury.jl:76-80, 323-324 */
double pfo_logdet(int d, int k, const double *sqrt_alpha, const double *V) {
    double lu = 0.0, lv = 0.0;
    for (int i = 0; i < d; ++i) lu += log(sqrt_alpha[i]);
    for (int c = 0; c < k; ++c) lv += log(V[c + k * c]);
    return 2.0 * (lu + lv);
}

/* lmul!(R::WoodburyPDRightFactor, x): x <- [V 0;0 I] Q' U x -- src/woodbury.jl:129-135 */
void pfo_lmul_R(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                const double *V, double *X, long N) {
    int k = d < m ? d : m;
    for (long c = 0; c < N; ++c)
        for (int i = 0; i < d; ++i) X[i + (long)d * c] *= sqrt_alpha[i];      /* lmul!(U) :131 */
    pfo_apply_q(d, k, QR, tau, 1, X, N);                                       /* lmul!(Q') :132 */
    for (long c = 0; c < N; ++c) {                                             /* lmul!(V, x[1:k]) :133 */
        double *x = X + (long)d * c;
        for (int a = 0; a < k; ++a) {
            double v = 0.0;
            for (int b = a; b < k; ++b) v += V[a + k * b] * x[b];
            x[a] = v;
        }
    }
}
/* lmul!(L::WoodburyPDLeftFactor, x): x <- U' Q [V' 0;0 I] x -- src/woodbury.jl:136-143 */
void pfo_lmul_L(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                const double *V, double *X, long N) {
    int k = d < m ? d : m;
    for (long c = 0; c < N; ++c) {                                             /* lmul!(V', x[1:k]) :139 */
        double *x = X + (long)d * c;
        for (int a = k - 1; a >= 0; --a) {
            double v = 0.0;
            for (int b = 0; b <= a; ++b) v += V[b + k * a] * x[b];
            x[a] = v;
        }
    }
    pfo_apply_q(d, k, QR, tau, 0, X, N);                                       /* lmul!(Q) :140 */
    for (long c = 0; c < N; ++c)
        for (int i = 0; i < d; ++i) X[i + (long)d * c] *= sqrt_alpha[i];      /* lmul!(U') :141 */
}
/* ldiv!(R, x): x <- U^{-1} Q [V^{-1} 0;0 I] x -- src/woodbury.jl:151-157 */
void pfo_ldiv_R(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                const double *V, double *X, long N) {
    int k = d < m ? d : m;
    for (long c = 0; c < N; ++c) {
        double *x = X + (long)d * c;
        for (int a = k - 1; a >= 0; --a) {      /* back substitution V z = x */
            double v = x[a];
            for (int b = a + 1; b < k; ++b) v -= V[a + k * b] * x[b];
            x[a] = v / V[a + k * a];
        }
    }
    pfo_apply_q(d, k, QR, tau, 0, X, N);
    for (long c = 0; c < N; ++c)
        for (int i = 0; i < d; ++i) X[i + (long)d * c] /= sqrt_alpha[i];
}
/* ldiv!(L, x): x <- [V'^{-1} 0;0 I] Q' U'^{-1} x -- src/woodbury.jl:158-165 */
void pfo_ldiv_L(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                const double *V, double *X, long N) {
    int k = d < m ? d : m;
    for (long c = 0; c < N; ++c)
        for (int i = 0; i < d; ++i) X[i + (long)d * c] /= sqrt_alpha[i];      /* :161 */
    pfo_apply_q(d, k, QR, tau, 1, X, N);                                       /* :162 */
    for (long c = 0; c < N; ++c) {                                             /* forward subst V' z = x :163 */
        double *x = X + (long)d * c;
        for (int a = 0; a < k; ++a) {
            double v = x[a];
            for (int b = 0; b < a; ++b) v -= V[b + k * a] * x[b];
            x[a] = v / V[a + k * a];
        }
    }
}
/* lmul!(W, x) = lmul!(F.L, lmul!(F.R, x)) -- src/woodbury.jl:64-68, 340-349 */
void pfo_mul_W(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
               const double *V, double *X, long N) {
    pfo_lmul_R(d, m, sqrt_alpha, QR, tau, V, X, N);
    pfo_lmul_L(d, m, sqrt_alpha, QR, tau, V, X, N);
}

/* fit_mvnormals mean: mu = muladd(Sigma, grad, theta) -- src/mvnormal.jl:14-21 */
void pfo_fit_mean(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                  const double *V, const double *theta, const double *grad, double *mu) {
    memcpy(mu, grad, sizeof(double) * d);
    pfo_mul_W(d, m, sqrt_alpha, QR, tau, V, mu, 1);
    for (int i = 0; i < d; ++i) mu[i] += theta[i];
}

/* rand_and_logpdf -- src/mvnormal.jl:24-39.  U (d x N) holds u ~ N(0,I) on entry, x on exit. */
void pfo_rand_and_logpdf(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                         const double *V, const double *mu, double logdet, long N, double *U,
                         double *logq) {
    for (long c = 0; c < N; ++c) {                                   /* unormsq before transform :31 */
        double s = 0.0;
        for (int i = 0; i < d; ++i) s += U[i + (long)d * c] * U[i + (long)d * c];
        logq[c] = s;
    }
    pfo_lmul_L(d, m, sqrt_alpha, QR, tau, V, U, N);                  /* unwhiten! :32 */
    for (long c = 0; c < N; ++c) {
        for (int i = 0; i < d; ++i) U[i + (long)d * c] += mu[i];     /* :33 */
        logq[c] = ((double)d * PFO_LOG2PI + logdet + logq[c]) / -2.0; /* :36 */
    }
}

/* Distributions.logpdf(MvNormal(mu, W), X) = -(d log2pi + logdet)/2 - invquad(W, x - mu)/2,
 * invquad through ldiv!(L) -- src/resample.jl:85-89, src/woodbury.jl:378-382,158-165,425-436.
 * X (d x N) is not modified. */
void pfo_logpdf_mvnormal(int d, int m, const double *sqrt_alpha, const double *QR, const double *tau,
                         const double *V, const double *mu, double logdet, long N, const double *X,
                         double *out) {
    double *z = (double *)malloc(sizeof(double) * d);
    for (long c = 0; c < N; ++c) {
        for (int i = 0; i < d; ++i) z[i] = X[i + (long)d * c] - mu[i];
        pfo_ldiv_L(d, m, sqrt_alpha, QR, tau, V, z, 1);
        double s = 0.0;
        for (int i = 0; i < d; ++i) s += z[i] * z[i];
        out[c] = -((double)d * PFO_LOG2PI + logdet) / 2.0 - s / 2.0;
    }
    free(z);
}

/* elbo_and_samples statistics -- src/elbo.jl:16-18 */
void pfo_elbo_stats(long N, const double *logp, const double *logq, double *logr, double *value,
                    double *se) {
    for (long i = 0; i < N; ++i) logr[i] = logp[i] - logq[i];
    double mean = sum_pairwise(logr, N) / (double)N;
    double var = sumsq_centered_pairwise(logr, N, mean) / (double)(N - 1);
    *value = mean;
    *se = sqrt(var / (double)N);
}

/* _findmax_skipnan -- src/utils.jl:55-72.  Returns the 1-based index (0 for empty input). */
long pfo_findmax_skipnan(long n, const double *x, double *maxval) {
    if (n == 0) { *maxval = NAN; return 0; }
    double xmax = x[0]; long imax = 1;
    for (long i = 1; i < n; ++i) {
        double xi = x[i];
        if (isnan(xi)) continue;
        if (isnan(xmax) || xi > xmax) { xmax = xi; imax = i + 1; }
    }
    *maxval = xmax;
    return imax;
}

/* ------------------------------------------------------------------------------------------ */
/* Built-in targets.  X is d x N column-major.                                  */
/* Gaussian family: logp(x) = offset - 1/2 [ sum_i a_i e_i^2 - || G (Wd' e) ||^2 ], e = x-mean */
/*   iso: a = 1, r = 0 (test/singlepath.jl:15); diag: a = 1/sigma^2, r = 0;                    */
/*   low-rank+diag: Sigma* = diag(sigma^2) + W W', Wd = W ./ sigma^2 (d x r col-major),         */
/*   G = inv(chol_lower(I + W' diag(1/sigma^2) W)) (r x r col-major, lower triangular).        */
/* ------------------------------------------------------------------------------------------ */
void pfo_logp_gauss(int d, int r, const double *mean, const double *a, const double *Wd,
                    const double *G, double offset, long N, const double *X, double *out) {
    double t[64], g[64];
    for (long c = 0; c < N; ++c) {
        const double *x = X + (long)d * c;
        double q = 0.0;
        for (int j = 0; j < r; ++j) t[j] = 0.0;
        for (int i = 0; i < d; ++i) {
            double e = x[i] - mean[i];
            q += a[i] * e * e;
            for (int j = 0; j < r; ++j) t[j] += Wd[i + (long)d * j] * e;
        }
        double corr = 0.0;
        for (int j = 0; j < r; ++j) {
            g[j] = 0.0;
            for (int l = 0; l <= j; ++l) g[j] += G[j + r * l] * t[l];
            corr += g[j] * g[j];
        }
        out[c] = offset - 0.5 * (q - corr);
    }
}
/* funnel: docs/src/examples/quickstart.md:229-234 */
void pfo_logp_funnel(int d, long N, const double *X, double *out) {
    for (long c = 0; c < N; ++c) {
        const double *x = X + (long)d * c;
        double tau = x[0], ss = 0.0;
        double e = exp(-tau / 2.0);
        for (int i = 1; i < d; ++i) { double b = x[i] * e; ss += b * b; }
        out[c] = ((tau / 3.0) * (tau / 3.0) + (double)(d - 1) * tau + ss) / -2.0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Counter-based RNG (this repo's replacement for Random.randn!, src/mvnormal.jl:30).          */
/* Philox4x32 (Salmon et al. 2011), key = 64-bit per-fit seed.  Normal number for              */
/* (row i, draw n): counter = (n, i/4, stream, 0) -> 4 x u32 -> one normal per word (rows       */
/* 4g .. 4g+3) through the tabulated inverse normal CDF below (round 1: Box-Muller pairs).      */
/* ------------------------------------------------------------------------------------------ */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
/* Philox4x32-R (Salmon, Moraes, Dror, Shaw 2011).  R = 10 everywhere a host can see the bits (seeds, resampling uniforms; pinned by
 * the Random123 known answers in tests/test_oracle_elbo_psis.py); the normal-generation stream uses R = PFO_NORMAL_ROUNDS = 7,
 * the crush-resistant minimum of the paper's table 2 (same round function and key schedule, three rounds fewer). */
#define PFO_NORMAL_ROUNDS 7
void pfo_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < rounds; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
void pfo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { pfo_philox4x32_r(ctr, key, 10, out); }
/* The standard-normal generator (replaces Julia's randn! of src/mvnormal.jl:30, whose Xoshiro/ziggurat stream cannot be
 * reproduced outside Julia -- SURVEY.md H3).  One 32-bit Philox word -> one normal through the piecewise-cubic inverse normal
 * CDF tabulated in pathfinder.jl_amd/csrc/pfmi_icdftab.h (generated by pathfinder.jl_amd/tools/gen_icdf_table.py, which
 * documents the construction; |Q(p) + Phi^-1(p)| <= 7.5e-10).  That header is DATA shared with the device code on purpose: the
 * table IS the definition of the generator, and because the evaluation uses only exactly rounded IEEE operations
 * (int -> double, fma) the GPU reproduces these normals BIT FOR BIT.  The table itself is pinned against
 * scipy.special.ndtri and by distribution tests in tests/test_oracle_rng.py. */
#include "../pathfinder.jl_amd/csrc/pfmi_icdftab.h"
static const double PFO_ICDF_TAB[PF_ICDF_ENTRIES][4] = { PF_ICDF_TABLE_ROWS };

/* Q ~ -Phi^-1(p) for p in (2^-65, 1/2); argument v = the polynomial variable of the table (gen_icdf_table.py): mag in the common
 * case (p = (mag + 1/2) 2^-32, the half is folded into the coefficients), 2^32 p in the tail.  The interval is read off the exponent
 * and the top PF_ICDF_B mantissa bits of v; the cubic is stored in global monomial form: three fma. */
double pfo_icdf_q(double v) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    uint32_t hi = (uint32_t)(bits >> 32);
    int idx = PF_ICDF_IDX0 - (int)(hi >> (20 - PF_ICDF_B));
    const double *c = PFO_ICDF_TAB[idx];
    return fma(fma(fma(c[3], v, c[2]), v, c[1]), v, c[0]);
}
/* one normal from the Philox word x; x2 = the matching word of the refinement call (counter word 3 = 1), only read when
 * mag < 2^PF_ICDF_TAILBITS (probability 2^-19) */
double pfo_icdf_normal(uint32_t x, uint32_t x2) {
    const uint32_t mag = x & 0x7FFFFFFFu;
    double v;
    if (mag >= (1u << PF_ICDF_TAILBITS)) v = (double)mag;                                        /* p = (mag + 1/2) 2^-32 */
    else v = ((double)(((uint64_t)mag << 32) | x2) + 0.5) * 0x1p-32;                              /* exact: < 2^44 */
    const double q = pfo_icdf_q(v);
    return (x >> 31) ? -q : q;
}
void pfo_randn4(uint64_t seed, uint32_t g, uint32_t n, uint32_t stream, double z[4]) {
    uint32_t ctr[4] = {n, g, stream, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4], x2[4] = {0u, 0u, 0u, 0u};
    pfo_philox4x32_r(ctr, key, PFO_NORMAL_ROUNDS, x);
    int tail = 0;
    for (int t = 0; t < 4; ++t) tail |= (x[t] & 0x7FFFFFFFu) < (1u << PF_ICDF_TAILBITS);
    if (tail) { ctr[3] = 1u; pfo_philox4x32_r(ctr, key, PFO_NORMAL_ROUNDS, x2); }
    for (int t = 0; t < 4; ++t) z[t] = pfo_icdf_normal(x[t], x2[t]);
}
/* fill U (d x N col-major) with the standard normals of draws n0 .. n0+N-1 of `seed` */
void pfo_randn_fill(uint64_t seed, int d, long n0, long N, double *U) {
    for (long n = 0; n < N; ++n)
        for (int g = 0; 4 * g < d; ++g) {
            double z[4];
            pfo_randn4(seed, (uint32_t)g, (uint32_t)(n0 + n), 0u, z);
            for (int t = 0; t < 4 && 4 * g + t < d; ++t) U[(4 * g + t) + (long)d * n] = z[t];
        }
}
/* 64 random bits for (seed, counter t, stream) -- used by the resampler and the seed hierarchy */
uint64_t pfo_rand_u64(uint64_t seed, uint64_t t, uint32_t stream) {
    uint32_t ctr[4] = {(uint32_t)t, (uint32_t)(t >> 32), stream, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    pfo_philox4x32_10(ctr, key, x);
    return (uint64_t)x[0] | ((uint64_t)x[1] << 32);
}

/* ------------------------------------------------------------------------------------------ */
/* PSIS -- restatement of PSIS.psis (third party, called at src/resample.jl:78).               */
/* Algorithm: Vehtari, Simpson, Gelman, Yao, Gabry (2024) + Zhang & Stephens (2009) GPD fit,   */
/* as implemented by PSIS.jl 0.9 (psis!, psis_tail!, fit_gpd with min_points = 30, prior 3,     */
/* shape prior-adjusted (k M + 5)/(M + 10)) and identically by ArviZ/loo.  reff = 1.            */
/* logw (S) in: log ratios; out: smoothed + normalised log weights.  weights = exp(logw).      */
/* ------------------------------------------------------------------------------------------ */
static int cmp_idx_by_val(const void *a, const void *b, void *ctx) {
    const double *v = (const double *)ctx;
    long ia = *(const long *)a, ib = *(const long *)b;
    if (v[ia] < v[ib]) return -1;
    if (v[ia] > v[ib]) return 1;
    return (ia > ib) - (ia < ib);   /* stable tie-break by index */
}
long pfo_psis_tail_length(long S) {
    long a = (S + 4) / 5;                                  /* cld(S, 5) */
    long b = (long)ceil(3.0 * sqrt((double)S));            /* ceil(3 sqrt(S / reff)) */
    return a < b ? a : b;
}
/* Zhang-Stephens fit on sorted-ascending x (n points > = 0).  Returns sigma, k (un-adjusted). */
void pfo_gpd_fit(long n, const double *x, double *sigma, double *kshape) {
    const double prior = 3.0;
    long m = 30 + (long)floor(sqrt((double)n));
    double xstar = x[(n + 2) / 4 - 1];                      /* first quartile  x[fld(n+2,4)] */
    double xmax = x[n - 1];
    double *theta = (double *)malloc(sizeof(double) * m);
    double *ll = (double *)malloc(sizeof(double) * m);
    for (long i = 0; i < m; ++i) {
        double p = ((double)(i + 1) - 0.5) / (double)m;
        theta[i] = 1.0 / xmax + (1.0 - sqrt(1.0 / p)) / (prior * xstar);
        double kk = 0.0;
        for (long t = 0; t < n; ++t) kk += log1p(-theta[i] * x[t]);
        kk /= (double)n;
        ll[i] = (double)n * (log(-theta[i] / kk) - kk - 1.0);
    }
    double lmax = -INFINITY;
    for (long i = 0; i < m; ++i) if (ll[i] > lmax) lmax = ll[i];
    double wsum = 0.0, tsum = 0.0;
    for (long i = 0; i < m; ++i) { double w = exp(ll[i] - lmax); wsum += w; tsum += w * theta[i]; }
    double th = tsum / wsum;
    double kk = 0.0;
    for (long t = 0; t < n; ++t) kk += log1p(-th * x[t]);
    kk /= (double)n;
    *kshape = kk;
    *sigma = -kk / th;
    free(theta); free(ll);
}
static double gpd_quantile(double p, double sigma, double k) {
    double nl = -log1p(-p);
    double z = (k == 0.0) ? nl : expm1(k * nl) / k;
    return sigma * z;
}
static double logsumexp(const double *x, long n) {
    double mx = -INFINITY;
    for (long i = 0; i < n; ++i) if (x[i] > mx) mx = x[i];
    if (!isfinite(mx)) return mx;
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += exp(x[i] - mx);
    return mx + log(s);
}
/* returns the tail length M; *pareto_k = NaN when no fit was possible */
long pfo_psis(long S, double *logw, double *weights, double *pareto_k) {
    *pareto_k = NAN;
    long M = pfo_psis_tail_length(S);
    if (S > 0 && M >= 5) {
        long *perm = (long *)malloc(sizeof(long) * S);
        for (long i = 0; i < S; ++i) perm[i] = i;
        qsort_r(perm, (size_t)S, sizeof(long), cmp_idx_by_val, logw);  /* ascending */
        long cutoff_ind = perm[S - M - 1];
        const long *tail = perm + (S - M);
        double logu = logw[cutoff_ind];
        int finite = 1;
        for (long i = 0; i < M; ++i) if (!isfinite(logw[tail[i]])) finite = 0;
        if (finite) {
            double lmax = logw[tail[M - 1]];
            double mu_scaled = exp(logu - lmax);
            double *w = (double *)malloc(sizeof(double) * M);
            int allzero = 1;
            for (long i = 0; i < M; ++i) {
                w[i] = exp(logw[tail[i]] - lmax) - mu_scaled;
                if (w[i] != 0.0) allzero = 0;
            }
            double sigma = NAN, k = NAN;
            if (!allzero) pfo_gpd_fit(M, w, &sigma, &k);
            if (isfinite(k)) k = (k * (double)M + 5.0) / ((double)M + 10.0);   /* prior adjust */
            *pareto_k = k;
            if (isfinite(k) && isfinite(sigma)) {
                for (long i = 0; i < M; ++i) {
                    double p = ((double)(i + 1) - 0.5) / (double)M;
                    double v = log(gpd_quantile(p, sigma, k) + mu_scaled);
                    if (v > 0.0) v = 0.0;
                    logw[tail[i]] = v + lmax;
                }
            }
            free(w);
        }
        free(perm);
    }
    double lse = logsumexp(logw, S);
    for (long i = 0; i < S; ++i) { logw[i] -= lse; weights[i] = exp(logw[i]); }
    return M;
}

/* ------------------------------------------------------------------------------------------ */
/* Resampling -- stands in for StatsBase.sample at src/resample.jl:61-66 (third party).        */
/* Deterministic fixed-point inverse-CDF ("direct") sampler:                                   */
/*   q_i = floor(w_i 2^62) (u64), C = inclusive prefix sums (exact integer arithmetic, hence   */
/*   independent of summation order / GPU count), Q = C[S-1];                                  */
/*   draw t: r = mulhi64(R_t, Q), index = first i with C[i] > r.                               */
/*   R_t is either 64 Philox bits (seed, t, stream 1) or floor(u_t 2^64) for given uniforms.   */
/* Indices returned 0-based.  Returns 0, or -1 when every weight is zero / non-finite.         */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t mulhi64(uint64_t a, uint64_t b) {
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}
uint64_t pfo_weight_to_fixed(double w) {
    if (!(w > 0.0)) return 0;
    if (w >= 1.0) return (uint64_t)1 << 62;
    return (uint64_t)floor(w * 4611686018427387904.0);  /* 2^62, exact scaling */
}
uint64_t pfo_uniform_to_bits(double u) {
    /* u in [0,1): top 53 bits */
    return ((uint64_t)floor(u * 9007199254740992.0)) << 11;
}
int pfo_sample_weighted(long S, const double *weights, long ndraws, uint64_t seed,
                        const double *uniforms /* may be NULL */, int64_t *idx) {
    uint64_t *C = (uint64_t *)malloc(sizeof(uint64_t) * (S > 0 ? S : 1));
    uint64_t acc = 0;
    for (long i = 0; i < S; ++i) { acc += pfo_weight_to_fixed(weights[i]); C[i] = acc; }
    if (S <= 0 || acc == 0) { free(C); return -1; }
    for (long t = 0; t < ndraws; ++t) {
        uint64_t R = uniforms ? pfo_uniform_to_bits(uniforms[t]) : pfo_rand_u64(seed, (uint64_t)t, 1u);
        uint64_t r = mulhi64(R, acc);
        long lo = 0, hi = S - 1;                 /* first i with C[i] > r */
        while (lo < hi) { long mid = (lo + hi) >> 1; if (C[mid] > r) hi = mid; else lo = mid + 1; }
        idx[t] = lo;
    }
    free(C);
    return 0;
}
/* uniform with replacement: StatsBase.sample(rng, 1:n, ndraws) when psis_result === nothing */
void pfo_sample_uniform(long S, long ndraws, uint64_t seed, const double *uniforms, int64_t *idx) {
    for (long t = 0; t < ndraws; ++t) {
        uint64_t R = uniforms ? pfo_uniform_to_bits(uniforms[t]) : pfo_rand_u64(seed, (uint64_t)t, 1u);
        idx[t] = (int64_t)mulhi64(R, (uint64_t)S);
    }
}
/* weighted WITHOUT replacement: Efraimidis-Spirakis keys e_i / w_i, e_i ~ Exp(1) from
 * Philox (seed, i, stream 2); the ndraws smallest keys, in ascending key order
 * (equivalent in distribution to StatsBase's efraimidis_aexpj_wsample_norep!). */
static int cmp_idx_by_val_d(const void *a, const void *b, void *ctx) { return cmp_idx_by_val(a, b, ctx); }
int pfo_sample_weighted_norep(long S, const double *weights, long ndraws, uint64_t seed, int64_t *idx) {
    if (ndraws > S) return -1;
    double *key = (double *)malloc(sizeof(double) * S);
    long *perm = (long *)malloc(sizeof(long) * S);
    long npos = 0;
    for (long i = 0; i < S; ++i) {
        uint64_t R = pfo_rand_u64(seed, (uint64_t)i, 2u);
        double u = ((double)(R >> 11) + 0.5) * 1.1102230246251565404e-16; /* (x+0.5) 2^-53 */
        double w = weights[i];
        key[i] = (w > 0.0) ? (-log(u) / w) : INFINITY;
        if (w > 0.0) npos++;
        perm[i] = i;
    }
    if (npos < ndraws) { free(key); free(perm); return -1; }
    qsort_r(perm, (size_t)S, sizeof(long), cmp_idx_by_val_d, key);
    for (long t = 0; t < ndraws; ++t) idx[t] = perm[t];
    free(key); free(perm);
    return 0;
}

/* StatsBase.direct_sample!(rng, 1:S, ProbabilityWeights(w, 1), x) -- the weighted branch StatsBase.sample takes for small
 * requests (src/resample.jl:61-66 calls StatsBase.sample(rng, axes(draws_all, 2), pweights, ndraws; replace)); literal loop of
 * StatsBase.sample(rng, wv):  t = rand(rng) * sum(wv); i = 1; cw = wv[1]; while cw < t && i < n; i += 1; cw += wv[i]; end.
 * uniforms[t] plays rand(rng); sum(wv) = 1 by construction.  0-based output.  (StatsBase is third party: restated from its
 * published source, parity unpinned -- see header.) */
void pfo_sample_direct(long S, const double *w, long ndraws, const double *uniforms, int64_t *idx) {
    for (long t = 0; t < ndraws; ++t) {
        const double thr = uniforms[t] * 1.0;
        long i = 0;
        double cw = w[0];
        while (cw < thr && i < S - 1) { i += 1; cw += w[i]; }
        idx[t] = i;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Whole-path driver used for parity at scale and as the CPU baseline:                         */
/* fit every point of one trace, run the ELBO over points 1..L with the Philox normals of      */
/* seeds[l], return elbo/se per point and the NaN-skipping argmax.  Mirrors _pathfinder,       */
/* src/singlepath.jl:301-308 -> src/mvnormal.jl:14-39 -> src/elbo.jl:1-20.                      */
/* target_kind: 0 gauss family, 1 funnel.  Per-point outputs have L+1 entries (entry 0: fit    */
/* only, elbo = NaN).  best_iter is 1-based over points 1..L (0 if L == 0).                    */
/* If draws_best != NULL it receives the d x N draws of the winning fit, logp_best/logq_best   */
/* their densities.                                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int kind, d, r;
    const double *mean, *a, *Wd, *G;
    double offset;
} pfo_target;

static void eval_target(const pfo_target *t, long N, const double *X, double *out) {
    if (t->kind == 1) pfo_logp_funnel(t->d, N, X, out);
    else pfo_logp_gauss(t->d, t->r, t->mean, t->a, t->Wd, t->G, t->offset, N, X, out);
}

int pfo_path_fit_elbo(int d, int L, const double *theta, const double *grad, int J, double eps,
                      int kind, int r, const double *mean, const double *a, const double *Wd,
                      const double *G, double offset, long N, const uint64_t *seeds,
                      double *mu_all /* (L+1)*d or NULL */, double *logdet_all, int *status_all,
                      int *jeff_all, double *elbo, double *se, long *best_iter, int *n_rejected,
                      double *draws_best, double *logp_best, double *logq_best) {
    pfo_target tg = {kind, d, r, mean, a, Wd, G, offset};
    int mmax = 2 * J;
    double *alpha_all = (double *)malloc(sizeof(double) * (size_t)(L + 1) * d);
    int *hist_src = (int *)malloc(sizeof(int) * (size_t)(L + 1) * (J > 0 ? J : 1));
    *n_rejected = pfo_lbfgs_history(d, L, theta, grad, J, eps, alpha_all, jeff_all, hist_src);
    double *S = (double *)malloc(sizeof(double) * (size_t)d * (J > 0 ? J : 1));
    double *Y = (double *)malloc(sizeof(double) * (size_t)d * (J > 0 ? J : 1));
    double *B = (double *)malloc(sizeof(double) * (size_t)d * (mmax > 0 ? mmax : 1));
    double *D = (double *)malloc(sizeof(double) * (size_t)(mmax > 0 ? mmax * mmax : 1));
    double *QR = (double *)malloc(sizeof(double) * (size_t)d * (mmax > 0 ? mmax : 1));
    double *tau = (double *)malloc(sizeof(double) * (size_t)(mmax > 0 ? mmax : 1));
    double *V = (double *)malloc(sizeof(double) * (size_t)(mmax > 0 ? mmax * mmax : 1));
    double *sa = (double *)malloc(sizeof(double) * d);
    double *mu = (double *)malloc(sizeof(double) * d);
    double *U = (double *)malloc(sizeof(double) * (size_t)d * (N > 0 ? N : 1));
    double *lp = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
    double *lq = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
    double *lr = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
    double best = NAN; long besti = 0;
    for (int l = 0; l <= L; ++l) {
        int j = jeff_all[l], m = 2 * j, k = d < m ? d : m;
        for (int c = 0; c < j; ++c) {
            int src = hist_src[(long)l * J + c];
            for (int i = 0; i < d; ++i) {
                S[i + (long)d * c] = theta[(long)(src + 1) * d + i] - theta[(long)src * d + i];
                Y[i + (long)d * c] = grad[(long)src * d + i] - grad[(long)(src + 1) * d + i];
            }
        }
        const double *alpha = alpha_all + (long)l * d;
        pfo_lbfgs_inverse_hessian(d, j, alpha, S, Y, B, D);
        int st = pfo_pdfactorize(d, m, alpha, B, D, sa, QR, tau, V);
        status_all[l] = st;
        elbo[l] = NAN; se[l] = NAN; logdet_all[l] = NAN;
        if (st != PFO_OK) { if (mu_all) for (int i = 0; i < d; ++i) mu_all[(long)l * d + i] = NAN; continue; }
        double logdet = pfo_logdet(d, k, sa, V);
        logdet_all[l] = logdet;
        pfo_fit_mean(d, m, sa, QR, tau, V, theta + (long)l * d, grad + (long)l * d, mu);
        if (mu_all) memcpy(mu_all + (long)l * d, mu, sizeof(double) * d);
        if (l == 0 || N <= 0) continue;
        pfo_randn_fill(seeds[l], d, 0, N, U);
        pfo_rand_and_logpdf(d, m, sa, QR, tau, V, mu, logdet, N, U, lq);
        eval_target(&tg, N, U, lp);
        pfo_elbo_stats(N, lp, lq, lr, &elbo[l], &se[l]);
        /* running _findmax_skipnan over points 1..L (src/utils.jl:57-72) */
        int take = 0;
        if (besti == 0) take = 1;
        else if (!isnan(elbo[l]) && (isnan(best) || elbo[l] > best)) take = 1;
        if (take) {
            best = elbo[l]; besti = l;
            if (draws_best) memcpy(draws_best, U, sizeof(double) * (size_t)d * N);
            if (logp_best) memcpy(logp_best, lp, sizeof(double) * N);
            if (logq_best) memcpy(logq_best, lq, sizeof(double) * N);
        }
    }
    *best_iter = besti;
    free(alpha_all); free(hist_src); free(S); free(Y); free(B); free(D); free(QR); free(tau);
    free(V); free(sa); free(mu); free(U); free(lp); free(lq); free(lr);
    return 0;
}

/* Multi-path fan-out (OpenMP over paths = the reference's ntasks loop, src/multipath.jl:190-208).
 * Traces are concatenated: path k owns points off[k] .. off[k+1]-1.  Returns total ELBO draws. */
long pfo_multipath_fit_elbo(int K, const long *off, int d, const double *theta, const double *grad,
                            int J, double eps, int kind, int r, const double *mean, const double *a,
                            const double *Wd, const double *G, double offset, long N,
                            const uint64_t *seeds, double *elbo, double *se, long *best_iter,
                            int *status_all, int *jeff_all, double *logdet_all, int *n_rejected,
                            int nthreads) {
    long total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total)
#endif
    for (int k = 0; k < K; ++k) {
        long p0 = off[k];
        int L = (int)(off[k + 1] - off[k] - 1);
        pfo_path_fit_elbo(d, L, theta + p0 * d, grad + p0 * d, J, eps, kind, r, mean, a, Wd, G, offset,
                          N, seeds + p0, NULL, logdet_all + p0, status_all + p0, jeff_all + p0,
                          elbo + p0, se + p0, &best_iter[k], &n_rejected[k], NULL, NULL, NULL);
        total += (long)L * N;
    }
    (void)nthreads;
    return total;
}

/* ------------------------------------------------------------------------------------------ */
/* L-BFGS trajectory generation: produces the hot path's INPUT.           */
/* The reference delegates to Optim.LBFGS + LineSearches.HagerZhang (src/optimize.jl:35-59,     */
/* src/Pathfinder.jl:29-35) -- third party, absent from /root/reference, no trajectory values   */
/* pinned by the reference's tests (test/optimize.jl:96-136 checks convergence + trace shape    */
/* only): PARITY UNPINNED against Optim.  What is restated here is this repo's own driver       */
/* (two-loop recursion with gamma = s'y / y'y, strong-Wolfe bracketing + bisection zoom,         */
/* maxiters 1000 as src/optimize.jl:40, stop when |g|_inf <= g_tol), the same algorithm the      */
/* device kernel pf_lbfgs_kernel runs; the trace (theta_l, logp_l, grad logp_l) has the layout   */
/* of OptimizationTrace (src/optimize.jl:94-100).                                                */
/* ------------------------------------------------------------------------------------------ */
/* f = -logp and its gradient at x */
static double neg_logp_grad(const pfo_target *t, const double *x, double *gout) {
    int d = t->d;
    if (t->kind == 1) {                          /* funnel, docs/src/examples/quickstart.md:229-234 */
        double tau = x[0], ss = 0.0, e = exp(-tau);
        for (int i = 1; i < d; ++i) ss += x[i] * x[i];
        gout[0] = 0.5 * (2.0 * tau / 9.0 + (double)(d - 1) - e * ss);
        for (int i = 1; i < d; ++i) gout[i] = e * x[i];
        return 0.5 * ((tau / 3.0) * (tau / 3.0) + (double)(d - 1) * tau + e * ss);
    }
    int r = t->r;
    double tt[64], gg[64], hh[64], q = 0.0, corr = 0.0;
    for (int j = 0; j < r; ++j) tt[j] = 0.0;
    for (int i = 0; i < d; ++i) {
        double e = x[i] - t->mean[i];
        q += t->a[i] * e * e;
        for (int j = 0; j < r; ++j) tt[j] += t->Wd[i + (long)d * j] * e;
    }
    for (int j = 0; j < r; ++j) {
        gg[j] = 0.0;
        for (int l = 0; l <= j; ++l) gg[j] += t->G[j + r * l] * tt[l];
        corr += gg[j] * gg[j];
    }
    for (int l = 0; l < r; ++l) {                /* hh = G' gg */
        hh[l] = 0.0;
        for (int j = l; j < r; ++j) hh[l] += t->G[j + r * l] * gg[j];
    }
    for (int i = 0; i < d; ++i) {
        double v = t->a[i] * (x[i] - t->mean[i]);
        for (int j = 0; j < r; ++j) v -= t->Wd[i + (long)d * j] * hh[j];
        gout[i] = v;
    }
    return 0.5 * (q - corr) - t->offset;
}

/* logp and its gradient at one point (checker for the trace the device optimiser records) */
double pfo_logp_grad(int kind, int d, int r, const double *mean, const double *a, const double *Wd, const double *G,
                     double offset, const double *x, double *grad) {
    pfo_target tg = {kind, d, r, mean, a, Wd, G, offset};
    double f = neg_logp_grad(&tg, x, grad);
    for (int i = 0; i < d; ++i) grad[i] = -grad[i];
    return -f;
}

typedef struct {
    const pfo_target *t;
    int d;
    const double *x, *p;
    double *xn, *gn;     /* last evaluated point and gradient (the "pack") */
    double fn;
    int evaluated;
} ls_ctx;

static double dotd(int d, const double *a, const double *b) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}
static void ls_phi(ls_ctx *c, double a, double *f, double *g) {
    for (int i = 0; i < c->d; ++i) c->xn[i] = c->x[i] + a * c->p[i];
    c->fn = neg_logp_grad(c->t, c->xn, c->gn);
    c->evaluated = 1;
    *f = c->fn;
    *g = dotd(c->d, c->gn, c->p);
}
static double ls_zoom(ls_ctx *c, double lo, double hi, double f_lo, double f_hi, double f0, double g0,
                      double c1, double c2) {
    double a = 0.5 * (lo + hi);
    (void)f_hi;
    for (int it = 0; it < 30; ++it) {
        double f, g;
        a = 0.5 * (lo + hi);
        ls_phi(c, a, &f, &g);
        if ((f > f0 + c1 * a * g0) || (f >= f_lo)) {
            hi = a;
        } else {
            if (fabs(g) <= -c2 * g0) return a;
            if (g * (hi - lo) >= 0) hi = lo;
            lo = a; f_lo = f;
        }
    }
    return a;
}
static double ls_search(ls_ctx *c, double f0, double g0, double a_init) {
    const double c1 = 1e-4, c2 = 0.9, amax = 1e10;
    double a_prev = 0.0, f_prev = f0, a = a_init;
    for (int it = 0; it < 25; ++it) {
        double f, g;
        ls_phi(c, a, &f, &g);
        if (!isfinite(f)) { a = 0.5 * (a_prev + a); continue; }
        if ((f > f0 + c1 * a * g0) || (it > 0 && f >= f_prev)) return ls_zoom(c, a_prev, a, f_prev, f, f0, g0, c1, c2);
        if (fabs(g) <= -c2 * g0) return a;
        if (g >= 0) return ls_zoom(c, a, a_prev, f, f_prev, f0, g0, c1, c2);
        a_prev = a; f_prev = f;
        a = 2 * a < amax ? 2 * a : amax;
    }
    return a;
}

/* Minimise f = -logp from x0.  pts/grads: (maxiters+1) x d point-major; lps: maxiters+1.
 * grads hold the gradient of the LOG DENSITY.  Returns the number of trace points (>= 1). */
int pfo_optimize_trace(int kind, int d, int r, const double *mean, const double *a, const double *Wd,
                       const double *G, double offset, const double *x0, int J, int maxiters, double g_tol,
                       double *pts, double *lps, double *grads) {
    pfo_target tg = {kind, d, r, mean, a, Wd, G, offset};
    double *x = (double *)malloc(sizeof(double) * d), *g = (double *)malloc(sizeof(double) * d);
    double *q = (double *)malloc(sizeof(double) * d), *p = (double *)malloc(sizeof(double) * d);
    double *xn = (double *)malloc(sizeof(double) * d), *gn = (double *)malloc(sizeof(double) * d);
    double *S = (double *)malloc(sizeof(double) * (size_t)d * J), *Y = (double *)malloc(sizeof(double) * (size_t)d * J);
    double al[64];
    int h = 0, n = 0;                                   /* history length (oldest first in slots 0..h-1) */
    memcpy(x, x0, sizeof(double) * d);
    double f = neg_logp_grad(&tg, x, g);
    memcpy(pts, x, sizeof(double) * d); lps[0] = -f;
    for (int i = 0; i < d; ++i) grads[i] = -g[i];
    n = 1;
    for (int it = 0; it < maxiters; ++it) {
        int fin = isfinite(f);
        double gmax = 0.0;
        for (int i = 0; i < d; ++i) { if (!isfinite(g[i])) fin = 0; if (fabs(g[i]) > gmax) gmax = fabs(g[i]); }
        if (!fin) break;
        if (gmax <= g_tol) break;
        memcpy(q, g, sizeof(double) * d);
        for (int c = h - 1; c >= 0; --c) {
            const double *s = S + (size_t)c * d, *y = Y + (size_t)c * d;
            double rho = 1.0 / dotd(d, y, s);
            al[c] = rho * dotd(d, s, q);
            for (int i = 0; i < d; ++i) q[i] -= al[c] * y[i];
        }
        if (h) {
            const double *s = S + (size_t)(h - 1) * d, *y = Y + (size_t)(h - 1) * d;
            double gam = dotd(d, s, y) / dotd(d, y, y);
            for (int i = 0; i < d; ++i) q[i] *= gam;
        }
        for (int c = 0; c < h; ++c) {
            const double *s = S + (size_t)c * d, *y = Y + (size_t)c * d;
            double rho = 1.0 / dotd(d, y, s);
            double b = rho * dotd(d, y, q);
            for (int i = 0; i < d; ++i) q[i] += (al[c] - b) * s[i];
        }
        for (int i = 0; i < d; ++i) p[i] = -q[i];
        double g0 = dotd(d, g, p);
        if (g0 >= 0) {
            h = 0;
            for (int i = 0; i < d; ++i) p[i] = -g[i];
            g0 = dotd(d, g, p);
        }
        double a0 = 1.0;
        if (!h) {
            double nrm = sqrt(dotd(d, g, g));
            if (nrm < 1e-300) nrm = 1e-300;
            a0 = 1.0 / nrm < 1.0 ? 1.0 / nrm : 1.0;
        }
        ls_ctx lc = {&tg, d, x, p, xn, gn, 0.0, 0};
        ls_search(&lc, f, g0, a0);
        if (!lc.evaluated) break;
        int ok = isfinite(lc.fn);
        for (int i = 0; i < d; ++i) if (!isfinite(gn[i])) ok = 0;
        if (!ok) {                                    /* src/optimize.jl:96-105: the offending iterate is recorded, then the run stops */
            memcpy(pts + (size_t)n * d, xn, sizeof(double) * d); lps[n] = -lc.fn;
            for (int i = 0; i < d; ++i) grads[(size_t)n * d + i] = -gn[i];
            ++n;
            break;
        }
        double ys = 0.0, yy = 0.0;
        int moved = 0;
        for (int i = 0; i < d; ++i) {
            double si = xn[i] - x[i], yi = gn[i] - g[i];
            ys += yi * si; yy += yi * yi;
            if (xn[i] != x[i]) moved = 1;
        }
        if (ys > 1e-10 * yy) {
            if (h == J) {
                memmove(S, S + d, sizeof(double) * (size_t)d * (J - 1));
                memmove(Y, Y + d, sizeof(double) * (size_t)d * (J - 1));
                h = J - 1;
            }
            for (int i = 0; i < d; ++i) { S[(size_t)h * d + i] = xn[i] - x[i]; Y[(size_t)h * d + i] = gn[i] - g[i]; }
            ++h;
        }
        memcpy(x, xn, sizeof(double) * d); memcpy(g, gn, sizeof(double) * d); f = lc.fn;
        memcpy(pts + (size_t)n * d, x, sizeof(double) * d); lps[n] = -f;
        for (int i = 0; i < d; ++i) grads[(size_t)n * d + i] = -g[i];
        ++n;
        if (!moved) break;
    }
    free(x); free(g); free(q); free(p); free(xn); free(gn); free(S); free(Y);
    return n;
}
