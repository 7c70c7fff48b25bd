"""The data flow of csrc/fit_tsqr_kernel.hip stated in NumPy: TSQR over row chunks + Householder reconstruction (Ballard et al. 2014; LAPACK dorhr_col)
reproduces the reflectors V, the compact-WY T and R that LAPACK's dgeqr2 / dlarft produce for the whole block, and the kernel's row-local formulas
(N_i = -K_i W_i U^-1, the W_i U^-1 term of a chunk's first m rows, the mean through Q_in S (V_c'V_c - I) head) give Vh and mu without a sweep.

    python tools/tsqr_hr_check.py        -> prints the deviations from dgeqrf / dlarft on a few shapes
    tests/test_tsqr_hr_math.py           -> asserts them (CPU)
"""
import numpy as np
import scipy.linalg as sl

def geqr2(A):
    """LAPACK dgeqr2/dlarfg convention. returns V (unit lower trapezoidal explicit), tau, R"""
    A = A.copy(); n, m = A.shape; k = min(n, m)
    tau = np.zeros(k)
    for c in range(k):
        alpha = A[c, c]; x = A[c+1:, c]; xn = np.sqrt(np.dot(x, x))
        if xn == 0.0:
            tau[c] = 0.0; continue
        beta = -np.copysign(np.hypot(alpha, xn), alpha)
        tau[c] = (beta - alpha) / beta
        A[c+1:, c] = x / (alpha - beta)
        A[c, c] = beta
        v = np.concatenate([[1.0], A[c+1:, c]])
        w = tau[c] * (v @ A[c:, c+1:])
        A[c:, c+1:] -= np.outer(v, w)
    R = np.triu(A[:k, :])
    V = np.tril(A[:, :k], -1); V[np.arange(k), np.arange(k)] = 1.0
    return V, tau, R

def larft(V, tau):
    k = V.shape[1]; T = np.zeros((k, k))
    for c in range(k):
        T[c, c] = tau[c]
        if c > 0: T[:c, c] = -tau[c] * T[:c, :c] @ (V[:, :c].T @ V[:, c])
    return T

def tsqr_hr(B, ug, CH):
    d, m = B.shape
    nch = (d + CH - 1) // CH
    Vs, Ts, Rs, hs = [], [], [], []
    for i in range(nch):
        Bi = B[i*CH:(i+1)*CH]
        V, tau, R = geqr2(np.column_stack([Bi, ug[i*CH:(i+1)*CH]])[:, :])   # augmented column rides along
        # only the first m reflectors (the augmented column is transformed, not factored)
        Vb, taub, Rb = geqr2(Bi)
        T = larft(Vb, taub)
        ui = ug[i*CH:(i+1)*CH] - Vb @ (T.T @ (Vb.T @ ug[i*CH:(i+1)*CH]))      # Q_i' ug_i
        Vs.append(Vb); Ts.append(T); Rs.append(Rb[:m]); hs.append(ui[:m])
    Rst = np.vstack(Rs); hst = np.concatenate(hs)
    Vt, taut, Rin = geqr2(Rst); Tt = larft(Vt, taut)
    # explicit first m columns of Q_top, and Q_top' hst
    E = np.zeros((Rst.shape[0], m)); E[:m] = np.eye(m)
    Qtop = E - Vt @ (Tt @ Vt[:m].T)
    head_in = (hst - Vt @ (Tt.T @ (Vt.T @ hst)))[:m]                            # Q_in' ug (before sign fix)
    D = np.where(np.diag(Rin) < 0, -1.0, 1.0)
    Qtop = Qtop * D; Rin = D[:, None] * Rin[:m]; head_in = D * head_in
    # Q_in rows of chunk i = [W_i; 0] - V_i (T_i V_i[top]' W_i)
    Qin = np.zeros((d, m))
    for i in range(nch):
        Wi = Qtop[i*m:(i+1)*m]
        Mi = Ts[i] @ (Vs[i][:m].T @ Wi)
        blk = -Vs[i] @ Mi; blk[:m] += Wi
        Qin[i*CH:(i+1)*CH] = blk
    # Householder reconstruction: modified LU of Qin - S (top m x m), S chosen on the fly
    A = Qin.copy(); S = np.zeros(m)
    for c in range(m):
        S[c] = -1.0 if A[c, c] >= 0 else 1.0            # s = -sign(q_cc)
        A[c, c] -= S[c]
        A[c+1:, c] /= A[c, c]
        A[c+1:, c+1:] -= np.outer(A[c+1:, c], A[c, c+1:])
    V = np.tril(A, -1); V[np.arange(m), np.arange(m)] = 1.0
    U = np.triu(A[:m])
    # Qin - S_ = V U  with S_ = diag(S);  Q_out = Qin S_ (Householder represented: Q_out = E - V T V1');  =>  Qin S - I = V U S = -V T V1'  => T = -U S V1^-T
    T = -(U * S[None, :]) @ np.linalg.inv(V[:m].T)
    R = S[:, None] * Rin
    head = S * head_in                                   # Q_out' ug
    return V, T, R, head, Qin * S[None, :]

def check_reconstruction(d, m, CH, seed=0, ill=True):
    """TSQR + reconstruction against dgeqr2 / dlarft (and SciPy's dgeqrf): max deviations of V, T, R (relative), Q'(U g), Q"""
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((d, m)) * np.exp(rng.standard_normal(m))[None, :]
    if ill and m > 3:
        B[:, 3] = B[:, 2] * 0.7 + 1e-3 * B[:, 3]              # a column of condition ~1e3
    ug = rng.standard_normal(d)
    V0, tau0, R0 = geqr2(B)
    T0 = larft(V0, tau0)
    (qr_raw, tau_l), R_l = sl.qr(B, mode='raw')
    assert np.allclose(np.tril(qr_raw, -1)[:, :m], np.tril(V0, -1)) and np.allclose(np.triu(qr_raw)[:m], R0[:m])   # geqr2() above IS LAPACK's convention
    V, T, R, head, Qout = tsqr_hr(B, ug, CH)
    Q0 = np.eye(d, m) - V0 @ (T0 @ V0[:m].T)
    return dict(V=np.abs(V - V0).max(), T=np.abs(T - T0).max(), R=np.abs(R - R0[:m]).max() / np.abs(R0).max(), head=np.abs(head - Q0.T @ ug).max(),
                Q=np.abs(Qout - Q0).max(), T_lower=np.abs(np.tril(T, -1)).max())


# ---- second check: the kernel's row-local formulas (N_i, Wu_i, y_i, wd_i) and the mean
def kernel_dataflow(B, ug, theta, sqa, Vc, CH):
    d, m = B.shape; nch = (d + CH - 1) // CH
    Vs, Ks, st = [], [], []
    for i in range(nch):
        Bi = np.zeros((CH, m)); ui = np.zeros(CH)
        n = min(CH, d - i*CH); Bi[:n] = B[i*CH:i*CH+n]; ui[:n] = ug[i*CH:i*CH+n]
        Vb, taub, Rb = geqr2(Bi); T = larft(Vb, taub)
        h = (ui - Vb @ (T.T @ (Vb.T @ ui)))[:m]
        Vs.append(Vb); Ks.append(T @ Vb[:m].T); st.append(np.column_stack([Rb[:m], h]))
    st = np.vstack(st)
    Vt, taut, Rin = geqr2(st[:, :m]); Tt = larft(Vt, taut)
    head = (st[:, m] - Vt @ (Tt.T @ (Vt.T @ st[:, m])))[:m]
    Kt = Tt @ Vt[:m].T
    W = np.eye(st.shape[0], m) - Vt @ Kt
    D = np.where(np.diag(Rin) < 0, -1.0, 1.0)
    W = W * D; Rin = D[:, None] * Rin[:m]; head = D * head
    M0 = Ks[0] @ W[:m]
    A = W[:m] - Vs[0][:m] @ M0
    S = np.zeros(m)
    for c in range(m):
        S[c] = -1.0 if A[c, c] >= 0 else 1.0
        A[c, c] -= S[c]; A[c+1:, c] /= A[c, c]; A[c+1:, c+1:] -= np.outer(A[c+1:, c], A[c, c+1:])
    L = np.tril(A, -1) + np.eye(m); U = np.triu(A)
    T = np.zeros((m, m))
    for a in range(m):
        for c in range(a, m):
            T[a, c] = -U[a, c] * S[c] - T[a, a:c] @ L[c, a:c]
    R = S[:, None] * Rin; hb = S * head
    h2 = Vc.T @ (Vc @ hb); sdel = S * (h2 - hb)
    Uinv = np.linalg.inv(U)
    Vout = np.zeros((d, m)); mu = np.zeros(d)
    for i in range(nch):
        Wi = W[i*m:(i+1)*m]; Mi = Ks[i] @ Wi
        Ni = -Mi @ Uinv; yi = -Mi @ sdel; Wu = Wi @ Uinv; wd = Wi @ sdel
        n = min(CH, d - i*CH)
        out = Vs[i] @ Ni; yv = Vs[i] @ yi
        out[:m] += Wu; yv[:m] += wd
        if i == 0: out[:m] = L
        Vout[i*CH:i*CH+n] = out[:n]
        mu[i*CH:i*CH+n] = theta[i*CH:i*CH+n] + sqa[i*CH:i*CH+n] * (ug[i*CH:i*CH+n] + yv[:n])
    return Vout, T, R, mu

def check_kernel_dataflow(d, m, CH, seed=1):
    """the kernel's row-local formulas against the sweep-based reference: V, T, R, mu"""
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((d, m)) * np.exp(rng.standard_normal(m))[None, :]
    ug = rng.standard_normal(d); theta = rng.standard_normal(d); sqa = np.exp(rng.standard_normal(d))
    Cm = rng.standard_normal((m, m)); Vc = np.linalg.cholesky(np.eye(m) + Cm @ Cm.T).T
    V0, tau0, R0 = geqr2(B)
    T0 = larft(V0, tau0)
    def Qt(x): return x - V0 @ (T0.T @ (V0.T @ x))
    def Q(x): return x - V0 @ (T0 @ (V0.T @ x))
    b = Qt(ug); b[:m] = Vc.T @ (Vc @ b[:m]); mu0 = theta + sqa * Q(b)
    V, T, R, mu = kernel_dataflow(B, ug, theta, sqa, Vc, CH)
    return dict(V=np.abs(V - V0).max(), T=np.abs(T - T0).max(), R=np.abs(R - R0[:m]).max() / np.abs(R0).max(), mu=np.abs(mu - mu0).max() / np.abs(mu0).max())


if __name__ == "__main__":
    np.set_printoptions(linewidth=200, precision=3)
    for d, m, CH in [(1000, 12, 256), (10000, 20, 2048), (3000, 8, 1024), (2500, 20, 512)]:
        print("reconstruction", d, m, CH, {k: float("%.2e" % v) for k, v in check_reconstruction(d, m, CH).items()})
    for d, m, CH in [(1000, 12, 256), (10000, 20, 1536), (2500, 20, 512)]:
        print("kernel data flow", d, m, CH, {k: float("%.2e" % v) for k, v in check_kernel_dataflow(d, m, CH).items()})
