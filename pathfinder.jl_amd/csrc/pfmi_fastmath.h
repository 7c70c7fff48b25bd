// pfmi_fastmath.h -- domain-restricted fp64 transcendental kernels for the in-kernel normal generator.
//
// The Box-Muller transform only ever sees u = (x + 0.5) 2^-32 in (0,1), so the general-purpose ocml
// log / sqrt / sincospi (denormal, inf, NaN, huge-argument handling) are replaced by short sequences:
//   -2 ln u : exponent split + 128-entry table {inv_c, -log(inv_c)} + degree-6 log1p polynomial (|r| < 2^-8)
//   sqrt    : v_rsq_f64 seed + two Goldschmidt/Newton refinements (argument in [2e-10, 45])
//   sin/cos(2 pi u): exact quadrant reduction + Taylor polynomials on |pi f| <= pi/4
// Accuracy ~1e-15 absolute on the normals (checked against the ocml path by tools/microbench.hip and by
// the GPU parity tests, whose oracle uses glibc libm).
#pragma once
#include "pfmi_logtab.h"

#ifdef __HIPCC__
__constant__ double PF_LOGTAB_DEV[128][2] = {
#define PF_ROW(i) {PF_LOGTAB_HOST[i][0], PF_LOGTAB_HOST[i][1]}
    PF_ROW(0), PF_ROW(1), PF_ROW(2), PF_ROW(3), PF_ROW(4), PF_ROW(5), PF_ROW(6), PF_ROW(7), PF_ROW(8), PF_ROW(9),
    PF_ROW(10), PF_ROW(11), PF_ROW(12), PF_ROW(13), PF_ROW(14), PF_ROW(15), PF_ROW(16), PF_ROW(17), PF_ROW(18), PF_ROW(19),
    PF_ROW(20), PF_ROW(21), PF_ROW(22), PF_ROW(23), PF_ROW(24), PF_ROW(25), PF_ROW(26), PF_ROW(27), PF_ROW(28), PF_ROW(29),
    PF_ROW(30), PF_ROW(31), PF_ROW(32), PF_ROW(33), PF_ROW(34), PF_ROW(35), PF_ROW(36), PF_ROW(37), PF_ROW(38), PF_ROW(39),
    PF_ROW(40), PF_ROW(41), PF_ROW(42), PF_ROW(43), PF_ROW(44), PF_ROW(45), PF_ROW(46), PF_ROW(47), PF_ROW(48), PF_ROW(49),
    PF_ROW(50), PF_ROW(51), PF_ROW(52), PF_ROW(53), PF_ROW(54), PF_ROW(55), PF_ROW(56), PF_ROW(57), PF_ROW(58), PF_ROW(59),
    PF_ROW(60), PF_ROW(61), PF_ROW(62), PF_ROW(63), PF_ROW(64), PF_ROW(65), PF_ROW(66), PF_ROW(67), PF_ROW(68), PF_ROW(69),
    PF_ROW(70), PF_ROW(71), PF_ROW(72), PF_ROW(73), PF_ROW(74), PF_ROW(75), PF_ROW(76), PF_ROW(77), PF_ROW(78), PF_ROW(79),
    PF_ROW(80), PF_ROW(81), PF_ROW(82), PF_ROW(83), PF_ROW(84), PF_ROW(85), PF_ROW(86), PF_ROW(87), PF_ROW(88), PF_ROW(89),
    PF_ROW(90), PF_ROW(91), PF_ROW(92), PF_ROW(93), PF_ROW(94), PF_ROW(95), PF_ROW(96), PF_ROW(97), PF_ROW(98), PF_ROW(99),
    PF_ROW(100), PF_ROW(101), PF_ROW(102), PF_ROW(103), PF_ROW(104), PF_ROW(105), PF_ROW(106), PF_ROW(107), PF_ROW(108), PF_ROW(109),
    PF_ROW(110), PF_ROW(111), PF_ROW(112), PF_ROW(113), PF_ROW(114), PF_ROW(115), PF_ROW(116), PF_ROW(117), PF_ROW(118), PF_ROW(119),
    PF_ROW(120), PF_ROW(121), PF_ROW(122), PF_ROW(123), PF_ROW(124), PF_ROW(125), PF_ROW(126), PF_ROW(127)
#undef PF_ROW
};

// copy the table into LDS (call by all threads of the block, then __syncthreads())
__device__ __forceinline__ void pf_logtab_load(double2 *tab) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tab[i] = make_double2(PF_LOGTAB_DEV[i][0], PF_LOGTAB_DEV[i][1]);
}

// -2 ln(u), u in (0,1) normal double
__device__ __forceinline__ double pf_neg2log_fast(double u, const double2 *tab) {
    const long long bits = __double_as_longlong(u);
    const int e = (int)(bits >> 52) - 1023;
    const int idx = (int)(bits >> 45) & 127;
    const double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    const double2 t = tab[idx];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, -1.0 / 7.0, 1.0 / 6.0);
    p = fma(r, p, -1.0 / 5.0);
    p = fma(r, p, 1.0 / 4.0);
    p = fma(r, p, -1.0 / 3.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, -1.0);
    // -2 ln u = -2 (e ln2 + t.y) + 2 r p',  p' = -(1 - r/2 + r^2/3 - ...)  => ln(1+r) = -r p
    const double l = fma((double)e, 0.6931471805599453094, t.y);
    return fma(2.0 * r, p, -2.0 * l);
}

// sqrt(x), x in [1e-10, 100]
__device__ __forceinline__ double pf_sqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);          // ~2^-26 relative
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    return fma(d, h, g);
}

// sin(2 pi u), cos(2 pi u), u in (0,1)
__device__ __forceinline__ void pf_sincos2pi_fast(double u, double &s, double &c) {
    const double t4 = 4.0 * u;                    // quarter turns, exact
    const double q = rint(t4);
    const double f = 0.5 * (t4 - q);              // half-turn fraction in [-0.25, 0.25], exact
    const int iq = (int)q;
    const double f2 = f * f;
    double ps = -2.1915353447830217e-05;
    ps = fma(ps, f2, 0.00046630280576761255);
    ps = fma(ps, f2, -0.0073704309457143504);
    ps = fma(ps, f2, 0.08214588661112823);
    ps = fma(ps, f2, -0.5992645293207921);
    ps = fma(ps, f2, 2.5501640398773455);
    ps = fma(ps, f2, -5.16771278004997);
    ps = fma(ps, f2, 3.141592653589793);
    ps *= f;
    double pc = 4.303069587032947e-06;
    pc = fma(pc, f2, -0.0001046381049248457);
    pc = fma(pc, f2, 0.0019295743094039231);
    pc = fma(pc, f2, -0.02580689139001406);
    pc = fma(pc, f2, 0.2353306303588932);
    pc = fma(pc, f2, -1.3352627688545895);
    pc = fma(pc, f2, 4.0587121264167685);
    pc = fma(pc, f2, -4.934802200544679);
    pc = fma(pc, f2, 1.0);
    const bool swap = (iq & 1) != 0;
    double ss = swap ? pc : ps;
    double cc = swap ? ps : pc;
    // quadrant signs: iq = 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s); 4 == 0
    const long long sflip = ((long long)(iq & 2)) << 62;
    const long long cflip = ((long long)((iq + 1) & 2)) << 62;
    s = __longlong_as_double(__double_as_longlong(ss) ^ sflip);
    c = __longlong_as_double(__double_as_longlong(cc) ^ cflip);
}

__device__ __forceinline__ void pf_boxmuller4_fast(const uint32_t (&x)[4], const double2 *tab, double (&z)[4]) {
    const double S = 2.3283064365386962890625e-10;  // 2^-32
    const double u0 = ((double)x[0] + 0.5) * S, u1 = ((double)x[1] + 0.5) * S;
    const double u2 = ((double)x[2] + 0.5) * S, u3 = ((double)x[3] + 0.5) * S;
    const double r0 = pf_sqrt_fast(pf_neg2log_fast(u0, tab)), r1 = pf_sqrt_fast(pf_neg2log_fast(u2, tab));
    double s, c;
    pf_sincos2pi_fast(u1, s, c); z[0] = r0 * c; z[1] = r0 * s;
    pf_sincos2pi_fast(u3, s, c); z[2] = r1 * c; z[3] = r1 * s;
}

// same stream as pf_randn4 (pfmi_common.h), fast transcendental path
__device__ __forceinline__ void pf_randn4_fast(uint64_t seed, uint32_t g, uint32_t n, uint32_t stream,
                                               const double2 *tab, double (&z)[4]) {
    uint32_t x[4];
    pf_philox4x32_10(n, g, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    pf_boxmuller4_fast(x, tab, z);
}

// ---- the same Box-Muller pair, cut into three phases of ~20 fp64 VALU instructions each so that a kernel can
//      interleave them with MFMA issue (elbo_mfma_kernel.hip).  pf_bm_a/b/c(x_radius, x_angle) == one pair of
//      pf_boxmuller4_fast, bit for bit.
struct PfBM {
    double v;        // -2 ln u_radius
    double rad;      // sqrt(v)
    double f, f2;    // half-turn fraction of the angle and its square
    double ps;       // partial sine polynomial
    int iq;          // quadrant
};
__device__ __forceinline__ void pf_bm_a(uint32_t xr, uint32_t xa, const double2 *tab, PfBM &st) {
    const double S = 2.3283064365386962890625e-10;  // 2^-32
    const double ur = ((double)xr + 0.5) * S, ua = ((double)xa + 0.5) * S;
    st.v = pf_neg2log_fast(ur, tab);
    const double t4 = 4.0 * ua;
    const double q = rint(t4);
    st.f = 0.5 * (t4 - q);
    st.iq = (int)q;
}
__device__ __forceinline__ void pf_bm_b(PfBM &st) {
    st.rad = pf_sqrt_fast(st.v);
    const double f2 = st.f * st.f;
    st.f2 = f2;
    double ps = -2.1915353447830217e-05;
    ps = fma(ps, f2, 0.00046630280576761255);
    ps = fma(ps, f2, -0.0073704309457143504);
    ps = fma(ps, f2, 0.08214588661112823);
    st.ps = ps;
}
__device__ __forceinline__ void pf_bm_c(const PfBM &st, double &z0, double &z1) {
    const double f2 = st.f2;
    double ps = st.ps;
    ps = fma(ps, f2, -0.5992645293207921);
    ps = fma(ps, f2, 2.5501640398773455);
    ps = fma(ps, f2, -5.16771278004997);
    ps = fma(ps, f2, 3.141592653589793);
    ps *= st.f;
    double pc = 4.303069587032947e-06;
    pc = fma(pc, f2, -0.0001046381049248457);
    pc = fma(pc, f2, 0.0019295743094039231);
    pc = fma(pc, f2, -0.02580689139001406);
    pc = fma(pc, f2, 0.2353306303588932);
    pc = fma(pc, f2, -1.3352627688545895);
    pc = fma(pc, f2, 4.0587121264167685);
    pc = fma(pc, f2, -4.934802200544679);
    pc = fma(pc, f2, 1.0);
    const bool swap = (st.iq & 1) != 0;
    const double ss = swap ? pc : ps;
    const double cc = swap ? ps : pc;
    const long long sflip = ((long long)(st.iq & 2)) << 62;
    const long long cflip = ((long long)((st.iq + 1) & 2)) << 62;
    const double s = __longlong_as_double(__double_as_longlong(ss) ^ sflip);
    const double c = __longlong_as_double(__double_as_longlong(cc) ^ cflip);
    z0 = st.rad * c;
    z1 = st.rad * s;
}
// two Philox4x32 rounds (the generator is split 5 x 2 rounds by the pipelined kernel)
__device__ __forceinline__ void pf_philox_2rounds(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                                  uint32_t &k0, uint32_t &k1) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// ---- eight-piece version of one Box-Muller pair (same operations, same order => same bits) for kernels that
//      interleave several independent chains (two pairs + the next Philox call) inside one scheduling region.
struct PfPair {
    double m, t4, r, ty, f, f2, p, v, g, h, rad, ps, pc;
    int e, idx, iq;
    __device__ __forceinline__ void s0(uint32_t xr, uint32_t xa) {
        const double S = 2.3283064365386962890625e-10;  // 2^-32
        const double ur = ((double)xr + 0.5) * S, ua = ((double)xa + 0.5) * S;
        const long long bits = __double_as_longlong(ur);
        e = (int)(bits >> 52) - 1023;
        idx = (int)(bits >> 45) & 127;
        m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
        t4 = 4.0 * ua;
    }
    __device__ __forceinline__ void s1(const double2 *tab) {
        const double2 t = tab[idx];
        ty = t.y;
        r = fma(m, t.x, -1.0);
        const double qq = rint(t4);
        f = 0.5 * (t4 - qq);
        iq = (int)qq;
    }
    __device__ __forceinline__ void s2() {
        double pp = fma(r, -1.0 / 7.0, 1.0 / 6.0);
        pp = fma(r, pp, -1.0 / 5.0);
        pp = fma(r, pp, 1.0 / 4.0);
        pp = fma(r, pp, -1.0 / 3.0);
        pp = fma(r, pp, 0.5);
        p = fma(r, pp, -1.0);
    }
    __device__ __forceinline__ void s3() {
        const double l = fma((double)e, 0.6931471805599453094, ty);
        v = fma(2.0 * r, p, -2.0 * l);
        const double y = __builtin_amdgcn_rsq(v);
        g = v * y;
        h = 0.5 * y;
        f2 = f * f;
    }
    __device__ __forceinline__ void s4() {
        const double rr = fma(-h, g, 0.5);
        g = fma(g, rr, g);
        h = fma(h, rr, h);
        const double dd = fma(-g, g, v);
        rad = fma(dd, h, g);
        double q = -2.1915353447830217e-05;
        q = fma(q, f2, 0.00046630280576761255);
        q = fma(q, f2, -0.0073704309457143504);
        ps = fma(q, f2, 0.08214588661112823);
    }
    __device__ __forceinline__ void s5() {
        double q = fma(ps, f2, -0.5992645293207921);
        q = fma(q, f2, 2.5501640398773455);
        q = fma(q, f2, -5.16771278004997);
        q = fma(q, f2, 3.141592653589793);
        ps = q * f;
        double w = 4.303069587032947e-06;
        w = fma(w, f2, -0.0001046381049248457);
        w = fma(w, f2, 0.0019295743094039231);
        pc = fma(w, f2, -0.02580689139001406);
    }
    __device__ __forceinline__ void s6() {
        double w = fma(pc, f2, 0.2353306303588932);
        w = fma(w, f2, -1.3352627688545895);
        w = fma(w, f2, 4.0587121264167685);
        w = fma(w, f2, -4.934802200544679);
        pc = fma(w, f2, 1.0);
    }
    __device__ __forceinline__ void s7(double &z0, double &z1) const {
        const bool swap = (iq & 1) != 0;
        const double ss = swap ? pc : ps;
        const double cc = swap ? ps : pc;
        const long long sflip = ((long long)(iq & 2)) << 62;
        const long long cflip = ((long long)((iq + 1) & 2)) << 62;
        const double sn = __longlong_as_double(__double_as_longlong(ss) ^ sflip);
        const double cs = __longlong_as_double(__double_as_longlong(cc) ^ cflip);
        z0 = rad * cs;
        z1 = rad * sn;
    }
};
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
#endif
