// sincos_exact.h -- float64 sin / cos that are BIT-IDENTICAL to the libm the reference runs on.
//
// Why: the classic-control dynamics (envs/classic_control/cartpole.py:180-181, pendulum.py:137, acrobot.py:281-283,
// continuous_mountain_car.py:163, mountain_car.py:144) call np.sin / np.cos on float64, which NumPy 2.x forwards to the C library:
// on x86-64 Linux with glibc 2.35 that is `__sin_fma` / `__cos_fma` (sysdeps/ieee754/dbl-64/s_sin.c compiled with -mfma, chosen by
// the IFUNC resolver on every CPU with FMA + AVX2 -- the build container and the GPU box alike).  Those routines are accurate to
// 0.55 ulp but NOT correctly rounded, so any other implementation (ocml included) differs in the last bit on ~1 argument in 10, and
// chaotic systems (Acrobot) amplify that to 1e-5 within ~360 steps.  Restating the SAME algorithm -- same range split, same
// 1/128-spaced double-double table (sincos_table.h), same polynomials, and the same fused multiply-adds the compiler contracted (read off
// the libm binary: each `fma` below is one vfmadd / vfnmadd there, each `*` `+` a separately rounded operation) -- makes the device results
// equal bit for bit, and with them whole classic-control trajectories (tests/test_sincos_exact.py: millions of arguments against the
// running libm on the CPU; tests/test_gpu_parity.py: array_equal against the oracle).
//
// Range: |x| < 105414350 (0x419921FB): beyond it glibc switches to its 768-bit Payne-Hanek reduction (__branred), which no environment
// of the path reaches (angles are wrapped or bounded by velocity limits x episode length); there this header falls back to ocml.
// Compile with -ffp-contract=off: every contraction is spelled out.
#pragma once
#include <math.h>
#include <stdint.h>

#include "sincos_table.h"

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define MI_SC_DEV __device__ __forceinline__
#define MI_SC_MEMBER __device__ __forceinline__
#define MI_SC_TABLE __device__
#else
#define MI_SC_DEV static inline
#define MI_SC_MEMBER inline
#define MI_SC_TABLE static
#endif

namespace mi_sincos {

MI_SC_TABLE const double kTable[440] = {MI_SINCOS_TABLE_VALUES};

// constants of s_sin.c / usncs.h (values as found in libm 2.35's .rodata)
constexpr double kBig = 0x1.8000000000000p+45;      // 1.5 * 2^45: adding it rounds |x| to a multiple of 1/128 whose index is the low word
constexpr double kSn3 = -0x1.5555555555515p-3, kSn5 = 0x1.11110e829872fp-7;
constexpr double kCs2 = 0x1.0000000000000p-1, kCs4 = -0x1.5555555555535p-5, kCs6 = 0x1.6c16bedd9e239p-10;
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ecep-7, kS3 = -0x1.a01a019db08b8p-13, kS4 = 0x1.71de27b9a7ed9p-19,
                 kS5 = -0x1.addffc2fcdf59p-26;
constexpr double kHp0 = 0x1.921fb54442d18p+0, kHp1 = 0x1.1a62633145c07p-54;  // pi/2 as a double-double
constexpr double kHpInv = 0x1.45f306dc9c883p-1, kToInt = 0x1.8000000000000p+52;
constexpr double kMp1 = 0x1.921fb58000000p+0, kMp2 = -0x1.dde973c000000p-27, kPp3 = -0x1.cb3b398000000p-55, kPp4 = -0x1.d747f23e32ed7p-83;
// HOT: the range reduction's constants read from eight doubles behind the 6-wide table (T6[kHotAt ..], fill_hot()) instead of written as literals.
// A float64 literal costs the instruction that uses it two scalar moves unless an SGPR pair holds it, and a kernel at the 106-SGPR limit (Acrobot:
// 14 reductions per step) re-materialises these on every use; a value that comes from memory the compiler keeps in vector registers across the loop.
constexpr int kHotAt = 660, kHotCount = 18;
MI_SC_DEV void fill_hot(double *T6) {
    double *h = T6 + kHotAt;
    h[0] = kHpInv, h[1] = kToInt, h[2] = kMp1, h[3] = kMp2, h[4] = kPp3, h[5] = kPp4, h[6] = kHp0, h[7] = kHp1;
    h[8] = kBig, h[9] = kSn3, h[10] = kSn5, h[11] = kCs4, h[12] = kCs6, h[13] = kS1, h[14] = kS2, h[15] = kS3, h[16] = kS4, h[17] = kS5;
}
template <bool HOT>
struct RedK {
    const double *h;
    MI_SC_MEMBER explicit RedK(const double *T6) : h(T6 + kHotAt) {}
    MI_SC_MEMBER double hp_inv() const { return HOT ? h[0] : kHpInv; }
    MI_SC_MEMBER double to_int() const { return HOT ? h[1] : kToInt; }
    MI_SC_MEMBER double mp1() const { return HOT ? h[2] : kMp1; }
    MI_SC_MEMBER double mp2() const { return HOT ? h[3] : kMp2; }
    MI_SC_MEMBER double pp3() const { return HOT ? h[4] : kPp3; }
    MI_SC_MEMBER double pp4() const { return HOT ? h[5] : kPp4; }
    MI_SC_MEMBER double hp0() const { return HOT ? h[6] : kHp0; }
    MI_SC_MEMBER double hp1() const { return HOT ? h[7] : kHp1; }
};
// ... and the constants of do_sin / do_cos / TAYLOR_SIN (POLY: a kernel chooses how many of its registers go to constants)
template <bool POLY>
struct PolyK {
    const double *h;
    MI_SC_MEMBER explicit PolyK(const double *T6) : h(T6 + kHotAt) {}
    MI_SC_MEMBER double big() const { return POLY ? h[8] : kBig; }
    MI_SC_MEMBER double sn3() const { return POLY ? h[9] : kSn3; }
    MI_SC_MEMBER double sn5() const { return POLY ? h[10] : kSn5; }
    MI_SC_MEMBER double cs4() const { return POLY ? h[11] : kCs4; }
    MI_SC_MEMBER double cs6() const { return POLY ? h[12] : kCs6; }
    MI_SC_MEMBER double s1() const { return POLY ? h[13] : kS1; }
    MI_SC_MEMBER double s2() const { return POLY ? h[14] : kS2; }
    MI_SC_MEMBER double s3() const { return POLY ? h[15] : kS3; }
    MI_SC_MEMBER double s4() const { return POLY ? h[16] : kS4; }
    MI_SC_MEMBER double s5() const { return POLY ? h[17] : kS5; }
};
#ifndef MI_SC_HOT_POLY
#define MI_SC_HOT_POLY 1
#endif

MI_SC_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// the same fused multiply-add for a Horner step whose multiplier AND addend are constants: on the device one v_fma_f64 with all three
// operands in VGPRs.  Left to itself the compiler copies the constant addend into a fresh register pair and accumulates into the copy
// (v_mov_b64 + v_fmac_f64, two issue slots in loops that are issue-bound); this way the coefficients stay in registers across a rollout loop.
// KASM = false: the plain builtin.  Inline asm is opaque to the compiler's hazard recognizer (GCNHazardRecognizer only protects the
// instructions it emitted itself); round 3 met a kernel -- Acrobot's fused rollout, the only classic kernel whose live values overflow into
// AGPRs (v_accvgpr_read / write around the asm) -- whose results with the asm form were NOT reproducible from launch to launch (1-ulp
// float32 flips in 15 % of the lanes, one hang) and with the builtin were.  So the asm form is used only by kernels that stay inside the
// VGPR file (tests/test_kernel_resources.py pins that), Acrobot takes the builtin.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_NO_FMA_K)
template <bool KASM = true>
MI_SC_DEV double fma_k(double a, double b, double c) {
    if constexpr (KASM) {
        double r;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    } else {
        return __builtin_fma(a, b, c);
    }
}
#else
template <bool KASM = true>
MI_SC_DEV double fma_k(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
MI_SC_DEV uint64_t bits(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    return v.u;
}
MI_SC_DEV double from_bits(uint64_t u) {
    union { double d; uint64_t u; } v;
    v.u = u;
    return v.d;
}
MI_SC_DEV double copysign_(double mag, double sgn) { return from_bits((bits(mag) & 0x7fffffffffffffffull) | (bits(sgn) & 0x8000000000000000ull)); }

// TAYLOR_SIN(xx, x, dx): x - x^3/3! + ... + x^9/9! with the correction of the low part dx
MI_SC_DEV double taylor_sin(double x, double dx) {
    const double xx = x * x;
    double p = fma_(xx, kS5, kS4);
    p = fma_(xx, p, kS3), p = fma_(xx, p, kS2), p = fma_(xx, p, kS1);
    const double t1 = fma_(x, p, -(0.5 * dx));
    const double t = fma_(t1, xx, dx);
    return x + t;
}
// do_sin: sin(x + dx) for |x| < 0.855469, x + dx a double-double
MI_SC_DEV double do_sin(const double *T, double x, double dx) {
    const double ax = fabs(x);
    if (ax < 0.126) return taylor_sin(x, dx);
    if (x <= 0) dx = -dx;
    const double u = kBig + ax;
    const double xr = ax - (u - kBig);
    const int k = (int)(uint32_t)bits(u) * 4;
    const double sn = T[k], ssn = T[k + 1], cs = T[k + 2], ccs = T[k + 3];
    const double xx = xr * xr;
    const double q = fma_(xx, kSn5, kSn3);
    const double si = fma_(xr * xx, q, dx);
    const double s = xr + si;
    double c0 = fma_(xx, kCs6, kCs4);
    c0 = fma_(xx, c0, kCs2);
    const double c = fma_(xr, dx, xx * c0);
    double cor = fma_(s, ccs, ssn);
    cor = fma_(-c, sn, cor);
    cor = fma_(s, cs, cor);
    return copysign_(sn + cor, x);
}
// do_cos: cos(x + dx) for |x| < 0.855469
MI_SC_DEV double do_cos(const double *T, double x, double dx) {
    if (x < 0) dx = -dx;
    const double ax = fabs(x);
    const double u = kBig + ax;
    const double xr = (ax - (u - kBig)) + dx;
    const int k = (int)(uint32_t)bits(u) * 4;
    const double sn = T[k], ssn = T[k + 1], cs = T[k + 2], ccs = T[k + 3];
    const double xx = xr * xr;
    const double q = fma_(xx, kSn5, kSn3);
    const double s = fma_(xr * xx, q, xr);
    double c0 = fma_(xx, kCs6, kCs4);
    c0 = fma_(xx, c0, kCs2);
    const double c = xx * c0;
    double cor = fma_(-s, ssn, ccs);
    cor = fma_(-c, cs, cor);
    cor = fma_(-s, sn, cor);
    return cs + cor;
}
// reduce_sincos: x = n pi/2 + (a + da), |a| <= pi/4, for 2.426 <= |x| < 105414350; returns n mod 4
MI_SC_DEV int reduce(double x, double &a, double &da) {
    const double t = fma_(x, kHpInv, kToInt);
    const double xn = t - kToInt;
    const int n = (int)((uint32_t)bits(t) & 3u);
    double y = fma_(-xn, kMp1, x);
    y = fma_(-xn, kMp2, y);
    const double t2 = fma_(-xn, kPp3, y);
    const double db = fma_(-kPp3, xn, y - t2);
    const double b = fma_(-xn, kPp4, t2);
    const double db2 = fma_(-xn, kPp4, t2 - b);
    a = b, da = db + db2;
    return n;
}
MI_SC_DEV double do_sincos(const double *T, double a, double da, int n) {
    const double r = (n & 1) ? do_cos(T, a, da) : do_sin(T, a, da);
    return (n & 2) ? -r : r;
}

// T: the table (kTable, or a copy of it in faster memory: the kernels of engine.hip keep one in LDS)
MI_SC_DEV double sin_exact(const double *T, double x) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                                           // |x| < 2^-26
    if (k < 0x3feb6000u) return do_sin(T, x, 0.0);                           // |x| < 0.855469
    if (k < 0x400368fdu) return copysign_(do_cos(T, kHp0 - fabs(x), kHp1), x);  // |x| < 2.426265
    if (k < 0x419921fbu) {                                                   // |x| < 105414350
        double a, da;
        const int n = reduce(x, a, da);
        return do_sincos(T, a, da, n);
    }
    return sin(x);
}
MI_SC_DEV double cos_exact(const double *T, double x) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;                                         // |x| < 2^-27
    if (k < 0x3feb6000u) return do_cos(T, x, 0.0);
    if (k < 0x400368fdu) {
        const double y = kHp0 - fabs(x);
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        return do_sin(T, a, da);
    }
    if (k < 0x419921fbu) {
        double a, da;
        const int n = reduce(x, a, da);
        return do_sincos(T, a, da, n + 1);
    }
    return cos(x);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same functions without data-dependent branches, for 64 lanes whose arguments fall in different ranges (a wavefront of Acrobots has
// lanes in all three of the ranges above on every call, and a divergent branch costs the SUM of its sides).  Two halves:
//   prep(): x -> a reduced argument a + da, |a| < 0.8555, whether the result is do_cos or do_sin of it, and whether it is negated; the
//           three candidate reductions are computed side by side and selected.
//   core(): do_sin / do_cos / TAYLOR_SIN of (a, da) in ONE instruction stream: the two table routines differ only in which table pair
//           plays which role and in where dx enters, so the table is read through a 6-wide layout {sn, ssn, cs, ccs, sn, ssn} at offset
//           0 (sin) or 2 (cos) and the few differing operands are selected.
// Every value is produced by the same operation on the same operands as in the branchy routines above (which stay as the readable
// statement of the algorithm and as the cross-check of tests/test_sincos_exact.py).
// T6: the 6-wide table (110 entries x 6 doubles; expand6() builds it from kTable).
MI_SC_DEV void expand6(const double *T4, double *T6, int entry) {
    const double sn = T4[4 * entry], ssn = T4[4 * entry + 1], cs = T4[4 * entry + 2], ccs = T4[4 * entry + 3];
    double *o = T6 + 6 * entry;
    o[0] = sn, o[1] = ssn, o[2] = cs, o[3] = ccs, o[4] = sn, o[5] = ssn;
}
MI_SC_DEV double sel(bool c, double a, double b) { return c ? a : b; }

template <bool KASM = true, bool HOT = false>
MI_SC_DEV double core(const double *T6, double a, double da, bool cm) {
    const double ax = fabs(a);
    const PolyK<HOT && MI_SC_HOT_POLY> P(T6);
    const double u = P.big() + ax;
    const double x0 = ax - (u - P.big());
    const int idx = (int)(uint32_t)bits(u) * 6 + (cm ? 2 : 0);
    const double p = T6[idx], pp = T6[idx + 1], q = T6[idx + 2], qq = T6[idx + 3];
    const bool flip = cm ? (a < 0) : (a <= 0);  // do_cos: if (x < 0) dx = -dx;  do_sin: if (x <= 0) dx = -dx
    const double dxs = flip ? -da : da;
    const double xr = cm ? x0 + dxs : x0;
    const double xx = xr * xr;
    const double poly_s = fma_k<KASM>(xx, P.sn5(), P.sn3());
    const double tt = fma_(xr * xx, poly_s, sel(cm, xr, dxs));
    const double s = cm ? -tt : xr + tt;  // cos: -s with s = fma(xr^3, q, xr);  sin: s = xr + fma(xr^3, q, dx)
    double c0 = fma_k<KASM>(xx, P.cs6(), P.cs4());
    c0 = fma_k<KASM>(xx, c0, kCs2);
    const double xc = xx * c0;
    const double c = cm ? xc : fma_(xr, dxs, xc);
    double cor = fma_(s, qq, pp);
    cor = fma_(-c, p, cor);
    cor = fma_(s, q, cor);
    double res = p + cor;
    // TAYLOR_SIN for the sin of |a| < 0.126 (signed a, dx as given)
    const double axx = a * a;
    double tp = fma_k<KASM>(axx, P.s5(), P.s4());
    tp = fma_k<KASM>(axx, tp, P.s3()), tp = fma_k<KASM>(axx, tp, P.s2()), tp = fma_k<KASM>(axx, tp, P.s1());
    const double t1 = fma_(a, tp, -(0.5 * da));
    const double rt = a + fma_(t1, axx, da);
    res = (!cm && ax < 0.126) ? rt : res;
    return cm ? res : copysign_(res, a);  // (the Taylor value has a's sign already, except that -0 + 0 = +0)
}

// reduced argument of sin(x) (want_cos = false) or cos(x) (true) for |x| < 105414350
template <bool KASM = true, bool HOT = false>
MI_SC_DEV void prep(const double *T6, double x, bool want_cos, double &a, double &da, bool &cm, bool &neg) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    const double ax = fabs(x);
    const RedK<HOT> K(T6);
    // |x| >= 2.426265: n pi/2 + (b + db)
    const double t = fma_k<KASM>(x, K.hp_inv(), K.to_int());
    const double xn = t - K.to_int();
    const uint32_t n = ((uint32_t)bits(t) + (want_cos ? 1u : 0u)) & 3u;
    double y = fma_(-xn, K.mp1(), x);
    y = fma_(-xn, K.mp2(), y);
    const double t2 = fma_(-xn, K.pp3(), y);
    const double db = fma_(-K.pp3(), xn, y - t2);
    const double b = fma_(-xn, K.pp4(), t2);
    const double db2 = fma_(-xn, K.pp4(), t2 - b);
    // 0.855469 <= |x| < 2.426265: sin = copysign(do_cos(pi/2 - |x|, lo), x);  cos = do_sin(two-sum of the same)
    const double ym = K.hp0() - ax;
    const double am = ym + K.hp1();
    const double dam = (ym - am) + K.hp1();
    const bool main = k < 0x3feb6000u, mid = k < 0x400368fdu;
    a = main ? x : (mid ? (want_cos ? am : ym) : b);
    da = main ? 0.0 : (mid ? (want_cos ? dam : K.hp1()) : db + db2);
    cm = main ? want_cos : (mid ? !want_cos : (n & 1u) != 0);
    neg = main ? false : (mid ? (!want_cos && x < 0) : (n & 2u) != 0);
}

// BOUNDED: the caller guarantees |x| < 105414336 (an angle that the environment wraps or clips), so the hand-over to the platform's sin / cos
// for huge arguments -- a test, a branch and a page of never-executed code per call site -- is left out.
template <bool BOUNDED = false, bool KASM = true, bool HOT = false>
MI_SC_DEV double sin_bf(const double *T6, double x) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (!BOUNDED && __builtin_expect(k >= 0x419921fbu, 0)) return sin(x);
    double a, da;
    bool cm, neg;
    prep<KASM, HOT>(T6, x, false, a, da, cm, neg);
    const double r = core<KASM, HOT>(T6, a, da, cm);
    return neg ? -r : r;
}
template <bool BOUNDED = false, bool KASM = true, bool HOT = false>
MI_SC_DEV double cos_bf(const double *T6, double x) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (!BOUNDED && __builtin_expect(k >= 0x419921fbu, 0)) return cos(x);
    double a, da;
    bool cm, neg;
    prep<KASM, HOT>(T6, x, true, a, da, cm, neg);
    const double r = core<KASM, HOT>(T6, a, da, cm);
    return neg ? -r : r;
}

// sin and cos of the same |x| < 0.855469 (CartPole's pole angle: the episode ends at 0.2095): one index, one table read, shared polynomials.
// With dx = 0 the additions of +-0 in do_sin / do_cos drop out (they change at most the sign of a zero that is then added to a non-zero).
template <bool KASM = true>
MI_SC_DEV void sincos_main(const double *T6, double x, double &sn_out, double &cs_out) {
    const double ax = fabs(x);
    const double u = kBig + ax;
    const double xr = ax - (u - kBig);
    const int idx = (int)(uint32_t)bits(u) * 6;
    const double sn = T6[idx], ssn = T6[idx + 1], cs = T6[idx + 2], ccs = T6[idx + 3];
    const double xx = xr * xr;
    const double poly_s = fma_k<KASM>(xx, kSn5, kSn3);
    const double x3 = xr * xx;
    const double s_sin = xr + x3 * poly_s;
    const double s_cos = fma_(x3, poly_s, xr);
    double c0 = fma_k<KASM>(xx, kCs6, kCs4);
    c0 = fma_k<KASM>(xx, c0, kCs2);
    const double c = xx * c0;
    double cor = fma_(s_sin, ccs, ssn);
    cor = fma_(-c, sn, cor);
    cor = fma_(s_sin, cs, cor);
    double rs = sn + cor;
    double cc = fma_(-s_cos, ssn, ccs);
    cc = fma_(-c, cs, cc);
    cc = fma_(-s_cos, sn, cc);
    cs_out = cs + cc;
    const double axx = x * x;
    double tp = fma_k<KASM>(axx, kS5, kS4);
    tp = fma_k<KASM>(axx, tp, kS3), tp = fma_k<KASM>(axx, tp, kS2), tp = fma_k<KASM>(axx, tp, kS1);
    const double rt = x + (x * tp) * axx;
    rs = ax < 0.126 ? rt : rs;
    sn_out = copysign_(rs, x);
}
// sin AND cos of one argument in any range (|x| < 105414350), branch-free, for lanes spread over all ranges (Pendulum's and Acrobot's angles).
// Whatever the range, glibc evaluates exactly ONE do_sin and ONE do_cos between the two results:
//   |x| < 0.855469:  sin = do_sin(x, 0),                         cos = do_cos(x, 0)
//   |x| < 2.426265:  sin = +-do_cos(pi/2 - |x|, lo),             cos = do_sin(two-sum of the same)
//   beyond:          x = n pi/2 + (b + db):  n even: sin = +-do_sin(b, db), cos = +-do_cos(b, db);  n odd: the two swap roles
// so the pair costs one shared reduction, one dedicated do_sin stream (TAYLOR_SIN selected in) and one dedicated do_cos stream -- none of the
// role selects the one-function core() above needs -- and two output selects.  Same operations on the same operands as sin_bf / cos_bf.
template <bool KASM = true, bool HOT = false>
MI_SC_DEV double do_sin_bf(const double *T6, double a, double da) {
    const double ax = fabs(a);
    const PolyK<HOT && MI_SC_HOT_POLY> P(T6);
    const double u = P.big() + ax;
    const double x0 = ax - (u - P.big());
    const int idx = (int)(uint32_t)bits(u) * 6;
    const double sn = T6[idx], ssn = T6[idx + 1], cs = T6[idx + 2], ccs = T6[idx + 3];
    const double dxs = (a <= 0) ? -da : da;
    const double xx = x0 * x0;
    const double q = fma_k<KASM>(xx, P.sn5(), P.sn3());
    const double si = fma_(x0 * xx, q, dxs);
    const double sv = x0 + si;
    double c0 = fma_k<KASM>(xx, P.cs6(), P.cs4());
    c0 = fma_k<KASM>(xx, c0, kCs2);
    const double c = fma_(x0, dxs, xx * c0);
    double cor = fma_(sv, ccs, ssn);
    cor = fma_(-c, sn, cor);
    cor = fma_(sv, cs, cor);
    double res = sn + cor;
    const double axx = a * a;
    double tp = fma_k<KASM>(axx, P.s5(), P.s4());
    tp = fma_k<KASM>(axx, tp, P.s3()), tp = fma_k<KASM>(axx, tp, P.s2()), tp = fma_k<KASM>(axx, tp, P.s1());
    const double t1 = fma_(a, tp, -(0.5 * da));
    const double rt = a + fma_(t1, axx, da);
    res = ax < 0.126 ? rt : res;
    return copysign_(res, a);
}
template <bool KASM = true, bool HOT = false>
MI_SC_DEV double do_cos_bf(const double *T6, double a, double da) {
    const double ax = fabs(a);
    const PolyK<HOT && MI_SC_HOT_POLY> P(T6);
    const double u = P.big() + ax;
    const double dxs = (a < 0) ? -da : da;
    const double xr = (ax - (u - P.big())) + dxs;
    const int idx = (int)(uint32_t)bits(u) * 6;
    const double sn = T6[idx], ssn = T6[idx + 1], cs = T6[idx + 2], ccs = T6[idx + 3];
    const double xx = xr * xr;
    const double q = fma_k<KASM>(xx, P.sn5(), P.sn3());
    const double sv = fma_(xr * xx, q, xr);
    double c0 = fma_k<KASM>(xx, P.cs6(), P.cs4());
    c0 = fma_k<KASM>(xx, c0, kCs2);
    const double c = xx * c0;
    double cor = fma_(-sv, ssn, ccs);
    cor = fma_(-c, cs, cor);
    cor = fma_(-sv, sn, cor);
    return cs + cor;
}
template <bool BOUNDED = false, bool KASM = true, bool HOT = false>
MI_SC_DEV void sincos_pair(const double *T6, double x, double &sn_out, double &cs_out) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (!BOUNDED && __builtin_expect(k >= 0x419921fbu, 0)) {
        sn_out = sin(x), cs_out = cos(x);
        return;
    }
    const double ax = fabs(x);
    const RedK<HOT> K(T6);
    // |x| >= 2.426265: n pi/2 + (b + db)
    const double t = fma_k<KASM>(x, K.hp_inv(), K.to_int());
    const double xn = t - K.to_int();
    const uint32_t n = (uint32_t)bits(t);
    double y = fma_(-xn, K.mp1(), x);
    y = fma_(-xn, K.mp2(), y);
    const double t2 = fma_(-xn, K.pp3(), y);
    const double db = fma_(-K.pp3(), xn, y - t2);
    const double b = fma_(-xn, K.pp4(), t2);
    const double db2 = fma_(-xn, K.pp4(), t2 - b);
    // 0.855469 <= |x| < 2.426265
    const double ym = K.hp0() - ax;
    const double am = ym + K.hp1();
    const double dam = (ym - am) + K.hp1();
    const bool main = k < 0x3feb6000u, mid = k < 0x400368fdu;
    const double as = main ? x : (mid ? am : b), das = main ? 0.0 : (mid ? dam : db + db2);
    const double ac = main ? x : (mid ? ym : b), dac = main ? 0.0 : (mid ? K.hp1() : db + db2);
    const double S = do_sin_bf<KASM, HOT>(T6, as, das), C = do_cos_bf<KASM, HOT>(T6, ac, dac);
    const bool swap = main ? false : (mid ? true : (n & 1u) != 0);
    const bool neg_s = main ? false : (mid ? (x < 0) : (n & 2u) != 0);
    const bool neg_c = (main || mid) ? false : ((n + 1u) & 2u) != 0;
    const double sv = swap ? C : S, cv = swap ? S : C;
    sn_out = neg_s ? -sv : sv, cs_out = neg_c ? -cv : cv;
}

// fmod(x, Y) for a positive compile-time modulus Y and |x| < 2^52 Y, exact like the C function (the remainder of a division is always
// representable) and branch-free: q = trunc(x (1 / Y)) is right or off by one (two roundings), the residual x - q Y is ONE fused
// multiply-add -- exact whenever the right q is used, because then |x - q Y| < Y -- and a wrong q shows as a residual outside [0, Y)
// (for x >= 0; mirrored for x < 0), which is redone with the neighbouring q.  ~25 instructions against ~56 of the general library routine
// (Pendulum's angle_normalize, pendulum.py:281-282).  Checked against the C library on millions of arguments by tests/test_sincos_exact.py.
template <class Y>
MI_SC_DEV double fmod_const(double x, Y) {
    constexpr double y = Y::value;
    const double ax = fabs(x);
#ifdef MI_FMOD_BY_DIVISION  // (A/B builds: scripts/build_variant.py ... -DMI_FMOD_BY_DIVISION)
    double q = trunc(ax / y);
#else
    constexpr double inv_y = 1.0 / y;
    double q = trunc(ax * inv_y);  // (two roundings of 2^-53 each: the product is within 1 of x / Y for |x| < 2^52 Y, so its integer part is right or off by one, like the quotient's -- and 1 instruction instead of 11)
#endif
    double r = fma_(-q, y, ax);
    const double qlo = q - 1.0, qhi = q + 1.0;
    const double rlo = fma_(-qlo, y, ax), rhi = fma_(-qhi, y, ax);
    r = (r < 0.0) ? rlo : ((r >= y) ? rhi : r);
    return copysign_(r, x);  // fmod carries the sign of x, also for a zero remainder
}

// MAIN_FIRST: the arguments of all lanes are expected inside |x| < 0.855469 (CartPole), worth a wavefront-uniform test for the short routine
template <bool BOUNDED = false, bool MAIN_FIRST = true, bool KASM = true, bool HOT = false>
MI_SC_DEV void sincos_bf(const double *T6, double x, double &sn_out, double &cs_out) {
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (MAIN_FIRST && k < 0x3feb6000u) {
        sincos_main<KASM>(T6, x, sn_out, cs_out);
    } else {
        sincos_pair<BOUNDED, KASM, HOT>(T6, x, sn_out, cs_out);
    }
}

}  // namespace mi_sincos
