// pow_exact.h -- x ** 2 the way the reference computes it: glibc's pow(x, 2.0) / powf(x, 2.0f), bit for bit.
//
// NumPy's scalar `**` forwards to libm (pendulum.py:131,135; acrobot.py:263-275), and glibc's pow is accurate to ~0.52 ulp, not correctly
// rounded: pow(x, 2.0) differs from the correctly rounded x * x for ~0.09 % of doubles (powf from x * x for ~0.07 % of floats).  One ulp
// once per thousand squares is nothing for Pendulum and the end of bit-identity for the chaotic Acrobot, so the squares the reference
// takes through `**` go through this restatement of e_pow.c / e_powf.c (the FMA build glibc selects on every x86-64 CPU with FMA + AVX2:
// each `fma_` below is one vfmadd in that binary, every other operation is rounded separately), specialised to the exponent 2:
//   pow:  log(x) = k ln2 + log(c_i) + log1p(r), r = z / c_i - 1 from a 128-entry table, as hi + lo (about 68 bits); 2 log(x) as ehi + elo;
//         exp by a 128-entry 2^(j/128) table and a degree-5 polynomial.
//   powf: log2 / exp2 in double precision with 16- and 32-entry tables, rounded to float at the end.
// The per-lane table reads go through pointers so that a kernel can keep the tables in LDS (envs_classic.h); the host harness passes the
// arrays below.
// Arguments outside the plain path (0, subnormal, inf, nan, |2 log x| >= 512) fall back to x * x: pow(x, 2.0) is exact there or the
// environments never produce them.  Tables: pow_tables.h (scripts/gen_pow_tables.py).  Checked against the running libm on millions of
// arguments by tests/test_pow_exact.py.  Compile with -ffp-contract=off.
#pragma once
#include <math.h>
#include <stdint.h>

#include "pow_tables.h"

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define MI_PW_DEV __device__ __forceinline__
#define MI_PW_TABLE __device__
#else
#define MI_PW_DEV static inline
#define MI_PW_TABLE static
#endif

namespace mi_pow {

MI_PW_TABLE const double kLogPoly[7] = {MI_POW_LOG_POLY_VALUES};
MI_PW_TABLE const double kLogTab[384] = {MI_POW_LOG_TAB_VALUES};   // {invc, logc, logctail} x 128
MI_PW_TABLE const double kExpPoly[4] = {MI_EXP_POLY_VALUES};       // C2 .. C5
MI_PW_TABLE const uint64_t kExpTab[256] = {MI_EXP_TAB_VALUES};     // {tail, scale bits} x 128
MI_PW_TABLE const double kLog2fTab[32] = {MI_POWF_LOG2_TAB_VALUES};  // {invc, logc} x 16
MI_PW_TABLE const double kLog2fPoly[5] = {MI_POWF_LOG2_POLY_VALUES};
MI_PW_TABLE const uint64_t kExp2fTab[32] = {MI_EXP2F_TAB_VALUES};
MI_PW_TABLE const double kExp2fPoly[3] = {MI_EXP2F_POLY_VALUES};

MI_PW_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a Horner step with constant multiplier and addend as ONE v_fma_f64 with register operands; KASM = false: the builtin (see sincos_exact.h fma_k)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_NO_FMA_K)
template <bool KASM = true>
MI_PW_DEV double fma_k(double a, double b, double c) {
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
MI_PW_DEV double fma_k(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
MI_PW_DEV uint64_t bits(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    return v.u;
}
MI_PW_DEV double from_bits(uint64_t u) {
    union { double d; uint64_t u; } v;
    v.u = u;
    return v.d;
}
MI_PW_DEV uint32_t bitsf(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    return v.u;
}
MI_PW_DEV float from_bitsf(uint32_t u) {
    union { float f; uint32_t u; } v;
    v.u = u;
    return v.f;
}

// pow(x, 2.0).  log_tab: kLogTab (or a copy), exp_tab: kExpTab (or a copy)
template <bool KASM = true>
MI_PW_DEV double square(const double *log_tab, const uint64_t *exp_tab, double x) {
    // Written without data-dependent branches (a wavefront runs alone on its SIMD: every s_cbranch / exec-mask pair is issue slots, and
    // the special cases below practically never occur): the main path is evaluated on whatever bits arrive -- the table indices are masked,
    // nothing traps -- and the rare results are selected in at the end.
    const uint64_t ix = bits(x) & 0x7fffffffffffffffull;  // x < 0 with an even integer exponent: pow(|x|, 2), no sign
    const uint32_t topx = (uint32_t)(ix >> 52);
    const bool special = topx - 1u >= 0x7feu;  // 0, subnormal, inf, nan: x * x
    // log_inline
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = from_bits(ix - (tmp & 0xfff0000000000000ull)), kd = (double)k;
    const double invc = log_tab[3 * i], logc = log_tab[3 * i + 1], logctail = log_tab[3 * i + 2];
    const double r = fma_(z, invc, -1.0);
    const double t1 = fma_(kd, MI_POW_LN2HI, logc);
    const double t2 = t1 + r;
    const double lo1 = fma_(kd, MI_POW_LN2LO, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = kLogPoly[0] * r;  // A[0] = -0.5
    const double ar2 = r * ar, ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma_(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double pa = fma_k<KASM>(r, kLogPoly[2], kLogPoly[1]), pb = fma_k<KASM>(r, kLogPoly[4], kLogPoly[3]), pc = fma_k<KASM>(r, kLogPoly[6], kLogPoly[5]);
    const double p = fma_(ar2, fma_(pc, ar2, pb), pa);
    const double lo = fma_(ar3, p, ((lo1 + lo2) + lo3) + lo4);
    const double lhi = hi + lo;
    const double ltail = (hi - lhi) + lo;
    // y log(x) with y = 2
    // (e_pow.c: ehi = y * lhi, elo = y * ltail + fma(y, lhi, -ehi); with y = 2 the product is exact, the inner fma is +0 and adding it
    //  changes no bit that survives the addition to rr below)
    const double ehi = 2.0 * lhi;
    const double elo = 2.0 * ltail;
    // exp_inline
    const uint32_t abstop = (uint32_t)(bits(ehi) >> 52) & 0x7ffu;
    const bool tiny = abstop < 0x3c9u;                                 // |2 log x| < 2^-54: x is 1 to working precision: 1.0 + ehi
    const bool huge = abstop - 0x3c9u >= 0x3fu && !tiny;               // |2 log x| >= 512: over / underflow range (never reached by the environments): x * x
    const double zz = fma_k<KASM>(ehi, MI_EXP_INVLN2N, MI_EXP_SHIFT);
    const uint64_t ki = bits(zz);
    const double kdd = zz - MI_EXP_SHIFT;
    double rr = fma_(kdd, MI_EXP_NEGLN2HIN, ehi);
    rr = fma_(kdd, MI_EXP_NEGLN2LON, rr);
    rr = elo + rr;
    const int idx = 2 * (int)(ki & 127);
    const uint64_t sbits = exp_tab[idx + 1] + (ki << 45);
    const double tail = from_bits(exp_tab[idx]);
    const double q23 = fma_k<KASM>(rr, kExpPoly[1], kExpPoly[0]);
    const double r2 = rr * rr;
    const double q45 = fma_k<KASM>(rr, kExpPoly[3], kExpPoly[2]);
    const double s1 = fma_(q23, r2, rr + tail);
    const double tmp2 = fma_(q45, r2 * r2, s1);
    const double scale = from_bits(sbits);
    double res = fma_(tmp2, scale, scale);
    res = tiny ? 1.0 + ehi : res;
    return (special || huge) ? x * x : res;
}

// ---- three squares for the price of (little more than) one ------------------------------------------------------------------------------
// x * x is exact as hi + lo (lo = fma(x, x, -hi)), and glibc's pow is within 0.509 ulp + 4e-5 ulp * |y log x| of the true value (the error
// budget in e_pow.c's header: 0.5 from the final rounding, 0.009 from exp_inline's polynomial and table tail, the rest is log_inline's
// relative error 1.3 * 2^-68 scaled by |y log x|).  So whenever the true square is at least 1/64 ulp away from the rounding boundary --
// |lo| <= 31/64 ulp(hi) -- and |2 log x| < 128, pow(x, 2.0) can only be hi, the correctly rounded product; measured on this libm: every
// one of the 0.084 % of arguments with pow(x, 2.0) != x * x has |lo| > (1/2 - 0.0087) ulp (tests/test_pow_exact.py samples that band).
// 31/32 of the arguments pass.  square3() evaluates the test for three arguments and the table routine above ONCE, on the first argument
// of each lane that did not pass (lanes without one recompute their last argument and keep hi), and again only while some lane of the
// wavefront still has one pending (a lane needs a second pass with probability 0.3 %, a wavefront in one group of six).
MI_PW_DEV bool square_is_plain(double x, double &hi) {
    hi = x * x;
    const double lo = fma_(x, x, -hi);
    // the exponent field of hi, taken one binade lower when the top 20 mantissa bits are zero: an exact power of two has the smaller ulp on its
    // lower side.  hi >= 0 or nan; zero, subnormal, inf and nan fail the range test (the table routine selects x * x for them)
    const uint32_t e = ((uint32_t)(bits(hi) >> 32) - 1u) & 0x7ff00000u;
    const double thr = from_bits((uint64_t)(e - 0x03510000u) << 32);  // 2^(E - 54) * 31/16 = 31/64 ulp(hi)
    const bool in_range = e - (843u << 20) < (361u << 20);           // 2^-180 <= hi < 2^181: |2 log x| < 128
    return in_range & (__builtin_fabs(lo) <= thr);
}
// (the loop condition is the lane's own: the compiler turns it into "while any lane is pending" with the others masked off.  A wavefront-uniform
//  condition through a ballot -- a convergent operation -- stops LLVM from unrolling the rollout loop that contains the call.)
#define MI_PW_ANY(p) (p)
template <bool KASM = true>
MI_PW_DEV void square3(const double *log_tab, const uint64_t *exp_tab, double a, double b, double c, double &ra, double &rb, double &rc) {
    bool fa = !square_is_plain(a, ra), fb = !square_is_plain(b, rb), fc = !square_is_plain(c, rc);
#pragma nounroll  // (the trip count is provably <= 3: left alone, the compiler lays out three copies of the table routine)
    do {
        const double r = square<KASM>(log_tab, exp_tab, fa ? a : (fb ? b : c));
        const bool wb = !fa && fb, wc = !fa && !fb && fc;
        ra = fa ? r : ra, rb = wb ? r : rb, rc = wc ? r : rc;
        fb = fb && !wb, fc = fc && !wc, fa = false;
    } while (MI_PW_ANY(fa || fb || fc));
}
// two squares, same scheme
template <bool KASM = true>
MI_PW_DEV void square2(const double *log_tab, const uint64_t *exp_tab, double a, double b, double &ra, double &rb) {
    bool fa = !square_is_plain(a, ra), fb = !square_is_plain(b, rb);
#pragma nounroll
    do {
        const double r = square<KASM>(log_tab, exp_tab, fa ? a : b);
        const bool wb = !fa && fb;
        ra = fa ? r : ra, rb = wb ? r : rb;
        fb = fb && !wb, fa = false;
    } while (MI_PW_ANY(fa || fb));
}

// powf(x, 2.0f).  log2_tab: kLog2fTab (or a copy), exp2_tab: kExp2fTab (or a copy)
template <bool KASM = true>
MI_PW_DEV float squaref(const double *log2_tab, const uint64_t *exp2_tab, float x) {
    const uint32_t ix = bitsf(x) & 0x7fffffffu;
    const bool special = ix - 0x00800000u >= 0x7f800000u - 0x00800000u;  // 0, subnormal, inf, nan: x * x (selected at the end, see square())
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const uint32_t top = tmp & 0xff800000u;
    const double z = (double)from_bitsf(ix - top), kd = (double)((int32_t)top >> 23);
    const double invc = log2_tab[2 * i], logc = log2_tab[2 * i + 1];
    const double r = fma_(z, invc, -1.0);
    const double y0 = logc + kd;
    const double a01 = fma_k<KASM>(r, kLog2fPoly[0], kLog2fPoly[1]), a23 = fma_k<KASM>(r, kLog2fPoly[2], kLog2fPoly[3]);
    const double r2 = r * r;
    double q = fma_(r, kLog2fPoly[4], y0);
    q = fma_(r2, a23, q);
    const double logx = fma_(a01, r2 * r2, q);
    const double ylogx = 2.0 * logx;
    const bool huge = ((bits(ylogx) >> 47) & 0xffffu) > 0x80beu;  // |y log2 x| >= 126: over / underflow range: x * x
    // exp2_inline
    const double kdd0 = ylogx + MI_EXP2F_SHIFT_SCALED;
    const uint64_t ki = bits(kdd0);
    const double kdd = kdd0 - MI_EXP2F_SHIFT_SCALED;
    const double rr = ylogx - kdd;
    const uint64_t t = exp2_tab[ki & 31] + (ki << 47);
    const double zq = fma_k<KASM>(rr, kExp2fPoly[0], kExp2fPoly[1]);
    const double rr2 = rr * rr;
    const double y2 = fma_k<KASM>(rr, kExp2fPoly[2], 1.0);
    const double y3 = fma_(zq, rr2, y2);
    const float res = (float)(y3 * from_bits(t));
    return (special || huge) ? x * x : res;
}

// ... and the float32 square.  powf evaluates exp2(2 log2 x) in double precision (relative error 1.27 * 2^-26 at worst, e_powf.c's header) and rounds once,
// so it can differ from the correctly rounded product only when the exact square -- hi + lo with lo = fmaf(x, x, -hi) -- lies next to a rounding boundary.
// EXHAUSTIVELY, over all 2^31 finite floats (tests/test_pow_exact.py, 4 s): powf(x, 2.0f) != x * x for 6 061 arguments per binade (0.072 %), and in every one of
// them the exact square is within 0.0017 ulp of the boundary.  The test below passes an argument when it is at least 1/64 ulp away (|lo| <= 31/64 ulp(hi)) and
// 2^-63 <= hi < 2^64: 31 of 32 arguments.  (Zeros, subnormals, infinities and NaNs fail the range test; the table routine selects x * x for them.)
MI_PW_DEV bool squaref_is_plain(float x, float &hi) {
    hi = x * x;
    const float lo = __builtin_fmaf(x, x, -hi);
    const uint32_t e = (bitsf(hi) - 1u) & 0x7f800000u;           // (one binade lower for an exact power of two: the smaller ulp is on its lower side)
    const float thr = from_bitsf(e - 0x0c080000u);                // 2^(E - 25) * 31/16 = 31/64 ulp(hi)
    const bool in_range = e - (64u << 23) < (127u << 23);        // 2^-63 <= hi < 2^64
    return in_range & (__builtin_fabsf(lo) <= thr);
}

}  // namespace mi_pow
