// envs_classic.h -- per-lane dynamics of the five classic-control environments for gfx950.
//
// One sub-environment per lane; the state lives in registers for the duration of a step (or of a whole fused
// rollout).  Arithmetic follows the reference expression by expression, in float64 except where NumPy-2 weak-scalar
// promotion makes the reference round to float32.  The translation unit is compiled with
// -ffp-contract=off so that no a*b+c is fused: the reference rounds every product.
//
// Every environment is a template over a math policy:
//   ExactMath (default): sin / cos are the reference's libm bit for bit (sincos_exact.h) and `x ** 2` on NumPy scalars is libm's pow / powf
//     bit for bit (pow_exact.h); fmod is exact by definition and everything else is IEEE arithmetic in the reference's order, so state,
//     rewards and flags equal the reference's float64 / float32 values EXACTLY, for whole episodes (tests/test_gpu_parity.py: array_equal).
//     The one place that cannot be reproduced is Acrobot's observation right after a reset, where NumPy evaluates cos / sin of a float32
//     array with its own SIMD float32 kernels (<= 1 float32 ulp; oracle and engine return the correctly rounded value there).
//   FastMath (opt-in, MI_CFG_FAST_MATH): ocml's sin / cos (<= 1 ulp from libm's) and x * x (correctly rounded, where glibc's pow is off by
//     one ulp on ~0.09 % of arguments).  Differences from the reference start at 1e-16 and grow with the system's own error amplification.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355env.h"
#include "pcg64_dev.h"
#include "pow_exact.h"
#include "sincos_exact.h"

namespace mi {

#define MI_DEV __device__ __forceinline__

constexpr double kPi = 3.141592653589793;

enum : uint32_t { kNeedsReset = 1u, kStateF32 = 2u };

// The tables of sincos_exact.h / pow_exact.h live in LDS for the lifetime of a kernel (5.2 KB + 5 KB + 0.5 KB; a kernel allocates only the
// ones its environment reads): every kernel that steps / resets / observes a classic env calls tables_init<E>() first.  A per-lane table
// index into global memory inside the rollout loop would tie the loop's loads to its stores (one in-order vmcnt on gfx950).
static __shared__ double g_trig6[660 + 18];  // the 6-wide sin / cos table and the reduction constants behind it (sincos_exact.h fill_hot)
static __shared__ double g_pow_log[384];
static __shared__ uint64_t g_pow_exp[256];
static __shared__ double g_powf_log2[32];
static __shared__ uint64_t g_powf_exp2[32];

struct TwoPi {
    static constexpr double value = 2 * kPi;
};
#ifndef MI_HOT_TRIG
#define MI_HOT_TRIG true
#endif

// Several IEEE quotients by ONE divisor.  The compiler's float64 division is v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 FMAs),
// q0 = a r, e = fma(-b, q0, a), v_div_fmas (= fma(e, r, q0) when nothing was scaled) and v_div_fixup: 11 instructions, one of them a 17-tick
// transcendental.  v_div_scale leaves both operands alone unless an exponent is extreme (divisor or quotient near the ends of the range, or a
// numerator below 2^-969), so for operands of ordinary magnitude the refined reciprocal depends on the divisor only and can be shared: the same
// operations on the same values, hence the same quotient bit for bit, in 5 + 4 per quotient.  The caller guarantees the ranges (ordinary()).
struct SharedDivisor {
    double b, r;
    MI_DEV explicit SharedDivisor(double divisor) : b(divisor) {
        const double r0 = __builtin_amdgcn_rcp(divisor);
        const double e0 = __builtin_fma(-divisor, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-divisor, r1, 1.0);
        r = __builtin_fma(r1, e1, r1);
    }
    MI_DEV double under(double a) const {  // a / b
        const double q0 = a * r;
        const double e = __builtin_fma(-b, q0, a);
        return __builtin_amdgcn_div_fixup(__builtin_fma(e, r, q0), b, a);
    }
    // 2^-511 <= |a| < 2^513: far inside what v_div_scale passes through for a divisor of ordinary size (a zero goes the long way round as well)
    static MI_DEV bool ordinary(double a) { return (((uint32_t)(mi_sincos::bits(a) >> 32) & 0x7fffffffu) - 0x20000000u) < 0x40000000u; }
};
// KASM: constant Horner steps as inline-asm v_fma_f64 (sincos_exact.h fma_k); false for kernels whose registers overflow into AGPRs
template <bool KASM>
struct ExactMathT {
    static constexpr bool EXACT = true;
    template <bool POW, bool POWF>
    static MI_DEV void init() {
        for (int e = threadIdx.x; e < 110; e += blockDim.x) mi_sincos::expand6(mi_sincos::kTable, g_trig6, e);
        if (threadIdx.x == 0) mi_sincos::fill_hot(g_trig6);
        if (POW) {
            for (int k = threadIdx.x; k < 384; k += blockDim.x) g_pow_log[k] = mi_pow::kLogTab[k];
            for (int k = threadIdx.x; k < 256; k += blockDim.x) g_pow_exp[k] = mi_pow::kExpTab[k];
        }
        if (POWF) {
            for (int k = threadIdx.x; k < 32; k += blockDim.x) g_powf_log2[k] = mi_pow::kLog2fTab[k], g_powf_exp2[k] = mi_pow::kExp2fTab[k];
        }
        __syncthreads();
    }
    // HOT: the range reduction's and the polynomials' constants are read from behind the table (sincos_exact.h fill_hot) instead of written as float64
    // literals: a literal costs two scalar moves at every use unless an SGPR pair holds it, and these kernels sit at the 106-SGPR limit; what comes
    // from LDS the compiler keeps in vector registers across the step loop.  (CartPole's short routine, sincos_main, keeps its literals.)
    static constexpr bool HOT = MI_HOT_TRIG;
    static MI_DEV double sin(double x) { return mi_sincos::sin_bf<false, KASM, HOT>(g_trig6, x); }
    static MI_DEV double cos(double x) { return mi_sincos::cos_bf<false, KASM, HOT>(g_trig6, x); }
    static MI_DEV void sincos(double x, double &s, double &c) { mi_sincos::sincos_bf<false, true, KASM>(g_trig6, x, s, c); }  // (CartPole: literals, its general path is rare)
    // the two halves of sincos() for a caller that defers the rare case: the short routine of |x| < 0.855469 evaluated unconditionally (its result
    // is only meaningful when in_main_range(x); an index outside the table reads zeros from LDS, no fault), and the test
    static MI_DEV void sincos_main_unchecked(double x, double &s, double &c) { mi_sincos::sincos_main<KASM>(g_trig6, x, s, c); }
    static MI_DEV bool in_main_range(double x) { return ((uint32_t)(mi_sincos::bits(x) >> 32) & 0x7fffffffu) < 0x3feb6000u; }
    static MI_DEV void sincos_spread(double x, double &s, double &c) { mi_sincos::sincos_bf<false, false, KASM, HOT>(g_trig6, x, s, c); }  // any range, no small-angle short cut
    // for angles the environment wraps or clips (|x| far below 1e8): no hand-over to the platform's huge-argument routine, and lanes
    // spread over all ranges (no wavefront-uniform short cut)
    static MI_DEV double sin_bounded(double x) { return mi_sincos::sin_bf<true, KASM, HOT>(g_trig6, x); }
    static MI_DEV double cos_bounded(double x) { return mi_sincos::cos_bf<true, KASM, HOT>(g_trig6, x); }
    static MI_DEV void sincos_bounded(double x, double &s, double &c) { mi_sincos::sincos_bf<true, false, KASM, HOT>(g_trig6, x, s, c); }
    static MI_DEV double cos_bounded_literals(double x) { return mi_sincos::cos_bf<true, KASM>(g_trig6, x); }  // (MountainCar: -0.8 % with HOT, profiles/r06_hot_constants_all_ab.txt)
    static MI_DEV double fmod_2pi(double x) { return mi_sincos::fmod_const(x, TwoPi()); }  // fmod(x, 2 pi): exact, like the library's, in half the instructions
    static MI_DEV double sq(double x) { return mi_pow::square<KASM>(g_pow_log, g_pow_exp, x); }    // np.float64 ** 2
    static MI_DEV float sqf(float x) { return mi_pow::squaref<KASM>(g_powf_log2, g_powf_exp2, x); }  // np.float32 ** 2
    // three / two np.float64 ** 2 at once: the table routine runs once per group for the lanes whose square is not provably the plain product (pow_exact.h)
    static MI_DEV void sq3(double a, double b, double c, double &ra, double &rb, double &rc) { mi_pow::square3<KASM>(g_pow_log, g_pow_exp, a, b, c, ra, rb, rc); }
    static MI_DEV void sq2(double a, double b, double &ra, double &rb) { mi_pow::square2<KASM>(g_pow_log, g_pow_exp, a, b, ra, rb); }
    // the test alone, for a caller that collects the arguments which need the table routine (the two-role Pendulum rollout, engine.hip): true = `hi` IS x ** 2
    static MI_DEV bool sq_is_plain(double x, double &hi) { return mi_pow::square_is_plain(x, hi); }
    static MI_DEV bool sqf_is_plain(float x, float &hi) { return mi_pow::squaref_is_plain(x, hi); }  // ... and for np.float32 ** 2
};
typedef ExactMathT<true> ExactMath;
typedef ExactMathT<false> ExactMathBuiltinFma;  // Acrobot: its kernels use AGPRs (see sincos_exact.h fma_k)
struct FastMath {
    static constexpr bool EXACT = false;
    template <bool POW, bool POWF>
    static MI_DEV void init() {}
    static MI_DEV double sin(double x) { return ::sin(x); }
    static MI_DEV double cos(double x) { return ::cos(x); }
    // |x| <= pi/4 (always the case for a CartPole inside its 12-degree termination band) needs no range reduction: minimax kernels on
    // [-pi/4, pi/4] (the fdlibm k_sin / k_cos coefficient sets, evaluated with FMAs; < 1 ulp).  Anything else goes through ocml's sincos.
    // The branch is wavefront-uniform in practice, so a CartPole wavefront never pays for the reduction code.
    static MI_DEV void sincos(double x, double &sn, double &cs) {
        if (__builtin_expect(!(fabs(x) <= 0.7853981633974483), 0)) {
            ::sincos(x, &sn, &cs);
            return;
        }
        auto fma_k = [](double a, double b, double c) { return mi_sincos::fma_k<true>(a, b, c); };
        const double z = x * x;
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        double rs = fma_k(z, S6, S5);
        rs = fma_k(z, rs, S4), rs = fma_k(z, rs, S3), rs = fma_k(z, rs, S2), rs = fma_k(z, rs, S1);
        sn = fma(z * x, rs, x);
        double rc = fma_k(z, C6, C5);
        rc = fma_k(z, rc, C4), rc = fma_k(z, rc, C3), rc = fma_k(z, rc, C2), rc = fma_k(z, rc, C1);
        const double hz = 0.5 * z;
        const double w = 1.0 - hz;
        cs = w + (((1.0 - w) - hz) + z * (z * rc));
    }
    static MI_DEV void sincos_spread(double x, double &s, double &c) { ::sincos(x, &s, &c); }
    static MI_DEV double sin_bounded(double x) { return ::sin(x); }
    static MI_DEV double cos_bounded(double x) { return ::cos(x); }
    static MI_DEV double cos_bounded_literals(double x) { return ::cos(x); }
    static MI_DEV void sincos_bounded(double x, double &s, double &c) { ::sincos(x, &s, &c); }
    static MI_DEV double fmod_2pi(double x) { return ::fmod(x, 2 * kPi); }
    static MI_DEV double sq(double x) { return x * x; }
    static MI_DEV float sqf(float x) { return x * x; }
    static MI_DEV void sq3(double a, double b, double c, double &ra, double &rb, double &rc) { ra = a * a, rb = b * b, rc = c * c; }
    static MI_DEV void sq2(double a, double b, double &ra, double &rb) { ra = a * a, rb = b * b; }
    static MI_DEV bool sq_is_plain(double x, double &hi) { return hi = x * x, true; }
    static MI_DEV bool sqf_is_plain(float x, float &hi) { return hi = x * x, true; }
};
template <class E>
MI_DEV void tables_init() {
    E::Math::template init<E::USES_POW, E::USES_POWF>();
}

// x / C for a compile-time constant C, bit-identical to the IEEE division for every |x| in [1e-300, 1e300]:
// q = x * RN(1/C) is within ~1.5 ulp, the FMA residual x - q*C is exact, and one correction step rounds correctly
// (Markstein); checked against x / C on 6e8 random doubles for C = 1.1.  Three FMAs instead of the ~11-instruction
// v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence; everything outside the range takes the real division.
template <class C>
MI_DEV double div_by_constant(double x, C) {
    constexpr double c = C::value, r = 1.0 / C::value;
    const double q = x * r;
    const double q2 = fma(fma(-q, c, x), r, q);
    const double ax = fabs(x);
    if (__builtin_expect(!(ax <= 1e300 && ax >= 1e-300), 0)) return x / c;
    return q2;
}

// the two halves of div_by_constant for a caller that tests several operands with ONE branch (CartPole's step: three divisions by total_mass)
template <class C>
MI_DEV double div_by_constant_unchecked(double x, C) {
    constexpr double c = C::value, r = 1.0 / C::value;
    const double q = x * r;
    return fma(fma(-q, c, x), r, q);
}
MI_DEV bool div_by_constant_in_range(double x) {
    const double ax = fabs(x);
    return (ax <= 1e300) & (ax >= 1e-300);
}

struct EnvParams {
    double p[16];
};

// What an environment carries from obs() to the next step() while its state sits in registers (fused rollouts): obs() evaluates sin / cos of
// the angles it reports, and the next step() starts from the very same angles -- with the exact libm routines an evaluation costs ~100
// instructions, so Pendulum and Acrobot keep the values (same bits: the same function of the same argument).  obs() always recomputes and
// refreshes; step() uses the values only if `ok` (false after a load from memory and after step() itself has moved the state).
struct NoTrig {};
struct PendulumTrig {
    double sn, cs;
    bool ok;
};
struct AcrobotTrig {
    double s1, c1, s2, c2;
    bool ok;
};
MI_DEV void trig_invalidate(NoTrig &) {}
MI_DEV void trig_invalidate(PendulumTrig &t) { t.ok = false; }
MI_DEV void trig_invalidate(AcrobotTrig &t) { t.ok = false; }

// What an action row of a Box space holds (mi_step_io.actions_dtype).  The reference hands the caller's rows to the scalar env as they are
// (vector/sync_vector_env.py:274), and NumPy 2's promotion then depends on what `action[0]` is: an np.float32 (rows of a float32 array: the
// space's dtype, what Box.sample() draws), an np.float64 (rows of a float64 array), or a Python float (rows of a list of lists), which is
// WEAK: np.float32 + float stays float32.
struct ActF32 {
    typedef float T;
    static constexpr int KIND = MI_F32;
};
struct ActF64 {
    typedef double T;
    static constexpr int KIND = MI_F64;
};
struct ActF64Weak {
    typedef double T;
    static constexpr int KIND = MI_F64_WEAK;
};

// ---------------------------------------------------------------------------------------------------------
// CartPole-v1: gymnasium/envs/classic_control/cartpole.py:119-247
// ---------------------------------------------------------------------------------------------------------
struct CartPoleTotalMass {
    static constexpr double value = 0.1 + 1.0;  // masspole + masscart (cartpole.py:127)
};
template <class M>
struct CartPoleT {
    typedef M Math;
    static constexpr bool SPLIT_TERMINAL = false;
    static constexpr bool USES_POW = false, USES_POWF = false;
    typedef CartPoleTotalMass TotalMass;
    static constexpr int S = 4, OBS = 4, N_ACTIONS = 2;
    static constexpr bool DISCRETE = true;
    static constexpr int ACT_KIND = MI_I64;
    static constexpr int ROLLOUT_UNROLL = 1;  // engine.hip rollout_kernel: steps per unrolled loop body
    static constexpr bool DUO_ROLLOUT = true;  // engine.hip rollout_duo_kernel (env + aux wavefront per 64 sub-environments): measured +13 %
    static constexpr int DUO_CHUNK = 8;  // steps per phase of the two-role kernel (round 6, after the phase overhead left the wavefronts: +3.0 % over 4; round 4 measured -2.4 %)
    static constexpr bool DUO_ACT_AHEAD = false;  // (profiles/r06_act_prefetch_ab.txt)
    // the reward of a (non-reset) step is a function of its terminated flag (step(), below): the aux role recomputes it instead of receiving it
    static constexpr bool REWARD_FROM_TERMINATED = true;
    static MI_DEV double reward_from_terminated(bool terminated, const EnvParams &P) {
        const bool sutton_barto = P.p[0] != 0.0;
        return terminated ? (sutton_barto ? -1.0 : 1.0) : (sutton_barto ? 0.0 : 1.0);
    }
    // Two-role rollout, round 6: the aux role CAN derive the observation row and the flags of a step from the state itself -- the env role then hands
    // over the two components the termination test reads (x, theta: cartpole.py:200-205) as the float64 they are and the two velocities as the float32
    // the observation holds, and neither converts an observation nor packs a flag word.  aux_unpack's test is step()'s, on the same doubles.
    // Measured (profiles/r06_duo_diet_ab.txt): what bounds the pair of wavefronts is the SUM of their instructions, and for CartPole the hand-over costs
    // the aux role what it saves the env role: 127.2 G against 128.3 G env-steps/s -- off here; MountainCar, whose state is the whole observation, gains 1.9 %.
    static constexpr bool AUX_DERIVES_FLAGS = false;
    static constexpr bool AUX_REWARD = false;  // (the reward comes from the flags here: REWARD_FROM_TERMINATED)
    static constexpr bool AUX_REWARD_OF_ACTION = false;
    template <class A>
    static MI_DEV double reward_of_action(bool, A) { return 0.0; }
    static constexpr int AUX_PRE = 1;
    static constexpr int AUX_PRE_SQUARES = 0;
    static MI_DEV uint32_t aux_pre(const double *, double *) { return 0u; }
    static MI_DEV double aux_square(double x) { return x; }
    static MI_DEV double aux_reward(const double *, int64_t) { return 0.0; }
    static constexpr int AUX_F64 = 2, AUX_F32 = 2;
    static MI_DEV void aux_pack(const double s[S], double w64[AUX_F64], float w32[AUX_F32]) {
        w64[0] = s[0], w64[1] = s[2], w32[0] = (float)s[1], w32[1] = (float)s[3];
    }
    static MI_DEV void aux_unpack(const double w64[AUX_F64], const float w32[AUX_F32], const EnvParams &, float o[OBS], bool &terminated) {
        const double theta_threshold = 12 * 2 * kPi / 360, x_threshold = 2.4;
        const double x = w64[0], theta = w64[1];
        o[0] = (float)x, o[1] = w32[0], o[2] = (float)theta, o[3] = w32[1];
        terminated = x < -x_threshold || x > x_threshold || theta < -theta_threshold || theta > theta_threshold;
    }
    typedef int64_t Act;

    static MI_DEV void default_bounds(double &b0, double &b1) { b0 = -0.05, b1 = 0.05; }

    // cartpole.py:242  state = np_random.uniform(low, high, size=(4,)); u[k] are the stream's next_double() values
    static constexpr int NDRAWS = 4;
    static MI_DEV void reset_u(const double u[NDRAWS], double s[S], uint32_t &flags, double b0, double b1) {
        const double range = b1 - b0;
#pragma unroll
        for (int k = 0; k < S; k++) s[k] = b0 + range * u[k];
        (void)flags;
    }
    typedef NoTrig Trig;
    static MI_DEV void obs(const double s[S], uint32_t, float o[OBS], Trig &) {
#pragma unroll
        for (int k = 0; k < OBS; k++) o[k] = (float)s[k];
    }
    static MI_DEV bool valid(Act a) { return a >= 0 && a < N_ACTIONS; }
    static MI_DEV Act sample(double u) { return (Act)(u * 2.0); }  // (random(N) * nvec).astype(int64)
    // the same value straight from the 64 random bits: u = (bits >> 11) * 2^-53, and u * 2.0 is exact, so
    // floor(u * 2) is the top bit
    static constexpr bool SAMPLE_FROM_BITS = true;
    static MI_DEV Act sample_bits(uint64_t bits) { return (Act)(bits >> 63); }
    // ... and straight from the generator's state words: the output is rotr64(hi ^ lo, hi >> 58), whose top bit is bit (63 + rot) mod 64 of hi ^ lo --
    // one 64-bit shift instead of the rotation's two and their merge
    static MI_DEV Act sample_state(uint64_t hi, uint64_t lo) {
        const unsigned rot = (unsigned)(hi >> 58);
        return (Act)(((hi ^ lo) >> ((rot + 63u) & 63u)) & 1ull);
    }

    struct Accel {
        double thetaacc, xacc;
    };
    // the rare lanes' accelerations through the general routines, OUT OF LINE: the hot path then carries neither this body's scalar registers nor the
    // spill of the saved exec mask around it
    static __device__ __attribute__((noinline)) Accel general_accel(double theta, double theta_dot, double force) {
        const double gravity = 9.8, masspole = 0.1, length = 0.5;
        const double polemass_length = masspole * length;
        double sintheta, costheta;
        M::sincos(theta, sintheta, costheta);
        const double temp = div_by_constant(force + polemass_length * (theta_dot * theta_dot) * sintheta, TotalMass());
        Accel r;
        r.thetaacc = (gravity * sintheta - costheta * temp) /
                     (length * (4.0 / 3.0 - div_by_constant(masspole * (costheta * costheta), TotalMass())));
        r.xacc = temp - div_by_constant(polemass_length * r.thetaacc * costheta, TotalMass());
        return r;
    }
    // cartpole.py:164-226: explicit Euler with the OLD velocities, all float64.
    static MI_DEV void step(double s[S], uint32_t &, Act action, const EnvParams &P, double &reward, bool &terminated, Trig &) {
        const double gravity = 9.8, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
        const double polemass_length = masspole * length;
        const double theta_threshold = 12 * 2 * kPi / 360;
        const double x_threshold = 2.4;
        double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const double force = action == 1 ? force_mag : -force_mag;
        double costheta, sintheta;  // cartpole.py:180-181 np.cos / np.sin
        double temp, thetaacc, xacc;
        if constexpr (M::EXACT) {
            // Round 6: ONE rare branch per step instead of four.  The step has four places where a lane may leave the common case -- an angle
            // outside the short sin / cos routine's range and the three operands of the constant divisions outside the range in which three
            // FMAs give the IEEE quotient -- and each cost a compare pair, two exec-mask instructions and a branch on a wavefront that issues one
            // instruction at a time.  The common case is now evaluated unconditionally (same operations on the same operands: same bits) and the
            // lanes that left it redo the step through the general routines.  (masspole * cos^2 needs no test of its own: inside the short
            // routine's range cos >= 0.65.)
            const bool main_range = M::in_main_range(theta);
            M::sincos_main_unchecked(theta, sintheta, costheta);
            const double t1 = force + polemass_length * (theta_dot * theta_dot) * sintheta;
            temp = div_by_constant_unchecked(t1, TotalMass());
            thetaacc = (gravity * sintheta - costheta * temp) /
                       (length * (4.0 / 3.0 - div_by_constant_unchecked(masspole * (costheta * costheta), TotalMass())));
            const double t3 = polemass_length * thetaacc * costheta;
            xacc = temp - div_by_constant_unchecked(t3, TotalMass());
            // (`&`, not `&&`: three flags combined by scalar instructions, no control flow of their own)
            const bool common = main_range & div_by_constant_in_range(t1) & div_by_constant_in_range(t3);
            if (__builtin_expect(!common, 0)) {
                const Accel r = general_accel(theta, theta_dot, force);
                thetaacc = r.thetaacc, xacc = r.xacc;
            }
        } else {
            M::sincos(theta, sintheta, costheta);
            temp = div_by_constant(force + polemass_length * (theta_dot * theta_dot) * sintheta, TotalMass());
            thetaacc = (gravity * sintheta - costheta * temp) /
                       (length * (4.0 / 3.0 - div_by_constant(masspole * (costheta * costheta), TotalMass())));
            xacc = temp - div_by_constant(polemass_length * thetaacc * costheta, TotalMass());
        }
        x = x + tau * x_dot;
        x_dot = x_dot + tau * xacc;
        theta = theta + tau * theta_dot;
        theta_dot = theta_dot + tau * thetaacc;
        s[0] = x, s[1] = x_dot, s[2] = theta, s[3] = theta_dot;
        terminated = x < -x_threshold || x > x_threshold || theta < -theta_threshold || theta > theta_threshold;
        const bool sutton_barto = P.p[0] != 0.0;
        reward = terminated ? (sutton_barto ? -1.0 : 1.0) : (sutton_barto ? 0.0 : 1.0);
    }
};

// ---------------------------------------------------------------------------------------------------------
// Pendulum-v1: gymnasium/envs/classic_control/pendulum.py:102-171,281-282
// ---------------------------------------------------------------------------------------------------------
template <class M, class AK = ActF32>
struct PendulumT {
    typedef M Math;
    static constexpr bool SPLIT_TERMINAL = false;
    static constexpr int ACT_KIND = AK::KIND;
    static constexpr bool USES_POW = true, USES_POWF = ACT_KIND == MI_F32;
    static constexpr int S = 2, OBS = 3;
    static constexpr bool DISCRETE = false;
    static constexpr int ROLLOUT_UNROLL = 1;
    // engine.hip rollout_duo_kernel (env + aux wavefront per 64 sub-environments).  Round 4 measured -0.5 % (and +-0 with the reward on the aux role) and
    // concluded that the instruction count is the limit; it was the kernel's own phase overhead (round 6: lane masks carried through the role branches).
    // Now the step is cut in two balanced halves: the env role advances the state and takes the observation (one exact sincos) plus ONE of the reward's
    // three exact pow() calls and the angle's normalisation (the exact fmod), the aux role evaluates the rest of the reward -- two pow and the sums, from
    // the normalised pre-step angle and the action it drew itself -- next to the policy, the episode statistics and the stores.  Same operations on the
    // same operands (reward_of == aux_reward o aux_pre).
    static constexpr bool DUO_ROLLOUT = true && ACT_KIND == MI_F32;  // (the float64-row instantiations never sample: one role; MI355ENV_ROLLOUT_DUO=0 is the A/B switch)
    static constexpr int DUO_CHUNK = 8;
    static constexpr bool DUO_ACT_AHEAD = false;  // (profiles/r06_act_prefetch_ab.txt)
    static constexpr bool REWARD_FROM_TERMINATED = false;
    static MI_DEV double reward_from_terminated(bool, const EnvParams &) { return 0.0; }
    static constexpr bool AUX_DERIVES_FLAGS = false;
    static constexpr int AUX_F64 = 1, AUX_F32 = 0;
    static MI_DEV void aux_pack(const double *, double *, float *) {}
    static MI_DEV void aux_unpack(const double *, const float *, const EnvParams &, float *, bool &) {}
    static constexpr bool AUX_REWARD = true;
    static constexpr bool AUX_REWARD_OF_ACTION = false;
    template <class A>
    static MI_DEV double reward_of_action(bool, A) { return 0.0; }
    static constexpr int AUX_PRE = 2;
    typedef typename AK::T Act;
    // pendulum.py:131 in two halves: what the env role hands over about the state BEFORE the step ...
    // (round 6, second cut: with the angle handed over as it is the aux role was the longer one -- 306 k against 229 k cycles of work per launch,
    //  profiles/r06_duo_timing.txt -- so angle_normalize moved to the env role as well)
    static MI_DEV double angle_normalize(double th) {  // pendulum.py:281-282  ((x + pi) % (2 pi)) - pi with Python's floor-modulo
        double md = M::fmod_2pi(th + kPi);
        if (md != 0.0) {
            if (md < 0.0) md += 2 * kPi;
        } else {
            md = 0.0;
        }
        return md - kPi;
    }
    // (round 6, third cut: the two float64 squares.  31 of 32 arguments' `** 2` is provably the plain product -- pow_exact.h square_is_plain -- so the env
    //  role hands over the squares it could take that way and the raw arguments otherwise, with one bit each saying which; the aux role runs the table
    //  routine once per PENDING argument of a chunk's 16, all lanes side by side: 2.7 passes per chunk on average instead of 16.)
    static constexpr int AUX_PRE_SQUARES = 2;  // pre[0 .. 1] become squares on the aux role
    static MI_DEV uint32_t aux_pre(const double s[S], double pre[AUX_PRE]) {
        const double an = angle_normalize(s[0]);
        double h0, h1;
        const bool p0 = M::sq_is_plain(an, h0), p1 = M::sq_is_plain(s[1], h1);
        pre[0] = p0 ? h0 : an, pre[1] = p1 ? h1 : s[1];
        return (p0 ? 0u : 1u) | (p1 ? 0u : 2u);
    }
    static MI_DEV double aux_square(double x) { return M::sq(x); }

    static MI_DEV void default_bounds(double &b0, double &b1) { b0 = kPi, b1 = 1.0; }  // DEFAULT_X, DEFAULT_Y

    // pendulum.py:149-167: high = [x_init, y_init], low = -high, uniform(low, high) -> theta, thetadot
    static constexpr int NDRAWS = 2;
    static MI_DEV void reset_u(const double u[NDRAWS], double s[S], uint32_t &, double x_init, double y_init) {
        s[0] = -x_init + (x_init - (-x_init)) * u[0];
        s[1] = -y_init + (y_init - (-y_init)) * u[1];
    }
    typedef PendulumTrig Trig;
    static MI_DEV void obs(const double s[S], uint32_t, float o[OBS], Trig &t) {
        // theta is never wrapped in the state (pendulum.py:141): the lanes of a wavefront spread over all argument ranges, so the short cut
        // for small angles (CartPole's) would only add its own instructions to the general path; the hand-over for huge angles stays
        // (|theta| grows by up to 0.4 per step: 1e8 is reachable without a TimeLimit)
        M::sincos_spread(s[0], t.sn, t.cs);
        t.ok = true;
        o[0] = (float)t.cs, o[1] = (float)t.sn, o[2] = (float)s[1];
    }
    static MI_DEV bool valid(Act) { return true; }
    static MI_DEV Act sample(double u) { return (Act)(-2.0 + (2.0 - (-2.0)) * u); }  // Box.sample: uniform(low, high).astype(f32)
    static constexpr bool SAMPLE_FROM_BITS = false;
    static MI_DEV Act sample_bits(uint64_t) { return 0; }
    static MI_DEV Act sample_state(uint64_t, uint64_t) { return 0; }

    static MI_DEV Act clip_torque(Act u) {  // np.clip(u, -2, 2)[0] stays np.float32 for a float32 row; a float64 row (or a list's) makes it an np.float64
        const double max_torque = 2.0;
        u = u < (Act)-max_torque ? (Act)-max_torque : u;
        u = u > (Act)max_torque ? (Act)max_torque : u;
        return u;
    }
    // pendulum.py:131  costs = angle_normalize(th) ** 2 + 0.1 * thdot ** 2 + 0.001 * (u ** 2), of the state BEFORE the step; reward = -costs.
    static MI_DEV double reward_of(double th, double thdot, Act action) {
        const Act u = clip_torque(action);
        const double an = angle_normalize(th);
        double cu;
        if constexpr (ACT_KIND == MI_F32)
            cu = (double)(0.001f * M::sqf(u));  // float32: 0.001 * (u ** 2); NumPy scalar ** is libm powf / pow (pendulum.py:131)
        else
            cu = 0.001 * M::sq(u);  // all float64
        double sq_an, sq_thdot;
        M::sq2(an, thdot, sq_an, sq_thdot);
        const double costs = sq_an + 0.1 * sq_thdot + cu;
        return -costs;
    }
    // ... and the reward from it on the aux role: reward_of with angle_normalize(th) ** 2 and thdot ** 2 already evaluated (the sums in the reference's order)
    static MI_DEV double aux_reward(const double pre[AUX_PRE], Act action) {
        const Act u = clip_torque(action);
        double cu;
        if constexpr (ACT_KIND == MI_F32)
            cu = (double)(0.001f * M::sqf(u));
        else
            cu = 0.001 * M::sq(u);
        const double costs = pre[0] + 0.1 * pre[1] + cu;
        return -costs;
    }
    // (round 6, fourth cut: the float32 `u ** 2`.  powf's square is provably the plain product for 31 of 32 floats too -- pow_exact.h squaref_is_plain, checked
    //  over all 2^32 of them -- so the aux role tests a chunk's clipped actions first, runs the table routine once per PENDING action, all lanes side by side,
    //  and reads the squares back in the step loop: engine.hip rollout_duo_kernel.)
    static constexpr bool AUX_ACT_SQUARE = M::EXACT && ACT_KIND == MI_F32;
    static MI_DEV float act_square_arg(Act action) { return (float)clip_torque(action); }
    static MI_DEV double aux_reward_squared(const double pre[AUX_PRE], float u_squared) {  // aux_reward with powf(u, 2) already evaluated
        const double cu = (double)(0.001f * u_squared);
        const double costs = pre[0] + 0.1 * pre[1] + cu;
        return -costs;
    }
    // pendulum.py:133-141 the dynamics alone
    static MI_DEV void advance(double s[S], Act action, const EnvParams &P, Trig &t) {
        const double max_speed = 8, dt = 0.05, m = 1.0, l = 1.0;
        const double g = P.p[0];
        const double th = s[0], thdot = s[1];
        const Act u = clip_torque(action);
        double tu;
        if constexpr (ACT_KIND == MI_F32)
            tu = (double)((float)(3.0 / (m * (l * l))) * u);  // float32: 3.0 / (m l^2) * u
        else
            tu = 3.0 / (m * (l * l)) * u;
        if (!t.ok) t.sn = M::sin(th);  // (else: the observation of the previous step evaluated sin of this very angle)
        const double sn = t.sn;
        t.ok = false;
        double newthdot = thdot + (3 * g / (2 * l) * sn + tu) * dt;
        newthdot = newthdot < -max_speed ? -max_speed : newthdot;
        newthdot = newthdot > max_speed ? max_speed : newthdot;
        const double newth = th + newthdot * dt;
        s[0] = newth, s[1] = newthdot;
    }
    static MI_DEV void step(double s[S], uint32_t &, Act action, const EnvParams &P, double &reward, bool &terminated, Trig &t) {
        reward = reward_of(s[0], s[1], action);
        advance(s, action, P, t);
        terminated = false;
    }
};

// ---------------------------------------------------------------------------------------------------------
// Acrobot-v1: gymnasium/envs/classic_control/acrobot.py:172-279,375-461
// ---------------------------------------------------------------------------------------------------------
template <class M>
struct AcrobotT {
    typedef M Math;
    // This environment's kernels need ~290 live registers: the compiler parks the overflow in AGPRs (v_accvgpr_read / write in the loop).  They
    // are instantiated with ExactMathBuiltinFma (engine.hip AcrobotMath): no inline-asm Horner steps next to AGPR traffic (sincos_exact.h fma_k).
#ifndef MI_ACROBOT_SPLIT_TERMINAL
#define MI_ACROBOT_SPLIT_TERMINAL 1
#endif
    static constexpr bool SPLIT_TERMINAL = MI_ACROBOT_SPLIT_TERMINAL != 0;  // fused rollouts: integrate(), obs(), terminal_after_obs() (engine.hip lane_step_fused)
    static constexpr bool USES_POW = true, USES_POWF = false;
    static constexpr int S = 4, OBS = 6, N_ACTIONS = 3;
    static constexpr bool DISCRETE = true;
    static constexpr int ACT_KIND = MI_I64;
    static constexpr int ROLLOUT_UNROLL = 2;  // engine.hip rollout_kernel: steps per unrolled loop body
    static constexpr bool DUO_ROLLOUT = false;  // engine.hip rollout_duo_kernel (env + aux wavefront per 64 sub-environments): measured +-0: 256 registers
    typedef int64_t Act;

    static MI_DEV void default_bounds(double &b0, double &b1) { b0 = -0.1, b1 = 0.1; }

    // acrobot.py:185-200: uniform(low, high, size=(4,)).astype(np.float32)
    static constexpr int NDRAWS = 4;
    static MI_DEV void reset_u(const double u[NDRAWS], double s[S], uint32_t &flags, double b0, double b1) {
        const double range = b1 - b0;
#pragma unroll
        for (int k = 0; k < S; k++) s[k] = (double)(float)(b0 + range * u[k]);
        flags |= kStateF32;
    }
    // acrobot.py:232-237 (after a reset NumPy evaluates these in float32; we return the correctly rounded value)
    typedef AcrobotTrig Trig;
    static MI_DEV void obs(const double s[S], uint32_t, float o[OBS], Trig &t) {
        M::sincos_bounded(s[0], t.s1, t.c1);
        M::sincos_bounded(s[1], t.s2, t.c2);
        t.ok = true;
        o[0] = (float)t.c1, o[1] = (float)t.s1, o[2] = (float)t.c2, o[3] = (float)t.s2, o[4] = (float)s[2], o[5] = (float)s[3];
    }
    static MI_DEV bool valid(Act a) { return a >= 0 && a < N_ACTIONS; }
    static MI_DEV Act sample(double u) { return (Act)(u * 3.0); }  // u * 3.0 rounds: not reducible to integer arithmetic
    static constexpr bool SAMPLE_FROM_BITS = false;
    static MI_DEV Act sample_bits(uint64_t) { return 0; }
    static MI_DEV Act sample_state(uint64_t, uint64_t) { return 0; }

    // acrobot.py:244-279, "book" dynamics; y = (theta1, theta2, dtheta1, dtheta2), a = torque.  The `** 2` on Python floats (lc1**2 = 0.25, ...)
    // are exact; the ones on np.float64 state components go through libm pow (M::sq).
    // known: sin / cos of y[1] if the caller already has them (the first RK4 stage starts at the state the last observation was taken of)
    static MI_DEV void dsdt(const double y[4], double a, double d[4], bool known = false, double s2k = 0, double c2k = 0) {
        const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
        const double theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
        double c2 = c2k, s2 = s2k;
        if (!known) M::sincos_bounded(theta2, s2, c2);
        const double d1 = m1 * (lc1 * lc1) + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2) + I1 + I2;
        const double d2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2;
        const double phi2 = m2 * lc2 * g * M::cos_bounded(theta1 + theta2 - kPi / 2.0);
        double sq_dtheta2, sq_dtheta1, sq_d2;  // dtheta2 ** 2, dtheta1 ** 2, d2 ** 2 (acrobot.py:263-275): one pass of the pow routine for the three
        M::sq3(dtheta2, dtheta1, d2, sq_dtheta2, sq_dtheta1, sq_d2);
        const double phi1 = -m2 * l1 * lc2 * sq_dtheta2 * s2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2 +
                            (m1 * lc1 + m2 * l1) * g * M::cos_bounded(theta1 - kPi / 2) + phi2;
        double ddtheta2, ddtheta1;
        if constexpr (M::EXACT) {
            // three of the four divisions are by d1, which lies in [2.5, 4.5] (|c2| <= 1); d2 in [0.75, 1.75] and its square are ordinary by
            // construction, the third numerator is tested
            const SharedDivisor by_d1(d1);
            ddtheta2 = (a + by_d1.under(d2) * phi1 - m2 * l1 * lc2 * sq_dtheta1 * s2 - phi2) / (m2 * (lc2 * lc2) + I2 - by_d1.under(sq_d2));
            const double n1 = -(d2 * ddtheta2 + phi1);
            ddtheta1 = by_d1.under(n1);
            if (__builtin_expect(!SharedDivisor::ordinary(n1), 0)) ddtheta1 = n1 / d1;
        } else {
            ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * sq_dtheta1 * s2 - phi2) / (m2 * (lc2 * lc2) + I2 - sq_d2 / d1);
            ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
        }
        d[0] = dtheta1, d[1] = dtheta2, d[2] = ddtheta1, d[3] = ddtheta2;
    }
    static MI_DEV double wrap(double x, double lo, double hi) {
        const double diff = hi - lo;
        while (x > hi) x = x - diff;
        while (x < lo) x = x + diff;
        return x;
    }
    static MI_DEV double bound(double x, double lo, double hi) {
        const double t = (lo > x) ? lo : x;
        return (hi < t) ? hi : t;
    }
    // acrobot.py:202-230 with rk4 (:415-461) over t = [0, 0.2]; the torque component has derivative 0
    static MI_DEV void step(double s[S], uint32_t &flags, Act action, const EnvParams &, double &reward, bool &terminated, Trig &tc) {
        integrate(s, flags, action, tc);
        terminated = (-M::cos_bounded(s[0]) - M::cos_bounded(s[1] + s[0])) > 1.0;
        reward = terminated ? 0.0 : -1.0;
    }
    // acrobot.py:239-242 on a state whose observation has just been taken: t holds cos / sin of s[0] and s[1] (the same libm cos of the same argument)
    static MI_DEV void terminal_after_obs(const double s[S], const Trig &t, double &reward, bool &terminated) {
        terminated = (-t.c1 - M::cos_bounded(s[1] + s[0])) > 1.0;
        reward = terminated ? 0.0 : -1.0;
    }
    static MI_DEV void integrate(double s[S], uint32_t &flags, Act action, Trig &tc) {
        const double dt = 0.2 - 0, dt2 = dt / 2.0, dt6 = dt / 6.0;
        const double a = action == 0 ? -1.0 : (action == 1 ? 0.0 : 1.0);
        double k1[4], k2[4], k3[4], k4[4], t[4];
        dsdt(s, a, k1, tc.ok, tc.s2, tc.c2);
        tc.ok = false;
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = s[i] + dt2 * k1[i];
        dsdt(t, a, k2);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = s[i] + dt2 * k2[i];
        dsdt(t, a, k3);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = s[i] + dt * k3[i];
        dsdt(t, a, k4);
        double ns[4];
#pragma unroll
        for (int i = 0; i < 4; i++) ns[i] = s[i] + dt6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        ns[0] = wrap(ns[0], -kPi, kPi);
        ns[1] = wrap(ns[1], -kPi, kPi);
        ns[2] = bound(ns[2], -4 * kPi, 4 * kPi);
        ns[3] = bound(ns[3], -9 * kPi, 9 * kPi);
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = ns[i];
        flags &= ~kStateF32;
    }
};

// ---------------------------------------------------------------------------------------------------------
// MountainCar-v0: gymnasium/envs/classic_control/mountain_car.py:108-170
// ---------------------------------------------------------------------------------------------------------
template <class M>
struct MountainCarT {
    typedef M Math;
    static constexpr bool SPLIT_TERMINAL = false;
    static constexpr bool USES_POW = false, USES_POWF = false;
    static constexpr int S = 2, OBS = 2, N_ACTIONS = 3;
    static constexpr bool DISCRETE = true;
    static constexpr int ACT_KIND = MI_I64;
    static constexpr int ROLLOUT_UNROLL = 1;  // engine.hip rollout_kernel: steps per unrolled loop body
    static constexpr bool DUO_ROLLOUT = true;  // engine.hip rollout_duo_kernel (env + aux wavefront per 64 sub-environments): measured +15 %
    static constexpr int DUO_CHUNK = 8;  // steps per phase of the two-role kernel (+2.6 % over 4)
    static constexpr bool DUO_ACT_AHEAD = true;   // +3.5 % here; CartPole -0.7 %, Pendulum -1 %, MountainCarContinuous -3.4 %
    static constexpr bool REWARD_FROM_TERMINATED = true;  // -1.0 every step (step(), below)
    static MI_DEV double reward_from_terminated(bool, const EnvParams &) { return -1.0; }
    // (see CartPoleT: the aux role of the two-role rollout derives observation and flags from the float64 state, mountain_car.py:139-142)
    static constexpr bool AUX_DERIVES_FLAGS = true;
    static constexpr bool AUX_REWARD = false;
    static constexpr bool AUX_REWARD_OF_ACTION = false;
    template <class A>
    static MI_DEV double reward_of_action(bool, A) { return 0.0; }
    static constexpr int AUX_PRE = 1;
    static constexpr int AUX_PRE_SQUARES = 0;
    static MI_DEV uint32_t aux_pre(const double *, double *) { return 0u; }
    static MI_DEV double aux_square(double x) { return x; }
    static MI_DEV double aux_reward(const double *, int64_t) { return 0.0; }
    static constexpr int AUX_F64 = 2, AUX_F32 = 0;
    static MI_DEV void aux_pack(const double s[S], double w64[AUX_F64], float *) { w64[0] = s[0], w64[1] = s[1]; }
    static MI_DEV void aux_unpack(const double w64[AUX_F64], const float *, const EnvParams &P, float o[OBS], bool &terminated) {
        const double goal_position = 0.5;
        o[0] = (float)w64[0], o[1] = (float)w64[1];
        terminated = w64[0] >= goal_position && w64[1] >= P.p[0];
    }
    typedef int64_t Act;

    static MI_DEV void default_bounds(double &b0, double &b1) { b0 = -0.6, b1 = -0.4; }
    static constexpr int NDRAWS = 1;
    static MI_DEV void reset_u(const double u[NDRAWS], double s[S], uint32_t &flags, double b0, double b1) {
        s[0] = b0 + (b1 - b0) * u[0];
        s[1] = 0.0;
        flags &= ~kStateF32;
    }
    typedef NoTrig Trig;
    static MI_DEV void obs(const double s[S], uint32_t, float o[OBS], Trig &) { o[0] = (float)s[0], o[1] = (float)s[1]; }
    static MI_DEV bool valid(Act a) { return a >= 0 && a < N_ACTIONS; }
    static MI_DEV Act sample(double u) { return (Act)(u * 3.0); }  // u * 3.0 rounds: not reducible to integer arithmetic
    static constexpr bool SAMPLE_FROM_BITS = false;
    static MI_DEV Act sample_bits(uint64_t) { return 0; }
    static MI_DEV Act sample_state(uint64_t, uint64_t) { return 0; }

    static MI_DEV void step(double s[S], uint32_t &, Act action, const EnvParams &P, double &reward, bool &terminated, Trig &) {
        const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.5;
        const double force = 0.001, gravity = 0.0025;
        double position = s[0], velocity = s[1];
        velocity += (double)(action - 1) * force + M::cos_bounded_literals(3 * position) * (-gravity);
        velocity = velocity < -max_speed ? -max_speed : velocity;
        velocity = velocity > max_speed ? max_speed : velocity;
        position += velocity;
        position = position < min_position ? min_position : position;
        position = position > max_position ? max_position : position;
        if (position == min_position && velocity < 0) velocity = 0;
        s[0] = position, s[1] = velocity;
        terminated = position >= goal_position && velocity >= P.p[0];
        reward = -1.0;
    }
};

// ---------------------------------------------------------------------------------------------------------
// MountainCarContinuous-v0: gymnasium/envs/classic_control/continuous_mountain_car.py:116-194
// The state is a float64 array right after reset and a float32 array from the first step on (":178"); NumPy-2
// promotion then makes most of the update float32 arithmetic.
// ---------------------------------------------------------------------------------------------------------
template <class M, class AK = ActF32>
struct MountainCarContinuousT {
    typedef M Math;
    static constexpr bool SPLIT_TERMINAL = false;
    static constexpr int ACT_KIND = AK::KIND;
    static constexpr bool USES_POW = ACT_KIND != MI_F32, USES_POWF = false;  // math.pow(action[0], 2): exact for a float32 value, libm's rounding for a float64 one
    static constexpr int S = 2, OBS = 2;
    static constexpr bool DISCRETE = false;
    static constexpr int ROLLOUT_UNROLL = 1;
    static constexpr bool DUO_ROLLOUT = true;  // engine.hip rollout_duo_kernel (env + aux wavefront per 64 sub-environments): measured +8 %
    static constexpr int DUO_CHUNK = 8;  // steps per phase of the two-role kernel (+4.2 % over 4)
    static constexpr bool DUO_ACT_AHEAD = false;  // (profiles/r06_act_prefetch_ab.txt)
    static constexpr bool REWARD_FROM_TERMINATED = false;
    static MI_DEV double reward_from_terminated(bool, const EnvParams &) { return 0.0; }
    // (the state's float32 / float64 phases and the reward's dependence on the action keep observation, reward and flags on the env role here)
    static constexpr bool AUX_DERIVES_FLAGS = false;
    static constexpr int AUX_F64 = 1, AUX_F32 = 0;
    static MI_DEV void aux_pack(const double *, double *, float *) {}
    static MI_DEV void aux_unpack(const double *, const float *, const EnvParams &, float *, bool &) {}
    static constexpr bool AUX_REWARD = false;
    // The reward reads the unclipped action and the terminated flag (continuous_mountain_car.py:174-176) -- both of which the aux role of the two-role
    // rollout holds itself (it drew the action, the flag comes over with the observation): round 6 takes the reward off the env role, which bounds
    // this kernel (150 k against 99 k cycles of work per launch, profiles/r06_duo_timing.txt), and its float64 ring out of LDS.
    static constexpr bool AUX_REWARD_OF_ACTION = true;
    template <class A>
    static MI_DEV double reward_of_action(bool terminated, A a0) {
        if constexpr (ACT_KIND == MI_F32) {
            const double a_d = (double)a0;
            return (terminated ? 100.0 : 0.0) - (a_d * a_d) * 0.1;  // (step_f32's last line)
        } else {
            return (terminated ? 100.0 : 0.0) - M::sq((double)a0) * 0.1;  // (step_f64's)
        }
    }
    static constexpr int AUX_PRE = 1;
    static constexpr int AUX_PRE_SQUARES = 0;
    static MI_DEV uint32_t aux_pre(const double *, double *) { return 0u; }
    static MI_DEV double aux_square(double x) { return x; }
    typedef typename AK::T Act;
    static MI_DEV double aux_reward(const double *, Act) { return 0.0; }

    static MI_DEV void default_bounds(double &b0, double &b1) { b0 = -0.6, b1 = -0.4; }
    static constexpr int NDRAWS = 1;
    static MI_DEV void reset_u(const double u[NDRAWS], double s[S], uint32_t &flags, double b0, double b1) {
        s[0] = b0 + (b1 - b0) * u[0];
        s[1] = 0.0;
        flags &= ~kStateF32;
    }
    typedef NoTrig Trig;
    static MI_DEV void obs(const double s[S], uint32_t, float o[OBS], Trig &) { o[0] = (float)s[0], o[1] = (float)s[1]; }
    static MI_DEV bool valid(Act) { return true; }
    static MI_DEV Act sample(double u) { return (Act)(float)(-1.0 + (1.0 - (-1.0)) * u); }
    static constexpr bool SAMPLE_FROM_BITS = false;
    static MI_DEV Act sample_bits(uint64_t) { return 0; }
    static MI_DEV Act sample_state(uint64_t, uint64_t) { return 0; }

    static MI_DEV void step(double s[S], uint32_t &flags, Act a0, const EnvParams &P, double &reward, bool &terminated, Trig &t) {
        if constexpr (ACT_KIND == MI_F32)
            step_f32(s, flags, a0, P, reward, terminated, t);
        else
            step_f64(s, flags, a0, P, reward, terminated, t);
    }
    static MI_DEV void step_f32(double s[S], uint32_t &flags, float a0, const EnvParams &P, double &reward, bool &terminated, Trig &) {
        const double min_action = -1.0, max_action = 1.0, min_position = -1.2, max_position = 0.6, max_speed = 0.07;
        const double goal_position = 0.45, power = 0.0015, goal_velocity = P.p[0];
        // force = min(max(action[0], -1.0), 1.0): an np.float32 unless out of range, then the Python float bound
        bool force_is_py = false;
        double force_py = 0.0;
        if (min_action > (double)a0) force_is_py = true, force_py = min_action;
        if (!force_is_py && max_action < (double)a0) force_is_py = true, force_py = max_action;
        double position, velocity;
        if (flags & kStateF32) {
            float p = (float)s[0], v = (float)s[1];
            const float three_p = 3.0f * p;
            const double g = 0.0025 * M::cos_bounded((double)three_p);
            if (force_is_py)
                v = v + (float)(force_py * power - g);
            else
                v = v + ((a0 * (float)power) - (float)g);
            v = v > (float)max_speed ? (float)max_speed : v;
            v = v < (float)-max_speed ? (float)-max_speed : v;
            p = p + v;
            p = p > (float)max_position ? (float)max_position : p;
            p = p < (float)min_position ? (float)min_position : p;
            if (p == (float)min_position && v < 0) v = 0;
            terminated = p >= (float)goal_position && v >= (float)goal_velocity;
            position = p, velocity = v;
        } else {
            double p = s[0], v = s[1];
            const double g = 0.0025 * M::cos_bounded(3 * p);
            if (force_is_py)
                v = v + (force_py * power - g);
            else
                v = v + (double)((a0 * (float)power) - (float)g);
            v = v > max_speed ? max_speed : v;
            v = v < -max_speed ? -max_speed : v;
            p = p + v;
            p = p > max_position ? max_position : p;
            p = p < min_position ? min_position : p;
            if (p == min_position && v < 0) v = 0;
            terminated = p >= goal_position && v >= goal_velocity;
            position = (double)(float)p, velocity = (double)(float)v;
        }
        const double a_d = (double)a0;
        reward = (terminated ? 100.0 : 0.0) - (a_d * a_d) * 0.1;  // math.pow(action[0], 2) * 0.1 (exact in double)
        s[0] = position, s[1] = velocity;
        flags |= kStateF32;
    }
    // A float64 action row (continuous_mountain_car.py:150-178 again): `force` is an np.float64 -- or, out of range and for the rows of a list
    // batch (ActF64Weak), a Python float.  `force * power - 0.0025 * math.cos(3 * position)` is then an np.float64 and `velocity += ...`
    // promotes the np.float32 velocity of a float32 state to float64 for the rest of the step; a Python float leaves it float32.  Each clamp
    // REPLACES the scalar by a Python float (weak again: `position += velocity` with an np.float32 position stays float32), and comparisons
    // with Python floats happen in the array scalar's own type.  vk / pk: 0 = np.float32, 1 = np.float64, 2 = Python float.
    static MI_DEV void step_f64(double s[S], uint32_t &flags, double a0, const EnvParams &P, double &reward, bool &terminated, Trig &) {
        const double min_action = -1.0, max_action = 1.0, min_position = -1.2, max_position = 0.6, max_speed = 0.07;
        const double goal_position = 0.45, power = 0.0015, goal_velocity = P.p[0];
        bool force_is_py = ACT_KIND == MI_F64_WEAK;
        double force = a0;
        if (min_action > a0) force_is_py = true, force = min_action;
        if (force == a0 && max_action < a0) force_is_py = true, force = max_action;
        const bool state_f32 = (flags & kStateF32) != 0;
        double p = s[0], v = s[1];
        const double g = state_f32 ? 0.0025 * M::cos_bounded((double)(3.0f * (float)p)) : 0.0025 * M::cos_bounded(3 * p);
        int vk = state_f32 ? 0 : 1, pk = vk;
        if (force_is_py) {
            if (vk == 0)
                v = (double)((float)v + (float)(force * power - g));
            else
                v = v + (force * power - g);
        } else {
            v = v + (force * power - g), vk = 1;
        }
        if (vk == 0 ? ((float)v > (float)max_speed) : (v > max_speed)) v = max_speed, vk = 2;
        if (vk == 0 ? ((float)v < (float)-max_speed) : (v < -max_speed)) v = -max_speed, vk = 2;
        if (pk == 0 && (vk == 0 || vk == 2))
            p = (double)((float)p + (float)v);
        else
            p = p + v, pk = 1;
        if (pk == 0 ? ((float)p > (float)max_position) : (p > max_position)) p = max_position, pk = 2;
        if (pk == 0 ? ((float)p < (float)min_position) : (p < min_position)) p = min_position, pk = 2;
        const bool at_wall = pk == 0 ? ((float)p == (float)min_position) : (p == min_position);
        if (at_wall && v < 0) v = 0, vk = 2;
        const bool p_ge = pk == 0 ? ((float)p >= (float)goal_position) : (p >= goal_position);
        const bool v_ge = vk == 0 ? ((float)v >= (float)goal_velocity) : (v >= goal_velocity);
        terminated = p_ge && v_ge;
        reward = (terminated ? 100.0 : 0.0) - M::sq(a0) * 0.1;  // math.pow(action[0], 2) * 0.1: libm pow, not the correctly rounded a0 * a0
        s[0] = (double)(float)p, s[1] = (double)(float)v;       // np.array([position, velocity], dtype=np.float32)
        flags |= kStateF32;
    }
};

typedef CartPoleT<ExactMath> CartPole;
typedef PendulumT<ExactMath> Pendulum;
typedef AcrobotT<ExactMathBuiltinFma> Acrobot;
typedef MountainCarT<ExactMath> MountainCar;
typedef MountainCarContinuousT<ExactMath> MountainCarContinuous;

}  // namespace mi
