// pcg64_dev.h -- NumPy-compatible PCG64 / SeedSequence arithmetic for gfx950 device code (and the host side of the
// library).  One generator per lane; the 128-bit LCG state lives in two 64-bit VGPR pairs, the multiply lowers to
// v_mad_u64_u32 chains.
//
// Reference behaviour reproduced (the arithmetic is NumPy's; the reference only calls it):
//   gymnasium/utils/seeding.py:39-41   Generator(PCG64(SeedSequence(seed)))
//   gymnasium/core.py:157-159          Env.reset(seed=...)
//   numpy PCG64: state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc; output = XSL-RR(new state);
//   random() = (out >> 11) * 2^-53; uniform(lo, hi) = lo + (hi - lo) * random()      
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD inline
#endif

namespace mi {

typedef unsigned __int128 u128;

MI_HD u128 make_u128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | (u128)lo; }
MI_HD u128 pcg_mult() { return make_u128(0x2360ED051FC65DA4ULL, 0x4385DF649FCCF645ULL); }
// Multiplicative inverse of pcg_mult() mod 2^128 (Newton: x <- x (2 - a x) doubles the number of correct low bits).
MI_HD constexpr u128 pcg_mult_inverse() {
    const u128 a = ((u128)0x2360ED051FC65DA4ULL << 64) | (u128)0x4385DF649FCCF645ULL;
    u128 x = a;  // a * a == 1 (mod 8) for odd a: 3 correct bits
    for (int it = 0; it < 7; it++) x = x * ((u128)2 - a * x);
    return x;
}
static_assert(pcg_mult_inverse() * (((u128)0x2360ED051FC65DA4ULL << 64) | (u128)0x4385DF649FCCF645ULL) == (u128)1, "inverse");

// mult * s + plus mod 2^128, limb by limb.  Left to `unsigned __int128` the device compiler zero-extends every partial product into a fresh register pair
// (~29 instructions for a run-time multiplier, 27 for the generator's constant); here the products accumulate in 64-bit multiply-adds whose addends
// cannot overflow where a carry would matter, the one carry that can occur (out of the middle limbs) is the addition's overflow flag, and the top limb's
// four cross products are 32-bit: 19 instructions.  Checked against the __int128 expression on 2e7 random and edge-pattern operands (tests/test_pcg_limbs.py).
//   T0 = a0 m0 + p0                       <= (2^32-1)^2 + 2^32-1 < 2^64
//   T1 = a1 m0 + (hi(T0) + p1) + a0 m1    first sum <= 2^64-1; the second may carry: c1, worth 2^96
//   H  = a2 m0 + a1 m1 + a0 m2 + hi(T1) + (p3:p2)   mod 2^64 (bits 64..127)
//   r3 = hi(H) + lo32(a3 m0 + a2 m1 + a1 m2 + a0 m3) + c1
MI_HD u128 pcg_muladd(u128 s, u128 mult, u128 plus) {
    const uint32_t a0 = (uint32_t)s, a1 = (uint32_t)(s >> 32), a2 = (uint32_t)(s >> 64), a3 = (uint32_t)(s >> 96);
    const uint32_t m0 = (uint32_t)mult, m1 = (uint32_t)(mult >> 32), m2 = (uint32_t)(mult >> 64), m3 = (uint32_t)(mult >> 96);
    const uint32_t p0 = (uint32_t)plus, p1 = (uint32_t)(plus >> 32);
    const uint64_t p23 = (uint64_t)(plus >> 64);
    const uint64_t t0 = (uint64_t)a0 * m0 + p0;
    const uint64_t t1a = (uint64_t)a1 * m0 + ((t0 >> 32) + p1);
    uint64_t t1;
    const bool c1 = __builtin_add_overflow((uint64_t)a0 * m1, t1a, &t1);
    uint64_t h = (uint64_t)a2 * m0 + ((t1 >> 32) + p23);
    h = (uint64_t)a1 * m1 + h;
    h = (uint64_t)a0 * m2 + h;
    const uint32_t x = a3 * m0 + a2 * m1 + a1 * m2 + a0 * m3;
    const uint32_t r3 = (uint32_t)(h >> 32) + x + (c1 ? 1u : 0u);
    return ((u128)r3 << 96) | ((u128)(uint32_t)h << 64) | ((u128)(uint32_t)t1 << 32) | (u128)(uint32_t)t0;
}

struct Pcg64 {
    u128 state;
    u128 inc;

#ifndef MI_PCG_LIMB_STEP  // (A/B builds: -DMI_PCG_LIMB_STEP=0)
#define MI_PCG_LIMB_STEP 1
#endif
    MI_HD void step() { state = MI_PCG_LIMB_STEP ? pcg_muladd(state, pcg_mult(), inc) : state * pcg_mult() + inc; }
    MI_HD void unstep() { state = (state - inc) * pcg_mult_inverse(); }  // exact inverse of step()
    MI_HD uint64_t next64() {
        step();
        uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        uint64_t x = hi ^ lo;
        unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((0u - rot) & 63u));
    }
    MI_HD double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    // Generator.uniform(low, high) = low + (high - low) * next_double; callers pass range = high - low.
    MI_HD double uniform(double low, double range) { return low + range * next_double(); }
    // pcg_setseq_128_srandom_r from SeedSequence.generate_state(4, uint64)
    MI_HD void srandom(const uint64_t w[4]) {
        u128 initstate = make_u128(w[0], w[1]);
        u128 initseq = make_u128(w[2], w[3]);
        state = 0;
        inc = (initseq << 1) | 1u;
        step();
        state += initstate;
        step();
    }
};

// One affine jump of the LCG: s -> mult * s + plus.
struct PcgJump {
    u128 mult;
    u128 plus;
};

// Brown's O(log delta) skip-ahead: the affine map equivalent to `delta` steps of (mult, inc).
MI_HD PcgJump pcg_jump(u128 inc, u128 delta) {
    u128 cur_mult = pcg_mult(), cur_plus = inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    PcgJump j = {acc_mult, acc_plus};
    return j;
}

// SeedSequence(entropy = non-negative int < 2^64).generate_state(4, uint64)  (numpy/random/bit_generator.pyx)
MI_HD uint32_t ss_hashmix(uint32_t value, uint32_t &hash_const) {
    value ^= hash_const;
    hash_const *= 0x931e8875u;
    value *= hash_const;
    value ^= value >> 16;
    return value;
}
MI_HD uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    r ^= r >> 16;
    return r;
}
MI_HD void seed_sequence_words(uint64_t seed, uint64_t out[4]) {
    const uint32_t e0 = (uint32_t)seed, e1 = (uint32_t)(seed >> 32);
    uint32_t pool[4];
    uint32_t hc = 0x43b0d7e5u;
    pool[0] = ss_hashmix(e0, hc);
    pool[1] = ss_hashmix(e1, hc);  // absent high word == 0 gives the same hash as an explicit 0 word
    pool[2] = ss_hashmix(0u, hc);
    pool[3] = ss_hashmix(0u, hc);
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    uint32_t st[8];
    hc = 0x8b51f9ddu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hc;
        hc *= 0x58f38dedu;
        v *= hc;
        v ^= v >> 16;
        st[i] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (uint64_t)st[2 * k] | ((uint64_t)st[2 * k + 1] << 32);
}

}  // namespace mi
