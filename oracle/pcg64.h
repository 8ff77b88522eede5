/*
 * oracle/pcg64.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product.
 *
 * CPU restatement of the random-number path the reference uses for env seeding:
 *   gymnasium/utils/seeding.py:39-41   np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
 *   gymnasium/core.py:157-159          Env.reset(seed=...) re-creates that generator
 * The arithmetic itself lives in NumPy (third party, not vendored in /root/reference; installed 2.2.6):
 *   SeedSequence  = numpy/random/bit_generator.pyx (O'Neill's seed_seq_fe128 variant)
 *   PCG64         = numpy/random/src/pcg64/pcg64.h  (pcg_setseq_128 + XSL-RR 128/64)
 *   uniform/random= numpy/random/src/distributions/distributions.c (next_double, random_uniform)
 * This file restates their published algorithms; it is pinned by tests/golden/rng_golden.npz, which is
 * generated from NumPy itself by tests/golden/make_golden.py (and by SURVEY.md Appendix B).
 */
#ifndef ORACLE_PCG64_H
#define ORACLE_PCG64_H
#include <stdint.h>

typedef unsigned __int128 orc_u128;

typedef struct {
    orc_u128 state;
    orc_u128 inc;
} orc_pcg64;

#define ORC_PCG_MULT ((((orc_u128)0x2360ED051FC65DA4ULL) << 64) | (orc_u128)0x4385DF649FCCF645ULL)

static inline void orc_pcg64_step(orc_pcg64 *r) { r->state = r->state * ORC_PCG_MULT + r->inc; }

/* pcg64_next64: advance, then XSL-RR of the NEW state. */
static inline uint64_t orc_pcg64_next64(orc_pcg64 *r) {
    orc_pcg64_step(r);
    uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63u));
}

/* next_double: 53 random bits scaled by 2^-53. */
static inline double orc_pcg64_double(orc_pcg64 *r) {
    return (double)(orc_pcg64_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

/* Generator.uniform(low, high): low + (high - low) * next_double (random_uniform(lower, range)). */
static inline double orc_pcg64_uniform(orc_pcg64 *r, double low, double high) {
    double range = high - low;
    return low + range * orc_pcg64_double(r);
}

/* pcg64_set_seed + pcg_setseq_128_srandom_r from four uint64 words (SeedSequence.generate_state(4, uint64)). */
static inline void orc_pcg64_srandom(orc_pcg64 *r, const uint64_t w[4]) {
    orc_u128 initstate = ((orc_u128)w[0] << 64) | w[1];
    orc_u128 initseq = ((orc_u128)w[2] << 64) | w[3];
    r->state = 0;
    r->inc = (initseq << 1) | 1u;
    orc_pcg64_step(r);
    r->state += initstate;
    orc_pcg64_step(r);
}

/* ---- SeedSequence(entropy=int).generate_state(4, np.uint64) ------------------------------------------- */
#define ORC_SS_XSHIFT 16
#define ORC_SS_INIT_A 0x43b0d7e5u
#define ORC_SS_MULT_A 0x931e8875u
#define ORC_SS_INIT_B 0x8b51f9ddu
#define ORC_SS_MULT_B 0x58f38dedu
#define ORC_SS_MIX_L 0xca01f9ddu
#define ORC_SS_MIX_R 0x4973f715u

static inline uint32_t orc_ss_hashmix(uint32_t value, uint32_t *hash_const) {
    value ^= *hash_const;
    *hash_const *= ORC_SS_MULT_A;
    value *= *hash_const;
    value ^= value >> ORC_SS_XSHIFT;
    return value;
}

static inline uint32_t orc_ss_mix(uint32_t x, uint32_t y) {
    uint32_t result = ORC_SS_MIX_L * x - ORC_SS_MIX_R * y;
    result ^= result >> ORC_SS_XSHIFT;
    return result;
}

/* entropy: little-endian uint32 words of the (non-negative) integer seed; seed 0 -> one word {0}. */
static inline void orc_seedseq_words(const uint32_t *entropy, int n_entropy, uint64_t out[4]) {
    uint32_t pool[4];
    uint32_t hash_const = ORC_SS_INIT_A;
    for (int i = 0; i < 4; i++) pool[i] = orc_ss_hashmix(i < n_entropy ? entropy[i] : 0u, &hash_const);
    for (int i_src = 0; i_src < 4; i_src++)
        for (int i_dst = 0; i_dst < 4; i_dst++)
            if (i_src != i_dst) pool[i_dst] = orc_ss_mix(pool[i_dst], orc_ss_hashmix(pool[i_src], &hash_const));
    for (int i_src = 4; i_src < n_entropy; i_src++)
        for (int i_dst = 0; i_dst < 4; i_dst++)
            pool[i_dst] = orc_ss_mix(pool[i_dst], orc_ss_hashmix(entropy[i_src], &hash_const));
    /* generate_state(8 x uint32) viewed as 4 x uint64 (little endian) */
    uint32_t st[8];
    hash_const = ORC_SS_INIT_B;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hash_const;
        hash_const *= ORC_SS_MULT_B;
        v *= hash_const;
        v ^= v >> ORC_SS_XSHIFT;
        st[i] = v;
    }
    for (int k = 0; k < 4; k++) out[k] = (uint64_t)st[2 * k] | ((uint64_t)st[2 * k + 1] << 32);
}

static inline void orc_pcg64_seed_u64(orc_pcg64 *r, uint64_t seed) {
    uint32_t e[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t w[4];
    orc_seedseq_words(e, e[1] ? 2 : 1, w);
    orc_pcg64_srandom(r, w);
}
#endif
