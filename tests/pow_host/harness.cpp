// TEST INFRASTRUCTURE: gymnasium_amd/csrc/pow_exact.h compiled for the host (g++ -mfma -ffp-contract=off), compared with the running libm
// by tests/test_pow_exact.py.
#include "../../gymnasium_amd/csrc/pow_exact.h"

#include <string.h>
static volatile double g_two = 2.0;  // keeps the compiler from folding pow(x, 2.0) into x * x
static inline uint64_t xs128(uint64_t s[2]) {
    uint64_t a = s[0], b = s[1];
    s[0] = b, a ^= a << 23, s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
}
// one random argument; mode 0: [-8, 8), 1: [-0.1, 0.1), 2: [-30, 30), 3: log-uniform 2^-60 .. 2^60, 4: next to 1 and sqrt(2) 2^k (hi near a power of two),
// 5: log-uniform 2^-100 .. 2^100 (beyond the fast path's range on both sides)
static inline double draw(uint64_t s[2], int mode) {
    const uint64_t u = xs128(s);
    const double f = (double)(u >> 11) * 0x1p-53;
    double x;
    switch (mode) {
    case 0: return f * 16.0 - 8.0;
    case 1: return f * 0.2 - 0.1;
    case 2: return f * 60.0 - 30.0;
    case 3: case 5: {
        const int span = mode == 3 ? 120 : 200;
        const uint64_t e = (u >> 52) % span + 1023 - span / 2, b = (u & 0x800fffffffffffffull) | (e << 52);
        memcpy(&x, &b, 8);
        return x;
    }
    default: {
        const double base = (u & 1) ? 1.0 : 1.4142135623730951;
        const int k = (int)((u >> 1) % 9) - 4;
        uint64_t b;
        x = __builtin_ldexp(base, k);
        memcpy(&b, &x, 8);
        b += (int64_t)((u >> 8) % 4097) - 2048;  // +- 2048 ulps around
        memcpy(&x, &b, 8);
        return x;
    }
    }
}
extern "C" {
__attribute__((visibility("default"))) void square3_batch(const double *x, double *out, long n3) {
    for (long i = 0; i < n3; i++)
        mi_pow::square3(mi_pow::kLogTab, mi_pow::kExpTab, x[3 * i], x[3 * i + 1], x[3 * i + 2], out[3 * i], out[3 * i + 1], out[3 * i + 2]);
}
__attribute__((visibility("default"))) void square2_batch(const double *x, double *out, long n2) {
    for (long i = 0; i < n2; i++) mi_pow::square2(mi_pow::kLogTab, mi_pow::kExpTab, x[2 * i], x[2 * i + 1], out[2 * i], out[2 * i + 1]);
}
__attribute__((visibility("default"))) void plain_batch(const double *x, unsigned char *out, long n) {
    double hi;
    for (long i = 0; i < n; i++) out[i] = mi_pow::square_is_plain(x[i], hi);
}
// Brute force against the running libm, all in C: n random arguments of the given kind; `band` > 0 keeps only arguments whose exact square lies
// within `band` ulp of a rounding boundary (importance sampling of the only region where pow(x, 2.0) != x * x can happen).
// out[0] = arguments examined, out[1] = passed square_is_plain, out[2] = passed but pow(x, 2.0) != x * x (must be 0), out[3] = pow != x * x at all,
// out[4] = square3 results that differ from libm (must be 0); *closest = the smallest distance to the boundary (in ulp) among the pow != x * x cases.
__attribute__((visibility("default"))) void square_brute(uint64_t seed, long n, int mode, double band, long *out, double *closest) {
    uint64_t s[2] = {seed * 0x9E3779B97F4A7C15ull + 1, seed ^ 0xD1B54A32D192ED03ull};
    for (int i = 0; i < 8; i++) xs128(s);
    double trip[3], far = 0.0;
    int nt = 0;
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    for (long i = 0; i < n; i++) {
        const double x = draw(s, mode);
        const double hi = x * x, lo = __builtin_fma(x, x, -hi);
        int e;
        frexp(hi, &e);
        const double dist = 0.5 - fabs(lo) / ldexp(1.0, e - 53);  // distance of the exact square from the rounding boundary, in ulp(hi)
        if (band > 0 && !(dist < band)) continue;
        out[0]++;
        double h2;
        const bool plain = mi_pow::square_is_plain(x, h2);
        const double ref = pow(x, g_two);
        out[1] += plain;
        if (ref != hi) {
            out[3]++;
            if (dist > far) far = dist;
            if (plain) out[2]++;
        }
        trip[nt++] = x;
        if (nt == 3) {
            double r[3];
            mi_pow::square3(mi_pow::kLogTab, mi_pow::kExpTab, trip[0], trip[1], trip[2], r[0], r[1], r[2]);
            for (int k = 0; k < 3; k++) {
                const double want = pow(trip[k], g_two);
                out[4] += memcmp(&want, &r[k], 8) != 0;
            }
            nt = 0;
        }
    }
    *closest = far;
}
__attribute__((visibility("default"))) void square_batch(const double *x, double *out, long n) {
    for (long i = 0; i < n; i++) out[i] = mi_pow::square(mi_pow::kLogTab, mi_pow::kExpTab, x[i]);
}
// Every float with bit pattern in [first, last): out[0] = examined, out[1] = passed squaref_is_plain, out[2] = passed but powf(x, 2.0f) != x * x (must be 0),
// out[3] = powf != x * x at all, out[4] = squaref() results with a normal square below 2^126 that differ from libm (must be 0); *closest = the largest distance to the rounding boundary
// (in ulp) among the powf != x * x cases with a normal square.
static volatile float g_twof = 2.0f;
__attribute__((visibility("default"))) void squaref_scan(uint32_t first, uint32_t last, long *out, double *closest) {
    double far = 0.0;
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    for (uint64_t b = first; b < last; b++) {
        const uint32_t u = (uint32_t)b;
        float x, h2;
        memcpy(&x, &u, 4);
        const float ref = powf(x, g_twof), hi = x * x, own = mi_pow::squaref(mi_pow::kLog2fTab, mi_pow::kExp2fTab, x);
        const bool plain = mi_pow::squaref_is_plain(x, h2);
        out[0]++, out[1] += plain;
        out[4] += memcmp(&ref, &own, 4) != 0 && hi >= 0x1p-126f && hi < 0x1p126f;  // (for |2 log2 x| >= 126 the routine returns x * x where libm still runs exp2: ties among subnormals, the last two binades -- no environment squares such a float)
        if (plain && memcmp(&h2, &hi, 4) != 0) out[2]++;
        if (memcmp(&ref, &hi, 4) != 0 && !(ref != ref)) {
            out[3]++;
            if (plain) out[2]++;
            if (hi >= 0x1p-126f && hi < __builtin_inff()) {
                const double s = (double)x * (double)x;
                int e;
                frexp((double)hi, &e);
                const double dist = 0.5 - fabs(s - (double)hi) / ldexp(1.0, e - 24);
                if (dist > far) far = dist;
            }
        }
    }
    *closest = far;
}
__attribute__((visibility("default"))) void squaref_batch(const float *x, float *out, long n) {
    for (long i = 0; i < n; i++) out[i] = mi_pow::squaref(mi_pow::kLog2fTab, mi_pow::kExp2fTab, x[i]);
}
}
