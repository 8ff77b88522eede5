// TEST INFRASTRUCTURE: gymnasium_amd/csrc/sincos_exact.h compiled for the host (g++ -mfma -ffp-contract=off: the header spells out every
// fused multiply-add), so that tests/test_sincos_exact.py can compare it with the running libm on millions of arguments.
#include "../../gymnasium_amd/csrc/sincos_exact.h"

extern "C" {
__attribute__((visibility("default"))) void sin_exact_batch(const double *x, double *out, long n) {
    for (long i = 0; i < n; i++) out[i] = mi_sincos::sin_exact(mi_sincos::kTable, x[i]);
}
__attribute__((visibility("default"))) void cos_exact_batch(const double *x, double *out, long n) {
    for (long i = 0; i < n; i++) out[i] = mi_sincos::cos_exact(mi_sincos::kTable, x[i]);
}
static const double *table6() {
    static double t6[660];
    static bool done = false;
    if (!done) {
        for (int e = 0; e < 110; e++) mi_sincos::expand6(mi_sincos::kTable, t6, e);
        done = true;
    }
    return t6;
}
// the branch-free forms the kernels call
__attribute__((visibility("default"))) void sin_bf_batch(const double *x, double *out, long n) {
    const double *t6 = table6();
    for (long i = 0; i < n; i++) out[i] = mi_sincos::sin_bf(t6, x[i]);
}
__attribute__((visibility("default"))) void cos_bf_batch(const double *x, double *out, long n) {
    const double *t6 = table6();
    for (long i = 0; i < n; i++) out[i] = mi_sincos::cos_bf(t6, x[i]);
}
__attribute__((visibility("default"))) void sincos_bf_batch(const double *x, double *s, double *c, long n) {
    const double *t6 = table6();
    for (long i = 0; i < n; i++) mi_sincos::sincos_bf(t6, x[i], s[i], c[i]);
}
// the one-reduction sin + cos of any range (Pendulum / Acrobot observations and dynamics)
__attribute__((visibility("default"))) void sincos_pair_batch(const double *x, double *s, double *c, long n) {
    const double *t6 = table6();
    for (long i = 0; i < n; i++) mi_sincos::sincos_pair<true>(t6, x[i], s[i], c[i]);
}
// ... with the reduction's and the polynomials' constants read from behind the table (HOT: what the Acrobot kernels instantiate)
static const double *table6_hot() {
    static double t6[mi_sincos::kHotAt + mi_sincos::kHotCount];
    static bool done = false;
    if (!done) {
        for (int e = 0; e < 110; e++) mi_sincos::expand6(mi_sincos::kTable, t6, e);
        mi_sincos::fill_hot(t6);
        done = true;
    }
    return t6;
}
__attribute__((visibility("default"))) void sin_bf_hot_batch(const double *x, double *out, long n) {
    const double *t6 = table6_hot();
    for (long i = 0; i < n; i++) out[i] = mi_sincos::sin_bf<true, false, true>(t6, x[i]);
}
__attribute__((visibility("default"))) void cos_bf_hot_batch(const double *x, double *out, long n) {
    const double *t6 = table6_hot();
    for (long i = 0; i < n; i++) out[i] = mi_sincos::cos_bf<true, false, true>(t6, x[i]);
}
__attribute__((visibility("default"))) void sincos_pair_hot_batch(const double *x, double *s, double *c, long n) {
    const double *t6 = table6_hot();
    for (long i = 0; i < n; i++) mi_sincos::sincos_pair<true, false, true>(t6, x[i], s[i], c[i]);
}
struct TwoPi {
    static constexpr double value = 6.283185307179586;
};
__attribute__((visibility("default"))) void fmod_2pi_batch(const double *x, double *out, long n) {
    for (long i = 0; i < n; i++) out[i] = mi_sincos::fmod_const(x[i], TwoPi());
}
__attribute__((visibility("default"))) void table_copy(double *out) {
    for (int i = 0; i < 440; i++) out[i] = mi_sincos::kTable[i];
}
}
