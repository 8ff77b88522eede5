// wrappers.hip -- the reference's stateful vector wrappers as device epilogues of the step path.
//
// What it replaces (gymnasium v1.4.0; all NumPy passes over the (N, ...) batch on one host core in the reference):
//   gymnasium/wrappers/utils.py:33-71                     RunningMeanStd.update / update_mean_var_count_from_moments
//   gymnasium/wrappers/vector/stateful_observation.py     NormalizeObservation.observations: (obs - mean) / sqrt(var + eps)
//   gymnasium/wrappers/vector/stateful_reward.py:140-176  NormalizeReward.step: discounted return per env, its running variance
//   gymnasium/wrappers/vector/vectorize_reward.py:115-151 ClipReward
// The batch never leaves HBM: the statistics of one batch are two column sums (shifted by the running mean, float64) reduced by
// a grid of partial sums + one combine kernel; the normalisation is fused into the pass that follows.  Arithmetic follows the
// reference expression by expression in the dtype NumPy uses there (float32 running statistics for float32 observations,
// float64 otherwise); only the batch mean / variance themselves are computed more accurately than the reference's float32 sums,
// which is why parity of these wrappers is stated as a tolerance (tests/test_gpu_wrappers.py), not bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <new>

#include "../../include/mi355env.h"
#include "wrappers_internal.h"

namespace mi_internal {
int set_error(int code, const char *msg);
}


namespace {
constexpr int kBlock = 256, kMaxGrid = 256;  // 64 Ki partial sums: the one-workgroup-per-column fold costs 85 us with 256 Ki of them

#define W_TRY(expr)                                                                                             \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            char buf[400];                                                                                      \
            snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return mi_internal::set_error(MI_ERR_HIP, buf);                                                     \
        }                                                                                                       \
    } while (0)

using mi_wrap::rd;

// Column sums of (x - shift_c) and (x - shift_c)^2 over the rows selected by `active` (nullptr = all), float64.
// Thread t owns flattened elements t, t + S, t + 2S, ... with S a multiple of `dim`, so it always sees column t % dim and
// consecutive threads read consecutive addresses.
template <class T>
__global__ __launch_bounds__(kBlock) void partial_sums(const T *x, const uint8_t *active, int active_is_done, const double *shift, int N, int dim,
                                                       long S, double *partial) {
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= S) return;
    const int c = (int)(t % dim);
    const double sh = shift[c];
    double s1 = 0, s2 = 0, cnt = 0;
    const long total = (long)N * dim;
    for (long e = t; e < total; e += S) {
        const long row = e / dim;
        if (active && ((active[row] != 0) == (active_is_done != 0))) continue;
        const double v = (double)x[e] - sh;
        s1 += v, s2 += v * v, cnt += 1;
    }
    partial[t] = s1, partial[S + t] = s2, partial[2 * S + t] = cnt;
}

// One block per column: combine the partials, then RunningMeanStd.update_from_moments in the dtype T of the running statistics.
// X = dtype of the batch (np.mean / np.var return it), T = dtype the running statistics are updated in.
template <class T, class X>
__global__ __launch_bounds__(kBlock) void combine_update(const double *partial, long S, int dim, double *mean, double *var, double *count,
                                                         int *rows_out) {
    __shared__ double sh[3][kBlock];
    const int c = blockIdx.x;
    double s1 = 0, s2 = 0, n = 0;
    for (long t = c + (long)threadIdx.x * dim; t < S; t += (long)kBlock * dim) s1 += partial[t], s2 += partial[S + t], n += partial[2 * S + t];
    sh[0][threadIdx.x] = s1, sh[1][threadIdx.x] = s2, sh[2][threadIdx.x] = n;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double rows = sh[2][0];
    if (c == 0) *rows_out = (int)rows;
    if (rows == 0) return;  // `if self._update_running_mean and np.any(active)`
    double m = mean[c], v = var[c];
    mi_wrap::update_column<T, X>(m, v, *count, sh[0][0], sh[1][0], rows);
    mean[c] = m, var[c] = v;
}
__global__ void bump_count(double *count, const int *rows) {
    if (*rows > 0) *count += (double)*rows;
}

// NormalizeObservation.observations: (obs - mean) / np.sqrt(var + epsilon) in the observation dtype, float32 output for float32
// statistics (stateful_observation.py: new_single_space dtype float32)
template <class XT, class T, class O>
__global__ __launch_bounds__(kBlock) void normalize_obs(const XT *x, const double *mean, const double *var, double eps, long total, int dim, O *out) {
    const long e = (long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % dim);
    const T num = (T)((T)x[e] - (T)mean[c]);
    const T den = (T)sqrt((double)(T)((T)var[c] + (T)eps));  // np.sqrt of a T array is correctly rounded in T
    out[e] = (O)(T)(num / den);
}

// NormalizeReward.step, stateful_reward.py:150-176
__global__ __launch_bounds__(kBlock) void accumulate_return(float *acc, const uint8_t *prev_done, const double *reward, const uint8_t *term, int N,
                                                            float gamma, int same_step) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const bool active = same_step || !prev_done[i];
    if (!active) return;
    // float32 array * Python float -> float32; * (1 - terminated) (int64) -> float64; + reward -> float64; stored as float32
    const float a = acc[i] * gamma;
    acc[i] = (float)((double)a * (term[i] ? 0.0 : 1.0) + reward[i]);
}
__global__ __launch_bounds__(kBlock) void finish_reward(float *acc, uint8_t *prev_done, const double *reward, const uint8_t *term,
                                                        const uint8_t *trunc, const double *var, double eps, int N, int same_step, double *out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const uint8_t done = (term[i] || trunc[i]) ? 1 : 0;
    prev_done[i] = done;
    if (same_step && done) acc[i] = 0.0f;
    out[i] = reward[i] / sqrt(var[0] + eps);
}
__global__ __launch_bounds__(kBlock) void clip_reward(const double *r, int N, double lo, double hi, int has_lo, int has_hi, double *out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    double v = r[i];
    if (has_lo) v = v < lo ? lo : v;  // np.clip(reward, min_reward, max_reward)
    if (has_hi) v = v > hi ? hi : v;
    out[i] = v;
}

long stride_for(int N, int dim) {
    long want = (long)kBlock * kMaxGrid;
    const long total = (long)N * dim;
    if (want > total) want = total;
    return ((want + dim - 1) / dim) * dim;
}

// x: [N][dim] device array of dtype xdtype; rows with active/done semantics as in partial_sums
int update_stats(mi_running_stats *s, hipStream_t st, const void *x, int xdtype, const uint8_t *mask, int mask_is_done, int N) {
    const long S = stride_for(N, s->dim);
    const int grid = (int)((S + kBlock - 1) / kBlock);
    if (xdtype == MI_F32)
        hipLaunchKernelGGL(partial_sums<float>, dim3(grid), dim3(kBlock), 0, st, (const float *)x, mask, mask_is_done, s->mean, N, s->dim, S, s->partial);
    else
        hipLaunchKernelGGL(partial_sums<double>, dim3(grid), dim3(kBlock), 0, st, (const double *)x, mask, mask_is_done, s->mean, N, s->dim, S, s->partial);
    if (s->dtype == MI_F32 && xdtype == MI_F32)
        hipLaunchKernelGGL((combine_update<float, float>), dim3(s->dim), dim3(kBlock), 0, st, s->partial, S, s->dim, s->mean, s->var, s->count, s->flag);
    else if (xdtype == MI_F32)
        hipLaunchKernelGGL((combine_update<double, float>), dim3(s->dim), dim3(kBlock), 0, st, s->partial, S, s->dim, s->mean, s->var, s->count, s->flag);
    else  // float64 batch: the statistics are float64 from the first update on (NumPy promotion)
        hipLaunchKernelGGL((combine_update<double, double>), dim3(s->dim), dim3(kBlock), 0, st, s->partial, S, s->dim, s->mean, s->var, s->count, s->flag);
    hipLaunchKernelGGL(bump_count, dim3(1), dim3(1), 0, st, s->count, s->flag);
    W_TRY(hipGetLastError());
    return MI_OK;
}
}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int mi_rms_create(int device, int dim, int dtype, double epsilon, mi_running_stats **out) {
    if (!out || dim < 1 || (dtype != MI_F32 && dtype != MI_F64)) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "bad mi_rms_create argument");
    if (mi_device_count() == 0) return mi_internal::set_error(MI_ERR_NO_DEVICE, "no HIP device visible: the wrappers run on the GPU only");
    W_TRY(hipSetDevice(device));
    mi_running_stats *s = new (std::nothrow) mi_running_stats();
    if (!s) return mi_internal::set_error(MI_ERR_HIP, "out of host memory");
    s->device = device, s->dim = dim, s->dtype = dtype;
    W_TRY(hipMalloc(&s->mean, sizeof(double) * dim));
    W_TRY(hipMalloc(&s->var, sizeof(double) * dim));
    W_TRY(hipMalloc(&s->count, sizeof(double)));
    W_TRY(hipMalloc(&s->mean2, sizeof(double) * dim));
    W_TRY(hipMalloc(&s->var2, sizeof(double) * dim));
    W_TRY(hipMalloc(&s->count2, sizeof(double)));
    W_TRY(hipMalloc(&s->flag, sizeof(int)));
    W_TRY(hipMalloc(&s->partial, sizeof(double) * 3 * ((size_t)kBlock * kMaxGrid + dim)));
    // RunningMeanStd.__init__ (wrappers/utils.py:37-41): mean = 0, var = 1, count = epsilon
    double *h = new double[2 * (size_t)dim + 1];
    for (int k = 0; k < dim; k++) h[k] = 0.0, h[dim + k] = 1.0;
    h[2 * dim] = epsilon;
    W_TRY(hipMemcpy(s->mean, h, sizeof(double) * dim, hipMemcpyHostToDevice));
    W_TRY(hipMemcpy(s->var, h + dim, sizeof(double) * dim, hipMemcpyHostToDevice));
    W_TRY(hipMemcpy(s->count, h + 2 * dim, sizeof(double), hipMemcpyHostToDevice));
    delete[] h;
    *out = s;
    return MI_OK;
}

void mi_rms_destroy(mi_running_stats *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(s->mean), (void)hipFree(s->var), (void)hipFree(s->count), (void)hipFree(s->partial), (void)hipFree(s->flag);
    (void)hipFree(s->mean2), (void)hipFree(s->var2), (void)hipFree(s->count2);
    delete s;
}

int mi_rms_get(mi_running_stats *s, void *hip_stream, double *mean, double *var, double *count) {
    if (!s) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "null statistics handle");
    W_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (mean) W_TRY(hipMemcpyAsync(mean, s->mean, sizeof(double) * s->dim, hipMemcpyDeviceToHost, st));
    if (var) W_TRY(hipMemcpyAsync(var, s->var, sizeof(double) * s->dim, hipMemcpyDeviceToHost, st));
    if (count) W_TRY(hipMemcpyAsync(count, s->count, sizeof(double), hipMemcpyDeviceToHost, st));
    W_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_rms_set(mi_running_stats *s, void *hip_stream, const double *mean, const double *var, const double *count) {
    if (!s) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "null statistics handle");
    W_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (mean) W_TRY(hipMemcpyAsync(s->mean, mean, sizeof(double) * s->dim, hipMemcpyHostToDevice, st));
    if (var) W_TRY(hipMemcpyAsync(s->var, var, sizeof(double) * s->dim, hipMemcpyHostToDevice, st));
    if (count) W_TRY(hipMemcpyAsync(s->count, count, sizeof(double), hipMemcpyHostToDevice, st));
    W_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_normalize_observation(mi_running_stats *s, void *hip_stream, const void *obs, int obs_dtype, int num_rows, double epsilon, int update,
                             void *out) {
    if (!s || !obs || !out || num_rows < 1) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "bad mi_normalize_observation argument");
    if (obs_dtype != MI_F32 && obs_dtype != MI_F64) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "observations must be float32 or float64");
    W_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (update) {
        const int rc = update_stats(s, st, obs, obs_dtype, nullptr, 0, num_rows);
        if (rc) return rc;
    }
    const long total = (long)num_rows * s->dim;
    const dim3 g((unsigned)((total + kBlock - 1) / kBlock)), b(kBlock);
    // arithmetic in the NumPy-promoted dtype of (observation, statistics); the result is cast to float32 (`.astype(np.float32)`)
    if (obs_dtype == MI_F32 && s->dtype == MI_F32)
        hipLaunchKernelGGL((normalize_obs<float, float, float>), g, b, 0, st, (const float *)obs, s->mean, s->var, epsilon, total, s->dim, (float *)out);
    else if (obs_dtype == MI_F32)
        hipLaunchKernelGGL((normalize_obs<float, double, float>), g, b, 0, st, (const float *)obs, s->mean, s->var, epsilon, total, s->dim, (float *)out);
    else
        hipLaunchKernelGGL((normalize_obs<double, double, float>), g, b, 0, st, (const double *)obs, s->mean, s->var, epsilon, total, s->dim, (float *)out);
    W_TRY(hipGetLastError());
    return MI_OK;
}

int mi_normalize_reward(mi_running_stats *return_rms, void *hip_stream, float *accumulated, uint8_t *prev_done, const double *reward,
                        const uint8_t *terminated, const uint8_t *truncated, int num_envs, double gamma, double epsilon, int same_step,
                        int update, double *out) {
    if (!return_rms || return_rms->dim != 1 || !accumulated || !prev_done || !reward || !terminated || !truncated || !out || num_envs < 1)
        return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "bad mi_normalize_reward argument");
    W_TRY(hipSetDevice(return_rms->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 g((unsigned)((num_envs + kBlock - 1) / kBlock)), b(kBlock);
    hipLaunchKernelGGL(accumulate_return, g, b, 0, st, accumulated, prev_done, reward, terminated, num_envs, (float)gamma, same_step);
    if (update) {
        const int rc = update_stats(return_rms, st, accumulated, MI_F32, same_step ? nullptr : prev_done, 1, num_envs);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(finish_reward, g, b, 0, st, accumulated, prev_done, reward, terminated, truncated, return_rms->var, epsilon, num_envs, same_step,
                       out);
    W_TRY(hipGetLastError());
    return MI_OK;
}

int mi_clip_reward(int device, void *hip_stream, const double *reward, int num_envs, const double *min_reward, const double *max_reward, double *out) {
    if (!reward || !out || num_envs < 1) return mi_internal::set_error(MI_ERR_INVALID_ARGUMENT, "bad mi_clip_reward argument");
    W_TRY(hipSetDevice(device));
    const dim3 g((unsigned)((num_envs + kBlock - 1) / kBlock)), b(kBlock);
    hipLaunchKernelGGL(clip_reward, g, b, 0, (hipStream_t)hip_stream, reward, num_envs, min_reward ? *min_reward : 0.0, max_reward ? *max_reward : 0.0,
                       min_reward != nullptr, max_reward != nullptr, out);
    W_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
#pragma GCC visibility pop
