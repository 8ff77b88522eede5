// wrappers_internal.h -- what wrappers.hip (the stand-alone wrapper passes) and engine.hip (the same wrappers as the output stage of the step
// kernel) share: the statistics handle and RunningMeanStd's update in the dtype NumPy computes it in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct mi_running_stats {
    int device, dim, dtype;  // dtype of the running mean / var: MI_F32 or MI_F64 (what NumPy's promotion gives in the reference)
    double *mean, *var;      // [dim] device, values always representable in `dtype`
    double *count;           // [1] device
    double *mean2, *var2, *count2;  // a second buffer set: the step epilogue (engine.hip) writes the updated statistics there while the
                                    // other workgroups still read the first, then the host swaps the two sets
    double *partial;         // [2][kPartials] device scratch
    int *flag;               // [1] device: number of rows of the last update (0 = the update was skipped)
};

namespace mi_wrap {

template <class T>
__device__ __forceinline__ double rd(double x) {  // round to the dtype NumPy holds the statistic in
    return (double)(T)x;
}

// RunningMeanStd.update -> update_mean_var_count_from_moments (gymnasium/wrappers/utils.py:43-71) for one column, from the float64 sums
// s1 = sum (x - mean), s2 = sum (x - mean)^2 over `rows` rows (shifted by the running mean for accuracy).  X = dtype of the batch
// (np.mean / np.var return it), T = dtype the running statistics are updated in; every operation is rounded to T where NumPy computes in T
// (count and batch_count are Python scalars).  Does not touch the count.
template <class T, class X>
__device__ __forceinline__ void update_column(double &mean, double &var, double count, double s1, double s2, double rows) {
    const double m1 = s1 / rows;
    const double batch_mean = rd<X>(mean + m1), batch_var = rd<X>(fmax(s2 / rows - m1 * m1, 0.0));
    const double tot = count + rows;
    const double delta = rd<T>(batch_mean - mean);
    const double new_mean = rd<T>(mean + rd<T>(rd<T>(delta * rows) / tot));
    const double m_a = rd<T>(var * count), m_b = rd<T>(batch_var * rows);
    const double M2 = rd<T>(rd<T>(m_a + m_b) + rd<T>(rd<T>(rd<T>(rd<T>(delta * delta) * count) * rows) / tot));
    mean = new_mean, var = rd<T>(M2 / tot);
}

}  // namespace mi_wrap
