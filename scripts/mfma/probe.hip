// Empirical operand map of v_mfma_f64_4x4x4_4b_f64: for every lane la, A = one-hot(la), B[l] = l + 1 -> out[o] = lb + 1 for the B lane paired with la in output o.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; la++) out[la * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, (double)(l + 1), 0.0, 0, 0, 0);
}
int main() {
    static double ho[64 * 64];
    double *dout;
    hipMalloc(&dout, sizeof(ho));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d ->", la);
        for (int o = 0; o < 64; o++)
            if (ho[la * 64 + o] != 0) printf("  out %2d <- B lane %2d", o, (int)ho[la * 64 + o] - 1);
        printf("\n");
    }
    return 0;
}
