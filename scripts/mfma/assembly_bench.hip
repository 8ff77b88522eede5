// A/B microbenchmark for the MFMA question of the cooperative MuJoCo kernel (docs/mujoco_design.md): the Newton solver's Hessian assembly
//     H = M + sum_c J_c^T W_c J_c        (J_c: 3 x NV contact-frame Jacobian, W_c: symmetric 3 x 3, NV = 14 padded to 16)
// for the FOUR sub-environments of a 16-lane-per-environment wavefront (Ant / HalfCheetah layout), with J_c and W_c on the LDS blackboard
// as mjx_coop.h assemble() publishes them and the result as one Hessian row per dof lane (what the Cholesky consumes):
//   valu: the shipped form -- lane = dof row i, for every contact t = W J_c[:, i], then H[i][j] += t . J_c[:, j] with broadcast LDS reads;
//   mfma: v_mfma_f64_4x4x4_4b_f64, one 4 x 4 output tile of each of the four sub-environments per instruction.  Its operand map (measured,
//         scripts/mfma/probe.hip -> profiles/r02_mfma_probe.txt) is  A[i][k] on lane 16 k + 4 blk + i,  B[k][j] on lane 16 k + 4 blk + j,
//         D[i][j] on lane 16 i + 4 blk + j:  a block's 16 values sit on four lanes of EACH 16-lane row, the k index runs ACROSS the rows
//         (= across the sub-environments of the wavefront), so no operand is where the cooperative kernel keeps it: operands are gathered
//         from LDS per lane, the accumulated tiles go back through LDS to become rows.
// Prints shader cycles per assembly (s_memtime, mean over wavefronts, one wavefront per SIMD like the product) and the largest difference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NV = 14, NP = 16, MAXC = 8, REP = 256;

struct Board {  // one sub-environment
    double J[MAXC][3][NP];
    double W[MAXC][3][3];
    double M[NP][NP];
    double H[NP][NP];
};

__device__ inline void fill(Board *bb, int e, int l16, int nc, unsigned seed) {
    // deterministic pseudo-random inputs, the same for both variants
    auto rnd = [&](unsigned a, unsigned b, unsigned c) {
        unsigned x = seed * 2654435761u ^ (a * 40503u + b * 9973u + c * 131u + (unsigned)e * 7919u);
        x ^= x >> 13, x *= 0x5bd1e995u, x ^= x >> 15;
        return (double)(x & 0xffffff) / 16777216.0 - 0.5;
    };
    for (int c = 0; c < nc; c++) {
        for (int k = 0; k < 3; k++) bb[e].J[c][k][l16] = l16 < NV ? rnd(c, k, l16) : 0.0;
        if (l16 < 9) {
            const int r = l16 / 3, q = l16 % 3;
            bb[e].W[c][r][q] = (r == q ? 2.0 : 0.0) + rnd(c, 100 + (r < q ? r : q), 200 + (r < q ? q : r));  // symmetric
        }
    }
    for (int j = 0; j < NP; j++) bb[e].M[l16][j] = (l16 == j ? 3.0 : 0.0) + 0.01 * rnd(999, l16 < j ? l16 : j, l16 < j ? j : l16);
}

template <bool MFMA>
__global__ __launch_bounds__(64) void bench(int nc, unsigned long long *cycles, double *out) {
    __shared__ Board bb[4];
    const int lane = threadIdx.x, e = lane >> 4, l16 = lane & 15;
    fill(bb, e, l16, nc, 12345u + blockIdx.x);
    __syncthreads();
    double H[NP];
    double check = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REP; rep++) {
        asm volatile("" ::: "memory");
        if (!MFMA) {
#pragma unroll
            for (int j = 0; j < NP; j++) H[j] = bb[e].M[l16][j];
#pragma unroll 1
            for (int c = 0; c < nc; c++) {
                const double j0 = bb[e].J[c][0][l16], j1 = bb[e].J[c][1][l16], j2 = bb[e].J[c][2][l16];
                const double (&W)[3][3] = bb[e].W[c];
                const double t0 = W[0][0] * j0 + W[0][1] * j1 + W[0][2] * j2, t1 = W[1][0] * j0 + W[1][1] * j1 + W[1][2] * j2,
                             t2 = W[2][0] * j0 + W[2][1] * j1 + W[2][2] * j2;
#pragma unroll
                for (int j = 0; j < NP; j++) H[j] += t0 * bb[e].J[c][0][j] + t1 * bb[e].J[c][1][j] + t2 * bb[e].J[c][2][j];
            }
        } else {
            // lane = (k, blk, x): k = lane >> 4 (the reduction index, 3 used), blk = (lane >> 2) & 3 (sub-environment), x = lane & 3
            const int k = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
            double acc[4][4];  // acc[I][Jt]: D[i][j] of tile (I, Jt) of sub-environment blk, with i = lane >> 4, j = lane & 3
#pragma unroll
            for (int I = 0; I < 4; I++)
#pragma unroll
                for (int Jt = 0; Jt < 4; Jt++) acc[I][Jt] = bb[blk].M[4 * I + k][4 * Jt + x];  // C operand: D layout, row index on lane >> 4
#pragma unroll 1
            for (int c = 0; c < nc; c++) {
                double a[4], b[4];
                const double w0 = k < 3 ? bb[blk].W[c][k][0] : 0.0, w1 = k < 3 ? bb[blk].W[c][k][1] : 0.0, w2 = k < 3 ? bb[blk].W[c][k][2] : 0.0;
#pragma unroll
                for (int I = 0; I < 4; I++) a[I] = k < 3 ? bb[blk].J[c][k][4 * I + x] : 0.0;  // A[i][k] = J[k][4 I + i]
#pragma unroll
                for (int Jt = 0; Jt < 4; Jt++)  // B[k][j] = (W J)[k][4 Jt + j]
                    b[Jt] = w0 * bb[blk].J[c][0][4 * Jt + x] + w1 * bb[blk].J[c][1][4 * Jt + x] + w2 * bb[blk].J[c][2][4 * Jt + x];
#pragma unroll
                for (int I = 0; I < 4; I++)
#pragma unroll
                    for (int Jt = 0; Jt < 4; Jt++) acc[I][Jt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[I], b[Jt], acc[I][Jt], 0, 0, 0);
            }
            // tiles -> rows: through the blackboard
#pragma unroll
            for (int I = 0; I < 4; I++)
#pragma unroll
                for (int Jt = 0; Jt < 4; Jt++) bb[blk].H[4 * I + k][4 * Jt + x] = acc[I][Jt];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < NP; j++) H[j] = bb[e].H[l16][j];
        }
#pragma unroll
        for (int j = 0; j < NP; j++) check += H[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    for (int j = 0; j < NP; j++) out[((size_t)blockIdx.x * 64 + lane) * NP + j] = H[j];
    if (check == 1.2345e301) out[0] = check;
}

int main(int argc, char **argv) {
    const int blocks = 1024;
    unsigned long long *dc;
    double *do_[2];
    hipMalloc(&dc, blocks * 8);
    for (int v = 0; v < 2; v++) hipMalloc(&do_[v], (size_t)blocks * 64 * NP * 8);
    std::vector<unsigned long long> hc(blocks);
    std::vector<double> ho[2] = {std::vector<double>((size_t)blocks * 64 * NP), std::vector<double>((size_t)blocks * 64 * NP)};
    printf("Hessian assembly H = M + sum_c J_c^T W_c J_c, NV = %d (padded to %d), 4 sub-environments per wavefront, %d wavefronts, %d repetitions\n", NV, NP, blocks, REP);
    printf("%-10s %16s %16s %10s %14s\n", "contacts", "valu cycles", "mfma cycles", "mfma/valu", "max |diff|");
    for (int nc : {1, 2, 4, 8}) {
        double cyc[2];
        for (int v = 0; v < 2; v++) {
            for (int warm = 0; warm < 2; warm++) {
                if (v == 0)
                    hipLaunchKernelGGL(bench<false>, dim3(blocks), dim3(64), 0, 0, nc, dc, do_[v]);
                else
                    hipLaunchKernelGGL(bench<true>, dim3(blocks), dim3(64), 0, 0, nc, dc, do_[v]);
                hipDeviceSynchronize();
            }
            hipMemcpy(hc.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
            hipMemcpy(ho[v].data(), do_[v], ho[v].size() * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto c : hc) s += (double)c;
            cyc[v] = s / blocks / REP;
        }
        double worst = 0;
        for (size_t i = 0; i < ho[0].size(); i++) {
            const size_t lane = (i / NP) % 64, j = i % NP;
            if ((lane & 15) < NV && j < NV) worst = fmax(worst, fabs(ho[0][i] - ho[1][i]));
        }
        printf("%-10d %16.0f %16.0f %10.3f %14.3e\n", nc, cyc[0], cyc[1], cyc[1] / cyc[0], worst);
    }
    return 0;
}
