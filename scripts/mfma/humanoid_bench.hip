// A/B microbenchmarks for the MFMA question on the Humanoid-sized pieces of the cooperative MuJoCo kernel (DESIGN.md
// section 7): one 32-lane group per sub-environment, two sub-environments per wavefront, one wavefront per SIMD -- the product's layout.
//
//   Part 1  Cholesky factor of the 23 x 23 mass matrix (padded 32), rows in registers, as mjx_coop.h chol_factor_lds ships it:
//     lds    the shipped form: per column the pivot column goes through LDS, every lane updates its own row (23 dependent columns)
//     mfma   right-looking blocked form, block 16: columns 0..15 by the same row sweep but WITHOUT touching A22 (rows / columns 16..22), then
//            A22 -= L21 L21^T as ONE 16 x 16 x 16 product on v_mfma_f64_16x16x4_f64 (4 instructions; the two sub-environments' 7 x 7 blocks
//            are packed block-diagonally into the tile, rows 0..6 and 8..14), tile -> rows through LDS, then columns 16..22 by the row sweep
//     tree   what the structure of M allows instead: M[i][j] != 0 only if j is an ancestor dof of i (or the reverse), so eliminating the dofs
//            leaves-first (MuJoCo's L^T D L order) has no fill-in and the two legs / two arms eliminate side by side: 13 dependent levels
//            instead of 23 columns, 185 instead of 276 multiply-adds per factor
//   Part 2  W = J (M^-1 J^T) for R = 15 constraint rows (5 contacts x 3 frame axes), J: R x 23, B = M^-1 J^T: 23 x R -- the matrix a
//           constraint-space Gauss-Seidel needs:
//     valu   the product's idiom: dof lane i holds J[:, i] and B[i][:]; every entry of W is a group reduction (R (R + 1) / 2 of them)
//     mfma   six v_mfma_f64_16x16x4_f64 (K = 23 padded to 24) per sub-environment, operands gathered from the blackboard
// Operand map of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): A[i][k] on lane i + 16 k, B[k][j] on lane j + 16 k, one double each;
// D[row][col]: col = lane & 15, row = (lane >> 4) + 4 reg.  Both parts check the MFMA result against the VALU one.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma/humanoid_bench.hip -o gpurun_out/humanoid_mfma_bench && gpurun_out/humanoid_mfma_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NV = 23, G = 32, REP = 64, R = 15;
typedef double v4d __attribute__((ext_vector_type(4)));

// Humanoid-v5 dof tree (gymnasium/envs/mujoco/assets/humanoid.xml): parent dof of every dof
__host__ __device__ constexpr int parent_of(int i) {
    constexpr int p[NV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 8, 13, 14, 15, 5, 17, 18, 5, 20, 21};
    return p[i];
}
__host__ __device__ constexpr bool is_ancestor(int a, int i) {  // a is i or an ancestor of i
    while (i >= 0) {
        if (i == a) return true;
        i = parent_of(i);
    }
    return false;
}
__host__ __device__ constexpr int depth_of(int i) {
    int d = 0;
    while (parent_of(i) >= 0) i = parent_of(i), d++;
    return d;
}
constexpr int kMaxDepth = 12;

struct Board {
    double Mt[NV][NV];   // M, column-wise like the product (lane i reads Mt[j][i])
    double col[2][NV];   // pivot column exchange
    double L[G][G];      // scratch: rows of L / tiles
    double S[16][17];    // MFMA tile -> rows
    double J[R + 1][24], Bm[24][R + 1], W[16][17];
};

__device__ inline void sync_group() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline double rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * 0.5, e, y);
}
__device__ inline double rnd(unsigned seed, unsigned a, unsigned b) {
    unsigned x = seed * 2654435761u ^ (a * 40503u + b * 9973u);
    x ^= x >> 13, x *= 0x5bd1e995u, x ^= x >> 15;
    return (double)(x & 0xffffff) / 16777216.0 - 0.5;
}
// an SPD matrix with the Humanoid's sparsity: M = sum over dofs k of v_k v_k^T with v_k supported on the ancestors of k, plus a diagonal
__device__ inline void fill(Board &bb, int lane, unsigned seed) {
    if (lane < NV) {
        for (int j = 0; j < NV; j++) {
            double m = (lane == j) ? 2.0 : 0.0;
            if (is_ancestor(lane, j) || is_ancestor(j, lane)) {
                for (int k = 0; k < NV; k++)
                    if (is_ancestor(lane, k) && is_ancestor(j, k)) m += (0.6 + rnd(seed, k, lane)) * (0.6 + rnd(seed, k, j)) * 0.3;
            }
            bb.Mt[j][lane] = m;
        }
    }
    for (int r = lane; r < R + 1; r += G)
        for (int k = 0; k < 24; k++) bb.J[r][k] = (r < R && k < NV) ? rnd(seed + 7, r, k) : 0.0, bb.Bm[k][r] = (r < R && k < NV) ? rnd(seed + 9, k, r) : 0.0;
    sync_group();
}

// ---- Part 1 ---------------------------------------------------------------------------------------------------------------------
// columns [K0, K1) of the row sweep; JMAX: the row update stops below this column (the blocked form leaves A22 alone in the first block)
template <int K0, int K1, int JMAX>
__device__ inline void sweep(Board &bb, double *A, double &idiag, int lane) {
#pragma unroll
    for (int k = K0; k < K1; k++) {
        double (&col)[NV] = bb.col[k & 1];
        if (lane >= k && lane < NV) col[lane] = A[k];
        sync_group();
        if (lane >= k && lane < NV) {
            double piv = col[k];
            piv = piv < 1e-15 ? 1e-15 : piv;
            const double inv = rsq(piv);
            const double lik = A[k] * inv;
            A[k] = lik;
            if (lane == k) idiag = inv;
            const double t = lik * inv;
#pragma unroll
            for (int j = k + 1; j < JMAX; j++) A[j] -= (j <= lane ? t : 0.0) * col[j];
        }
    }
}
__device__ inline void factor_lds(Board &bb, double *A, double &idiag, int lane) { sweep<0, NV, NV>(bb, A, idiag, lane); }

// bb0: the blackboard of the wavefront's first sub-environment (the tile packs both)
__device__ inline void factor_mfma(Board *boards, int grp, double *A, double &idiag, int lane) {
    Board &bb = boards[grp];
    sweep<0, 16, 16>(bb, A, idiag, lane);
    // rows 16..22 now hold L21 in A[0..15]; publish them, form the 7 x 7 Schur updates of both sub-environments in one tile
    if (lane >= 16 && lane < NV) {
#pragma unroll
        for (int k = 0; k < 16; k++) bb.L[lane - 16][k] = A[k];
    }
    sync_group();
    const int wl = threadIdx.x & 63, i = wl & 15, kk = wl >> 4;
    v4d acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const double a = (i & 7) < 7 ? boards[i >> 3].L[i & 7][4 * t + kk] : 0.0;  // A[i][k] = B[k][i] = L21 of sub-environment i / 8, row i % 8
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) boards[0].S[kk + 4 * rg][i] = acc[rg];
    sync_group();
    if (lane >= 16 && lane < NV) {
#pragma unroll
        for (int j = 16; j < NV; j++) A[j] -= (j <= lane) ? boards[0].S[8 * grp + lane - 16][8 * grp + j - 16] : 0.0;
    }
    sweep<16, NV, NV>(bb, A, idiag, lane);
}

// the blocked form with the Schur update on the VALU, as fused multiply-add chains over k = 0..15 starting from zero: if v_mfma_f64_16x16x4_f64
// accumulates its four products in k order into the C operand with one rounding each, this is bit-identical to factor_mfma -- what a
// fallback for a partially active wavefront (one sub-environment in its autoreset step) would have to compute to keep trajectories
// independent of the neighbour in the wavefront
__device__ inline void factor_chain(Board *boards, int grp, double *A, double &idiag, int lane) {
    Board &bb = boards[grp];
    sweep<0, 16, 16>(bb, A, idiag, lane);
    if (lane >= 16 && lane < NV) {
#pragma unroll
        for (int k = 0; k < 16; k++) bb.L[lane - 16][k] = A[k];
    }
    sync_group();
    if (lane >= 16 && lane < NV) {
#pragma unroll
        for (int j = 16; j < NV; j++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) acc = fma(bb.L[lane - 16][k], bb.L[j - 16][k], acc);
            A[j] -= (j <= lane) ? acc : 0.0;
        }
    }
    sweep<16, NV, NV>(bb, A, idiag, lane);
}

// leaves-first elimination without fill-in (MuJoCo's mj_factorM order): afterwards A[j] (j a proper ancestor of lane) = L[lane][j] of
// M = L^T D L with unit-diagonal L, and A[lane] = D[lane].  depth / descendants of the lane come from per-lane constants (tables in the
// real kernel); which dofs sit on a level and who their ancestors are is compile-time (template recursion: K, J are constants).
template <int K, int J>
__device__ inline void tree_update(const Board &bb, double *A, double lki, int lane) {  // A[j] -= l_ki M[k][j] over the ancestors j of k
    if constexpr (J < K) {
        if constexpr (is_ancestor(J, K)) A[J] -= lki * bb.L[K][J];  // (lki = 0 unless this lane is an ancestor of K; entries J > lane of a row are never read)
        tree_update<K, J + 1>(bb, A, lki, lane);
    }
}
template <int LEVEL, int K>
__device__ inline void tree_eliminate(const Board &bb, double *A, int lane, unsigned my_desc) {  // every dof K of this level
    if constexpr (K < NV) {
        if constexpr (depth_of(K) == LEVEL) {
            const bool mine = (my_desc >> K) & 1u;
            const double lki = mine ? bb.L[K][lane < NV ? lane : 0] / bb.L[K][K] : 0.0;
            tree_update<K, 0>(bb, A, lki, lane);
        }
        tree_eliminate<LEVEL, K + 1>(bb, A, lane, my_desc);
    }
}
template <int LEVEL>
__device__ inline void tree_level(Board &bb, double *A, int lane, int my_depth, unsigned my_desc) {
    const bool final_row = lane < NV && my_depth == LEVEL;
    if (final_row) {  // the rows of this level have received every update: publish them (unscaled), then scale them to L
#pragma unroll
        for (int j = 0; j < NV; j++) bb.L[lane][j] = A[j];  // (uncompressed row: a row of depth d has d + 1 entries that matter)
    }
    sync_group();
    if (final_row) {
        const double inv = 1.0 / bb.L[lane][lane];
#pragma unroll
        for (int j = 0; j < NV; j++) A[j] *= (j < lane) ? inv : 1.0;
    }
    tree_eliminate<LEVEL, 0>(bb, A, lane, my_desc);
    if constexpr (LEVEL > 0) tree_level<LEVEL - 1>(bb, A, lane, my_depth, my_desc);
}
__device__ inline void factor_tree(Board &bb, double *A, double &idiag, int lane, int my_depth, unsigned my_desc) {
    tree_level<kMaxDepth>(bb, A, lane, my_depth, my_desc);
    idiag = A[0];
}

// ---- Part 2 ---------------------------------------------------------------------------------------------------------------------
// the product's group reduction (mjx_coop.h group_sum<32>): v_permlane16_swap folds the two 16-lane rows, then DPP row rotations
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double group_sum(double v) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    v = v + dpp_mov<0x128>(v);
    v = v + dpp_mov<0x124>(v);
    v = v + dpp_mov<0x122>(v);
    v = v + dpp_mov<0x121>(v);
    return v;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void bench_factor(unsigned long long *cycles, double *out) {
    __shared__ Board boards[2];
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    Board &bb = boards[grp];
    fill(bb, lane, 1234u + blockIdx.x * 2 + grp);
    double A[NV], idiag = 0, check = 0;
    int my_depth = 0;
    unsigned my_desc = 0;
    for (int k = 0; k < NV; k++) {
        if (lane == k) my_depth = depth_of(k);
        if (lane < k && is_ancestor(lane, k)) my_desc |= 1u << k;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REP; rep++) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NV; j++) A[j] = bb.Mt[j][lane < NV ? lane : 0];
        if (VARIANT == 0) factor_lds(bb, A, idiag, lane);
        if (VARIANT == 1) factor_mfma(boards, grp, A, idiag, lane);
        if (VARIANT == 2) factor_tree(bb, A, idiag, lane, my_depth, my_desc);
        if (VARIANT == 3) factor_chain(boards, grp, A, idiag, lane);
        check += A[3] + idiag;
        sync_group();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x] = t1 - t0;
    if (lane < NV)
        for (int j = 0; j < NV; j++) out[((size_t)blockIdx.x * 2 + grp) * NV * NV + lane * NV + j] = j <= lane ? A[j] : 0.0;
    if (check == 1.2345e301) out[0] = check;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void bench_w(unsigned long long *cycles, double *out) {
    __shared__ Board boards[2];
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    Board &bb = boards[grp];
    fill(bb, lane, 4321u + blockIdx.x * 2 + grp);
    double check = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REP; rep++) {
        asm volatile("" ::: "memory");
        if (VARIANT == 0) {
            // dof lane i: its column of J and its row of B in registers (as the product keeps jcol / b), every W entry a group reduction;
            // W is symmetric only for B = M^-1 J^T with the true M^-1: the full R x R product is formed, as the MFMA variant does
            double jc[R], bi[R];
#pragma unroll
            for (int r = 0; r < R; r++) jc[r] = lane < NV ? bb.J[r][lane] : 0.0, bi[r] = lane < NV ? bb.Bm[lane][r] : 0.0;
#pragma unroll 1
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int s = 0; s < R; s++) {
                    if (s > r) continue;  // W is symmetric for the true M^-1: the product forms the lower triangle (120 reductions)
                    const double w = group_sum(jc[r] * bi[s]);
                    if (lane == 0) bb.W[r][s] = w, bb.W[s][r] = w;
                }
            }
        } else {
            const int wl = threadIdx.x & 63, i = wl & 15, kk = wl >> 4;
#pragma unroll
            for (int e = 0; e < 2; e++) {  // one 16 x 16 x 24 product per sub-environment
                v4d acc = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; t++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(boards[e].J[i][4 * t + kk], boards[e].Bm[4 * t + kk][i], acc, 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; rg++) boards[e].W[kk + 4 * rg][i] = acc[rg];
            }
        }
        sync_group();
        check += bb.W[lane & 7][3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x] = t1 - t0;
    for (int r = lane; r < R; r += G)
        for (int s = 0; s < R; s++) out[((size_t)blockIdx.x * 2 + grp) * R * R + r * R + s] = bb.W[r][s];
    if (check == 1.2345e301) out[0] = check;
}

template <class K>
double run(K kernel, int blocks, unsigned long long *dc, double *dout, std::vector<double> &host) {
    std::vector<unsigned long long> hc(blocks);
    for (int warm = 0; warm < 2; warm++) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, dc, dout);
        hipDeviceSynchronize();
    }
    hipMemcpy(hc.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(host.data(), dout, host.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto c : hc) s += (double)c;
    return s / blocks / REP;
}

int main() {
    const int blocks = 1024;
    unsigned long long *dc;
    double *dout;
    hipMalloc(&dc, blocks * 8);
    const size_t nf = (size_t)blocks * 2 * NV * NV, nw = (size_t)blocks * 2 * R * R;
    hipMalloc(&dout, (nf > nw ? nf : nw) * 8);
    std::vector<double> f0(nf), f1(nf), f2(nf), f3(nf), w0(nw), w1(nw);
    printf("Humanoid-sized MFMA A/B (NV = %d, 2 sub-environments per wavefront, %d wavefronts = one per SIMD, %d repetitions); shader cycles per call and wavefront\n", NV, blocks, REP);
    const double c0 = run(bench_factor<0>, blocks, dc, dout, f0), c1 = run(bench_factor<1>, blocks, dc, dout, f1), c2 = run(bench_factor<2>, blocks, dc, dout, f2);
    const double c3 = run(bench_factor<3>, blocks, dc, dout, f3);
    double d01 = 0, d13 = 0;
    size_t nbits = 0;
    for (size_t k = 0; k < nf; k++) d01 = fmax(d01, fabs(f0[k] - f1[k])), d13 = fmax(d13, fabs(f1[k] - f3[k])), nbits += f1[k] != f3[k];
    // tree form: check L^T D L == M on the host against the LL^T factor: compare the products
    double dtree = 0;
    for (int b = 0; b < 4; b++) {
        const double *Lc = &f0[(size_t)b * NV * NV], *Lt = &f2[(size_t)b * NV * NV];
        for (int i = 0; i < NV; i++)
            for (int j = 0; j <= i; j++) {
                double mc = 0, mt = 0;
                for (int k = 0; k <= j; k++) mc += Lc[i * NV + k] * Lc[j * NV + k];  // L L^T
                for (int k = i; k < NV; k++) {  // L^T D L with unit-diagonal L, D = Lt[k][k]
                    const double lki = k == i ? 1.0 : Lt[k * NV + i], lkj = k == j ? 1.0 : Lt[k * NV + j];
                    mt += lki * Lt[k * NV + k] * lkj;
                }
                dtree = fmax(dtree, fabs(mc - mt));
            }
    }
    printf("Part 1  Cholesky factor of M (23 x 23)\n");
    printf("  %-44s %10.0f cycles\n", "lds   (shipped: 23 dependent columns)", c0);
    printf("  %-44s %10.0f cycles   x%.3f of lds   max |L diff| %.2e\n", "mfma  (block 16, Schur update on MFMA)", c1, c1 / c0, d01);
    printf("  %-44s %10.0f cycles   x%.3f of lds   max |L^T D L - L L^T| %.2e\n", "tree  (leaves first, 13 levels, no fill-in)", c2, c2 / c0, dtree);
    printf("  %-44s %10.0f cycles   x%.3f of lds   vs mfma: max |L diff| %.2e, %zu of %zu entries differ in any bit\n", "chain (block 16, Schur update as VALU fma chains)", c3, c3 / c0, d13, nbits, nf);
    const double v0 = run(bench_w<0>, blocks, dc, dout, w0), v1 = run(bench_w<1>, blocks, dc, dout, w1);
    double dw = 0;
    for (size_t k = 0; k < nw; k++) {  // (the random B of the benchmark is not M^-1 J^T of a symmetric M^-1: compare the triangle both variants computed)
        const size_t r = (k / R) % R, c = k % R;
        if (c <= r) dw = fmax(dw, fabs(w0[k] - w1[k]));
    }
    printf("Part 2  W = J (M^-1 J^T), %d x %d rows (5 contacts)\n", R, R);
    printf("  %-44s %10.0f cycles\n", "valu  (group reductions, the product's idiom)", v0);
    printf("  %-44s %10.0f cycles   x%.3f of valu  max |W diff| %.2e\n", "mfma  (6 x v_mfma_f64_16x16x4_f64 per env)", v1, v1 / v0, dw);
    return 0;
}
