// Micro-benchmark (development tool, not product): issue cost and dependent-chain latency of the VALU instructions the env kernels are
// made of, ONE wavefront per SIMD (the occupancy of the 65536-lane CartPole rollout and of the cooperative MuJoCo kernel).
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench_valu.bin scripts/ubench_valu.hip ; prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITERS 2000

// each kernel: out[0] = cycles (s_memtime) for ITERS * 8 instructions, per wave
#define KERNEL_F64(NAME, ASM_I, ASM_D)                                                                                              \
    __global__ void NAME##_indep(double *sink, unsigned long long *cyc) {                                                           \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;         \
        double b = 1.0000001, c = 1e-9;                                                                                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITERS; i++) {                                                                                           \
            asm volatile(ASM_I(0) ASM_I(1) ASM_I(2) ASM_I(3) ASM_I(4) ASM_I(5) ASM_I(6) ASM_I(7)                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                           \
                         : "v"(b), "v"(c));                                                                                         \
        }                                                                                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                                                       \
        sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                            \
    }                                                                                                                               \
    __global__ void NAME##_dep(double *sink, unsigned long long *cyc) {                                                             \
        double a0 = threadIdx.x;                                                                                                    \
        double b = 1.0000001, c = 1e-9;                                                                                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITERS; i++) {                                                                                           \
            asm volatile(ASM_D ASM_D ASM_D ASM_D ASM_D ASM_D ASM_D ASM_D : "+v"(a0) : "v"(b), "v"(c));                              \
        }                                                                                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                                                       \
        sink[blockIdx.x * 64 + threadIdx.x] = a0;                                                                                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                            \
    }

#define FMA_I(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define FMA_D "v_fma_f64 %0, %0, %1, %2\n"
#define ADD_I(k) "v_add_f64 %" #k ", %" #k ", %9\n"
#define ADD_D "v_add_f64 %0, %0, %2\n"
#define MUL_I(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define MUL_D "v_mul_f64 %0, %0, %1\n"
#define RCP_I(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define RCP_D "v_rcp_f64 %0, %0\n"
#define LDEXP_I(k) "v_ldexp_f64 %" #k ", %" #k ", 1\n"
#define LDEXP_D "v_ldexp_f64 %0, %0, 1\n"
#define MAD64_I(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n"
#define MAD64_D "v_mad_u64_u32 %0, vcc, %0, %1, %0\n"
#define LSHLADD_I(k) "v_lshl_add_u64 %" #k ", %" #k ", 1, %8\n"
#define LSHLADD_D "v_lshl_add_u64 %0, %0, 1, %1\n"
#define MOV64_I(k) "v_mov_b64 %" #k ", %8\n"
#define MOV64_D "v_mov_b64 %0, %1\n"
KERNEL_F64(fma_f64, FMA_I, FMA_D)
KERNEL_F64(add_f64, ADD_I, ADD_D)
KERNEL_F64(mul_f64, MUL_I, MUL_D)
KERNEL_F64(rcp_f64, RCP_I, RCP_D)
KERNEL_F64(ldexp_f64, LDEXP_I, LDEXP_D)
KERNEL_F64(lshl_add_u64, LSHLADD_I, LSHLADD_D)
KERNEL_F64(mov_b64, MOV64_I, MOV64_D)

// 64-bit accumulators with 32-bit multiplicands in the low halves of b / c (clobbers vcc)
#define KERNEL_MAD(NAME, ASM_I, ASM_D)                                                                                              \
    __global__ void NAME##_indep(double *sink, unsigned long long *cyc) {                                                           \
        unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned b = 0x9e3779b9u + threadIdx.x, c = 0x85ebca6bu;                                                                    \
        unsigned long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITERS; i++) {                                                                                           \
            asm volatile(ASM_I(0) ASM_I(1) ASM_I(2) ASM_I(3) ASM_I(4) ASM_I(5) ASM_I(6) ASM_I(7)                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                           \
                         : "v"(b), "v"(c)                                                                                           \
                         : "vcc");                                                                                                  \
        }                                                                                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                                                       \
        sink[blockIdx.x * 64 + threadIdx.x] = (double)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                                     \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                            \
    }
KERNEL_MAD(mad_u64_u32, MAD64_I, MAD64_D)

// 32-bit integer / select ops
#define KERNEL_U32(NAME, ASM_I)                                                                                                     \
    __global__ void NAME##_indep(double *sink, unsigned long long *cyc) {                                                           \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;       \
        unsigned b = 0x9e3779b9u + threadIdx.x, c = 0x85ebca6bu;                                                                    \
        unsigned long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITERS; i++) {                                                                                           \
            asm volatile(ASM_I(0) ASM_I(1) ASM_I(2) ASM_I(3) ASM_I(4) ASM_I(5) ASM_I(6) ASM_I(7)                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                           \
                         : "v"(b), "v"(c)                                                                                           \
                         : "vcc", "scc", "s20", "s21");                                                                                    \
        }                                                                                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                                                       \
        sink[blockIdx.x * 64 + threadIdx.x] = (double)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                                     \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                            \
    }
#define MULLO_I(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define MULHI_I(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define MUL24_I(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define MAD24_I(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define ADD32_I(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define ADDCO_I(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n"
#define CNDMASK_I(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define XOR_I(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define ALIGNBIT_I(k) "v_alignbit_b32 %" #k ", %" #k ", %8, %9\n"
#define FMA32_I(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
KERNEL_U32(mul_lo_u32, MULLO_I)
KERNEL_U32(mul_hi_u32, MULHI_I)
KERNEL_U32(mul_u32_u24, MUL24_I)
KERNEL_U32(mad_u32_u24, MAD24_I)
KERNEL_U32(add_u32, ADD32_I)
KERNEL_U32(add_co_u32, ADDCO_I)
KERNEL_U32(cndmask_b32, CNDMASK_I)
KERNEL_U32(xor_b32, XOR_I)
KERNEL_U32(alignbit_b32, ALIGNBIT_I)
KERNEL_U32(fma_f32, FMA32_I)
#define CNDMASK64_I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define CNDMASKSAME_I(k) "v_cndmask_b32 %" #k ", %8, %9, vcc\n"
#define CMPCND_I(k) "v_cmp_lt_u32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define CMP_I(k) "v_cmp_lt_u32 vcc, %" #k ", %8\n"
#define CMP64_I(k) "v_cmp_lt_u32 s[20:21], %" #k ", %8\n"
#define MOV32_I(k) "v_mov_b32 %" #k ", %8\n"
#define ADD3_I(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define OR_I(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define LSHR_I(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define ADDC_I(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %8, vcc\n"
#define SALU_I(k) "s_add_u32 s20, s20, 1\n"
#define SNOP_I(k) "s_nop 0\n"
#define BITOP_I(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define CNDMASK64VCC_I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n"
#define CND_FMA3_I(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\nv_xor_b32 %" #k ", %" #k ", %9\nv_xor_b32 %" #k ", %" #k ", %8\nv_xor_b32 %" #k ", %" #k ", %9\n"
#define CND64_XOR3_I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\nv_xor_b32 %" #k ", %" #k ", %9\nv_xor_b32 %" #k ", %" #k ", %8\nv_xor_b32 %" #k ", %" #k ", %9\n"
#define CNDSDWA_I(k) "v_cndmask_b32_e64 %" #k ", %8, %9, vcc\n"
KERNEL_U32(cndmask_e64_vcc, CNDMASK64VCC_I)
KERNEL_U32(cnd_vcc_plus_3xor, CND_FMA3_I)
KERNEL_U32(cnd_sgpr_plus_3xor, CND64_XOR3_I)
KERNEL_U32(cndmask_e64_sgpr, CNDMASK64_I)
KERNEL_U32(cndmask_nodep, CNDMASKSAME_I)
KERNEL_U32(cmp_then_cndmask, CMPCND_I)
KERNEL_U32(cmp_vcc, CMP_I)
KERNEL_U32(cmp_sgpr, CMP64_I)
KERNEL_U32(mov_b32, MOV32_I)
KERNEL_U32(add3_u32, ADD3_I)
KERNEL_U32(or_b32, OR_I)
KERNEL_U32(lshrrev_b32, LSHR_I)
KERNEL_U32(addc_co_u32, ADDC_I)
KERNEL_U32(s_add_u32, SALU_I)
KERNEL_U32(s_nop0, SNOP_I)
KERNEL_U32(and_or_b32, BITOP_I)
#define CVT32_I(k) "v_cvt_f32_f64 %" #k ", %8\n"
#define CMPF64_I(k) "v_cmp_lt_f64 vcc, %8, %9\n"
#define CVTF64U_I(k) "v_cvt_f64_u32 %" #k ", %8\n"
#define DIVSCALE_I(k) "v_div_scale_f64 %" #k ", vcc, %8, %9, %8\n"
#define DIVFMAS_I(k) "v_div_fmas_f64 %" #k ", %" #k ", %8, %9\n"
#define DIVFIX_I(k) "v_div_fixup_f64 %" #k ", %" #k ", %8, %9\n"
#define LSHR64_I(k) "v_lshrrev_b64 %" #k ", 3, %" #k "\n"
#define FMAC_I(k) "v_fmac_f64 %" #k ", %8, %9\n"
#define FMAK_I(k) "v_fma_f64 %" #k ", %" #k ", %8, 0x3ff0000000100000\n"
#define ADDK_I(k) "v_add_f64 %" #k ", %" #k ", 0x3ff0000000100000\n"
#define ADDS_I(k) "v_add_f64 %" #k ", %" #k ", s[22:23]\n"
#define RSQ_I(k) "v_rsq_f64 %" #k ", %" #k "\n"
#define SQRT_I(k) "v_sqrt_f64 %" #k ", %" #k "\n"

#define KERNEL_F64I(NAME, ASM_I)                                                                                                    \
    __global__ void NAME##_indep(double *sink, unsigned long long *cyc) {                                                           \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;         \
        double b = 1.0000001, c = 1.5;                                                                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITERS; i++) {                                                                                           \
            asm volatile(ASM_I(0) ASM_I(1) ASM_I(2) ASM_I(3) ASM_I(4) ASM_I(5) ASM_I(6) ASM_I(7)                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                           \
                         : "v"(b), "v"(c)                                                                                           \
                         : "vcc", "scc", "s20", "s21", "s22", "s23");                                                                      \
        }                                                                                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                                                       \
        sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                            \
    }
KERNEL_F64I(cmp_lt_f64, CMPF64_I)
KERNEL_F64I(div_scale_f64, DIVSCALE_I)
KERNEL_F64I(div_fmas_f64, DIVFMAS_I)
KERNEL_F64I(div_fixup_f64, DIVFIX_I)
KERNEL_F64I(lshrrev_b64, LSHR64_I)
KERNEL_F64I(fmac_f64, FMAC_I)
KERNEL_F64I(add_f64_sgpr, ADDS_I)
KERNEL_F64I(rsq_f64, RSQ_I)
KERNEL_F64I(sqrt_f64, SQRT_I)

typedef void (*kern_t)(double *, unsigned long long *);
static double run(kern_t k, int blocks, double *sink, unsigned long long *dcyc) {
    std::vector<unsigned long long> h(blocks);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, sink, dcyc);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, sink, dcyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dcyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / blocks / (ITERS * 8.0);
}

int main(int argc, char **argv) {
    // argv[1] = wavefronts per SIMD (default 1).  With W > 1 the per-wavefront cost of an instruction is W x its share of the SIMD's vector pipe:
    // cost / W is the pipe time of the instruction (4 cycles for a full-rate one), which one wavefront alone (issue cadence ~5 cycles) cannot show.
    const int per_simd = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = 1024 * per_simd;  // 64-lane workgroups; 1024 SIMDs
    double *sink;
    unsigned long long *dcyc;
    hipMalloc(&sink, sizeof(double) * blocks * 64);
    hipMalloc(&dcyc, sizeof(unsigned long long) * blocks);
    printf("%-16s %10s %10s   (s_memtime cycles per wave-instruction, %d wavefront(s) per SIMD)\n", "instruction", "8 chains", "1 chain", per_simd);
#define ROW2(N) printf("%-16s %10.2f %10.2f\n", #N, run(N##_indep, blocks, sink, dcyc), run(N##_dep, blocks, sink, dcyc));
#define ROW1(N) printf("%-16s %10.2f %10s\n", #N, run(N##_indep, blocks, sink, dcyc), "-");
    ROW2(fma_f64) ROW2(add_f64) ROW2(mul_f64) ROW2(rcp_f64) ROW2(ldexp_f64) ROW2(lshl_add_u64) ROW2(mov_b64)
    ROW1(mad_u64_u32) ROW1(mul_lo_u32) ROW1(mul_hi_u32) ROW1(mul_u32_u24) ROW1(mad_u32_u24) ROW1(add_u32) ROW1(add_co_u32)
    ROW1(cndmask_b32) ROW1(xor_b32) ROW1(alignbit_b32) ROW1(fma_f32)
    ROW1(cndmask_e64_vcc)
    printf("%-16s %10.2f   (4 instructions: cndmask vcc + 3 dependent xor)\n", "cnd_vcc+3xor", 4 * run(cnd_vcc_plus_3xor_indep, blocks, sink, dcyc));
    printf("%-16s %10.2f   (4 instructions: cndmask sgpr + 3 dependent xor)\n", "cnd_sgpr+3xor", 4 * run(cnd_sgpr_plus_3xor_indep, blocks, sink, dcyc));
    ROW1(cndmask_e64_sgpr) ROW1(cndmask_nodep) ROW1(cmp_vcc) ROW1(cmp_sgpr) ROW1(mov_b32) ROW1(add3_u32) ROW1(or_b32) ROW1(lshrrev_b32)
    ROW1(addc_co_u32) ROW1(s_add_u32) ROW1(s_nop0) ROW1(and_or_b32)
    printf("%-16s %10.2f   (two instructions: v_cmp + v_cndmask on its vcc)\n", "cmp+cndmask", 2 * run(cmp_then_cndmask_indep, blocks, sink, dcyc));
    ROW1(cmp_lt_f64) ROW1(div_scale_f64) ROW1(div_fmas_f64) ROW1(div_fixup_f64) ROW1(lshrrev_b64)
    ROW1(fmac_f64) ROW1(add_f64_sgpr) ROW1(rsq_f64) ROW1(sqrt_f64)
    return 0;
}
