// Issue-cost microbenchmark for the instruction mix of the attention kernels' VALU phases (round 4): what one SIMD pays, in
// shader cycles, per wave-instruction of each kind -- alone (one wave per SIMD), with a second wave of the SAME stream, and
// beside a second wave that issues bare MFMAs (the co-resident head of msst_bwd4.hip / the attention role of msst_fwd3.hip).
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_microbench.hip -o gpurun_out/valu_microbench
// output: one JSON object per line on stdout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// eight independent destinations per group, sources constant: no dependences, pure issue rate
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)

enum Op { ADD_F32, FMA_F32, PK_MUL_F32, PK_FMA_F32, PK_ADD_F32, EXP_F32, RCP_F32, MUL_LO_U32, MUL_HI_U32, MUL_U24, MAD_U24, CVT_PK_BF16,
          CNDMASK, CMP_GE_U32, CMP_CND, PERMLANE32_SWAP, DPP_MOV, BFE_U32, AND_B32, LSHL_OR, BITOP3, READLANE, MAX3_F32, MOV_B32, XOR_SHR, ADD3_U32,
          MAD_U64_U32, PERM_B32, ALIGNBIT, MFMA32, MFMA16, NOPS, N_OPS };
static const char* op_names[N_OPS] = {"v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32", "v_mul_lo_u32",
    "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_cvt_pk_bf16_f32", "v_cndmask_b32(vcc)", "v_cmp_ge_u32(vcc)", "v_cmp+v_cndmask pair", "v_permlane32_swap",
    "v_mov_b32 dpp row_shr", "v_bfe_u32", "v_and_b32", "v_lshl_or_b32", "v_bitop3_b32", "v_readlane_b32", "v_max3_f32", "v_mov_b32", "v_xor(x, x>>k) as lshrrev+xor",
    "v_add3_u32", "v_mad_u64_u32", "v_perm_b32", "v_alignbit_b32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "s_nop 0"};

template <int OP>
__device__ __forceinline__ void body8(float (&f)[8], unsigned (&u)[8], float fa, float fb, unsigned ua, unsigned ub, double (&pd)[8], double pc, float (&p)[8][2], f32x16 (&acc)[4], f32x4 (&acc4)[8], bf16x8 ma, bf16x8 mb) {
    unsigned tmp;
#define T_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
#define T_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
#define T_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pd[i]) : "v"(pc));
#define T_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pd[i]) : "v"(pc));
#define T_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pd[i]) : "v"(pc));
#define T_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
#define T_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define T_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define T_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define T_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define T_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub));
#define T_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(u[i]) : "v"(fb));
#define T_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(ub));
#define T_CMP(i) asm volatile("v_cmp_ge_u32 vcc, %0, %1" :: "v"(ua), "v"(ub) : "vcc");
#define T_CMPCND(i) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : "vcc");
#define T_PSWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
#define T_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define T_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 25" : "+v"(u[i]));
#define T_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define T_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(ub));
#define T_BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(ua), "v"(ub));
#define T_RDL(i) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(ua) : "s20");
#define T_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
#define T_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define T_XSH(i) asm volatile("v_lshrrev_b32 %1, 15, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(u[i]), "=&v"(tmp));
#define T_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub));
#define T_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(*(unsigned long long*)p[i]) : "v"(ua), "v"(ub) : "vcc");
#define T_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub));
#define T_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %0, 15" : "+v"(u[i]));
#define T_NOP(i) asm volatile("s_nop 0");
    if constexpr (OP == ADD_F32) { R8(T_ADD) }
    else if constexpr (OP == FMA_F32) { R8(T_FMA) }
    else if constexpr (OP == PK_MUL_F32) { R8(T_PKMUL) }
    else if constexpr (OP == PK_FMA_F32) { R8(T_PKFMA) }
    else if constexpr (OP == PK_ADD_F32) { R8(T_PKADD) }
    else if constexpr (OP == EXP_F32) { R8(T_EXP) }
    else if constexpr (OP == RCP_F32) { R8(T_RCP) }
    else if constexpr (OP == MUL_LO_U32) { R8(T_MULLO) }
    else if constexpr (OP == MUL_HI_U32) { R8(T_MULHI) }
    else if constexpr (OP == MUL_U24) { R8(T_MUL24) }
    else if constexpr (OP == MAD_U24) { R8(T_MAD24) }
    else if constexpr (OP == CVT_PK_BF16) { R8(T_CVT) }
    else if constexpr (OP == CNDMASK) { R8(T_CND) }
    else if constexpr (OP == CMP_GE_U32) { R8(T_CMP) }
    else if constexpr (OP == CMP_CND) { R8(T_CMPCND) }
    else if constexpr (OP == PERMLANE32_SWAP) { R8(T_PSWAP) }
    else if constexpr (OP == DPP_MOV) { R8(T_DPP) }
    else if constexpr (OP == BFE_U32) { R8(T_BFE) }
    else if constexpr (OP == AND_B32) { R8(T_AND) }
    else if constexpr (OP == LSHL_OR) { R8(T_LSHLOR) }
    else if constexpr (OP == BITOP3) { R8(T_BITOP) }
    else if constexpr (OP == READLANE) { R8(T_RDL) }
    else if constexpr (OP == MAX3_F32) { R8(T_MAX3) }
    else if constexpr (OP == MOV_B32) { R8(T_MOV) }
    else if constexpr (OP == XOR_SHR) { R8(T_XSH) }
    else if constexpr (OP == ADD3_U32) { R8(T_ADD3) }
    else if constexpr (OP == MAD_U64_U32) { R8(T_MAD64) }
    else if constexpr (OP == PERM_B32) { R8(T_PERM) }
    else if constexpr (OP == ALIGNBIT) { R8(T_ALIGN) }
    else if constexpr (OP == NOPS) { R8(T_NOP) }
    else if constexpr (OP == MFMA32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, acc[i & 3], 0, 0, 0);
    } else if constexpr (OP == MFMA16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, acc4[i], 0, 0, 0);
    }
}

// waves [0, nA) of a SIMD pair run OP_A, the others OP_B: blockDim = 64 * 4 * wps (wave w sits on SIMD w % 4; waves w < 4 run OP_A when
// split, the rest OP_B).  clk[2 * role + 0] = cycles of lane 0 of the first wave of that role, [.. + 1] = groups of 8 it issued.
template <int OP_A, int OP_B>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, int iters, unsigned long long* clk, int split) {
    float f[8]; unsigned u[8]; float p[8][2]; f32x16 acc[4]; f32x4 acc4[8]; double pd[8]; double pc;
    const int wv = threadIdx.x >> 6;
    const bool roleB = split && (wv & 4);
    bf16x8 ma, mb;
    for (int i = 0; i < 8; ++i) { f[i] = 0.5f + i; u[i] = threadIdx.x * 7 + i; p[i][0] = 0.25f * i; p[i][1] = 1.f + i; ma[i] = (__bf16)(0.01f * (threadIdx.x + i)); mb[i] = (__bf16)(0.02f * i); }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
    float fa = 0.37f + threadIdx.x * 1e-3f, fb = 1.0001f; unsigned ua = threadIdx.x * 2654435761u, ub = 0x9E3779B1u;
    for (int i = 0; i < 8; ++i) { float2 t = {1.0f + 1e-6f * i, 1.0f - 1e-6f * i}; pd[i] = __builtin_bit_cast(double, t); }
    { float2 t = {1.0000001f, 0.9999999f}; pc = __builtin_bit_cast(double, t); }
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    asm volatile("" : "+v"(fa), "+v"(fb), "+v"(ua), "+v"(ub), "+v"(pc));
    __shared__ volatile int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long nb = 0;
    if (!roleB) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) body8<OP_A>(f, u, fa, fb, ua, ub, pd, pc, p, acc, acc4, ma, mb);
        }
        if (split && threadIdx.x == 0) done = 1;
    } else {
        // the background role runs for as long as the measured one does and counts what it got done
        while (!done) {
#pragma unroll
            for (int r = 0; r < 4; ++r) body8<OP_B>(f, u, fa, fb, ua, ub, pd, pc, p, acc, acc4, ma, mb);
            ++nb;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && wv == 0) clk[0] = t1 - t0;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && !roleB) atomicMax(&clk[3], t1 - t0);
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && wv == 4) { clk[1] = t1 - t0; clk[2] = nb; }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i] + p[i][0] + p[i][1] + acc4[i][0] + (float)pd[i];
    for (int i = 0; i < 4; ++i) s += acc[i][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}


// ---- mixes inside one wave's stream: MODE 0 = 1 MFMA32 + K v_add_f32 (independent), 1 = dependent v_add_f32 chain (K ignored), 2 = alternating v_add_f32 / v_and_b32,
// 3 = alternating v_add_f32 / ds_read_b64 (no wait), 4 = alternating v_add_f32 / s_nop 0, 5 = 1 MFMA16 + K v_add_f32, 6 = alternating v_add_f32 / v_cvt_pk_bf16_f32,
// 7 = groups of [4 v_add_f32, s_nop 1]
template <int MODE, int K>
__global__ __launch_bounds__(1024) void mix_kernel(float* out, int iters, unsigned long long* clk) {
    float f[8]; unsigned u[8]; f32x16 acc[4]; f32x4 acc4[8];
    bf16x8 ma, mb;
    __shared__ double lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < 8; ++i) { f[i] = 0.5f + i; u[i] = threadIdx.x * 7 + i; ma[i] = (__bf16)(0.01f * (threadIdx.x + i)); mb[i] = (__bf16)(0.02f * i); }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
    float fa = 0.37f + threadIdx.x * 1e-3f, fb = 1.0001f; unsigned ua = threadIdx.x * 2654435761u, ub = 0x9E3779B1u;
    unsigned la = (threadIdx.x & 63) * 8;
    double dd[8];
    asm volatile("" : "+v"(fa), "+v"(fb), "+v"(ua), "+v"(ub), "+v"(la));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (MODE == 0) {
                acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, acc[g & 3], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(fb));
            } else if constexpr (MODE == 5) {
                acc4[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, acc4[g], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(fb));
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(fb));
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 2; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(2 * g + k) & 7]) : "v"(fb)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[(2 * g + k) & 7]) : "v"(ub)); }
            } else if constexpr (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 2; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(2 * g + k) & 7]) : "v"(fb)); asm volatile("ds_read_b64 %0, %1" : "=v"(dd[(2 * g + k) & 7]) : "v"(la)); }
                if (g == 7) asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (MODE == 4) {
#pragma unroll
                for (int k = 0; k < 2; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(2 * g + k) & 7]) : "v"(fb)); asm volatile("s_nop 0"); }
            } else if constexpr (MODE == 6) {
#pragma unroll
                for (int k = 0; k < 2; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(2 * g + k) & 7]) : "v"(fb)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[(2 * g + k) & 7]) : "v"(f[(2 * g + k) & 7]), "v"(fb)); }
            } else if constexpr (MODE == 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(2 * g + k) & 7]) : "v"(fb));
                asm volatile("s_nop 1");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i] + acc4[i][0] + (MODE == 3 ? (float)dd[i] : 0.f);
    for (int i = 0; i < 4; ++i) s += acc[i][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE, int K>
static int run_mix(const char* name, float* out, unsigned long long* clk) {
    const int iters = 2000;
    printf("{\"mix\": \"%s\", \"K\": %d, \"cyc_per_group\": {", name, K);
    for (int w = 1; w <= 4; ++w) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((mix_kernel<MODE, K>), dim3(256), dim3(256 * w), 0, 0, out, iters, clk);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        }
        printf("%s\"%d_waves_per_simd\": %.2f", w > 1 ? ", " : "", w, (double)h / (iters * 8.0));
    }
    printf("}}\n");
    fflush(stdout);
    return 0;
}

typedef void (*kern_t)(float*, int, unsigned long long*, int);
template <int OP> struct Tab { static kern_t same() { return &issue_kernel<OP, OP>; } static kern_t vs32() { return &issue_kernel<OP, MFMA32>; } static kern_t vs16() { return &issue_kernel<OP, MFMA16>; }
                               static kern_t vsadd() { return &issue_kernel<OP, ADD_F32>; } };
template <int... I> static void fill(kern_t (*t)[4], std::integer_sequence<int, I...>) {
    ((t[I][0] = Tab<I>::same(), t[I][1] = Tab<I>::vs32(), t[I][2] = Tab<I>::vs16(), t[I][3] = Tab<I>::vsadd()), ...);
}

int main() {
    static kern_t tab[N_OPS][4];
    fill(tab, std::make_integer_sequence<int, N_OPS>{});
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 512 * 1024 * sizeof(float)));
    CK(hipMalloc(&clk, 32));
    const int iters = 2000;   // x 32 instructions
    // waves per SIMD sweep of plain streams (cycles per wave-instruction as seen by wave 0)
    {
        const int ops[8] = {ADD_F32, PK_MUL_F32, EXP_F32, CVT_PK_BF16, MUL_LO_U32, CMP_CND, MFMA32, NOPS};
        for (int oi = 0; oi < 8; ++oi) {
            printf("{\"sweep\": \"%s\", \"cyc_per_instr\": {", op_names[ops[oi]]);
            for (int w = 1; w <= 4; ++w) {
                unsigned long long h[4] = {0, 0, 0, 0};
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(clk, 0, 32));
                    hipLaunchKernelGGL(tab[ops[oi]][0], dim3(256), dim3(256 * w), 0, 0, out, iters, clk, 0);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
                }
                printf("%s\"%d_waves_per_simd\": [%.2f, %.2f]", w > 1 ? ", " : "", w, (double)h[0] / (iters * 32.0), (double)h[3] / (iters * 32.0));
            }
            printf("}, \"note\": \"[wave 0, slowest wave]\"}\n");
        }
    }
    run_mix<0, 0>("group = 1 mfma32 + K v_add_f32", out, clk); run_mix<0, 2>("group = 1 mfma32 + K v_add_f32", out, clk); run_mix<0, 4>("group = 1 mfma32 + K v_add_f32", out, clk);
    run_mix<0, 6>("group = 1 mfma32 + K v_add_f32", out, clk); run_mix<0, 8>("group = 1 mfma32 + K v_add_f32", out, clk); run_mix<0, 12>("group = 1 mfma32 + K v_add_f32", out, clk);
    run_mix<5, 0>("group = 1 mfma16 + K v_add_f32", out, clk); run_mix<5, 2>("group = 1 mfma16 + K v_add_f32", out, clk); run_mix<5, 4>("group = 1 mfma16 + K v_add_f32", out, clk);
    run_mix<1, 0>("group = 4 dependent v_add_f32", out, clk);
    run_mix<2, 0>("group = 2 x (v_add_f32, v_and_b32)", out, clk);
    run_mix<3, 0>("group = 2 x (v_add_f32, ds_read_b64)", out, clk);
    run_mix<4, 0>("group = 2 x (v_add_f32, s_nop 0)", out, clk);
    run_mix<6, 0>("group = 2 x (v_add_f32, v_cvt_pk_bf16_f32)", out, clk);
    run_mix<7, 0>("group = 4 v_add_f32 + s_nop 1", out, clk);
    if (getenv("VALU_MB_SHORT")) return 0;
    for (int op = 0; op < N_OPS; ++op) {
        double r[5][2] = {};
        // 0: one wave per SIMD; 1: two waves per SIMD, same stream; 2: beside a wave of 32x32x16 MFMAs; 3: beside 16x16x32 MFMAs; 4: beside v_add_f32
        for (int cfg = 0; cfg < 5; ++cfg) {
            const int threads = cfg == 0 ? 256 : 512, split = cfg >= 2;
            kern_t k = tab[op][cfg < 2 ? 0 : cfg - 1];
            unsigned long long h[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(clk, 0, 32));
                hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, clk, split);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
            }
            r[cfg][0] = (double)h[0] / (iters * 32.0);
            r[cfg][1] = split ? (h[2] ? (double)h[1] / (h[2] * 32.0) : 0.0) : (double)h[1] / (iters * 32.0);
        }
        printf("{\"op\": \"%s\", \"cyc_alone\": %.2f, \"cyc_two_same\": %.2f, \"beside_mfma32\": {\"op\": %.2f, \"mfma\": %.2f}, \"beside_mfma16\": {\"op\": %.2f, \"mfma\": %.2f}, "
               "\"beside_v_add\": {\"op\": %.2f, \"add\": %.2f}}\n", op_names[op], r[0][0], r[1][0], r[2][0], r[2][1], r[3][0], r[3][1], r[4][0], r[4][1]);
        fflush(stdout);
    }
    return 0;
}
