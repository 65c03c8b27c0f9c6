// Gate for "attention backward on ONE wave per SIMD with the 512-register file" (VERDICT r4 item 1): replay the kernel's real
// per-tile instruction mix -- per role and tile 98 v_mfma_f32_32x32x16_bf16 + 598 VALU + 172 LDS (+ SALU), LABNOTES round 4 census of
// msst_bwd4.hip -- from one wave per SIMD that carries BOTH heads' streams interleaved (two tiles in flight: 196 MFMAs per tile
// pair), hand-placed as evenly as it can be: after every MFMA the same number of fillers.  Every filler chain is INDEPENDENT
// (eight rotating destinations, constant sources, no LDS waits inside the loop): no dependency or latency stall is modelled, so
// the number is the issue-rate FLOOR of such a kernel, not an estimate of a real one.
// Compare with the shipped two-waves-per-SIMD kernel: 13.0-13.2 k shader cycles per tile pair (tools/stamps_bwd4.py).
// The gate: >= 25 % fewer cycles (<= 9.8 k) or the structure is not worth building.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gate_microbench.hip -o /tmp/gate_microbench
// output: one JSON object per line
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// one MFMA gap: NV VALU (mix per 8: 3 v_fma_f32, 1 v_cvt_pk_bf16_f32, 1 v_cndmask, 1 v_and / bitop, 1 v_pk_mul_f32 or v_mul, 1 v_exp every 4th gap),
// NL LDS (alternating ds_read_b128 / ds_write_b64, no waits), NS SALU (s_mov)
template <int NV, int NL, int NS, bool PK>
__device__ __forceinline__ void gap(int g, float (&f)[8], unsigned (&u)[8], double (&pd)[4], float fa, float fb, unsigned ub, double pc,
                                    unsigned la, f32x16 (&r)[2], double (&w)[2]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = (g * NV + k) & 7;
        switch ((g * NV + k) % 8) {
            case 0: case 1: case 2: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(fa), "v"(fb)); break;
            case 3: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(f[j]), "v"(fb)); break;
            case 4: asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(u[j]) : "v"(ub)); break;
            case 5: asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(u[j]) : "v"(ub)); break;
            case 6:
                if (PK) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pd[j & 3]) : "v"(pc));
                else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fb));
                break;
            default:
                if ((g & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[j]));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fb));
        }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        if ((g * NL + k) & 1) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(w[k & 1]) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<float4*>(&r[k & 1]))) : "v"(la) : "memory");
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) asm volatile("s_mov_b32 s22, 0x1234" ::: "s22");   // (an SALU issue slot that leaves SCC alone: the loop's own compare / branch live around these)
}

// W2 (round 6): the SHIPPED structure instead -- two waves per SIMD (512 threads), each carrying ONE head's stream of
// a tile (98 MFMAs + its fillers), every chain independent as above: what the scalar-class diet (2 -> 1 scalar instruction per MFMA)
// can buy a kernel that already overlaps two instruction streams per SIMD.
template <int NV, int NL, int NS, bool PK, bool W2 = false>
__global__ __launch_bounds__(W2 ? 512 : 256, 1) void gate_kernel(float* out, int pairs, unsigned long long* clk) {
    __shared__ double lds[2048];
    float f[8]; unsigned u[8]; double pd[4]; f32x16 acc[8]; f32x16 r[2]; double w[2];
    bf16x8 ma, mb;
    for (int i = 0; i < 8; ++i) { f[i] = 0.5f + i; u[i] = threadIdx.x * 7 + i; ma[i] = (__bf16)(0.01f * ((threadIdx.x * 13 + i * 7) & 63) - 0.3f); mb[i] = (__bf16)(0.02f * ((threadIdx.x * 5 + i * 3) & 31) - 0.3f); }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 4; ++i) { float2 t = {1.0f + 1e-6f * i, 1.0f - 1e-6f * i}; pd[i] = __builtin_bit_cast(double, t); }
    double pc; { float2 t = {1.0000001f, 0.9999999f}; pc = __builtin_bit_cast(double, t); }
    w[0] = 1.0; w[1] = 2.0;
    for (int i = threadIdx.x; i < 2048; i += (W2 ? 512 : 256)) lds[i] = i;
    float fa = 0.37f + threadIdx.x * 1e-3f, fb = 1.0001f; unsigned ub = 0x9E3779B1u;
    unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x & 63) * 16;
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    asm volatile("" : "+v"(fa), "+v"(fb), "+v"(ub), "+v"(pc), "+v"(la));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < pairs; ++it) {
        // one tile pair = 196 MFMAs: 24 groups of 8 (eight independent accumulators) + 4
#pragma unroll 1
        for (int grp = 0; grp < (W2 ? 12 : 24); ++grp) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(ma), "v"(mb));   // (opaque: as a builtin the compiler deletes the MFMAs of the mixed streams)
                gap<NV, NL, NS, PK>(g, f, u, pd, fa, fb, ub, pc, la, r, w);
            }
        }
#pragma unroll
        for (int g = 0; g < (W2 ? 2 : 4); ++g) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(ma), "v"(mb));   // (opaque: as a builtin the compiler deletes the MFMAs of the mixed streams)
            gap<NV, NL, NS, PK>(g, f, u, pd, fa, fb, ub, pc, la, r, w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i] + acc[i][3] + (i < 4 ? (float)pd[i] : 0.f);
    s += r[0][0] + r[1][0];
    out[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
}

template <int NV, int NL, int NS, bool PK, bool W2 = false>
static int run(const char* what, float* out, unsigned long long* clk) {
    const int pairs = 40;
    unsigned long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((gate_kernel<NV, NL, NS, PK, W2>), dim3(256), dim3(W2 ? 512 : 256), 0, 0, out, pairs, clk);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    }
    const double cyc = (double)h / pairs;
    printf("{\"waves_per_simd\": %d, \"mix\": \"%s\", \"per_mfma\": {\"valu\": %d, \"lds\": %d, \"salu\": %d}, \"packed_f32\": %s, \"per_tile_pair\": {\"mfma\": 196, \"valu\": %d, \"lds\": %d, \"salu\": %d}, "
           "\"cycles_per_tile_pair\": %.0f, \"cycles_per_mfma\": %.1f, \"vs_shipped_13100\": %.3f}\n",
           W2 ? 2 : 1, what, NV, NL, NS, PK ? "true" : "false", 196 * NV, 196 * NL, 196 * NS, cyc, cyc / 196.0, cyc / 13100.0);
    fflush(stdout);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 256 * 256 * sizeof(float)));
    CK(hipMalloc(&clk, 8));
    run<0, 0, 0, false>("bare MFMAs", out, clk);
    run<4, 0, 0, false>("4 VALU per gap (what one 32x32x16 hides)", out, clk);
    run<4, 1, 1, false>("trimmed stream: 784 VALU + 196 LDS + 196 SALU (a hand-written kernel at 2/3 of today's count)", out, clk);
    run<5, 2, 1, false>("940 VALU (both heads merged, saved lse) + 392 LDS + 196 SALU", out, clk);
    run<6, 2, 1, false>("today's mix: 1196 VALU + 344 LDS (here 392) + SALU, scalar f32", out, clk);
    run<6, 2, 1, true>("today's mix with packed f32 in it", out, clk);
    run<6, 2, 2, false>("today's mix + the compiler's SALU (s_waitcnt / s_nop / branches: ~2 per gap)", out, clk);
    run<8, 2, 2, false>("8 VALU per gap", out, clk);
    // round 6: two waves per SIMD, one head's stream each (the shipped structure, stall free) -- today's mix by the round-5 census (4.8 VALU /
    // 1.7 LDS / 2.1 scalar-class per MFMA) and the verdict's target (one scalar-class instruction per MFMA)
    run<0, 0, 0, false, true>("two waves: bare MFMAs", out, clk);
    run<5, 2, 2, false, true>("two waves: today's mix (5 VALU + 2 LDS + 2 scalar per MFMA)", out, clk);
    run<5, 2, 1, false, true>("two waves: target mix (5 VALU + 2 LDS + 1 scalar per MFMA)", out, clk);
    run<5, 2, 0, false, true>("two waves: no scalar instruction at all", out, clk);
    run<4, 2, 1, false, true>("two waves: 4 VALU + 2 LDS + 1 scalar", out, clk);
    run<4, 1, 1, false, true>("two waves: 4 VALU + 1 LDS + 1 scalar", out, clk);
    return 0;
}
