// Box microbenchmark ("confirm on box with a microbench and report both"):
//   * dense bf16 MFMA peak   (back-to-back v_mfma_f32_32x32x16_bf16 / 16x16x32 on independent accumulators)
//   * HBM streaming bandwidth (float4 copy, 2 GiB moved)
//   * LDS wave-instruction cost of the fragment access patterns the attention backward uses (cycles per
//     wave-instruction as seen by one CU with 4 or 8 waves resident), for candidate tile layouts
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/peak_microbench.hip -o gpurun_out/peak_microbench
// output: one JSON object on stdout (kept as profiles/rNN_peak_microbench.json)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------ MFMA peak
// MODE: 0 = zero operands, 1 = one small constant, 2 = a smooth per-lane pattern (round 3's), 3 = hashed full-range values in [-1, 1)
// (the chip clocks to its power budget: the same instruction stream runs at a different clock on different data)
__device__ __forceinline__ float mb_rand(unsigned x) {
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}
template <int SHAPE, int MODE>
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters, unsigned long long* clk) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 16 + i;
        a[i] = (__bf16)(MODE == 0 ? 0.f : MODE == 1 ? 0.5f : MODE == 2 ? 0.001f * (threadIdx.x + i) : mb_rand(id));
        b[i] = (__bf16)(MODE == 0 ? 0.f : MODE == 1 ? 0.25f : MODE == 2 ? 0.002f * (threadIdx.x - i) : mb_rand(id + 8));
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if constexpr (SHAPE == 32) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - w0; }
    } else {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - w0; }
    }
}

// the same loop on v_mfma_f32_16x16x32_f16 (round 6: the forward's operands are IEEE half): does the wider multiplier cost clock under the power cap?
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void mfma_peak_f16(float* out, int iters, unsigned long long* clk) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 16 + i;
        a[i] = (_Float16)(MODE == 0 ? 0.f : mb_rand(id));
        b[i] = (_Float16)(MODE == 0 ? 0.f : mb_rand(id + 8));
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c7, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - w0; }
}

// ------------------------------------------------------------------ HBM copy
__global__ __launch_bounds__(256) void copy4(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// the same streams with U 16-byte requests in flight per lane (the product's HBM-bound kernels keep 6 to 28 in flight): copy,
// read-only (sum) and write-only
template <int U>
__global__ __launch_bounds__(256) void copy4u(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
}
template <int U>
__global__ __launch_bounds__(256) void read4u(const f32x4* __restrict__ src, float* __restrict__ sink, long n) {
    const long stride = (long)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc = acc + v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ __launch_bounds__(256) void write4(f32x4* __restrict__ dst, long n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = v;
}

// ------------------------------------------------------------------ L2-resident weight-fragment loads (1 KB per wave-instruction)
// every wave streams fragment-packed 1 KB pieces (lane * 16 B) of a small buffer that stays in L2; cycles per wave-instruction per CU
__global__ __launch_bounds__(512) void l2_frag(const u32x4* __restrict__ w, int nfrag, unsigned long long* cyc, unsigned* sink, int iters) {
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    int f = (blockIdx.x * 7 + wave * 13) % nfrag;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 r[12];
#pragma unroll
        for (int n = 0; n < 12; ++n) { r[n] = w[(long)((f + n) % nfrag) * 64 + l]; }
#pragma unroll
        for (int n = 0; n < 12; ++n) acc ^= r[n][0] ^ r[n][3];
        f = (f + 12) % nfrag;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (l == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ------------------------------------------------------------------ LDS patterns
// f(row) of the XOR-swizzled 128-byte-row layout: slot' = slot ^ f(row), f = (r1 << 2) | (r3 << 1) | r2
__device__ __forceinline__ int fswz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }

enum Pat {
    B128_32_S144 = 0,    // k-contiguous b128, 32-row fragment (32x32x16 operand), row stride 144 B
    B128_32_SWZ128,      // same, 128-B rows + XOR swizzle
    B128_16_S144,        // k-contiguous b128, 16-row fragment (16x16x32 operand), row stride 144 B (today's LDH)
    B128_16_SWZ128,
    TR_32_S144,          // transposed b64, 32-column fragment, 4 consecutive k-rows, row stride 144 B
    TR_32_SWZ128,        // same on 128-B rows + XOR swizzle
    TR_32_S192,          // same, 192-B rows unswizzled (96-wide arrays)
    TR_32_S144_R4,       // k-rows {c, c+4, c+8, c+12}, row stride 144 B
    TR_16_S144,          // today's ld_ks: 16-column fragments, k-rows 8g + (i >> 2), stride 144 B
    TR_16_S144_PERM,     // today's ld_ks_perm: k-rows 4g + (i >> 2)
    B128_32_S192_SWZ,    // k-contiguous b128, 32-row fragment, 192-B rows + XOR (r3 r2) inside the 64-B quarter
    B128_32_S208,        // same, 208-B rows (13 slots), unswizzled
    W64_32_S144,         // C-tile store: lane (j = l & 31, hi) writes 8 B at row j, element 4 hi (+8 q), row stride 144 B
    W64_32_SWZ128,       // same on the swizzled 128-B rows
    W64_16_S144,         // today's st_nat: lane (c = l & 15, g) writes 8 B at row c, element 4 g
    PAT_COUNT
};

template <int PAT>
__device__ __forceinline__ unsigned pat_addr(int l, int n) {   // byte address of wave-instruction n (n varies the immediate part)
    const int hi = l >> 5, u = (l >> 4) & 1, i = l & 15, g = l >> 4;
    switch (PAT) {
        case B128_32_S144: return (l & 31) * 144 + ((n & 3) * 2 + hi) * 16;
        case B128_32_SWZ128: { const int r = l & 31; return r * 128 + ((((n & 3) * 2 + hi) ^ fswz(r)) << 4); }
        case B128_16_S144: return i * 144 + ((n & 1) * 4 + g) * 16;
        case B128_16_SWZ128: return i * 128 + ((((n & 1) * 4 + g) ^ fswz(i)) << 4);
        case TR_32_S144: { const int kr = 8 * hi + 4 * (n & 1) + (i >> 2); return kr * 144 + (((n >> 1) & 1) * 32 + 16 * u + 4 * (i & 3)) * 2; }
        case TR_32_SWZ128: {
            const int kr = 8 * hi + 4 * (n & 1) + (i >> 2);
            const int e = ((n >> 1) & 1) * 32 + 16 * u + 4 * (i & 3);   // element (column) index
            return kr * 128 + (((e >> 3) ^ fswz(kr)) << 4) + (e & 4) * 2;
        }
        case TR_32_S192: { const int kr = 8 * hi + 4 * (n & 1) + (i >> 2); return kr * 192 + ((n >> 1) % 3 * 32 + 16 * u + 4 * (i & 3)) * 2; }
        case TR_32_S144_R4: { const int kr = 2 * hi + (n & 1) + 4 * (i >> 2); return kr * 144 + (((n >> 1) & 1) * 32 + 16 * u + 4 * (i & 3)) * 2; }
        case TR_16_S144: { const int kr = 8 * g + (i >> 2) + 4 * (n & 1); return kr * 144 + (((n >> 1) & 3) * 16 + 4 * (i & 3)) * 2; }
        case TR_16_S144_PERM: { const int kr = 4 * g + (i >> 2) + 16 * (n & 1); return kr * 144 + (((n >> 1) & 3) * 16 + 4 * (i & 3)) * 2; }
        case B128_32_S192_SWZ: {
            const int r = l & 31, s = (n % 6) * 2 + hi;
            return r * 192 + (((s & ~3) | ((s & 3) ^ ((((r >> 3) & 1) << 1) | ((r >> 2) & 1)))) << 4);
        }
        case B128_32_S208: return (l & 31) * 208 + ((n % 6) * 2 + hi) * 16;
        case W64_32_S144: return (l & 31) * 144 + (4 * hi + 8 * (n & 3)) * 2;
        case W64_32_SWZ128: { const int r = l & 31, e = 4 * hi + 8 * (n & 3) + 32 * ((n >> 2) & 1); return r * 128 + (((e >> 3) ^ fswz(r)) << 4) + (e & 4) * 2; }
        case W64_16_S144: return i * 144 + (4 * g + 16 * (n & 3)) * 2;
    }
    return 0;
}

template <int PAT>
__global__ __launch_bounds__(512) void lds_pat(unsigned long long* cyc, unsigned* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int l = threadIdx.x & 63;
    unsigned a[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) a[n] = pat_addr<PAT>(l, n) + (threadIdx.x >> 6) * 0;   // all waves read the same tile
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        constexpr bool WR = (PAT == W64_32_S144 || PAT == W64_32_SWZ128 || PAT == W64_16_S144);
        constexpr bool TR = (PAT >= TR_32_S144 && PAT <= TR_16_S144_PERM);
        if constexpr (WR) {
            u32x2 v = {acc, (unsigned)it};
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("ds_write_b64 %0, %1" :: "v"(a[n & 7]), "v"(v) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (TR) {
            u32x2 r[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[n]) : "v"(a[n & 7]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) { asm volatile("" : "+v"(r[n])); acc ^= r[n][0] ^ r[n][1]; }
        } else {
            u32x4 r[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("ds_read_b128 %0, %1" : "=v"(r[n]) : "v"(a[n & 7]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) { asm volatile("" : "+v"(r[n])); acc ^= r[n][0] ^ r[n][3]; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int PAT>
static int run_pat(const char* name, int waves, unsigned long long* d_cyc, unsigned* d_sink, bool last) {
    const int iters = 256, grid = 256;
    hipLaunchKernelGGL(lds_pat<PAT>, dim3(grid), dim3(64 * waves), 65536, 0, d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(lds_pat<PAT>, dim3(grid), dim3(64 * waves), 65536, 0, d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * waves);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    // per workgroup (= per CU): slowest wave's cycles / (instructions issued by all its waves)
    std::vector<double> per;
    for (int b = 0; b < grid; ++b) {
        unsigned long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = std::max(mx, h[b * waves + w]);
        per.push_back((double)mx / ((double)iters * 16 * waves));
    }
    std::sort(per.begin(), per.end());
    printf("    \"%s_w%d\": %.2f%s\n", name, waves, per[per.size() / 2], last ? "" : ",");
    return 0;
}

#define RUN(P, last) do { if (run_pat<P>(#P, 4, d_cyc, d_sink, false)) return 1; if (run_pat<P>(#P, 8, d_cyc, d_sink, last)) return 1; } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* d_out;
    CK(hipMalloc(&d_out, 4096 * 256 * 4));
    // MFMA: 256 CUs x 2 workgroups x 4 waves; four operand fills (DVFS: the clock follows the data)
    {
        const int grid = 512, iters = 20000;
        unsigned long long* d_clk;
        CK(hipMalloc(&d_clk, 16));
        const double f32 = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16, f16 = (double)grid * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
        double tf32[4], tf16[4], mhz[4];
        auto run = [&](auto kern, double flops, double& tf, double* clock_mhz) -> int {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                float ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, iters, d_clk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            tf = flops / best * 1e-9;
            if (clock_mhz) {
                unsigned long long h[2];
                CK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
                *clock_mhz = (double)h[0] / ((double)h[1] / 100.0);   // shader cycles per microsecond of the 100 MHz wall clock
            }
            return 0;
        };
        if (run(mfma_peak<32, 0>, f32, tf32[0], &mhz[0]) || run(mfma_peak<32, 1>, f32, tf32[1], &mhz[1]) || run(mfma_peak<32, 2>, f32, tf32[2], &mhz[2]) ||
            run(mfma_peak<32, 3>, f32, tf32[3], &mhz[3]) || run(mfma_peak<16, 0>, f16, tf16[0], nullptr) || run(mfma_peak<16, 3>, f16, tf16[3], nullptr)) return 1;
        printf("  \"mfma_bf16_32x32x16_tflops\": %.1f, \"mfma_bf16_16x16x32_tflops\": %.1f, \"mfma_nominal_tflops\": 2500,\n", tf32[3], tf16[3]);
        printf("  \"mfma_bf16_32x32x16_by_operands\": {\"zero\": %.1f, \"constant\": %.1f, \"smooth_pattern\": %.1f, \"random\": %.1f},\n", tf32[0], tf32[1], tf32[2], tf32[3]);
        printf("  \"mfma_shader_clock_mhz_by_operands\": {\"zero\": %.0f, \"constant\": %.0f, \"smooth_pattern\": %.0f, \"random\": %.0f},\n", mhz[0], mhz[1], mhz[2], mhz[3]);
        printf("  \"mfma_bf16_16x16x32_by_operands\": {\"zero\": %.1f, \"random\": %.1f},\n", tf16[0], tf16[3]);
        {
            double th[2], mh[2], tb = 0, mbz = 0;
            if (run(mfma_peak_f16<0>, f16, th[0], &mh[0]) || run(mfma_peak_f16<3>, f16, th[1], &mh[1]) || run(mfma_peak<16, 3>, f16, tb, &mbz)) return 1;
            printf("  \"mfma_f16_16x16x32_by_operands\": {\"zero\": %.1f, \"random\": %.1f}, \"mfma_f16_16x16x32_clock_mhz\": {\"zero\": %.0f, \"random\": %.0f}, "
                   "\"mfma_bf16_16x16x32_random_again\": {\"tflops\": %.1f, \"clock_mhz\": %.0f},\n", th[0], th[1], mh[0], mh[1], tb, mbz);
        }
        CK(hipFree(d_clk));
    }
    // HBM copy: 1 GiB read + 1 GiB written
    {
        const long n = (1L << 30) / 16;
        f32x4 *s, *d;
        CK(hipMalloc(&s, n * 16)); CK(hipMalloc(&d, n * 16));
        CK(hipMemset(s, 1, n * 16));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, s, d, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("  \"hbm_copy_gbps\": %.0f, \"hbm_nominal_gbps\": 8000,\n", 2.0 * n * 16 / best * 1e-6);
        // deeper streams: U requests in flight per lane, 256 x 16 workgroups
        auto timeit = [&](auto launch) -> float {
            float bestl = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                float ms;
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                bestl = std::min(bestl, ms);
            }
            return bestl;
        };
        const float c4 = timeit([&] { hipLaunchKernelGGL(copy4u<4>, dim3(256 * 16), dim3(256), 0, 0, s, d, n); });
        const float c8 = timeit([&] { hipLaunchKernelGGL(copy4u<8>, dim3(256 * 16), dim3(256), 0, 0, s, d, n); });
        const float r8 = timeit([&] { hipLaunchKernelGGL(read4u<8>, dim3(256 * 16), dim3(256), 0, 0, s, d_out, n); });
        const float r16 = timeit([&] { hipLaunchKernelGGL(read4u<16>, dim3(256 * 16), dim3(256), 0, 0, s, d_out, n); });
        const float w1 = timeit([&] { hipLaunchKernelGGL(write4, dim3(256 * 16), dim3(256), 0, 0, d, n); });
        printf("  \"hbm_copy_gbps_4_in_flight\": %.0f, \"hbm_copy_gbps_8_in_flight\": %.0f, \"hbm_read_gbps_8_in_flight\": %.0f, \"hbm_read_gbps_16_in_flight\": %.0f, \"hbm_write_gbps\": %.0f,\n",
               2.0 * n * 16 / c4 * 1e-6, 2.0 * n * 16 / c8 * 1e-6, 1.0 * n * 16 / r8 * 1e-6, 1.0 * n * 16 / r16 * 1e-6, 1.0 * n * 16 / w1 * 1e-6);
        CK(hipFree(s)); CK(hipFree(d));
    }
    // L2-resident fragment loads: 288 KB buffer (one layer's Wqkv in bf16), every CU streaming
    {
        const int nfrag = 288;
        u32x4* w; unsigned long long* d_cyc; unsigned* d_sink;
        CK(hipMalloc(&w, nfrag * 1024)); CK(hipMemset(w, 1, nfrag * 1024));
        CK(hipMalloc(&d_cyc, 512 * 8 * 8)); CK(hipMalloc(&d_sink, 512 * 512 * 4));
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int waves = cfg == 0 ? 4 : 8, grid = cfg == 2 ? 512 : 256, iters = 64;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(l2_frag, dim3(grid), dim3(64 * waves), 0, 0, w, nfrag, d_cyc, d_sink, iters); CK(hipDeviceSynchronize()); }
            std::vector<unsigned long long> h(grid * waves);
            CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> per;
            for (int b = 0; b < grid; ++b) { unsigned long long mx = 0; for (int wv = 0; wv < waves; ++wv) mx = std::max(mx, h[b * waves + wv]); per.push_back((double)mx / (iters * 12.0 * waves * (grid / 256))); }
            std::sort(per.begin(), per.end());
            printf("  \"l2_frag_load_cycles_per_KB_per_cu_w%d_g%d\": %.1f,\n", waves, grid, per[per.size() / 2]);
        }
    }
    // LDS access patterns
    {
        unsigned long long* d_cyc; unsigned* d_sink;
        CK(hipMalloc(&d_cyc, 256 * 8 * 8)); CK(hipMalloc(&d_sink, 256 * 512 * 4));
        printf("  \"lds_cycles_per_wave_instruction\": {\n");
        RUN(B128_32_S144, false); RUN(B128_32_SWZ128, false); RUN(B128_16_S144, false); RUN(B128_16_SWZ128, false);
        RUN(TR_32_S144, false); RUN(TR_32_SWZ128, false); RUN(TR_32_S192, false); RUN(TR_32_S144_R4, false);
        RUN(TR_16_S144, false); RUN(TR_16_S144_PERM, false); RUN(B128_32_S192_SWZ, false); RUN(B128_32_S208, false);
        RUN(W64_32_S144, false); RUN(W64_32_SWZ128, false); RUN(W64_16_S144, true);
        printf("  }\n");
    }
    printf("}\n");
    return 0;
}
