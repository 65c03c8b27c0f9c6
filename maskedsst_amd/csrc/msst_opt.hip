// Fused AdamW over the flat parameter buffer (one launch for all ~350 parameter tensors).
// Semantics of torch.optim.AdamW as configured by the reference (src/utils.py:36-45; pretrain.py:69):
//   g' = clamp(g * gscale, -clamp, clamp)      (pretrain.py:71-73 value clamp; gscale = 1/world for DP mean)
//   p  = p * (1 - lr*wd);  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <atomic>
#include <algorithm>
#include <chrono>
#include "msst_dev.h"
#include "msst_kernels.h"
#include <math.h>

namespace msst {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float clampv, float gscale) {
    const long n4 = n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = gg[e] * gscale;
            if (clampv > 0.f) gr = fminf(fmaxf(gr, -clampv), clampv);
            float pv = pp[e] * (1.f - lr * wd);
            const float mn = b1 * mm[e] + (1.f - b1) * gr;
            const float vn = b2 * vv[e] + (1.f - b2) * gr * gr;
            const float denom = sqrtf(vn) / bc2_sqrt + eps;
            pv -= step_size * (mn / denom);
            pp[e] = pv; mm[e] = mn; vv[e] = vn;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gr = g[i] * gscale;
        if (clampv > 0.f) gr = fminf(fmaxf(gr, -clampv), clampv);
        float pv = p[i] * (1.f - lr * wd);
        const float mn = b1 * m[i] + (1.f - b1) * gr;
        const float vn = b2 * v[i] + (1.f - b2) * gr * gr;
        pv -= step_size * (mn / (sqrtf(vn) / bc2_sqrt + eps));
        p[i] = pv; m[i] = mn; v[i] = vn;
    }
}

int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                 float wd, int step, float clamp, float gscale, hipStream_t st) {
    if (n <= 0) return 0;
    const float bc1 = 1.f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    ProfScope ps(K_ADAMW, st);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1,
                       bc2_sqrt, clamp, gscale);
    return (int)hipGetLastError();
}

// CU-occupancy probe for the data-parallel overlap: `nblocks` workgroups that do nothing but hold a CU each
// -- they declare all 160 KB of its LDS, so no workgroup that uses LDS (every MFMA kernel here) fits beside one, like a CU
// lost to a communication kernel's channel workgroup -- until `us` microseconds of the constant-rate device clock have passed.
// (__launch_bounds__ only CAPS the register allocation; the handful of registers this kernel uses excludes nobody.)  bench.py --cu-thief runs it on a
// side stream under the backward to measure what the static grids of the MFMA kernels lose to RCCL's channels.
__global__ __launch_bounds__(256, 2) void cu_thief_kernel(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        acc = acc * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0x12345u) sink[0] = acc;   // never true in practice; keeps the loop alive
}

int launch_cu_thief(int nblocks, int us, unsigned* sink, hipStream_t st) {
    if (nblocks < 1 || us < 1) return 0;
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cu_thief_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // wall_clock64 ticks at 100 MHz on gfx9
    hipLaunchKernelGGL(cu_thief_kernel, dim3(nblocks), dim3(256), 160 * 1024, st, (unsigned long long)us * 100ull, sink);
    return (int)hipGetLastError();
}

// ---- box probe (bench.py: VERDICT r5 item 4) ----
// What THIS box sustains, measured in the run that quotes it: (1) back-to-back v_mfma_f32_32x32x16_bf16 on hashed full-range
// operands, two waves per SIMD on every CU -- the chip clocks to its power budget, so the shader clock such a stream holds (and with
// it the attainable MFMA rate) differs from box to box and from the 2.4 GHz nominal; the clock is read as shader cycles
// (s_memtime domain) per tick of the constant 100 MHz wall clock; (2) a read-only stream over `bytes` of scratch (16 requests of
// 16 bytes in flight per lane).  Same loops as tools/peak_microbench.hip.
typedef float pb_f32x16 __attribute__((ext_vector_type(16)));
typedef float pb_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pb_rand(unsigned x) {
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

__global__ __launch_bounds__(256) void probe_mfma_kernel(float* out, int iters, unsigned long long* clk) {
    pb_bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 16 + i;
        a[i] = (__bf16)pb_rand(id);
        b[i] = (__bf16)pb_rand(id + 8);
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    pb_f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
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
}

__global__ __launch_bounds__(256) void probe_read_kernel(const pb_f32x4* __restrict__ src, float* __restrict__ sink, long n) {
    constexpr int U = 16;
    const long stride = (long)gridDim.x * 256;
    pb_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        pb_f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc = acc + v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// out[0] = TFLOP/s of the MFMA loop (median of the last reps), out[1] = shader clock in MHz while it ran, out[2] = GB/s of the read
// stream (best rep), out[3] = seconds the probe took.  scratch: >= 1 MiB (MFMA sinks in its first 1 MiB; the read stream covers all
// of it: give it >= 1 GiB so that the 256 MB memory-side cache does not serve it).  Synchronises the stream.
int launch_box_probe(double* out, void* scratch, long bytes, hipStream_t st) {
    if (!out || !scratch || bytes < (1L << 20)) return MSST_ERR_BADARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MSST_ERR_BADARG;
    const auto wall0 = std::chrono::steady_clock::now();
    float* sink = (float*)scratch;
    unsigned long long* clk = (unsigned long long*)((char*)scratch + 512 * 256 * 4);
    const int grid = 512, iters = 12000, reps = 12;
    const double flops = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    float ms[reps];
    double mhz = 0.0;
    int rc = 0;
    for (int r = 0; r < reps && !rc; ++r) {
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(grid), dim3(256), 0, st, sink, iters, clk);
        hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms[r], e0, e1) != hipSuccess) rc = MSST_ERR_BADARG;
    }
    if (!rc) {
        unsigned long long h[2] = {0, 1};
        if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = MSST_ERR_BADARG;
        mhz = (double)h[0] / ((double)h[1] / 100.0);
        // the clock settles over the first launches (the power controller reacts in milliseconds): median of the second half
        float tail[reps / 2];
        for (int i = 0; i < reps / 2; ++i) tail[i] = ms[reps / 2 + i];
        std::sort(tail, tail + reps / 2);
        out[0] = flops / tail[reps / 4] * 1e-9;
        out[1] = mhz;
    }
    if (!rc) {
        const long n = bytes / 16;
        float best = 1e30f;
        for (int r = 0; r < 4 && !rc; ++r) {
            float t;
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(probe_read_kernel, dim3(256 * 16), dim3(256), 0, st, (const pb_f32x4*)scratch, sink, n);
            hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) rc = MSST_ERR_BADARG;
            if (t < best) best = t;
        }
        const long covered = n / (16L * 256 * 16 * 256) * (16L * 256 * 16 * 256);
        out[2] = (double)covered * 16 / best * 1e-6;
    }
    out[3] = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (!rc) rc = (int)hipGetLastError();
    return rc;
}

}  // namespace msst
