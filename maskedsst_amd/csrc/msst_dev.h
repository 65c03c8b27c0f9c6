// Device-side primitives shared by the MaskedSST gfx950 kernels.
//
// Two precision policies drive one set of kernel templates:
//   PF32  -- exact fp32 operands on v_mfma_f32_16x16x4_f32   (parity mode, 1e-4 vs the CPU oracle)
//   PBF16 -- bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate (throughput mode)
// Both instructions share the C/D fragment layout (col = lane&15, row = 4*(lane>>4)+reg), so all
// epilogues (softmax, LayerNorm, GELU, residuals) are written once.
//
// GEMM convention used everywhere ("NT"): C[i][j] += sum_k A[i][k] * B[j][k].
//   A fragment: lane holds A[i0 + (lane&15)][k-slice (lane>>4)]
//   B fragment: lane holds B[j0 + (lane&15)][k-slice (lane>>4)]
//   C fragment: lane holds C[i0 + 4*(lane>>4) + r][j0 + (lane&15)], r = 0..3
// so the cheap ("natural") store of a C tile is OUT[j][i..i+3]: 4 consecutive i for one j.
// Weights are normally the A operand and activations the B operand, which makes activation
// outputs row-major [token][feature] with 16-byte stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msst {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;
typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));

// fp32 -> bf16, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ s16x4 f2bf4(f32x4 c) {
    return __builtin_bit_cast(s16x4, __builtin_convertvector(c, hbf16x4));
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> IEEE half, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950): the fp16-operand forward (MSST_FWD_HALF)
typedef _Float16 hf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16_t f2h(float f) {
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ s16x4 f2h4(f32x4 c) {
    return __builtin_bit_cast(s16x4, __builtin_convertvector(c, hf16x4));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float gelu_erf(float x);
__device__ __forceinline__ float gelu_erf_grad(float x);
__device__ __forceinline__ float gelu_fast(float x);
__device__ __forceinline__ float gelu_fast_grad(float x);
__device__ __forceinline__ void gelu_fast_both(float x, float& g, float& dg);

// ------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------
struct PF32 {
    typedef float elem;
    typedef float frag;
    static constexpr int KS = 4;    // k per MFMA
    static constexpr int UNROLL = 2;          // k-loop unroll (24 / 16 steps per GEMM)
    static constexpr int WAVES_PER_SIMD = 1;  // launch-bounds occupancy target (LDS allows 1 WG/CU)
    static constexpr int WAVES_BWD_ATTN = 1;
    static constexpr int WAVES_BWD_MLP = 1;
    static constexpr int PADE = 4;  // LDS row padding in elements (16 bytes)
    static __device__ __forceinline__ elem cvt(float f) { return f; }
    static __device__ __forceinline__ float up(elem e) { return e; }
    static __device__ __forceinline__ frag ones() { return 1.0f; }   // operand fragment of ones (column sums on the matrix cores)
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float gelu(float x) { return gelu_erf(x); }
    static __device__ __forceinline__ float gelu_grad(float x) { return gelu_erf_grad(x); }
    static __device__ __forceinline__ void gelu_both(float x, float& g, float& dg) { g = gelu_erf(x); dg = gelu_erf_grad(x); }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // k-contiguous operand: element (row, k) at p[row*ld + k]
    static __device__ __forceinline__ frag ld_kc(const elem* p, int ld) {
        const int l = lane_id();
        return p[(l & 15) * ld + (l >> 4)];
    }
    // k-strided operand: element (row, k) at p[k*ld + row]
    static __device__ __forceinline__ frag ld_ks(const elem* p, int ld) {
        const int l = lane_id();
        return p[(l >> 4) * ld + (l & 15)];
    }
    // (the bf16 policy has k-permuted variants of the LDS operand loads; in fp32 they are the plain ones)
    static __device__ __forceinline__ frag ld_kc_perm(const elem* p, int ld) { return ld_kc(p, ld); }
    static __device__ __forceinline__ frag ld_ks_perm(const elem* p, int ld) { return ld_ks(p, ld); }
    // global weight fragment: rows row0..+15, k-slice starting at k0 of a [R][K] matrix (fp32: row-major)
    static __device__ __forceinline__ frag ld_w(const elem* w, int K, int row0, int k0) {
        return ld_kc(w + (long)row0 * K + k0, K);
    }
    // natural store of a C tile: OUT[j][i0..i0+3], p -> OUT[j0][i0]
    static __device__ __forceinline__ void st_nat(elem* p, int ld, f32x4 c) {
        const int l = lane_id();
        *reinterpret_cast<f32x4*>(p + (l & 15) * ld + 4 * (l >> 4)) = c;
    }
    // transposed store: OUT[i][j], p -> OUT[i0][j0]
    static __device__ __forceinline__ void st_tr(elem* p, int ld, f32x4 c) {
        const int l = lane_id();
        elem* q = p + (4 * (l >> 4)) * ld + (l & 15);
        q[0] = c[0]; q[ld] = c[1]; q[2 * ld] = c[2]; q[3 * ld] = c[3];
    }
};

struct PBF16 {
    typedef bf16_t elem;
    typedef s16x8 frag;
    static constexpr int KS = 32;
#ifndef MSST_BF_UNROLL
#define MSST_BF_UNROLL 2
#endif
    static constexpr int UNROLL = MSST_BF_UNROLL;   // k-loops have 2-3 steps; full unrolling only inflates registers
    static constexpr int WAVES_PER_SIMD = 2;
    static constexpr int WAVES_BWD_ATTN = 2;  // 2 workgroups per CU (LDS 76 KB each)
#ifndef MSST_MLP2
#define MSST_MLP2 1
#endif
    static constexpr int WAVES_BWD_MLP = MSST_MLP2 ? 2 : 1;   // MSST_MLP2: two workgroups per CU (w1T fragments from L2, no row prefetch)
    static constexpr int PADE = 8;  // 16 bytes
    static __device__ __forceinline__ elem cvt(float f) { return f2bf(f); }
    static __device__ __forceinline__ float up(elem e) { return bf2f(e); }
    static __device__ __forceinline__ frag ones() {
        s16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (short)0x3F80;   // bf16 1.0
        return r;
    }
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }   // v_exp_f32; probabilities are rounded to bf16 anyway
    static __device__ __forceinline__ float gelu(float x) { return gelu_fast(x); }
    static __device__ __forceinline__ float gelu_grad(float x) { return gelu_fast_grad(x); }
    static __device__ __forceinline__ void gelu_both(float x, float& g, float& dg) { gelu_fast_both(x, g, dg); }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag ld_kc(const elem* p, int ld) {
        const int l = lane_id();
        return *reinterpret_cast<const s16x8*>(p + (l & 15) * ld + 8 * (l >> 4));
    }
    // global weight fragment of a [R][K] matrix.  bf16 weights are stored FRAGMENT-PACKED by
    // msst_prep_weights: the 1 KB fragment (16 rows x 32 k) f = (row0/16)*(K/32) + k0/32 is contiguous
    // in lane order, so one global_load_dwordx4 per lane reads 8 full cache lines instead of 16 halves.
    static __device__ __forceinline__ frag ld_w(const elem* w, int K, int row0, int k0) {
        // Buffer load: descriptor (w) and fragment offset live in SGPRs (row0 / k0 are wave uniform in every
        // caller), the only VGPR is the lane offset shared by all weight loads -- no 64-bit address pair per fragment.
        const int f = (row0 >> 4) * (K >> 5) + (k0 >> 5);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(w), 0, 0x7fffffff, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_id() * 16, f * 1024, 0);
        return __builtin_bit_cast(s16x8, v);
    }
    // k-strided operand in LDS: element (row, k) at p[k*ld + row].  Two ds_read_b64_tr_b16: the 16
    // lanes of a group fetch a [4 k][16 row] block (lane i supplies the address of k-row i>>2,
    // column chunk (i&3)*4) and each receives column i = its 4 consecutive-k values
    // (semantics probed on gfx950 with tools/probe_tr.hip).  ld must be a multiple of 4.
    static __device__ __forceinline__ frag ld_ks(const elem* p, int ld) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int l = lane_id();
        const elem* q = p + (8 * (l >> 4) + ((l & 15) >> 2)) * ld + (l & 3) * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * ld));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    // ---- operands for an MFMA whose OTHER operand is pack2(C tile 2m, C tile 2m+1) of a previous MFMA ----
    // Such a register-built operand holds, in lane (c, g), the contraction indices {4g..4g+3} and {16+4g..16+4g+3}
    // of the 32-chunk (a permutation of the natural 8g..8g+7, harmless when both operands use it).
    // k-contiguous LDS operand in that order: p -> element (row 0, first k of the chunk)
    static __device__ __forceinline__ frag ld_kc_perm(const elem* p, int ld) {
        const int l = lane_id();
        const elem* q = p + (l & 15) * ld + 4 * (l >> 4);
        const s16x4 lo = *reinterpret_cast<const s16x4*>(q);
        const s16x4 hi = *reinterpret_cast<const s16x4*>(q + 16);
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    // k-strided LDS operand (element (row, k) at p[k*ld + row]) in that order
    static __device__ __forceinline__ frag ld_ks_perm(const elem* p, int ld) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int l = lane_id();
        const elem* q = p + (4 * (l >> 4) + ((l & 15) >> 2)) * ld + (l & 3) * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 16 * ld));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    static __device__ __forceinline__ frag pack2(f32x4 lo, f32x4 hi) {
        const s16x4 a = f2bf4(lo), b = f2bf4(hi);
        s16x8 r;
        r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
        r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
        return r;
    }
    static __device__ __forceinline__ void st_nat(elem* p, int ld, f32x4 c) {
        const int l = lane_id();
        *reinterpret_cast<s16x4*>(p + (l & 15) * ld + 4 * (l >> 4)) = f2bf4(c);
    }
    static __device__ __forceinline__ void st_tr(elem* p, int ld, f32x4 c) {
        const int l = lane_id();
        elem* q = p + (4 * (l >> 4)) * ld + (l & 15);
        q[0] = f2bf(c[0]); q[ld] = f2bf(c[1]); q[2 * ld] = f2bf(c[2]); q[3 * ld] = f2bf(c[3]);
    }
};

// ------------------------------------------------------------------------------------------
// reductions over the 4 lane groups that share one C-fragment column (lanes l, l^16, l^32, l^48)
// ------------------------------------------------------------------------------------------
// v_permlane16_swap / v_permlane32_swap (gfx950): with vdst == src == v the two results are
// (even rows | even rows) and (odd rows | odd rows) resp. (low half | low half), (high | high), so
// their sum / max equals v (+) shfl_xor(v, 16 / 32) on every lane -- VALU only, no LDS round trip.
__device__ __forceinline__ float colgroup_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// (v_max_f32 spelled out: fmaxf on the swapped words -- opaque integers to the compiler -- gets a canonicalising v_max x, x, x in
// front of each operand, four more instructions on the softmax's dependent chain)
__device__ __forceinline__ float vmax_raw(float x, float y) {
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    return m;
}
__device__ __forceinline__ float colgroup_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// reduction over the 16 lanes that share one lane group (lanes with equal lane>>4)
__device__ __forceinline__ float rowgroup_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// sum over the 4 lanes of a quad (lanes l, l^1, l^2, l^3), result on every lane: two DPP quad_perm moves on the
// VALU -- __shfl_xor compiles to ds_bpermute_b32, one LDS round trip per step
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

// Software pipeline over N fully unrolled steps: the operands of step s are requested D steps before its MFMAs
// (register slots are the caller's, indexed s % (D + 1)).  The scheduling fences pin the order: left alone, the
// compiler sinks every LDS read next to its use (read / s_waitcnt lgkmcnt(0) / MFMA) once registers are tight.
#define MSST_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N, int D, class Issue, class Exec>
__device__ __forceinline__ void swpipe(Issue issue, Exec exec) {
#pragma unroll
    for (int s = 0; s < D && s < N; ++s) issue(s);
#pragma unroll
    for (int s = 0; s < N; ++s) {
        if (s + D < N) issue(s + D);
        MSST_SCHED_FENCE();
        exec(s);
        MSST_SCHED_FENCE();
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx gelu_erf
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), which
// would serialise the global prefetches that are deliberately left in flight across phases.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// async global -> LDS copy of one 1 KB fragment-packed weight fragment (16 B per lane, LDS image is
// lane-linear = exactly the packed layout).  Completion is tracked by vmcnt.
__device__ __forceinline__ void dma_frag(const void* gsrc_frag, void* lds_dst_frag) {
    const char* src = reinterpret_cast<const char*>(gsrc_frag) + lane_id() * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_frag, 16, 0, 0);
}
// Same copy, but opaque to the compiler: a builtin LDS-DMA makes it wait (vmcnt(0)) before ANY later LDS read,
// which serialises the copy with the GEMM it is meant to run under.  The caller orders the consumers itself
// with wait_vm0() + lds_barrier().  lds_dst_frag must be wave uniform.
__device__ __forceinline__ void dma_frag_async(const void* gsrc_frag, void* lds_dst_frag) {
    const char* src = reinterpret_cast<const char*>(gsrc_frag) + lane_id() * 16;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dst_frag;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "memory", "m0");
}
// Same, wave-uniform fragment address in SGPRs + a 32-bit lane offset (lane * 16) in one VGPR shared by all copies:
// no 64-bit address pair per fragment for the compiler to hoist out of the tile loop and spill.
__device__ __forceinline__ void dma_frag_async_s(const void* gsrc_frag_uniform, void* lds_dst_frag, int lane_off) {
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dst_frag;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(lane_off), "s"(gsrc_frag_uniform) : "memory", "m0");
}
// Pull the 128-byte line holding p towards this XCD's L2 without tying up a register: a 4-byte LDS-DMA into a
// 256-byte scratch area nobody reads.  Counts in vmcnt like any load (in-order return), so place it where the
// next vmcnt wait belongs to a request of similar latency.
__device__ __forceinline__ void l2_touch(const void* p, void* lds_dummy) {
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dummy;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"(dst), "v"(p) : "memory", "m0");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// kernel-study cycle stamps are compiled in only with -DMSST_STAMPS (tools/stamps*.py); production builds carry none
#ifdef MSST_STAMPS
#define STAMP(i) do { if (stamp_on) { a.stamps[(i)] = __builtin_readcyclecounter(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// Dropout (reference sites vit_spatial_spectral.py:38,40,57,62).  Counter-based, stateless: the keep
// mask of 4 consecutive elements is a pure function of (seed, site key, element-group index), so the
// backward regenerates the forward's mask from the same indices and nothing is stored.
// keep <=> 16 random bits >= thr, thr = round(p * 65536); kept values are scaled by 1 / (1 - p).
// (tests/dropout.py holds the bit-identical numpy restatement used to feed the oracle the same masks.)
// ------------------------------------------------------------------------------------------
struct Drop {
    unsigned seed, thr;   // thr == 0: dropout off
    float scale;
    int layer;            // block index (0 .. 2*depth-1)
};
// four consecutive features of a saved mid-residual row as fp32: the row is fp32 (16-byte load) or, X1B (MSST_X1_BF16), bf16
// (8-byte load, widened by a shift: exact)
template <bool X1B>
__device__ __forceinline__ f32x4 ld_x1_4(const float* base, long tok, int m0) {
    if constexpr (X1B) {
        const s16x4 h = *reinterpret_cast<const s16x4*>(reinterpret_cast<const unsigned short*>(base) + tok * 96 + m0);
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __builtin_bit_cast(float, (unsigned)(unsigned short)h[i] << 16);
        return r;
    } else {
        return *reinterpret_cast<const f32x4*>(base + tok * 96 + m0);
    }
}

// 64 mask bits (four 16-bit fields) of element group `group`: three 32-bit multiplies (the quarter-rate
// instruction here) instead of the six of a double murmur finaliser; keep rates / field and neighbour correlations
// checked in numpy over 4M groups (|corr| < 2e-3).
__device__ __forceinline__ void drop_bits(const Drop& d, int site, unsigned group, unsigned& a, unsigned& b) {
    unsigned x = group ^ (d.seed ^ ((unsigned)(d.layer * 4 + site) * 0x9E3779B9U));
    x *= 0x9E3779B1U; x ^= x >> 15;
    x *= 0x85EBCA6BU; x ^= x >> 13;
    a = x;
    b = x * 0xC2B2AE35U; b ^= b >> 16;
}
// site: 1 = attention probabilities, 2 = to_out, 3 = MLP hidden, 4 = MLP out
// The field tests are written on the full word: hi field >= thr <=> word >= thr << 16, lo field >= thr <=> word << 16 >= thr << 16
// (same decisions, no field extraction: one shift for the two lo fields instead of two ands and two shifts).
__device__ __forceinline__ f32x4 drop4(const Drop& d, int site, unsigned group, f32x4 v) {
    unsigned a, b;
    drop_bits(d, site, group, a, b);
    const unsigned t16 = d.thr << 16;
    f32x4 r;
    r[0] = (a << 16) >= t16 ? v[0] * d.scale : 0.f;
    r[1] = a >= t16 ? v[1] * d.scale : 0.f;
    r[2] = (b << 16) >= t16 ? v[2] * d.scale : 0.f;
    r[3] = b >= t16 ? v[3] * d.scale : 0.f;
    return r;
}
// same masks, kept values NOT scaled (the caller folded 1 / (1 - p) into a factor it applies anyway)
__device__ __forceinline__ f32x4 drop4_noscale(const Drop& d, int site, unsigned group, f32x4 v) {
    unsigned a, b;
    drop_bits(d, site, group, a, b);
    const unsigned t16 = d.thr << 16;
    f32x4 r;
    r[0] = (a << 16) >= t16 ? v[0] : 0.f;
    r[1] = a >= t16 ? v[1] : 0.f;
    r[2] = (b << 16) >= t16 ? v[2] : 0.f;
    r[3] = b >= t16 ? v[3] : 0.f;
    return r;
}
// same, also returning the four keep decisions as bits 0..3 (the attention backward reuses them for dP)
__device__ __forceinline__ f32x4 drop4_keep(const Drop& d, int site, unsigned group, f32x4 v, unsigned& keep) {
    unsigned a, b;
    drop_bits(d, site, group, a, b);
    const bool k0 = (a & 0xffffU) >= d.thr, k1 = (a >> 16) >= d.thr, k2 = (b & 0xffffU) >= d.thr, k3 = (b >> 16) >= d.thr;
    keep = (unsigned)k0 | ((unsigned)k1 << 1) | ((unsigned)k2 << 2) | ((unsigned)k3 << 3);
    f32x4 r;
    r[0] = k0 ? v[0] * d.scale : 0.f; r[1] = k1 ? v[1] * d.scale : 0.f;
    r[2] = k2 ? v[2] * d.scale : 0.f; r[3] = k3 ? v[3] * d.scale : 0.f;
    return r;
}
__device__ __forceinline__ f32x4 drop4_bits(const Drop& d, unsigned keep, f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (keep >> i) & 1u ? v[i] * d.scale : 0.f;
    return r;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), ~15 VALU ops with v_rcp / v_exp instead of the
// ~250-cycle libm erff; used by the bf16 kernels (the fp32 parity kernels keep erff)
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // (v_rcp_f32, 1 ulp: __frcp_rn is a ten-instruction IEEE division)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_fast_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// value and derivative together (MLP backward): the erf's exp(-(x / sqrt 2)^2) IS the density's exp(-x^2 / 2) -- one v_exp, not two
__device__ __forceinline__ void gelu_fast_both(float x, float& g, float& dg) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __expf(-ax * ax);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
    g = x * cdf;
    dg = cdf + x * (0.39894228040143267794f * e);
}

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// ------------------------------------------------------------------------------------------
// tile -> token-row mapping.  A tile is 64 rows = TS whole sequences of L tokens (TS = 64 / L).
//   mode 0 (spatial):  sequence q = (b, c);  row (q, p) -> token b*T + c*N + p   = q*N + p
//   mode 1 (spectral): sequence q = (b, n);  row (q, p) -> token b*T + p*N + n
// Rows past TS*L, or sequences past nseq, are padding: loaded as zero, never stored.
// ------------------------------------------------------------------------------------------
struct TileMap {
    int mode, L, TS, N, T, nseq;
    int nshift;   // log2(N) when N is a power of two (spectral token index without an integer division), else -1
    // loop-invariant part of a row: (sequence slot s, position p); s < 0 marks a padding row
    __device__ __forceinline__ int2 row_sp(int r) const {
        const int s = r / L;
        return s >= TS ? make_int2(-1, 0) : make_int2(s, r - s * L);
    }
    __device__ __forceinline__ long token_sp(int tile, int2 sp) const {
        if (sp.x < 0) return -1;
        const int q = tile * TS + sp.x;
        if (q >= nseq) return -1;
        if (mode == 0) return (long)q * N + sp.y;
        const int b = nshift >= 0 ? (q >> nshift) : q / N, n = q - b * N;
        return (long)b * T + (long)sp.y * N + n;
    }
    __device__ __forceinline__ long token(int tile, int r) const {
        const int s = r / L;
        if (s >= TS) return -1;
        const int q = tile * TS + s;
        if (q >= nseq) return -1;
        const int p = r - s * L;
        if (mode == 0) return (long)q * N + p;
        const int b = q / N, n = q - b * N;
        return (long)b * T + (long)p * N + n;
    }
};

}  // namespace msst
