// bf16 row-local seam of the backward, fused ACROSS blocks (round 4): the LN1 backward of block i and the MLP-half backward
// of block i - 1 in one launch (reference vit_spatial_spectral.py:22-44,100-104: PreNorm, FeedForward and the two residual
// adds are row-local; a15 of SURVEY.md section 8).
//
//   dx_i     = dx1_i + LN1_bwd(sum_p part_i[p]; x_i)                (block_bwd_ln1_kernel's job)
//   dx1_{i-1} = dx_i + LN2_bwd(W1^T dhpre; x1_{i-1}),  dW1 / dW2 / ...   (block_bwd_mlp_kernel's job, dy = dx_i)
//
// Unfused, dx_i is written by one HBM-roofline kernel (126 MB at B = 256) and read back by the next; fused, it lives in one
// 24.6 KB LDS tile: 1.10 GB -> 0.82 GB per block boundary.
//
// The round-1 attempt at this fusion ran both halves on the SAME waves and needed 512 registers (the LN half's whole-tile
// prefetch on top of the MLP half's accumulators), so every spill reload waited for the prefetch in flight.  Here the two
// halves are ROLES of one 512-thread workgroup (one per CU):
//   * waves 4-7 ("L"): thread <-> (row, 12 features) as in block_bwd_ln1; they own every HBM read of the LN half -- the four
//     bf16 head-pair partials, x_i, dx1_i: 1.5 KB per row, requested a whole tile ahead into 96 registers -- and are one
//     tile AHEAD of the M waves: while those run walk step k, the L waves compute dx of step k + 1 into the other half of a
//     double-buffered fp32 LDS tile and re-request each 8-row pass for step k + 2 as soon as it is consumed; behind barrier B1
//     they copy the finished dx1 rows of step k (left in LDS by the M waves) out as contiguous 1 KB pieces, together with
//     their dropped bf16 copy for the attention half.
//   * waves 0-3 ("M"): wave <-> 16 rows, the MLP backward exactly as block_bwd_mlp_kernel (LN2 / MLP recompute from x1,
//     dh, d(LN2 out), LN2 backward, dW1 / dW2 / bias gradients on the matrix cores), except that dy comes from the LDS tile
//     and all three weight matrices sit in LDS.  Their only HBM read is x1 (384 B per row, requested one phase ahead); they
//     store nothing to HBM.
// Two barriers per tile, executed by both roles.  Gradient slabs: one MLP slab and one LN1 slab per workgroup (256 instead
// of 512 + 256).
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"

#ifndef MSST_B5_REV
#define MSST_B5_REV 1   // walk the tiles from the last to the first: the attention kernel wrote its last partials most recently (memory-side cache)
#endif
#ifndef MSST_B5_X1EARLY
#define MSST_B5_X1EARLY 0   // 1: the M waves request the next step's x1 rows at the top of phase 1 (a whole tile ahead): +24 live registers, spills
#endif
#ifndef MSST_B5_PASSPIPE
#define MSST_B5_PASSPIPE 1
#endif
#ifndef MSST_B5_PRIO
#define MSST_B5_PRIO 0   // 1: the M waves (the critical chain) run at s_setprio 1 against the L wave that shares their SIMD
#endif

#ifdef MSST_STAMPS
#define B5_STAMP(i) do { if (stamp_on) a.stamps[16 * wave + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define B5_STAMP(i) do { } while (0)
#endif

namespace msst {

namespace {

typedef PBF16 P;
typedef bf16_t elem;
typedef s16x8 frag;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct LmSmem {
    static constexpr int LDX = 96 + 8, LDH = 64 + 8, LDF = 100;
    elem xn2[64][LDX];
    elem dy[64][LDX];
    elem h[64][LDH];
    elem dhp[64][LDH];
    float dxf[2][64][LDF];   // dx_i (fp32) of the tile the M waves process now / next, by walk-step parity
    float out[64][LDF];      // dx1 of block i - 1 (fp32) of the tile the M waves just finished: the L waves copy it out as whole rows
    float gam1[96];          // ln1_g of block i
    float bet1[96];          // (XNB) ln1_b / ln1_g of block i
    float ig1[96];           // (XNB) 1 / ln1_g
    int qt[8];               // (QUEUE) tile of walk step k at [k & 7]
    float lnp[256];          // ln2_g | ln2_b | b1 of block i - 1
    char wl[36 * 1024];      // [w1 12 | w2T 12 | w1T 12] fragments of 1 KB (block i - 1)
};

// sum over 8 consecutive lanes (the threads of a row in the L mapping), on every lane
__device__ __forceinline__ float oct_sum5(float v) {
    v = quad_sum(v);
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
}

}  // namespace

// NP: partial buffers the attention backward of block i wrote (one per head pair: heads / 2; or one per head)
// QUEUE (data parallel, opt-in): tiles drawn from one agent-scope counter instead of the static partition tile = workgroup +
// k grid, so that a workgroup whose CU is held by a communication kernel draws fewer tiles instead of running its whole share
// behind the others (see msst_bwd4.hip).  L wave 0 draws five walk steps ahead and publishes through an eight-entry LDS ring.
// XNB (MSST_LN1_FROM_XN, round 6): xhat of LN1 = (saved bf16 LN1 row - beta) / gamma and rstd from the forward's statistics buffer
// instead of the fp32 block input x re-read and re-normalised: 196 instead of 384 bytes per row, no mean / variance reductions.
template <int NP, bool DROP, bool QUEUE, bool X1B, bool XNB>
__global__ __launch_bounds__(512, 2) void block_bwd_ln1mlp_kernel(LnMlpArgs a) {
    typedef LmSmem SM;
    constexpr int KS = 32, LDX = SM::LDX, LDH = SM::LDH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int ntiles = (int)((a.ntok + 63) / 64);
    const int G = (int)gridDim.x;
    // k-th tile of this workgroup's walk
    const int nmine = (ntiles - 1 - (int)blockIdx.x) / G + 1;   // static partition: blockIdx.x < ntiles (launcher)
    // tile of walk step k (static: valid for k < nmine; queue: any value >= ntiles ends the walk)
    auto tile_at = [&](int k) -> int {
        if (QUEUE) return sm.qt[k & 7];
        return MSST_B5_REV ? (int)blockIdx.x + (nmine - 1 - k) * G : (int)blockIdx.x + k * G;
    };
    auto valid_step = [&](int k) -> bool { return QUEUE ? sm.qt[k & 7] < ntiles : k < nmine; };
    auto tile_of_draw = [&](int d) -> int { return d < ntiles ? (MSST_B5_REV ? ntiles - 1 - d : d) : 0x7fffffff; };
    int qpend = 0;   // (QUEUE, lane 0 of L wave 0) the draw in flight
    if (QUEUE && tid == 256) {
        const int r = __hip_atomic_fetch_add(a.queue, 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // walk steps 0 .. 4
#pragma unroll
        for (int i = 0; i < 5; ++i) sm.qt[i] = tile_of_draw(r + i);
        qpend = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // walk step 5
    }

    // ---- common prologue: block i - 1's small vectors and the three MLP weight matrices -> LDS ----
    if (tid < 96) { sm.lnp[tid] = a.w.ln2_g[tid]; sm.lnp[96 + tid] = a.w.ln2_b[tid]; if (tid < 64) sm.lnp[192 + tid] = a.w.b1[tid]; }
    if (tid >= 256 && tid < 352) {
        const float g_ = a.ln1_g[tid - 256];
        sm.gam1[tid - 256] = g_;
        if (XNB) { const float ig_ = 1.0f / g_; sm.ig1[tid - 256] = ig_; sm.bet1[tid - 256] = a.ln1_b[tid - 256] * ig_; }   // (bet1: beta / gamma)
    }
    {
        const elem* w1 = reinterpret_cast<const elem*>(a.w.w1);
        const elem* w1T = reinterpret_cast<const elem*>(a.w.w1T);
        const elem* w2T = reinterpret_cast<const elem*>(a.w.w2T);
#pragma unroll
        for (int i5 = 0; i5 < 5; ++i5) {
            const int f = wave * 5 + i5;
            if (f < 36) {
                const char* src = f < 12 ? reinterpret_cast<const char*>(w1) + f * 1024
                                : f < 24 ? reinterpret_cast<const char*>(w2T) + (f - 12) * 1024
                                         : reinterpret_cast<const char*>(w1T) + (f - 24) * 1024;
                dma_frag(src, sm.wl + f * 1024);
            }
        }
        wait_vm0();
    }
    if (QUEUE) __syncthreads();   // the first five tiles are in the ring

    if (wave >= 4) {
        // =====================================================================================================
        // L role: LN1 backward of block i, one tile ahead of the M waves
        // =====================================================================================================
        const int lw = wave - 4, r8 = l >> 3, p = l & 7;
        const elem* parts = reinterpret_cast<const elem*>(a.dxn_part);
        // the 12 features of a thread: 8 p .. 8 p + 7 and 64 + 4 p .. 64 + 4 p + 3 (16-byte aligned in the fp32 AND the bf16 rows)
        float dg[12], db[12], dbo[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { dg[i] = 0.f; db[i] = 0.f; dbo[i] = 0.f; }
        f32x4 xq[2][3], dq[2][3];
        u32x4 xna[2];   // (XNB, instead of xq) bf16 LN1 row: features 8 p .. 8 p + 7,
        u32x2 xnb[2];   //   features 64 + 4 p .. + 3,
        float xrs[2];   //   the row's rstd
        u32x4 pa[2][NP];
        u32x2 pb[2][NP];
        // buffer loads: descriptors in SGPRs, four 32-bit lane offsets per pass (64-bit address pairs per request cost registers the
        // whole-tile prefetch does not have)
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(XNB ? const_cast<void*>(a.xn) : (void*)const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.rstd), 0, XNB ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dx1, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(parts), 0, 0x7fffffff, 0x00020000);
        const int pstride = (int)(a.ntok * 192);   // bytes between partial buffers (launcher: nparts * ntok * 192 < 2^31)
        auto issue = [&](int k, int ps0, int ps1) {
            const bool live = valid_step(k);
            const int tile_ = live ? tile_at(k) : 0;
#pragma unroll
            for (int ps = ps0; ps < ps1; ++ps) {
                const int tok = tile_ * 64 + lw * 16 + ps * 8 + r8;
                // requested unconditionally (a definition on every path), masked at use; past the end of the walk every lane asks for row 0
                const int tokc = (tok < (int)a.ntok && live) ? tok : 0;
                const int vo1 = tokc * 384 + 32 * p, vo2 = tokc * 384 + 256 + 16 * p;
                const int vp1 = tokc * 192 + 16 * p, vp2 = tokc * 192 + 128 + 8 * p;
                if constexpr (XNB) {
                    xna[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vp1, 0, 0);
                    xnb[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, vp2, 0, 0);
                    xrs[ps] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_r, tokc * 4, 0, 0));
                } else {
                    xq[ps][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo1, 0, 0));
                    xq[ps][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo1 + 16, 0, 0));
                    xq[ps][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo2, 0, 0));
                }
                dq[ps][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo1, 0, 0));
                dq[ps][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo1 + 16, 0, 0));
                dq[ps][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo2, 0, 0));
#pragma unroll
                for (int k2 = 0; k2 < NP; ++k2) {
                    pa[ps][k2] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, vp1, k2 * pstride, 0);
                    pb[ps][k2] = __builtin_amdgcn_raw_buffer_load_b64(rs_p, vp2, k2 * pstride, 0);
                }
            }
        };
        // dx of tile k -> sm.dxf[k & 1] (the M waves read that buffer during walk step k, the other one during step k - 1)
        auto compute = [&](int k, int ps0, int ps1) {
            const int tile_ = tile_at(k);
            float (*dxf)[SM::LDF] = sm.dxf[k & 1];
#pragma unroll
            for (int ps = ps0; ps < ps1; ++ps) {
                const long tok = (long)tile_ * 64 + lw * 16 + ps * 8 + r8;
                const bool valid = tok < a.ntok;
                float v[12], d1[12], dn[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if constexpr (!XNB) v[i] = valid ? xq[ps][i >> 2][i & 3] : 0.f;
                    d1[i] = valid ? dq[ps][i >> 2][i & 3] : 0.f;
                    dn[i] = 0.f;
                }
#pragma unroll
                for (int k2 = 0; k2 < NP; ++k2) {
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const unsigned u = pa[ps][k2][w2];
                        dn[2 * w2] += __uint_as_float(u << 16);
                        dn[2 * w2 + 1] += __uint_as_float(u & 0xffff0000u);
                    }
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const unsigned u = pb[ps][k2][w2];
                        dn[8 + 2 * w2] += __uint_as_float(u << 16);
                        dn[8 + 2 * w2 + 1] += __uint_as_float(u & 0xffff0000u);
                    }
                }
                if (!valid) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) dn[i] = 0.f;
                }
                float rstd;
                if constexpr (XNB) {
                    // xhat from the forward's own LN1 row: (row - beta) / gamma = row * ig - beta * ig; the row's rstd from its statistics
                    // buffer.  (The table index is opaque per pass: left visible, the 24 table values are hoisted out of the walk and spilled.)
                    int pq = p;
                    asm volatile("" : "+v"(pq));
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int f0 = q < 2 ? 8 * pq + 4 * q : 64 + 4 * pq;
                        const f32x4 i4 = *reinterpret_cast<const f32x4*>(&sm.ig1[f0]), b4 = *reinterpret_cast<const f32x4*>(&sm.bet1[f0]);
                        const unsigned u0 = q < 2 ? xna[ps][2 * q] : xnb[ps][0], u1 = q < 2 ? xna[ps][2 * q + 1] : xnb[ps][1];
                        // (rows past the end hold row 0's finite values: their dn and rstd are zero, nothing of them is used)
                        v[4 * q] = fmaf(__uint_as_float(u0 << 16), i4[0], -b4[0]);
                        v[4 * q + 1] = fmaf(__uint_as_float(u0 & 0xffff0000u), i4[1], -b4[1]);
                        v[4 * q + 2] = fmaf(__uint_as_float(u1 << 16), i4[2], -b4[2]);
                        v[4 * q + 3] = fmaf(__uint_as_float(u1 & 0xffff0000u), i4[3], -b4[3]);
                    }
                    rstd = xrs[ps];
                } else {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) s += v[i];
                    const float mean = oct_sum5(s) * (1.f / 96.f);
                    float vs = 0.f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) { const float d = v[i] - mean; vs += d * d; }
                    rstd = rsqrtf(oct_sum5(vs) * (1.f / 96.f) + 1e-5f);
#pragma unroll
                    for (int i = 0; i < 12; ++i) v[i] = (v[i] - mean) * rstd;
                }
                float g1 = 0.f, g2 = 0.f;
                float gam[12];
                {
                    int pg = p;
                    if (XNB) asm volatile("" : "+v"(pg));
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(&sm.gam1[8 * pg]), gb = *reinterpret_cast<const f32x4*>(&sm.gam1[8 * pg + 4]),
                                gc = *reinterpret_cast<const f32x4*>(&sm.gam1[64 + 4 * pg]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gam[e] = ga[e]; gam[4 + e] = gb[e]; gam[8 + e] = gc[e]; }
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const float xh = v[i];
                    dg[i] += dn[i] * xh;
                    db[i] += dn[i];
                    dn[i] *= gam[i];
                    g1 += dn[i];
                    g2 += dn[i] * xh;
                }
                g1 = oct_sum5(g1) * (1.f / 96.f);
                g2 = oct_sum5(g2) * (1.f / 96.f);
                float* row = &dxf[lw * 16 + ps * 8 + r8][0];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    f32x4 t4 = {d1[4 * q], d1[4 * q + 1], d1[4 * q + 2], d1[4 * q + 3]};
                    f32x4 tm4 = t4, o4;
                    // site 2 backward (to_out dropout of block i): group index = token * 24 + feature / 4
                    if (DROP && valid) tm4 = drop4(a.drop_i, 2, (unsigned)(tok * 24 + (q < 2 ? 2 * p + q : 16 + p)), tm4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o4[e] = valid ? t4[e] + rstd * (dn[4 * q + e] - g1 - v[4 * q + e] * g2) : 0.f;
                        dbo[4 * q + e] += tm4[e];   // to_out bias gradient = column sum of d(attention output before dropout)
                    }
                    *reinterpret_cast<f32x4*>(row + (q < 2 ? 8 * p + 4 * q : 64 + 4 * p)) = o4;
                }
            }
        };
        // dx1 of block i - 1 and its dropped bf16 copy (the attention half's operand; site 2 of block i - 1), tile of walk step k:
        // the M waves left the fp32 rows in sm.out; they leave the CU as contiguous 1 KB per wave-instruction (the tile is 24 KB /
        // 12 KB of consecutive addresses), not as the 64-byte pieces of the MFMA C layout
        auto writeout = [&](int k) {
            const int tile_ = tile_at(k);
            const long tok0 = (long)tile_ * 64;
            const int li = lw * 64 + l;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ch = j * 256 + li, r = ch / 24, q = ch - r * 24;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&sm.out[r][4 * q]);
                if (tok0 + r < a.ntok) *reinterpret_cast<f32x4*>(a.dx1 + tok0 * 96 + (long)ch * 4) = v4;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ch = j * 256 + li, r = ch / 12, q = ch - r * 12;
                f32x4 lo = *reinterpret_cast<const f32x4*>(&sm.out[r][8 * q]), hi = *reinterpret_cast<const f32x4*>(&sm.out[r][8 * q + 4]);
                const long tok = tok0 + r;
                if (DROP) {
                    lo = drop4(a.drop_p, 2, (unsigned)(tok * 24 + 2 * q), lo);
                    hi = drop4(a.drop_p, 2, (unsigned)(tok * 24 + 2 * q + 1), hi);
                }
                if (tok < a.ntok) *reinterpret_cast<frag*>(reinterpret_cast<bf16_t*>(a.dab) + tok0 * 96 + (long)ch * 8) = P::pack2(lo, hi);
            }
        };
        // prologue: dx of the first tile
        issue(0, 0, 2);
        compute(0, 0, 2);
        issue(1, 0, 2);
        __syncthreads();
        for (int k = 0; valid_step(k); ++k) {
            // the draw issued a step ago (walk step k + 5) is published here, visible behind B1; the next one goes out
            if (QUEUE && lw == 0 && l == 0) {
                sm.qt[(k + 5) & 7] = tile_of_draw(qpend);
                qpend = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // M waves: walk step k (they read sm.dxf[k & 1] before B1).  Here: dx of step k + 1 into the other buffer, then the
            // rows of step k + 2 are requested -- a whole tile period ahead of their use
#ifdef MSST_STAMPS
            const bool stamp_on = a.stamps && blockIdx.x == 100 && l == 0 && k == nmine / 2;
#endif
            B5_STAMP(0);
            if (valid_step(k + 1)) {
#if MSST_B5_PASSPIPE
                // pass by pass: the registers of a pass are re-requested (walk step k + 2) as soon as it is consumed, so that
                // something is in flight all the time (all 28 requests behind the whole computation left the memory pipe idle half the tile)
                compute(k + 1, 0, 1); issue(k + 2, 0, 1); B5_STAMP(1); compute(k + 1, 1, 2); issue(k + 2, 1, 2);
#else
                compute(k + 1, 0, 2); B5_STAMP(1); issue(k + 2, 0, 2);
#endif
            }
            B5_STAMP(2);
            lds_barrier();   // B1
            B5_STAMP(3);
            writeout(k);     // (the M waves are in their weight-gradient phase: sm.out is stable until B2)
            B5_STAMP(4);
            lds_barrier();   // B2
            B5_STAMP(5);
        }
        // ---- LN1 gamma / beta / to_out bias gradients: sum over the 8 rows of a pass (lanes l ^ 8, 16, 32), then over the L waves ----
        __syncthreads();   // (S1) the M waves are done with every LDS tile
        float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][3][96], over xn2
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            float s0 = dg[i], s1 = db[i], s2 = dbo[i];
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
            if (r8 == 0) {
                const int f = i < 8 ? 8 * p + i : 64 + 4 * p + (i - 8);
                red[(lw * 3 + 0) * 96 + f] = s0; red[(lw * 3 + 1) * 96 + f] = s1; red[(lw * 3 + 2) * 96 + f] = s2;
            }
        }
        __syncthreads();   // (S2)
        const int t2 = tid - 256;
        for (int j = t2; j < 288; j += 256) {
            const int which = j / 96, m = j - which * 96;
            a.slab_ln1[(long)blockIdx.x * 288 + j] = (red[(0 * 3 + which) * 96 + m] + red[(1 * 3 + which) * 96 + m]) +
                                                    (red[(2 * 3 + which) * 96 + m] + red[(3 * 3 + which) * 96 + m]);
        }
        __syncthreads();   // (S3)
        __syncthreads();   // (S4)
        return;
    }

    // =========================================================================================================
    // M role: MLP half of block i - 1, backward (block_bwd_mlp_kernel with dy = sm.dxf)
    // =========================================================================================================
    const int g = l >> 4, c = l & 15;
    f32x4 dW1[6], dW2[6];   // dW1: C[i = n in tile wave][j = m tile jt]; dW2: C[i = m tile it][j = n in tile wave]
    float dgam[6][4], dbet[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        dW1[i] = zero4(); dW2[i] = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) { dgam[i][r] = 0.f; dbet[i][r] = 0.f; }
    }
    f32x4 db1a = zero4(), db2a = zero4(), db2b = zero4();
    const float* lnp = sm.lnp;
    auto wfrag = [&](int which, int K, int row0, int k0) -> frag {
        const int f = which * 12 + (row0 >> 4) * (K >> 5) + (k0 >> 5);
        return *reinterpret_cast<const frag*>(sm.wl + f * 1024 + l * 16);
    };
    f32x4 xrow[6];
    auto request_rows = [&](int k) {
        const int tile_ = valid_step(k) ? tile_at(k) : 0;
        const long t_ = (long)tile_ * 64 + wave * 16 + c;
        const long tokc = t_ < a.ntok ? t_ : 0;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) xrow[mt] = ld_x1_4<X1B>(a.x1, tokc, mt * 16 + 4 * g);
    };
    request_rows(0);
    __syncthreads();
    if (MSST_B5_PRIO) __builtin_amdgcn_s_setprio(1);
    for (int k = 0; valid_step(k); ++k) {
        const int tile = tile_at(k);
        const long tok = (long)tile * 64 + wave * 16 + c;
        const bool valid = tok < a.ntok;
#ifdef MSST_STAMPS
        const bool stamp_on = a.stamps && blockIdx.x == 100 && l == 0 && k == nmine / 2;
#endif
        B5_STAMP(0);
        float xhat[6][4];
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            const f32x4 xr = valid ? xrow[mt] : zero4();
            f32x4 dr = *reinterpret_cast<const f32x4*>(&sm.dxf[k & 1][wave * 16 + c][m0]);   // (zero for rows past the end)
#pragma unroll
            for (int r = 0; r < 4; ++r) { xhat[mt][r] = xr[r]; s1 += xr[r]; }
            // site 4 (MLP-out dropout of block i - 1): the FeedForward branch sees the masked gradient, the residual the raw one
            if (DROP && valid) dr = drop4(a.drop_p, 4, (unsigned)(tok * 24 + (m0 >> 2)), dr);
            P::st_nat(&sm.dy[wave * 16][mt * 16], LDX, dr);
        }
#if MSST_B5_X1EARLY
        // the x1 rows of the NEXT walk step: their registers are free from here on (copied into xhat), and a whole tile period
        // hides the HBM round trip (requested behind B1 they had one weight-gradient phase, ~1 us, to arrive)
        MSST_SCHED_FENCE();
        request_rows(k + 1);
        MSST_SCHED_FENCE();
#endif
        s1 = colgroup_sum(s1);
        const float mean = s1 * (1.f / 96.f);
        float vs = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float dd = xhat[mt][r] - mean; vs += dd * dd; }
        vs = colgroup_sum(vs);
        const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 n4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xhat[mt][r] = (xhat[mt][r] - mean) * rstd;
                n4[r] = xhat[mt][r] * lnp[m0 + r] + lnp[96 + m0 + r];
            }
            P::st_nat(&sm.xn2[wave * 16][mt * 16], LDX, n4);
        }
        __builtin_amdgcn_wave_barrier();
        B5_STAMP(1);
        // h_pre = W1 xn2 + b1, dh = W2^T dy   (C[i = n][j = row])
        f32x4 hp[4], dh[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { hp[nt] = zero4(); dh[nt] = zero4(); }
#pragma unroll
        for (int k0 = 0; k0 < 96; k0 += KS) {
            const frag xb = P::ld_kc(&sm.xn2[wave * 16][k0], LDX);
            const frag dbf = P::ld_kc(&sm.dy[wave * 16][k0], LDX);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                hp[nt] = P::mma(wfrag(0, 96, nt * 16, k0), xb, hp[nt]);
                dh[nt] = P::mma(wfrag(1, 96, nt * 16, k0), dbf, dh[nt]);
            }
        }
        B5_STAMP(2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = nt * 16 + 4 * g;
            f32x4 hv, dv;
            f32x4 dhm = dh[nt];
            unsigned keep3 = 0xfu;   // site 3: one hash for the backward mask here and the forward mask below
            if (DROP && valid) dhm = drop4_keep(a.drop_p, 3, (unsigned)(tok * 16 + (n0 >> 2)), dhm, keep3);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pre = hp[nt][r] + lnp[192 + n0 + r];
                float gv, gg;
                P::gelu_both(pre, gv, gg);
                hv[r] = gv;
                dv[r] = dhm[r] * gg;
            }
            if (DROP && valid) hv = drop4_bits(a.drop_p, keep3, hv);
            P::st_nat(&sm.h[wave * 16][nt * 16], LDH, hv);
            P::st_nat(&sm.dhp[wave * 16][nt * 16], LDH, dv);
        }
        __builtin_amdgcn_wave_barrier();
        B5_STAMP(3);
        // d(xn2) = W1^T dh_pre   (C[i = m][j = row])
        f32x4 dxn[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) dxn[mt] = zero4();
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += KS) {
            const frag hb = P::ld_kc(&sm.dhp[wave * 16][k0], LDH);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) dxn[mt] = P::mma(wfrag(2, 64, mt * 16, k0), hb, dxn[mt]);
        }
        B5_STAMP(4);
        // LN2 backward + residual
        float g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dn = dxn[mt][r];
                dgam[mt][r] += dn * xhat[mt][r];
                dbet[mt][r] += dn;
                const float dgv = dn * lnp[m0 + r];
                dxn[mt][r] = dgv;
                g1 += dgv;
                g2 += dgv * xhat[mt][r];
            }
        }
        g1 = colgroup_sum(g1) * (1.f / 96.f);
        g2 = colgroup_sum(g2) * (1.f / 96.f);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            const f32x4 dyv = *reinterpret_cast<const f32x4*>(&sm.dxf[k & 1][wave * 16 + c][m0]);   // the raw dx_i again (residual path)
            f32x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = dyv[r] + rstd * (dxn[mt][r] - g1 - xhat[mt][r] * g2);
            *reinterpret_cast<f32x4*>(&sm.out[wave * 16 + c][m0]) = o4;   // (rows past the end: never copied out)
        }
        B5_STAMP(5);
        lds_barrier();   // B1
        B5_STAMP(6);
#if !MSST_B5_X1EARLY
        request_rows(k + 1);
#endif
        // ---------------- phase 2: weight grads over the 64 rows of the tile ----------------
        {
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += KS) {
            const frag ah = P::ld_ks(&sm.dhp[k0][wave * 16], LDH);  // A[i = n][k = row]
            const frag bh = P::ld_ks(&sm.h[k0][wave * 16], LDH);    // B[j = n][k = row]
            const frag one = P::ones();
            db1a = P::mma(ah, one, db1a);                                          // C[i = n][j = *] = sum_row dhp[row][n]
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const frag dyt = P::ld_ks(&sm.dy[k0][t * 16], LDX);
                dW1[t] = P::mma(ah, P::ld_ks(&sm.xn2[k0][t * 16], LDX), dW1[t]);  // C[i = n][j = m]
                dW2[t] = P::mma(dyt, bh, dW2[t]);                                  // C[i = m][j = n]
                if (t == wave) db2a = P::mma(dyt, one, db2a);                      // C[i = m][j = *] = sum_row dy[row][m]
                if (t == wave + 4) db2b = P::mma(dyt, one, db2b);
            }
        }
        }
        B5_STAMP(7);
        lds_barrier();   // B2
        B5_STAMP(8);
    }

    // ---------------- write this workgroup's MLP slab ----------------
    float* slab = a.slab_mlp + (long)blockIdx.x * MSST_MLP_SLAB_N;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            slab[(wave * 16 + 4 * g + r) * 96 + t * 16 + c] = dW1[t][r];          // dW1[n][m]
            slab[6144 + (t * 16 + 4 * g + r) * 64 + wave * 16 + c] = dW2[t][r];   // dW2[m][n]
        }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            slab[12288 + wave * 16 + 4 * g + r] = db1a[r];                                   // db1
            slab[12288 + 64 + wave * 16 + 4 * g + r] = db2a[r];                              // db2, feature tile wave
            if (wave < 2) slab[12288 + 64 + (wave + 4) * 16 + 4 * g + r] = db2b[r];          // db2, feature tile wave + 4
        }
    }
    __syncthreads();   // (S1)
    __syncthreads();   // (S2) the L waves' reduction uses the front of the LDS
    __syncthreads();   // (S3)
    // LN2 gamma / beta: sum over the 16 rows of the wave, then over the M waves through LDS
    float* red = reinterpret_cast<float*>(smem_raw);  // [4 waves][2][96]
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = rowgroup_sum(dgam[mt][r]);
            const float sb = rowgroup_sum(dbet[mt][r]);
            if (c == 0) {
                red[(wave * 2 + 0) * 96 + mt * 16 + 4 * g + r] = sg;
                red[(wave * 2 + 1) * 96 + mt * 16 + 4 * g + r] = sb;
            }
        }
    __syncthreads();   // (S4)
    if (tid < 192) {
        const int which = tid / 96, m = tid - which * 96;
        slab[12288 + 160 + which * 96 + m] = (red[(0 * 2 + which) * 96 + m] + red[(1 * 2 + which) * 96 + m]) +
                                             (red[(2 * 2 + which) * 96 + m] + red[(3 * 2 + which) * 96 + m]);
    }
}

int launch_block_bwd_ln1mlp(const LnMlpArgs& a, int grid, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    const size_t smem = sizeof(LmSmem);
    const int ntiles = (int)((a.ntok + 63) / 64);
    if (grid < 1 || a.nparts < 1 || a.nparts > 4 || (long)a.nparts * a.ntok * 192 >= 0x7ffffff0L || a.ntok * 384 >= 0x7ffffff0L) return MSST_ERR_UNSUPPORTED;
    const bool xnb = a.xn != nullptr;   // MSST_LN1_FROM_XN: instantiated for bf16 x1 rows only (both are the role-split forward's products)
    if (xnb && (!a.rstd || !a.ln1_b || !a.x1_bf16)) return MSST_ERR_UNSUPPORTED;
    if (grid > ntiles) grid = ntiles;
#ifdef MSST_B5_ONLY_BENCH   // (kernel-study compile: the bench shape's instances only, tools/kres.py)
#define MSST_B5_FOR_ALL(X) X(4, true)
#else
#define MSST_B5_FOR_ALL(X) X(1, false) X(1, true) X(2, false) X(2, true) X(3, false) X(3, true) X(4, false) X(4, true)
#endif
    if (!attr_set) {
#define MSST_B5_ATTR1(np, dr, q, xb, xn) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_ln1mlp_kernel<np, dr, q, xb, xn>), \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); if (e != hipSuccess) return (int)e; }
#define MSST_B5_ATTR(np, dr) MSST_B5_ATTR1(np, dr, false, false, false) MSST_B5_ATTR1(np, dr, true, false, false) MSST_B5_ATTR1(np, dr, false, true, false) \
                             MSST_B5_ATTR1(np, dr, true, true, false) MSST_B5_ATTR1(np, dr, false, true, true) MSST_B5_ATTR1(np, dr, true, true, true)
        MSST_B5_FOR_ALL(MSST_B5_ATTR)
        attr_set = true;
    }
    ProfScope ps(K_BWD_LN1MLP, st);
    const bool dr = a.drop_i.thr != 0 || a.drop_p.thr != 0;
#define MSST_B5_LAUNCH1(np, drv, q, xb, xn) hipLaunchKernelGGL((block_bwd_ln1mlp_kernel<np, drv, q, xb, xn>), dim3(grid), dim3(512), smem, st, a)
#define MSST_B5_LAUNCH2(np, drv, q) { if (xnb) MSST_B5_LAUNCH1(np, drv, q, true, true); else if (a.x1_bf16) MSST_B5_LAUNCH1(np, drv, q, true, false); \
                                      else MSST_B5_LAUNCH1(np, drv, q, false, false); }
#define MSST_B5_LAUNCH(np, drv) if (a.nparts == np && dr == drv) { if (a.queue) MSST_B5_LAUNCH2(np, drv, true) else MSST_B5_LAUNCH2(np, drv, false) }
    MSST_B5_FOR_ALL(MSST_B5_LAUNCH)
    return (int)hipGetLastError();
}

}  // namespace msst
