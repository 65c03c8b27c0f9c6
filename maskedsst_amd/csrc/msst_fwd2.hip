// bf16 forward of one transformer block, head-per-wave (reference vit_spatial_spectral.py:22-104:
// PreNorm + Attention + residual, PreNorm + FeedForward + residual; a7-a10 of SURVEY.md section 8).
//
// Same math, same dropout streams and the same HBM interface as block_fwd_bf16_kernel (msst_fwd.hip);
// what changes is who does what on the CU:
//   * one 512-thread workgroup per CU, wave h <-> attention head h of the current 64-row tile;
//   * a wave computes q, k, v of ITS head straight into MFMA operand registers and keeps the whole
//     attention (S, softmax, P, O, out-projection of the head) in registers: the C-layout result of one
//     MFMA is fed to the next one as a B (or A) operand by packing two 16-row C tiles into one 32-deep
//     k-chunk.  That permutes the contraction index inside the chunk, which is harmless as long as the
//     other operand uses the same permutation -- for q.k^T and P.V both operands are built the same way;
//     for the out-projection the rows of W_v are gathered so that O comes out in natural order.
//   * no barrier and no LDS round trip between LN1 and the end of the out-projection (the tuned
//     4-wave kernel has two barriers and five LDS round trips per head);
//   * the heads meet in ONE bf16 tile O [64 rows][8 x 64 channels] in LDS; the out-projection is then a K = 512
//     GEMM split over the 8 waves as (half of the features, pair of 16-row tiles, half of K): 48 MFMAs per wave,
//     each weight fragment serves two row tiles.  The two K halves are combined by a single exchange of three
//     fp32 C tiles per wave through a lane-linear buffer, after which wave (row tile, feature half) holds its
//     48 features of 16 finished rows in registers: bias, dropout, residual, the LN2 statistics (combined with
//     the other feature half through 2 floats per row) and the MLP residual never touch LDS.  (The first version
//     summed eight fp32 per-head partials through LDS: 288 four-byte LDS operations per wave and a row-wise
//     re-read -- 44 % of the tile time.)
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"
#include <type_traits>

#ifndef MSST_F2_STAMP_TID
#define MSST_F2_STAMP_TID 0
#endif
#ifndef MSST_F2_PADX
#define MSST_F2_PADX 16
#endif
#ifndef MSST_F2_PADH
#define MSST_F2_PADH 8
#endif
#ifndef MSST_F2_RING
#define MSST_F2_RING 4
#endif
#ifndef MSST_F2_PRIO
#define MSST_F2_PRIO 0
#endif
#ifndef MSST_F2_RELOAD
#define MSST_F2_RELOAD 1   // weight-fragment offsets are re-derived on the scalar unit where they are used (the head / wave-role indices pass
                           // through an opaque scalar register): hoisted out of the tile loop they were 178 SGPRs spilled to VGPR lanes,
                           // i.e. ~190 v_readlane per wave and tile in a kernel whose limiter is the VALU stream
#endif
#ifndef MSST_F2_EXP
#define MSST_F2_EXP 0   // timing experiments (wrong results): 1 = every q / k / v weight request reads the same two fragments, 2 = no softmax
                        // arithmetic (scores pass through), 8 = every out-projection weight request reads fragment 0, 16 = no token arithmetic (every tile reads and
                        // writes the rows of tile 0), 32 / 64 = as 8, for the waves of the second / first K half only
#endif
#ifndef MSST_F2_SKIP
#define MSST_F2_SKIP 0   // measured: skipping the masked score tiles of spectral blocks costs more in branches than it saves (+1.5 %)
#endif

#ifndef MSST_F2_STAMPSEL
#define MSST_F2_STAMPSEL 0x00c09   // which of the stamps 0 .. 17 a -DMSST_STAMPS build carries.  All of them at once make the kernel spill to scratch, and
                                   // a spill reload waits for every prefetch in flight: phase times of such a build are artefacts.  Subsets that compile
                                   // without spills: 0x00c09 (tile start, LN1 done, attention done, O barrier passed), 0x3f001 (tile start, out-projection
                                   // ... end); check `grep -c scratch_` of the -save-temps assembly before trusting another one
#endif
#define F2_STAMP(i) do { if ((MSST_F2_STAMPSEL >> (i)) & 1) STAMP(i); } while (0)

namespace msst {

namespace {

// sum over 8 consecutive lanes (a row of the row-wise LN1 phase), on every lane: quad DPP moves, then the other quad of
// the 8 through row_half_mirror (all four lanes of a quad already agree)
__device__ __forceinline__ float oct_sum(float v) {
    v = quad_sum(v);
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
}

typedef PBF16 P;
typedef bf16_t elem;
typedef s16x8 frag;

struct Fwd2Smem {
    static constexpr int LDX = 96 + MSST_F2_PADX;   // bf16 rows of 96 + pad (row stride = 2 mod 4 sixteen-byte slots: conflict-free b128 fragment reads)
    static constexpr int LDH = 64 + MSST_F2_PADH;
    static constexpr int LDO = 512 + 16;  // 66 slots = 2 mod 4
    elem xn[64][LDX];                     // LN1(x), later LN2(x1)
    elem ob[64][LDO];                     // attention output of the 8 heads, [row][h * 64 + channel]
    f32x4 xch[8][3][64];                  // K-half exchange: [receiving wave][C tile][lane]
    float2 st[8][16];                     // LN2 partial statistics (mean, M2 over 48 features): [wave][row in tile]
    elem hb[64][LDH];                     // GELU(W1 .) of the MLP
    unsigned rowmap[64];                  // tile row -> (sequence slot << 16 | position), 0xffff = padding row: tile invariant
    int seqb[2][64];                      // token of position 0 of every sequence slot of the tile being processed / the next one (by tile parity)
};

__device__ __forceinline__ int sopaque(int v) {
#if MSST_F2_RELOAD
    asm volatile("" : "+s"(v));
#endif
    return v;
}

__device__ __forceinline__ frag pack2(f32x4 lo, f32x4 hi) {
    const s16x4 a = f2bf4(lo), b = f2bf4(hi);
    frag r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// fragment of 16 GATHERED rows of a fragment-packed [R][K] weight: lane column c reads row
// row32 + 8 (c / 4) + c % 4 + 4 hi  (voff carries the lane part, see below)
__device__ __forceinline__ frag ld_w_gather(const elem* w, int K, int row32, int k0, int voff) {
    const int f = (row32 >> 4) * (K >> 5) + (k0 >> 5);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(w), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, f * 1024, 0);
    return __builtin_bit_cast(frag, v);
}

// weight-fragment pair number pi of head h (compile-time constant after unrolling): the stream a wave consumes per tile is
//   0..5 q   6..11 k   (pair = fragments of channel tiles 2m, 2m+1 at k-step ks; pi = 3 (2 which + m) + ks)
//   12..17 gathered v  (pair = low / high channel halves of block mm at k-step ks; pi = 12 + 3 mm + ks)
__device__ __forceinline__ void load_pair(int pi, frag (&out)[2], const elem* wqkv, const elem* wout, int H, int h,
                                          const int (&voff)[2]) {
    if (MSST_F2_EXP & 1) {
        out[0] = P::ld_w(wqkv, 96, 0, 0);
        out[1] = P::ld_w(wqkv, 96, 16, 0);
    } else if (pi < 12) {
        const int st = pi / 3, ks = pi % 3;
        const int r0 = ((st >> 1) * H + h) * 64 + (st & 1) * 32;
        out[0] = P::ld_w(wqkv, 96, r0, ks * 32);
        out[1] = P::ld_w(wqkv, 96, r0 + 16, ks * 32);
    } else {
        const int mm = (pi - 12) / 3, ks = (pi - 12) % 3;
        const int r32 = (2 * H + h) * 64 + mm * 32;
        out[0] = ld_w_gather(wqkv, 96, r32, ks * 32, voff[0]);
        out[1] = ld_w_gather(wqkv, 96, r32, ks * 32, voff[1]);
    }
}

}  // namespace

// DROP: dropout compiled in / out (a uniform run-time test at every site splits the instruction stream into basic blocks the
// scheduler cannot interleave across)
template <bool DROP>
__global__ __launch_bounds__(512, 2) void block_fwd_hw_kernel(BlockArgs a) {
    typedef Fwd2Smem SM;
    constexpr int LDX = SM::LDX, LDH = SM::LDH, LDO = SM::LDO;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const int H = a.H, inner = H * 64, h = wave;
    const elem* wqkv = reinterpret_cast<const elem*>(a.w.wqkv);
    const elem* wout = reinterpret_cast<const elem*>(a.w.wout);
    const elem* w1 = reinterpret_cast<const elem*>(a.w.w1);
    const elem* w2 = reinterpret_cast<const elem*>(a.w.w2);
    const TileMap tm = a.tm;
    const int L = tm.L;

    // small parameter vectors and the MLP weights stay in LDS for the life of the workgroup
    float* lnp = reinterpret_cast<float*>(smem_raw + sizeof(SM));   // ln1_g | ln1_b | bo | ln2_g | ln2_b | b2 | b1
    if (tid < 96) {
        lnp[tid] = a.w.ln1_g[tid]; lnp[96 + tid] = a.w.ln1_b[tid]; lnp[192 + tid] = a.w.bo[tid];
        lnp[288 + tid] = a.w.ln2_g[tid]; lnp[384 + tid] = a.w.ln2_b[tid]; lnp[480 + tid] = a.w.b2[tid];
        if (tid < 64) lnp[576 + tid] = a.w.b1[tid];
    }
    char* wmlp = smem_raw + sizeof(SM) + 640 * sizeof(float);       // [w1: 12 frags | w2: 12 frags]
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int f = wave * 3 + i3;
        dma_frag(f < 12 ? reinterpret_cast<const char*>(w1) + f * 1024 : reinterpret_cast<const char*>(w2) + (f - 12) * 1024,
                 wmlp + f * 1024);
    }
    wait_vm0();
    __syncthreads();

    // row-wise phases (LN1, residual + LN2): thread <-> (row tid / 8, 12 features)
    const int2 sp_ln = tm.row_sp(tid >> 3);
    // Token arithmetic through two small LDS tables instead of two integer divisions per thread, three times per tile (420 instructions
    // between the last wave's attention and the "O complete" barrier): the row map is tile invariant, the sequence bases of a tile are
    // computed by 64 threads one tile ahead.
#ifndef MSST_F2_TOKTAB
#define MSST_F2_TOKTAB 1
#endif
    auto fill_seq = [&](int par, int tile_) {
        if (tid < 64) {
            const int q = tile_ * tm.TS + tid;
            int base = -1;
            if (tile_ < a.ntiles && tid < tm.TS && q < tm.nseq) {
                if (tm.mode == 0) base = q * tm.N;
                else { const int b = tm.nshift >= 0 ? (q >> tm.nshift) : q / tm.N; base = b * tm.T + (q - b * tm.N); }
            }
            sm.seqb[par][tid] = base;
        }
    };
    auto tok_of = [&](int par, int r) -> long {
        const unsigned sp = sm.rowmap[r];
        const int base = sm.seqb[par][min((int)(sp >> 16), 63)];
        return ((sp >> 16) == 0xffffu || base < 0) ? -1 : (long)(base + (int)(sp & 0xffffu) * (tm.mode == 0 ? 1 : tm.N));
    };
    if (tid < 64) {
        const int sq = tid / L, ps = tid - sq * L;
        sm.rowmap[tid] = ((unsigned)(sq >= tm.TS ? 0xffff : sq) << 16) | (unsigned)ps;
    }
    fill_seq(0, blockIdx.x);
    __syncthreads();
    int par = 0;
    // out-projection: wave <-> (feature half mh, row-tile pair rh, K half kh); it ends up owning row tile tt = 2 rh + kh
    // of feature half mh, and keeps that role through LN2 and the MLP
    const int mh = wave & 1, rh = (wave >> 1) & 1, kh = wave >> 2;
    const int tt = 2 * rh + kh, half = mh;
    // lane offsets of the gathered W_v fragments (low / high half of each group of 8 channels)
    int voff[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int rs = 8 * (c >> 2) + (c & 3) + 4 * hi;   // 0..31
        voff[hi] = ((rs >> 4) * 3) * 1024 + (g * 16 + (rs & 15)) * 16;   // K = 96 -> 3 fragments per 16 rows
    }
    int qlo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qlo[j] = ((j * 16 + c) / L) * L;
    // short sequences (spectral blocks): query tile j only meets the key tiles that overlap the sequences of its 16 rows --
    // bit t of need[j]; every other 16 x 16 score tile is masked anyway and is skipped (MFMAs, exps, dropout hashes)
    unsigned need[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s_lo = (16 * j) / L, s_hi = min((16 * j + 15) / L, tm.TS - 1);
        const int k_lo = s_lo * L, k_hi = (s_hi + 1) * L - 1;   // s_lo >= TS (all-padding query tile): empty range
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) m |= (unsigned)(s_lo <= s_hi && 16 * t <= k_hi && 16 * t + 15 >= k_lo) << t;
        need[j] = MSST_F2_SKIP ? m : 0xfu;
    }

    f32x4 xv[3];   // this thread's 12 row values of the tile to process (prefetched one tile ahead)
    {
        const long tok0 = tm.token_sp(blockIdx.x, sp_ln);
#pragma unroll
        for (int i = 0; i < 3; ++i) xv[i] = tok0 >= 0 ? reinterpret_cast<const f32x4*>(a.x + tok0 * 96 + (tid & 7) * 12)[i] : zero4();
    }

    constexpr int NR = MSST_F2_RING;
    static_assert(NR == 4 || 18 % NR == 0, "the out-projection / next-tile hand-over below assumes a ring of four pairs");   // pairs in flight: a pair is requested NR * 8 MFMAs of this wave before its use
    frag ring[NR][2];
    // pair to request into slot pi % NR once pair pi is consumed: NR ahead in the 18-pair stream of a tile, wrapping to the
    // next tile's first pairs (pair p always lives in slot p % NR: automatic when NR divides 18, hand-placed for NR = 4)
    auto next_pair = [](int pi) { return 18 % NR == 0 ? (pi + NR) % 18 : (pi + NR < 18 ? pi + NR : pi % NR); };
#pragma unroll
    for (int pi = 0; pi < NR; ++pi) load_pair(pi, ring[pi], wqkv, wout, H, h, voff);

#ifdef MSST_STAMPS
    const bool stamp_wg = (a.dbg & 8) && blockIdx.x == 100 && tid == MSST_F2_STAMP_TID;
#endif
    if (MSST_F2_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every age arbitration otherwise
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, par ^= 1) {
#ifdef MSST_STAMPS
        const bool stamp_on = stamp_wg && tile == blockIdx.x + 8 * (int)gridDim.x;   // a mid-walk tile
#endif
        F2_STAMP(0);
        int t1 = threadIdx.x;
        asm volatile("" : "+v"(t1));
        const int lr = t1 >> 3, part = t1 & 7;
        // ---------------- LN1 -> xn ----------------
        {
            float v[12];
#pragma unroll
            for (int i = 0; i < 3; ++i) { v[4*i] = xv[i][0]; v[4*i+1] = xv[i][1]; v[4*i+2] = xv[i][2]; v[4*i+3] = xv[i][3]; }
            if (MSST_F2_TOKTAB) fill_seq(par ^ 1, tile + (int)gridDim.x);   // (read behind the "O complete" barrier and by the next tile)
            const long tok_ln = (MSST_F2_EXP & 16) ? (long)lr : MSST_F2_TOKTAB ? tok_of(par, lr) : tm.token_sp(tile, tm.row_sp(lr));
            if (tok_ln < 0) {   // padding row: the prefetch read a clamped address, normalise zeros
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = 0.f;
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) s += v[i];
            const float mean = oct_sum(s) * (1.f / 96.f);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float d = v[i] - mean; vs += d * d; }
            const float rstd = rsqrtf(oct_sum(vs) * (1.f / 96.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 n4;
#pragma unroll
                for (int e = 0; e < 4; ++e) n4[e] = (v[4*i+e] - mean) * rstd * lnp[part * 12 + 4*i+e] + lnp[96 + part * 12 + 4*i+e];
                const s16x4 nb = f2bf4(n4);
                *reinterpret_cast<s16x4*>(&sm.xn[lr][part * 12 + 4 * i]) = nb;
                // the same bf16 rows go to HBM for the attention backward (192 contiguous bytes per row from 8 threads)
                if (a.xn_out && tok_ln >= 0) *reinterpret_cast<s16x4*>(reinterpret_cast<elem*>(a.xn_out) + tok_ln * 96 + part * 12 + 4 * i) = nb;
            }
        }
        lds_barrier();
        if (MSST_F2_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        // ================= head h, all in this wave's registers =================
        // Weight fragments arrive through a ring of six fragment pairs (see load_pair): the pair consumed now was
        // requested six pairs = 48 MFMAs of this wave ago.
        frag qB[4][2], kA[4][2], vA[4][2];
        {
            frag xf[4][3];   // LN1(x) as operand fragments: [16-row tile][k-step]
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) xf[t][ks] = P::ld_kc(&sm.xn[t * 16][ks * 32], LDX);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                F2_STAMP(3 + st);
                if (st < 4) {
                    // q and k: C[i = channel][j = row]; channel tiles (2m, 2m+1) -> k-chunk m of the operand
                    const int m = st & 1;
                    f32x4 ca[4], cb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { ca[t] = zero4(); cb[t] = zero4(); }
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const int pi = 3 * st + ks;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            ca[t] = P::mma(ring[pi % NR][0], xf[t][ks], ca[t]);
                            cb[t] = P::mma(ring[pi % NR][1], xf[t][ks], cb[t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        load_pair(next_pair(pi), ring[pi % NR], wqkv, wout, H, sopaque(h), voff);   // past pair 17: the next tile's first pairs
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (st < 2) qB[t][m] = pack2(ca[t], cb[t]); else kA[t][m] = pack2(ca[t], cb[t]);
                    }
                } else {
                    // v^T: C[i = row][j = gathered channel]; row tiles (2m, 2m+1) -> k-chunk m (keys)
                    const int mm = st - 4;
                    f32x4 cl[4], ch[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { cl[t] = zero4(); ch[t] = zero4(); }
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const int pi = 3 * st + ks;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            cl[t] = P::mma(xf[t][ks], ring[pi % NR][0], cl[t]);
                            ch[t] = P::mma(xf[t][ks], ring[pi % NR][1], ch[t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        load_pair(next_pair(pi), ring[pi % NR], wqkv, wout, H, sopaque(h), voff);   // past pair 17: the next tile's first pairs
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    vA[2 * mm][0] = pack2(cl[0], cl[1]);     vA[2 * mm][1] = pack2(cl[2], cl[3]);
                    vA[2 * mm + 1][0] = pack2(ch[0], ch[1]); vA[2 * mm + 1][1] = pack2(ch[2], ch[3]);
                }
            }
        }
        F2_STAMP(9);
        if (MSST_F2_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        // ================= attention of head h for the four query tiles; O rows go to the shared bf16 tile =================
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 s[4];
            f32x4 o[4];
            const float cs = a.scale * 1.44269504088896340736f;
            if (L == 64) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s[t] = P::mma(kA[t][0], qB[j][0], zero4());   // C[i = key][j = query]
                    s[t] = P::mma(kA[t][1], qB[j][1], s[t]);
                }
                if (MSST_F2_EXP & 2) {
                    const frag p0 = pack2(s[0], s[1]), p1 = pack2(s[2], s[3]);
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        o[dd] = P::mma(vA[dd][0], p0, zero4());
                        o[dd] = P::mma(vA[dd][1], p1, o[dd]);
                    }
                } else {
                // softmax over the 64 keys.  exp(scale (s - max)) = exp2(s c - max c), c = scale log2 e: one FMA + one v_exp
                // per element, nothing to mask
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
                mx = colgroup_max(mx);
                const float mc = mx * cs;
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -mc)); s[t][r] = e; sum += e; }
                sum = colgroup_sum(sum);
                const float inv = (DROP ? a.drop.scale : 1.f) * __builtin_amdgcn_rcpf(sum);   // the dropout scale rides on the normalisation
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s[t] = s[t] * inv;
                    if (DROP)
                        s[t] = drop4_noscale(a.drop, 1, (unsigned)(((tile * H + h) * 64 + j * 16 + c) * 16 + t * 4 + g), s[t]);
                }
                const frag p0 = pack2(s[0], s[1]), p1 = pack2(s[2], s[3]);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    o[dd] = P::mma(vA[dd][0], p0, zero4());       // C[i = gathered channel][j = query]
                    o[dd] = P::mma(vA[dd][1], p1, o[dd]);
                }
                }
            } else {
                // several short sequences per tile: keys outside the query's own sequence are masked; key tiles that no row of
                // this query tile can see are skipped altogether
                const unsigned nm = need[j];
                const int lo = qlo[j], hi = lo + L;
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((nm >> t) & 1u) {
                        s[t] = P::mma(kA[t][0], qB[j][0], zero4());
                        s[t] = P::mma(kA[t][1], qB[j][1], s[t]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = t * 16 + 4 * g + r;
                            const float v = (key >= lo && key < hi) ? s[t][r] : -INFINITY;
                            s[t][r] = v;
                            mx = fmaxf(mx, v);
                        }
                    }
                }
                mx = colgroup_max(mx);
                const float mc = mx * cs;
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((nm >> t) & 1u) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -mc)); s[t][r] = e; sum += e; }
                    } else {
                        s[t] = zero4();
                    }
                }
                sum = colgroup_sum(sum);
                const float inv = (DROP ? a.drop.scale : 1.f) * __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((nm >> t) & 1u) {
                        s[t] = s[t] * inv;
                        if (DROP)
                            s[t] = drop4_noscale(a.drop, 1, (unsigned)(((tile * H + h) * 64 + j * 16 + c) * 16 + t * 4 + g), s[t]);
                    }
                }
                const frag p0 = pack2(s[0], s[1]), p1 = pack2(s[2], s[3]);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    o[dd] = zero4();
                    if (nm & 3u) o[dd] = P::mma(vA[dd][0], p0, o[dd]);
                    if (nm & 12u) o[dd] = P::mma(vA[dd][1], p1, o[dd]);
                }
            }
            // pack2(o[2u], o[2u + 1]) holds, in lane (c, g), the natural channels 32 u + 8 g .. + 7 of query row 16 j + c:
            // one 16-byte store per half
            int l4 = threadIdx.x & 63;
            asm volatile("" : "+v"(l4));
            elem* orow = &sm.ob[j * 16 + (l4 & 15)][h * 64 + 8 * (l4 >> 4)];
            *reinterpret_cast<frag*>(orow) = pack2(o[0], o[1]);
            *reinterpret_cast<frag*>(orow + 32) = pack2(o[2], o[3]);
        }
        F2_STAMP(10);
        // rows owned from here on: lane (c, g) of wave (tt, half) <-> row 16 tt + c, features 16 (3 half + i) + 4 g .. + 3.
        // Residual values of this tile (L2 hits) and the next tile's rows for LN1 are requested now: q / k / v registers are
        // free.  Per-thread indices are re-derived from a laundered lane id so that none stays live across the head phase.
        int l3 = threadIdx.x & 63;
        asm volatile("" : "+v"(l3));
        const int g3 = l3 >> 4, c3 = l3 & 15;
        const long tok = (MSST_F2_EXP & 16) ? (long)(tt * 16 + c3) : MSST_F2_TOKTAB ? tok_of(par, tt * 16 + c3) : tm.token_sp(tile, tm.row_sp(tt * 16 + c3));
        f32x4 xr[3];
        {
            const float* xrow = a.x + (tok >= 0 ? tok : 0) * 96 + 48 * half + 4 * g3;
#pragma unroll
            for (int i = 0; i < 3; ++i) xr[i] = *reinterpret_cast<const f32x4*>(xrow + 16 * i);
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int nt = tile + gridDim.x;
            const long tokn = (MSST_F2_EXP & 16) ? (long)(t2 >> 3) : MSST_F2_TOKTAB ? tok_of(par ^ 1, t2 >> 3) : nt < a.ntiles ? tm.token_sp(nt, tm.row_sp(t2 >> 3)) : -1;
            const float* xn_row = a.x + (tokn >= 0 ? tokn : 0) * 96 + (t2 & 7) * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) xv[i] = reinterpret_cast<const f32x4*>(xn_row)[i];
        }
        // all 24 out-projection weight fragments of this wave's (feature half, K half) are requested before the barrier:
        // the wait for the slowest head hides their L2 round trip (q / k / v registers are free)
        frag fw[8][3];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int i = 0; i < 3; ++i) fw[s8][i] = ((MSST_F2_EXP & 8) || ((MSST_F2_EXP & 32) && kh == 1) || ((MSST_F2_EXP & 64) && kh == 0)) ? P::ld_w(wout, inner, 0, 0) : P::ld_w(wout, inner, (3 * sopaque(mh) + i) * 16, (8 * sopaque(kh) + s8) * 32);
        lds_barrier();   // O complete
        F2_STAMP(11);
        // ---------------- out-projection: C[i = feature][j = row], K = 512 split in two ----------------
        f32x4 acc[2][3];   // [row tile 2 rh + jj][feature tile 3 mh + i]
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[jj][i] = zero4();
        {
            frag fo[4][2];   // O row fragments of a k-step, requested three k-steps ahead
            swpipe<8, 3>(
                [&](int s8) {
                    const int k0 = (8 * kh + s8) * 32;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) fo[s8 % 4][jj] = P::ld_kc(&sm.ob[(2 * rh + jj) * 16][k0], LDO);
                },
                [&](int s8) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc[jj][i] = P::mma(fw[s8][i], fo[s8 % 4][jj], acc[jj][i]);
                });
        }
        // the other K half of the row tile this wave gives away goes to its owner (wave ^ 4), lane-linear
        {
            f32x4* dst = &sm.xch[wave ^ 4][0][l3];
            if (kh == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) dst[i * 64] = acc[1][i];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) dst[i * 64] = acc[0][i];
            }
        }
        F2_STAMP(12);
        lds_barrier();
        F2_STAMP(13);
        // ---------------- bias, dropout, residual -> x1;  LN2 -> xn  (all in registers) ----------------
        f32x4 x1r[3];   // x1 of the owned rows / features: stays in registers until the end of the MLP
        {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m0 = (3 * half + i) * 16 + 4 * g3;
                f32x4 o4 = (kh == 0 ? acc[0][i] : acc[1][i]) + sm.xch[wave][i][l3];
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] += lnp[192 + m0 + r];
                if (DROP && tok >= 0) o4 = drop4(a.drop, 2, (unsigned)(tok * 24 + (m0 >> 2)), o4);
                o4 = o4 + xr[i];
                x1r[i] = o4;
                s1 += (o4[0] + o4[1]) + (o4[2] + o4[3]);
                if (a.x1 && tok >= 0) *reinterpret_cast<f32x4*>(a.x1 + tok * 96 + m0) = o4;
            }
            // statistics of the 48 owned features of row c3 (sum over the 4 lane groups), then combined with the other half
            const float mean_w = colgroup_sum(s1) * (1.f / 48.f);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = x1r[i][r] - mean_w; m2 += d * d; }
            m2 = colgroup_sum(m2);
            if (g3 == 0) sm.st[wave][c3] = make_float2(mean_w, m2);
            lds_barrier();
            const float2 other = sm.st[wave ^ 1][c3];
            const float mean = 0.5f * (mean_w + other.x);
            const float dm = mean_w - other.x;
            const float var = (m2 + other.y + dm * dm * 24.f) * (1.f / 96.f);   // Chan's pairwise combination, n = 48 + 48
            const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m0 = (3 * half + i) * 16 + 4 * g3;
                f32x4 n4;
#pragma unroll
                for (int r = 0; r < 4; ++r) n4[r] = (x1r[i][r] - mean) * rstd * lnp[288 + m0 + r] + lnp[384 + m0 + r];
                *reinterpret_cast<s16x4*>(&sm.xn[tt * 16 + c3][m0]) = f2bf4(n4);
            }
        }
        F2_STAMP(14);
        lds_barrier();
        F2_STAMP(15);
        // ---------------- MLP: wave <-> (row tile tt, half of the outputs) ----------------
        {
            f32x4 hh[2];
            hh[0] = zero4(); hh[1] = zero4();
            frag xb[3], w1f[2][3];   // all nine operand fragments requested before the first MFMA
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                xb[ks] = P::ld_kc(&sm.xn[tt * 16][ks * 32], LDX);
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) w1f[jn][ks] = *reinterpret_cast<const frag*>(wmlp + ((2 * half + jn) * 3 + ks) * 1024 + l3 * 16);
            }
            MSST_SCHED_FENCE();
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) hh[jn] = P::mma(w1f[jn][ks], xb[ks], hh[jn]);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int n0 = (2 * half + jn) * 16 + 4 * g3;
#pragma unroll
                for (int r = 0; r < 4; ++r) hh[jn][r] = gelu_fast(hh[jn][r] + lnp[576 + n0 + r]);
                if (DROP && tok >= 0) hh[jn] = drop4(a.drop, 3, (unsigned)(tok * 16 + (n0 >> 2)), hh[jn]);
                P::st_nat(&sm.hb[tt * 16][(2 * half + jn) * 16], LDH, hh[jn]);
            }
        }
        lds_barrier();
        {
            f32x4 yy[3];
#pragma unroll
            for (int jm = 0; jm < 3; ++jm) yy[jm] = zero4();
            frag hbf[2], w2f[3][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                hbf[ks] = P::ld_kc(&sm.hb[tt * 16][ks * 32], LDH);
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) w2f[jm][ks] = *reinterpret_cast<const frag*>(wmlp + (12 + (3 * half + jm) * 2 + ks) * 1024 + l3 * 16);
            }
            MSST_SCHED_FENCE();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) yy[jm] = P::mma(w2f[jm][ks], hbf[ks], yy[jm]);
            if (tok >= 0) {
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) {
                    const int m0 = (3 * half + jm) * 16 + 4 * g3;
                    f32x4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = yy[jm][r] + lnp[480 + m0 + r];
                    if (DROP) o4 = drop4(a.drop, 4, (unsigned)(tok * 24 + (m0 >> 2)), o4);
                    o4 = o4 + x1r[jm];
                    *reinterpret_cast<f32x4*>(a.y + tok * 96 + m0) = o4;
                }
            }
        }
        F2_STAMP(16);
        // no barrier at the end of the tile: the next LN1 writes xn, last read before the barrier that precedes the second
        // MLP GEMM; ob / xch / st / hb are rewritten only after later barriers of the next tile
        F2_STAMP(17);
    }
}

int launch_block_fwd_hw(const BlockArgs& a, int grid, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    const size_t smem = sizeof(Fwd2Smem) + 640 * sizeof(float) + 24 * 1024;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_hw_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_hw_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ProfScope ps(K_BLOCK_FWD, st);
    if (a.drop.thr) hipLaunchKernelGGL(block_fwd_hw_kernel<true>, dim3(grid), dim3(512), smem, st, a);
    else hipLaunchKernelGGL(block_fwd_hw_kernel<false>, dim3(grid), dim3(512), smem, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
