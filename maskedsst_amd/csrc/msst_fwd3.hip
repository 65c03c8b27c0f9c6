// bf16 forward of one transformer block, ROLE-SPLIT (round 4; reference vit_spatial_spectral.py:22-104: PreNorm + Attention +
// residual, PreNorm + FeedForward + residual; a7-a10 of SURVEY.md section 8).  Same math, same dropout streams and the same
// HBM interface as block_fwd_hw_kernel (msst_fwd2.hip); what changes is WHEN the phases of a tile run.
//
// In msst_fwd2.hip all eight waves of the workgroup walk the same phase sequence in lockstep: the head phase (q / k / v
// projections + attention: ~11 k of the 26 k-cycle tile, where both waves of a SIMD fight for its VALU) and then five row-local
// phases separated by barriers (LN1, out-projection, LN2, two MLP GEMMs: ~13 k cycles for 0.8 k cycles of MFMA work -- LDS /
// L2 round trips and barriers nobody fills).  Here the two kinds of work are ROLES, one wave of each per SIMD:
//   * waves 0-3 ("A"): the head phase only, for TWO heads per tile (head w in round 0, head w + 4 in round 1): the whole
//     attention of a head in registers exactly as in msst_fwd2.hip; O rows go to the round's half of the bf16 O tile.
//   * waves 4-7 ("R"): everything row-local, software-pipelined over three tiles: while the A waves compute tile k they run
//     LN1 of tile k + 1, the first K half of the out-projection of tile k (as soon as round 0's O rows are complete) and the
//     second K half + bias / dropout / residual / LN2 / MLP of tile k - 1.  R wave (mh, rh) owns rows 32 rh .. + 31 x features
//     48 mh .. + 47 of the out-projection (each Wout fragment serves two row tiles, the full K runs on one wave: no K-half
//     exchange), the same rows x hidden half mh of the first MLP GEMM and x output half mh of the second.
// Four barriers per tile (the middle and the end of each round) are the only synchronisation: every cross-wave hand-over of
// the R pipeline (LN2 statistics of the two feature halves, LN2 rows, GELU rows) is placed across one of them.
// Measured: 293 -> 256 us per launch (B = 256, EnMAP shape; 252-262 us for the spatial launches, 266-274 for the spectral ones).
// What bounds it: the A wave is ONE in-order instruction stream per SIMD (11.7 k cycles per head alone: projections 3.1 k, their
// weight stream 1.4 k, attention MFMAs 1.3 k, softmax arithmetic 1.2 k, packing / stores / barriers the rest) and every piece costs about
// its own issue time (timing-only builds, round 4: LABNOTES.md); the R waves' path is 178 us on its own.  DESIGN.md section 5, LABNOTES.md round 4.
//
//   interval   A waves (tile k)                 R waves
//   q0         round 0: projections, j = 0, 1   out-projection K half 1 of tile k-1 (O of round 1), bias / dropout / +x -> x1,
//                                               partial LN2 statistics
//   q1         round 0: j = 2, 3                LN2 of tile k-1 -> XN2; the 24 Wout fragments of K half 0 requested
//   q2         round 1: projections, j = 0, 1   out-projection K half 0 of tile k (O of round 0); MLP GEMM 1 + GELU of tile k-1
//   q3         round 1: j = 2, 3                MLP GEMM 2 of tile k-1 -> y; LN1 of tile k+1 -> XN[(k+1) & 1]
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"
#include <type_traits>

#ifndef MSST_F3_RING
#define MSST_F3_RING 4
#endif
#ifndef MSST_F3_RPRIO
#define MSST_F3_RPRIO 1
#endif
#ifndef MSST_F3_BAND
#define MSST_F3_BAND 1   // spectral blocks: skip the score tiles outside the band j - 1 .. j + 1 at compile time
#endif
#ifndef MSST_F3_GROUP
#define MSST_F3_GROUP 10   // stack walk: a workgroup's tiles are cut into nmine / MSST_F3_GROUP groups; measured 3: +1.8 %, 6: -1.4 %, 10: -2.5 %, 100: -1.7 % (bench brackets, against per-block launches)
#endif
#if defined(MSST_LAB) && !defined(MSST_LAB_X1OLD)
#define MSST_LAB_X1OLD 0
#endif
// kernel-study builds only (-DMSST_LAB: msst_version() < 0, refused by the product loader): what each of the forward's saved extras costs
// -- MSST_LAB_NOLSE / MSST_LAB_NORSTD compile the softmax-statistics store / the LN1 rstd store out (tools/fwd_time.py, round 6)
#if !defined(MSST_LAB) || !defined(MSST_LAB_NOLSE)
#define MSST_F3_LSE_STORE 1
#else
#define MSST_F3_LSE_STORE 0
#endif
#if !defined(MSST_LAB) || !defined(MSST_LAB_NORSTD)
#define MSST_F3_RSTD_STORE 1
#else
#define MSST_F3_RSTD_STORE 0
#endif
#ifndef MSST_F3_RPRIO02
#define MSST_F3_RPRIO02 MSST_F3_RPRIO   // priority of the R waves in q0 / q2, where they finish early and wait for the A waves (q1 / q3, where the A waves wait for them: MSST_F3_RPRIO)
#endif
#ifndef MSST_F3_LN1Q
#define MSST_F3_LN1Q 3   // interval in which the R waves run LN1 of tile k + 1: 3 = at the end of q3 (rows requested in q2), 2 = at the end of q2 (rows
                         // requested at its start), the interval whose barrier the R waves otherwise sit out
#endif

#ifdef MSST_STAMPS
#define F3_STAMP(i) do { if (stamp_on) a.stamps[16 * wv + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define F3_STAMP(i) do { } while (0)
#endif

namespace msst {

namespace {

typedef PBF16 P;
typedef bf16_t elem;
typedef s16x8 frag;

struct Fwd3Smem {
    static constexpr int LDX = 96 + 16;    // row stride = 2 mod 4 sixteen-byte slots: conflict-free b128 fragment reads
    static constexpr int LDH = 64 + 8;
    static constexpr int LDO = 256 + 16;   // 34 slots = 2 mod 4
    elem xn[2][64][LDX];                   // LN1(x) of the tile of walk step k (by parity)
    elem ob[2][64][LDO];                   // attention output of round r: [row][local head * 64 + channel]
    elem xn2[64][LDX];                     // LN2(x1)
    elem hb[64][LDH];                      // GELU(W1 .) of the MLP
    float2 st[4][32];                      // LN2 partial statistics (mean, M2 over 48 features): [R wave][row of its 32]
    unsigned rowmap[64];                   // tile row -> (sequence slot << 16 | position), 0xffff = padding row: tile invariant
    unsigned long long vm[64];             // key-validity mask of a query row: bit k set <=> key row k belongs to the query's sequence (tile invariant)
    int seqb[4][64];                       // token of position 0 of every sequence slot, tiles of walk steps k - 1 .. k + 1 (by k & 3)
    float lnp[2][640];                     // ln1_g | ln1_b | bo | ln2_g | ln2_b | b2 | b1 of the block of a step; [1]: STACK only (the block a step pipeline
                                           // runs into while the stages behind it still work for the block before)
    float lseb[2][256];                    // softmax statistics of the tile in flight: [round][A wave (head 4 round + wave)][row]; the R waves copy them out (round 6)
    char wmlp[24 * 1024];                  // [w1: 12 frags | w2: 12 frags]
    const char* blktab[MSST_MAX_STACK][16];   // (STACK) every per-block operand of the run, in StackBlk's member order: block 0's + block x its stride
    unsigned steptab[1024];                // (STACK) walk step -> local tile index | block << 16 | lnp slot << 24 | idle << 31
};
constexpr int F3_MAX_STEPS = 1024;
constexpr int BT_WQKV = 0, BT_W1 = 2, BT_LN1G = 4, BT_X = 11;   // blktab columns: wqkv wout | w1 w2 | ln1_g ln1_b bo ln2_g ln2_b b1 b2 | x y x1 xn_out lse_out

// what a walk step works on (STACK: the step's block of the run; else the launch's one block)
struct StepBlk {
    const elem* wqkv; const elem* wout;
    const float* x; float* y; float* x1; elem* xn_out;   // (the statistics buffer: sm.blktab / a.lse_out where it is used)
    int layer;
};

__device__ __forceinline__ int sopaque3(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// HALF (MSST_FWD_HALF, round 6): the GEMM operands of the forward are IEEE half instead of bf16 -- same 16-bit containers, same fragment
// layouts, v_mfma_f32_16x16x32_f16 instead of ..._bf16 (same rate), v_cvt_pk_f16_f32 instead of v_cvt_pk_bf16_f32.  What is SAVED for the
// backward (LN1 rows, centred x1 rows) stays bf16: the backward kernels are bf16.
template <bool HALF>
__device__ __forceinline__ s16x4 cv4(f32x4 c) {
    if constexpr (HALF) return f2h4(c); else return f2bf4(c);
}
template <bool HALF>
__device__ __forceinline__ f32x4 mma16(frag a, frag b, f32x4 c) {
    if constexpr (HALF) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    } else return P::mma(a, b, c);
}
template <bool HALF>
__device__ __forceinline__ frag pack2f(f32x4 lo, f32x4 hi) {
    const s16x4 a = cv4<HALF>(lo), b = cv4<HALF>(hi);
    frag r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// fragment of 16 GATHERED rows of a fragment-packed [R][K] weight (see msst_fwd2.hip)
__device__ __forceinline__ frag ld_w_gather3(const elem* w, int K, int row32, int k0, int voff) {
    const int f = (row32 >> 4) * (K >> 5) + (k0 >> 5);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(w), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, f * 1024, 0);
    return __builtin_bit_cast(frag, v);
}

// weight-fragment pair number pi of head h: 0..5 q, 6..11 k, 12..17 gathered v (see msst_fwd2.hip).  Fragment f of the packed
// [3 H 64][96] matrix starts at byte 1024 f; rows 16 i .. of block c (0 q, 1 k, 2 v) of head h are fragments ((c H + h) 4 + i) 3 + ks:
// byte offset (c H + h) 12288 + 3072 i + 1024 ks.  hb = h * 12288 and HB = H * 12288 are wave-uniform values the caller keeps in two
// SGPRs per round: one s_add per request instead of the five-instruction recomputation from h (MSST_F3_WOFF = 0: the first version).
#ifndef MSST_F3_WOFF
#define MSST_F3_WOFF 1
#endif
__device__ __forceinline__ frag ld_wb3(const elem* w, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(w), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void load_pair3(int pi, frag (&out)[2], const elem* wqkv, int H, int h, const int (&voff)[2], int hb, int HB, int l16) {
    if (MSST_F3_WOFF) {
        if (pi < 12) {
            const int st = pi / 3, ks = pi % 3;
            const int o = hb + (st >> 1) * HB + (st & 1) * 6144 + ks * 1024;
            out[0] = ld_wb3(wqkv, l16, o);
            out[1] = ld_wb3(wqkv, l16, o + 3072);
        } else {
            const int mm = (pi - 12) / 3, ks = (pi - 12) % 3;
            const int o = hb + 2 * HB + mm * 6144 + ks * 1024;
            out[0] = ld_wb3(wqkv, voff[0], o);
            out[1] = ld_wb3(wqkv, voff[1], o);
        }
    } else if (pi < 12) {
        const int st = pi / 3, ks = pi % 3;
        const int r0 = ((st >> 1) * H + h) * 64 + (st & 1) * 32;
        out[0] = P::ld_w(wqkv, 96, r0, ks * 32);
        out[1] = P::ld_w(wqkv, 96, r0 + 16, ks * 32);
    } else {
        const int mm = (pi - 12) / 3, ks = (pi - 12) % 3;
        const int r32 = (2 * H + h) * 64 + mm * 32;
        out[0] = ld_w_gather3(wqkv, 96, r32, ks * 32, voff[0]);
        out[1] = ld_w_gather3(wqkv, 96, r32, ks * 32, voff[1]);
    }
}

}  // namespace

// STACK: ONE launch runs a RUN of blocks of one stack (same mode) -- the blocks of a stack never mix tiles, so a workgroup can take
// a tile through block after block.  The walk is over (tile, block) steps instead of tiles: the tiles of a workgroup are cut into
// groups of MSST_F3_GROUP or more (one group when it has fewer; a lone group of 1 or 2 is padded with idle steps to 3); a group goes
// through block 0, then block 1, ... so that two blocks of one tile are at least three steps apart -- the distance at which the rows
// a step's MLP stores (q3 of the step after it) have left the memory queue of their waves before LN1 of the next block requests
// them (q2 of the step after that).  Everything else is the per-block kernel: same arithmetic, same stores, bit-identical results;
// what goes away is the prologue + pipeline fill / drain of eleven of twelve launches.  Measured (tools/fwd_ab.py, LABNOTES round 5):
// 6.9 us per block saved, 4 % per step lost (block switches every group: MLP weights and small vectors reloaded, the weight ring
// running into lines that left L2; 150 more scalar / LDS instructions per step in the R waves) -- ahead below ~14 tiles per
// workgroup (batch 64, 5 / 6 tiles: -5.5 %; Houston shape: -7.3 %), behind above (batch 256 at the EnMAP shape, 20 / 22: +1.3 %): the host picks.
template <bool DROP, bool STACK, bool HALF>
__global__ __launch_bounds__(512, 2) void block_fwd_rs_kernel(typename std::conditional<STACK, StackArgs, BlockArgs>::type args) {
    const BlockArgs& a = [&]() -> const BlockArgs& { if constexpr (STACK) return args.base; else return args; }();
    typedef Fwd3Smem SM;
    constexpr int LDX = SM::LDX, LDH = SM::LDH, LDO = SM::LDO;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);

    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const int H = a.H, inner = H * 64;
    const TileMap tm = a.tm;
    const int L = tm.L;
    const int G = (int)gridDim.x;
    const int nmine = ((int)blockIdx.x < a.ntiles) ? (a.ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    // ---- the walk: steps 0 .. nsteps - 1 ----
    int nsteps = nmine;
    if constexpr (STACK) {
        // groups of 3 .. 5 tiles (a lone group of 1 or 2 is padded with idle steps to 3); group j walks its tiles block by block
        const int nblk = args.nblk;
        const int ng = max(1, nmine / MSST_F3_GROUP), gbase = nmine / ng, gextra = nmine - gbase * ng;   // group j holds gbase + (j < gextra) tiles
        const int p1 = max(gbase + 1, 3), p0 = max(gbase, 3);                                // ... and takes nblk * p steps
        nsteps = nmine ? nblk * (gextra * p1 + (ng - gextra) * p0) : 0;
        for (int s_ = tid; s_ < nsteps; s_ += 512) {
            const int head = gextra * nblk * p1;
            int j, r, pp, sz, before;
            if (s_ < head) { j = s_ / (nblk * p1); r = s_ - j * nblk * p1; pp = p1; sz = gbase + 1; before = j * (gbase + 1); }
            else { j = gextra + (s_ - head) / (nblk * p0); r = s_ - head - (j - gextra) * nblk * p0; pp = p0; sz = gbase; before = gextra * (gbase + 1) + (j - gextra) * gbase; }
            const int blk = r / pp, pos = r - blk * pp;
            const unsigned idle = pos >= sz ? 1u : 0u;
            sm.steptab[s_] = (unsigned)(before + min(pos, sz - 1)) | ((unsigned)blk << 16) | ((unsigned)((j * nblk + blk) & 1) << 24) | (idle << 31);
        }
    }
    if constexpr (STACK) {
        // per-block operands -> LDS, straight from the kernel-argument segment (StackBlk is sixteen pointers, StackStride sixteen strides, same order)
        static_assert(offsetof(StackBlk, lse_out) == 15 * 8 && offsetof(StackStride, lse_out) == 15 * 4, "blktab follows StackBlk's member order");
        for (int e = tid; e < args.nblk * 16; e += 512) {
            const int j = e >> 4, f = e & 15;
            auto kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
            const char* p0 = *(const char* const __attribute__((address_space(4)))*)(kp + offsetof(StackArgs, b0) + f * 8);
            const int stride = *(const int __attribute__((address_space(4)))*)(kp + offsetof(StackArgs, st) + f * 4);
            const char* pj = p0 ? p0 + (long)j * stride : nullptr;
            if (f == BT_X && j) pj = reinterpret_cast<const char*>(args.x_rest) + (long)j * args.st.y;
            sm.blktab[j][f] = pj;
        }
    }
    // The step words of the steps a loop iteration touches (k - 2 .. k + 2) are carried in scalars from iteration to iteration (one
    // LDS read per step, a step ahead): a lookup per use -- table read, readfirstlane, then a pointer fetch that depends on it, in
    // every stage of the R waves -- cost the first version of the stack walk more than the launches it saved (275 vs 268 us per block).
    constexpr unsigned F3_IDLE = 0x80000000u;
    auto load_word = [&](int k) -> unsigned {
        if constexpr (STACK) return (k >= 0 && k < nsteps) ? (unsigned)__builtin_amdgcn_readfirstlane((int)sm.steptab[k]) : F3_IDLE;
        else return (k >= 0 && k < nmine) ? (unsigned)k : F3_IDLE;
    };
    int kc = 0;                                   // the step the cached words are centred on: the walk loops count in it, so that k - kc below folds to a constant
    unsigned swm2 = F3_IDLE, swm1 = F3_IDLE, sw0 = F3_IDLE, swp1 = F3_IDLE, swp2 = F3_IDLE;
    auto step_word = [&](int k) -> unsigned {
        if constexpr (!STACK) return load_word(k);
        else { const int d = k - kc; return d == 0 ? sw0 : d == 1 ? swp1 : d == -1 ? swm1 : d == 2 ? swp2 : d == -2 ? swm2 : load_word(k); }
    };
    auto words_init = [&]() { kc = 0; swm2 = F3_IDLE; swm1 = F3_IDLE; sw0 = load_word(0); swp1 = load_word(1); swp2 = load_word(2); };
    auto words_advance = [&]() { swm2 = swm1; swm1 = sw0; sw0 = swp1; swp1 = swp2; swp2 = load_word(kc + 3); };   // (last statement of an iteration: kc is still the old step)
    // tile of walk step k; < 0: no tile (outside the walk, or an idle step)
    auto tile_at = [&](int k) -> int {
        const unsigned w_ = step_word(k);
        return (w_ >> 31) ? -1 : (int)blockIdx.x + (int)(w_ & 0xffffu) * G;
    };
    auto blk_at = [&](int k) -> int { if constexpr (STACK) return (int)((step_word(k) >> 16) & 0xffu); else return 0; };
    auto slot_at = [&](int k) -> int { if constexpr (STACK) return (int)((step_word(k) >> 24) & 1u); else return 0; };
    auto in_walk = [&](int k) -> bool { return k >= 0 && k < nsteps; };
    // does the walk move on to another block between steps k0 and k1 = k0 + 1 (another group's block 0 included)?
    auto epoch_differs = [&](int k1, int k0) -> bool {
        if constexpr (STACK) return in_walk(k1) && in_walk(k0) && ((step_word(k1) ^ step_word(k0)) & 0x01ff0000u) != 0u;
        else return false;
    };
    // per-block pointers: block 0's + block index x a byte stride (the host checked that every array of the call is affine in the
    // block index: msst_block_fwd_stack) -- scalar arithmetic, no table in memory
    auto at_blk = [&](const void* p0, int stride, int blk) -> const char* { return reinterpret_cast<const char*>(p0) + (long)blk * (long)stride; };
    auto step_blk = [&](int k) -> StepBlk {
        StepBlk q;
        if constexpr (STACK) {
            const int j = blk_at(k);
            const StackBlk& b0 = args.b0;
            const StackStride& st = args.st;
            q.wqkv = reinterpret_cast<const elem*>(at_blk(b0.wqkv, st.wqkv, j)); q.wout = reinterpret_cast<const elem*>(at_blk(b0.wout, st.wout, j));
            // (the row pointers are used by the R waves only, as the base of per-lane addresses: from the table in LDS -- the walk of
            // the R waves spilled 77 scalar registers to VGPR lanes with all of the run's operands held in scalars, 247 v_readlane a step)
            q.x = reinterpret_cast<const float*>(sm.blktab[j][BT_X]);   // block j > 0 reads block j - 1's y
            q.y = reinterpret_cast<float*>(const_cast<char*>(sm.blktab[j][BT_X + 1]));
            q.x1 = reinterpret_cast<float*>(const_cast<char*>(sm.blktab[j][BT_X + 2]));
            q.xn_out = reinterpret_cast<elem*>(const_cast<char*>(sm.blktab[j][BT_X + 3]));
            q.layer = b0.layer + j;
        } else {
            q.wqkv = reinterpret_cast<const elem*>(a.w.wqkv); q.wout = reinterpret_cast<const elem*>(a.w.wout);
            q.x = a.x; q.y = a.y; q.x1 = a.x1; q.xn_out = reinterpret_cast<elem*>(a.xn_out); q.layer = a.drop.layer;
        }
        return q;
    };
    auto drop_at = [&](int k) -> Drop { Drop d = a.drop; if constexpr (STACK) d.layer = args.b0.layer + blk_at(k); return d; };
    auto lnp_at = [&](int k) -> const float* { return sm.lnp[slot_at(k)]; };
    // small parameter vectors of block `blk` -> lnp[slot] (96 threads, t96 = 0 .. 95)
    struct LnpRegs { float v[7]; };
    auto fetch_lnp = [&](int blk, int t96) -> LnpRegs {
        const float *g1, *b1_, *bo, *g2, *b2_, *bb2, *bb1;
        if constexpr (STACK) {
            auto fp_ = [&](int f) { return reinterpret_cast<const float*>(sm.blktab[blk][f]); };
            g1 = fp_(BT_LN1G); b1_ = fp_(BT_LN1G + 1); bo = fp_(BT_LN1G + 2); g2 = fp_(BT_LN1G + 3); b2_ = fp_(BT_LN1G + 4); bb1 = fp_(BT_LN1G + 5); bb2 = fp_(BT_LN1G + 6);
        }
        else { g1 = a.w.ln1_g; b1_ = a.w.ln1_b; bo = a.w.bo; g2 = a.w.ln2_g; b2_ = a.w.ln2_b; bb2 = a.w.b2; bb1 = a.w.b1; }
        LnpRegs r;
        r.v[0] = g1[t96]; r.v[1] = b1_[t96]; r.v[2] = bo[t96]; r.v[3] = g2[t96]; r.v[4] = b2_[t96]; r.v[5] = bb2[t96];
        r.v[6] = bb1[t96 < 64 ? t96 : 0];
        return r;
    };
    auto commit_lnp = [&](int slot, int t96, const LnpRegs& r) {
        float* lnp = sm.lnp[slot];
#pragma unroll
        for (int i = 0; i < 6; ++i) lnp[96 * i + t96] = r.v[i];
        if (t96 < 64) lnp[576 + t96] = r.v[6];
    };
    auto load_lnp = [&](int slot, int blk, int t96) { commit_lnp(slot, t96, fetch_lnp(blk, t96)); };
    auto mlp_w = [&](int blk, int which) -> const char* {
        if constexpr (STACK) return sm.blktab[blk][BT_W1 + which];
        else return reinterpret_cast<const char*>(which ? a.w.w2 : a.w.w1);
    };

    // ---- common prologue: small parameter vectors, MLP weights, token tables ----
    if (STACK) __syncthreads();   // the step table
    words_init();
    if (tid < 96) load_lnp(slot_at(0), blk_at(0), tid);
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int f = wv * 3 + i3;
        dma_frag(f < 12 ? mlp_w(blk_at(0), 0) + f * 1024 : mlp_w(blk_at(0), 1) + (f - 12) * 1024, sm.wmlp + f * 1024);
    }
    wait_vm0();
    if (tid < 64) {
        const int sq = tid / L, ps = tid - sq * L;
        sm.rowmap[tid] = ((unsigned)(sq >= tm.TS ? 0xffff : sq) << 16) | (unsigned)ps;
    }
    // sequence bases of walk step k (tile tile_at(k)) -> seqb[k & 3]; steps outside the walk: all -1 (padding)
    auto fill_seq = [&](int k, int t64) {
        const int tile_ = tile_at(k);
        const int q = tile_ * tm.TS + t64;
        int base = -1;
        if (tile_ >= 0 && t64 < tm.TS && q < tm.nseq) {
            if (tm.mode == 0) base = q * tm.N;
            else { const int b = tm.nshift >= 0 ? (q >> tm.nshift) : q / tm.N; base = b * tm.T + (q - b * tm.N); }
        }
        sm.seqb[k & 3][t64] = base;
    };
    if (tid < 64) { fill_seq(0, tid); fill_seq(1, tid); }
    if (tid >= 64 && tid < 128) {
        const int r = tid - 64, sq = r / L;
        unsigned long long m = ~0ull;   // (padding rows see every key: finite garbage that is never stored, not NaN)
        if (sq < tm.TS && L < 64) m = ((1ull << L) - 1ull) << (sq * L);
        sm.vm[r] = m;
    }
    __syncthreads();
    auto tok_of = [&](int k, int r) -> long {
        const unsigned sp = sm.rowmap[r];
        const int base = sm.seqb[k & 3][min((int)(sp >> 16), 63)];
        return ((sp >> 16) == 0xffffu || base < 0) ? -1 : (long)(base + (int)(sp & 0xffffu) * (tm.mode == 0 ? 1 : tm.N));
    };

    if (wv < 4) {
        // =====================================================================================================
        // A role: heads wv (round 0) and wv + 4 (round 1) of every tile, all in registers (msst_fwd2.hip's head phase)
        // =====================================================================================================
        int voff[2];
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int rs = 8 * (c >> 2) + (c & 3) + 4 * hi;   // 0..31
            voff[hi] = ((rs >> 4) * 3) * 1024 + (g * 16 + (rs & 15)) * 16;   // K = 96 -> 3 fragments per 16 rows
        }
        // do the 16 queries of tile j only meet the key tiles {0,1} {0,1,2} {1,2,3} {2,3}?  (true for every L <= 21: a sequence
        // that starts in tile j - 1 or j ends before tile j + 2 starts; wave uniform, tile invariant)
        bool band = L < 64 && MSST_F3_BAND;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s_lo = (16 * j) / L, s_hi = min((16 * j + 15) / L, tm.TS - 1);
            const int k_lo = s_lo * L, k_hi = (s_hi + 1) * L - 1;   // s_lo > s_hi: an all-padding query tile, anything goes
            const unsigned allow = j == 0 ? 0x3u : j == 1 ? 0x7u : j == 2 ? 0xeu : 0xcu;
            for (int t = 0; t < 4; ++t)
                if (s_lo <= s_hi && 16 * t <= k_hi && 16 * t + 15 >= k_lo && !((allow >> t) & 1u)) band = false;
        }
        constexpr int NR = MSST_F3_RING;
        static_assert(NR >= 2 && NR <= 9, "ring depth");
        frag ring[NR][2];
        // pair to request into slot pi % NR once pair pi is consumed: NR ahead in the 18-pair stream of a head, wrapping to the
        // first pairs of the NEXT head of this wave (the other round's)
        // (pair p of a head always lives in slot p % NR: for pi >= 18 - NR the freed slots pi % NR run through 0 .. NR - 1 exactly once)
        auto next_pair = [](int pi) { return pi + NR < 18 ? pi + NR : pi % NR; };
        const int HB = H * 12288, l16 = l * 16;
        {
            const elem* wq0 = step_blk(0).wqkv;
#pragma unroll
            for (int pi = 0; pi < NR; ++pi) load_pair3(pi, ring[pi], wq0, H, wv, voff, wv * 12288, HB, l16);
        }

        __syncthreads();   // (P1) LN1 of the first tile is in XN[0]
        for (kc = 0; kc < nsteps; ++kc) {
            const int k = kc;
            const int tile = tile_at(k);   // (STACK: < 0 for an idle step -- its rows are all padding, nothing of it is stored)
            const StepBlk sb = step_blk(k);
            const elem* wqkv = sb.wqkv;
            const elem* wqkv_next = (STACK && k + 1 < nsteps) ? step_blk(k + 1).wqkv : sb.wqkv;   // where the weight ring wraps to at the end of round 1
            const Drop drop_k = drop_at(k);
#ifdef MSST_STAMPS
            const bool stamp_on = (a.dbg & 8) && a.stamps && blockIdx.x == 100 && l == 0 && k == nsteps / 2;
#endif
            F3_STAMP(0);
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const elem* wq_wrap = rd == 1 ? wqkv_next : wqkv;   // the pairs past this head's last belong to the wave's next head
                const int h = wv + 4 * rd;                  // this round's head
                const int hn = wv + 4 * (1 - rd);           // the head whose first pairs follow in the weight stream
                const int hb_c = sopaque3(h * 12288), hb_n = sopaque3(hn * 12288);   // (opaque per round: folded into 36 per-pair constants they would be hoisted out of the walk and spilled)
                frag qB[4][2], kA[4][2], vA[4][2];
                {
                    frag xf[4][3];   // LN1(x) as operand fragments: [16-row tile][k-step]
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks) xf[t][ks] = P::ld_kc(&sm.xn[k & 1][t * 16][ks * 32], LDX);
#pragma unroll
                    for (int st = 0; st < 6; ++st) {
                        if (st < 4) {
                            const int m = st & 1;
                            f32x4 ca[4], cb[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) { ca[t] = zero4(); cb[t] = zero4(); }
#pragma unroll
                            for (int ks = 0; ks < 3; ++ks) {
                                const int pi = 3 * st + ks;
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    ca[t] = mma16<HALF>(ring[pi % NR][0], xf[t][ks], ca[t]);
                                    cb[t] = mma16<HALF>(ring[pi % NR][1], xf[t][ks], cb[t]);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                load_pair3(next_pair(pi), ring[pi % NR], pi + NR < 18 ? wqkv : wq_wrap, H, MSST_F3_WOFF ? 0 : sopaque3(pi + NR < 18 ? h : hn), voff, pi + NR < 18 ? hb_c : hb_n, HB, l16);
                                __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                if (st < 2) qB[t][m] = pack2f<HALF>(ca[t], cb[t]); else kA[t][m] = pack2f<HALF>(ca[t], cb[t]);
                            }
                        } else {
                            const int mm = st - 4;
                            f32x4 cl[4], ch[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) { cl[t] = zero4(); ch[t] = zero4(); }
#pragma unroll
                            for (int ks = 0; ks < 3; ++ks) {
                                const int pi = 3 * st + ks;
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    cl[t] = mma16<HALF>(xf[t][ks], ring[pi % NR][0], cl[t]);
                                    ch[t] = mma16<HALF>(xf[t][ks], ring[pi % NR][1], ch[t]);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                load_pair3(next_pair(pi), ring[pi % NR], pi + NR < 18 ? wqkv : wq_wrap, H, MSST_F3_WOFF ? 0 : sopaque3(pi + NR < 18 ? h : hn), voff, pi + NR < 18 ? hb_c : hb_n, HB, l16);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            vA[2 * mm][0] = pack2f<HALF>(cl[0], cl[1]);     vA[2 * mm][1] = pack2f<HALF>(cl[2], cl[3]);
                            vA[2 * mm + 1][0] = pack2f<HALF>(ch[0], ch[1]); vA[2 * mm + 1][1] = pack2f<HALF>(ch[2], ch[3]);
                        }
                    }
                }
                F3_STAMP(1 + 5 * rd);   // projections done
#if defined(MSST_LAB) && defined(MSST_LAB_QKV)
                // kernel-study build only (tools/gate_qkv.py): what handing q / k / v to the backward would cost the forward -- the head's
                // 24 operand fragments (24.5 KB per tile and head, 1 GB per block at the bench shape) stored to a scratch
                if (a.stamps && tile >= 0) {
                    char* qs = reinterpret_cast<char*>(a.stamps) + (long)(tile * H + h) * 24576 + (threadIdx.x & 63) * 16;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            *reinterpret_cast<frag*>(qs + (2 * t + m) * 1024) = qB[t][m];
                            *reinterpret_cast<frag*>(qs + 8192 + (2 * t + m) * 1024) = kA[t][m];
                            *reinterpret_cast<frag*>(qs + 16384 + (2 * t + m) * 1024) = vA[t][m];
                        }
                }
#endif
                // ---- attention of head h, TWO query tiles at a time, phase by phase: the A wave is alone on its SIMD's VALU most of
                // the time, so the latency of its own dependent chain (S MFMAs -> max -> exp -> sum -> 1 / x -> dropout -> P V MFMAs ->
                // pack) is what it waits for; two independent query tiles in flight fill those slots.  O rows go to the round's
                // half of the O tile. ----
                const float cs = a.scale * 1.44269504088896340736f;
                // NM0 / NM1: key tiles (bit t = keys 16 t .. + 15) the two query tiles of the pass can see at all -- 0xf in the spatial
                // blocks; in the spectral blocks (several short sequences per tile) every other 16 x 16 score tile is masked anyway and
                // is skipped at COMPILE time (MFMAs, exps, dropout hashes, and the P V MFMAs of a fully skipped 32-key chunk): for the
                // 20-token sequences of the EnMAP shape 10 of the 16 score tiles remain.  (A run-time test per tile breaks the pass
                // into basic blocks and costs more than it saves -- msst_fwd2.hip's MSST_F2_SKIP, LABNOTES round 1.)
                auto attn_pair = [&](auto jp_c, auto masked_c, auto nm0_c, auto nm1_c) {
                    constexpr int jp = decltype(jp_c)::value;
                    constexpr bool MASKED = decltype(masked_c)::value;
                    constexpr unsigned NM[2] = {(unsigned)decltype(nm0_c)::value, (unsigned)decltype(nm1_c)::value};
                    f32x4 s[2][4];
                    f32x4 o[2][4];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (!((NM[u] >> t) & 1u)) { s[u][t] = zero4(); continue; }
                            s[u][t] = mma16<HALF>(kA[t][0], qB[2 * jp + u][0], zero4());   // C[i = key][j = query]
                            s[u][t] = mma16<HALF>(kA[t][1], qB[2 * jp + u][1], s[u][t]);
                        }
                    {
                        // bit position of key 16 t + 4 g + r inside its 32-bit half of a 64-bit row mask: 16 (t & 1) + 4 g + r
                        int lq = threadIdx.x & 63;
                        asm volatile("" : "+v"(lq));
                        const int cq = lq & 15, sh0 = 4 * (lq >> 4);
                        float mx[2] = {-INFINITY, -INFINITY};
                        if (!MASKED) {
#pragma unroll
                            for (int u = 0; u < 2; ++u)
#pragma unroll
                                for (int t = 0; t < 4; ++t)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) mx[u] = fmaxf(mx[u], s[u][t][r]);
                        } else {
                            // keys outside the query's own sequence are masked.  The validity of a key is one bit of the query row's
                            // 64-bit mask (LDS, tile invariant): a sign-extending 1-bit field extract gives 0 / ~0 and v_bfi selects
                            // score or -inf -- two full-rate instructions per score and no lane masks (as compares against [lo, hi)
                            // the 64 lane masks of a wave were hoisted into SGPR pairs and spilled)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const unsigned long long vmq = sm.vm[(2 * jp + u) * 16 + cq];
                                const int vlo = (int)(unsigned)vmq, vhi = (int)(unsigned)(vmq >> 32);
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    if (!((NM[u] >> t) & 1u)) continue;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const int m = __builtin_amdgcn_sbfe(t < 2 ? vlo : vhi, 16 * (t & 1) + sh0 + r, 1);   // 0 or -1
                                        const float v = __int_as_float((__float_as_int(s[u][t][r]) & m) | (~m & (int)0xff800000));
                                        s[u][t][r] = v;
                                        mx[u] = fmaxf(mx[u], v);
                                    }
                                }
                            }
                        }
                        float mc[2], sum[2] = {0.f, 0.f}, inv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) mc[u] = colgroup_max(mx[u]) * cs;
                        // exp(scale (s - max)) = exp2(s c - max c), c = scale log2 e: one FMA + one v_exp per element
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                if (!((NM[u] >> t) & 1u)) continue;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(s[u][t][r], cs, -mc[u])); s[u][t][r] = e; sum[u] += e; }
                            }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float st = colgroup_sum(sum[u]);
                            inv[u] = (DROP ? drop_k.scale : 1.f) * __builtin_amdgcn_rcpf(st);   // the dropout scale rides on the normalisation
                            // saved for the backward: p = exp2(s c - lse) with lse = max c + log2(sum) -- 256 bytes per (tile, head).  Handed to
                            // the R waves through LDS (a query's value is replicated over the four lane groups, which all write it); they copy a
                            // round's 1 KB out as whole lines in the next interval.  (Round 5 stored it from here with one buffer store per
                            // query tile: a store sits in this wave's in-order memory counter in front of every later weight-fragment wait --
                            // 6 us of the forward's 256, tools/fwd_time.py with -DMSST_LAB_NOLSE.)
                            if (MSST_F3_LSE_STORE) sm.lseb[rd][wv * 64 + (2 * jp + u) * 16 + cq] = mc[u] + __builtin_amdgcn_logf(st);
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            // site 1: drop4(site 1, ((tile H + h) 64 + query) 16 + t 4 + g).  (The R waves hashing the keep bits into 64-bit row
                            // masks in LDS for the A waves to AND in measured 267 -> 290 us, LABNOTES round 4: their 32 hashes per lane and tile
                            // lengthen the q1 / q3 intervals.)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                if (!((NM[u] >> t) & 1u)) continue;
                                s[u][t] = s[u][t] * inv[u];
                                if (DROP) {
                                    s[u][t] = drop4_noscale(drop_k, 1, (unsigned)(((tile * H + h) * 64 + (2 * jp + u) * 16 + cq) * 16 + t * 4 + (sh0 >> 2)), s[u][t]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const frag p0 = pack2f<HALF>(s[u][0], s[u][1]), p1 = pack2f<HALF>(s[u][2], s[u][3]);
#pragma unroll
                        for (int dd = 0; dd < 4; ++dd) {
                            o[u][dd] = zero4();
                            if (NM[u] & 3u) o[u][dd] = mma16<HALF>(vA[dd][0], p0, o[u][dd]);       // C[i = gathered channel][j = query]
                            if (NM[u] & 12u) o[u][dd] = mma16<HALF>(vA[dd][1], p1, o[u][dd]);
                        }
                    }
                    // pack2(o[2u'], o[2u' + 1]) holds, in lane (c, g), the natural channels 32 u' + 8 g .. + 7 of query row 16 j + c
                    int l4 = threadIdx.x & 63;
                    asm volatile("" : "+v"(l4));
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        elem* orow = &sm.ob[rd][(2 * jp + u) * 16 + (l4 & 15)][wv * 64 + 8 * (l4 >> 4)];
                        *reinterpret_cast<frag*>(orow) = pack2f<HALF>(o[u][0], o[u][1]);
                        *reinterpret_cast<frag*>(orow + 32) = pack2f<HALF>(o[u][2], o[u][3]);
                    }
                    // q0 / q2 (middle of the round), q1 / q3 (its end: the round's O rows are complete)
                    F3_STAMP(2 + 5 * rd + 2 * jp);
                    lds_barrier();
                    F3_STAMP(3 + 5 * rd + 2 * jp);
                };
                typedef std::integral_constant<int, 0> I0;
                typedef std::integral_constant<int, 1> I1;
                if (L == 64) {
                    attn_pair(I0{}, std::false_type{}, std::integral_constant<int, 0xf>{}, std::integral_constant<int, 0xf>{});
                    attn_pair(I1{}, std::false_type{}, std::integral_constant<int, 0xf>{}, std::integral_constant<int, 0xf>{});
                } else if (band) {   // every query tile j only meets key tiles j - 1 .. j + 1 (checked against the row map in the prologue)
                    attn_pair(I0{}, std::true_type{}, std::integral_constant<int, 0x3>{}, std::integral_constant<int, 0x7>{});
                    attn_pair(I1{}, std::true_type{}, std::integral_constant<int, 0xe>{}, std::integral_constant<int, 0xc>{});
                } else {
                    attn_pair(I0{}, std::true_type{}, std::integral_constant<int, 0xf>{}, std::integral_constant<int, 0xf>{});
                    attn_pair(I1{}, std::true_type{}, std::integral_constant<int, 0xf>{}, std::integral_constant<int, 0xf>{});
                }
            }
            if constexpr (STACK) words_advance();
        }
        // (the A waves' walk ends here)
        // the R waves finish the last tile: four more intervals
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_barrier();
        return;
    }

    // =========================================================================================================
    // R role: row-local work, pipelined over tiles k - 1 (second half of the out-projection .. MLP), k (first half of the
    // out-projection) and k + 1 (LN1)
    // =========================================================================================================
    const int rw = wv - 4, mh = rw & 1, rh = rw >> 1;
    const int rt = tid - 256;                   // 0..255: LN1 thread <-> (row rt / 4, 24 features 16 (i / 4) + 4 (rt % 4) + i % 4)

    // ---- LN1 of walk step k: x rows -> XN[k & 1] (+ the bf16 rows to HBM for the attention backward) ----
    f32x4 xv[6];   // this thread's 24 row values of the tile LN1 processes next (requested one interval ahead)
    auto request_ln1 = [&](int k) {
        int rt = (int)threadIdx.x - 256;
        asm volatile("" : "+v"(rt));
        const long tok = tok_of(k, rt >> 2);
        const float* xrow = step_blk(k).x + (tok >= 0 ? tok : 0) * 96 + 4 * (rt & 3);
#pragma unroll
        for (int i = 0; i < 6; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xrow + 16 * i);
    };
    auto ln1 = [&](int k) {
        int rt = (int)threadIdx.x - 256;
        asm volatile("" : "+v"(rt));
        const int lr = rt >> 2, part = rt & 3;
        const long tok = tok_of(k, lr);
        const float* lnp = lnp_at(k);
        elem* const xn_out = step_blk(k).xn_out;
        // statistics buffer of the step's block: [tiles][H][64] lse | [tokens] rstd of this LN1 (MSST_SAVED_RSTD: the fused row-local
        // backward rebuilds xhat from the bf16 rows below and this value instead of re-reading x)
        float* stats;
        if constexpr (STACK) stats = reinterpret_cast<float*>(const_cast<char*>(sm.blktab[blk_at(k)][BT_X + 4]));
        else stats = a.lse_out;
        float v[24];
#pragma unroll
        for (int i = 0; i < 6; ++i) { v[4*i] = xv[i][0]; v[4*i+1] = xv[i][1]; v[4*i+2] = xv[i][2]; v[4*i+3] = xv[i][3]; }
        if (tok < 0) {   // padding row: the request read a clamped address, normalise zeros
#pragma unroll
            for (int i = 0; i < 24; ++i) v[i] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) s += v[i];
        const float mean = quad_sum(s) * (1.f / 96.f);
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) { const float d = v[i] - mean; vs += d * d; }
        const float rstd = rsqrtf(quad_sum(vs) * (1.f / 96.f) + 1e-5f);
        if (MSST_F3_RSTD_STORE && stats && tok >= 0 && part == 0) stats[(long)a.ntiles * H * 64 + tok] = rstd;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int f0 = 16 * i + 4 * part;
            f32x4 n4;
#pragma unroll
            for (int e = 0; e < 4; ++e) n4[e] = (v[4*i+e] - mean) * rstd * lnp[f0 + e] + lnp[96 + f0 + e];
            const s16x4 nb = f2bf4(n4);   // (the backward's rows are bf16 in either mode)
            *reinterpret_cast<s16x4*>(&sm.xn[k & 1][lr][f0]) = HALF ? cv4<HALF>(n4) : nb;
            if (xn_out && tok >= 0) *reinterpret_cast<s16x4*>(xn_out + tok * 96 + f0) = nb;
        }
    };

    // out-projection state of the tile in flight: rows 32 rh + 16 jj + c, features 48 mh + 16 i + 4 g .. + 3
    f32x4 acc[2][3];
    f32x4 x1r[2][3];   // x1 of the owned rows / features of tile k - 1: in registers until the end of its MLP
    f32x4 xr[2][3];    // residual x values of the tile whose out-projection completes next
    float mean_w[2], m2_w[2], mean_r[2] = {0.f, 0.f};
    // per-thread indices are re-derived from a laundered lane id inside every stage: derived once, the compiler hoists two dozen
    // lane-dependent addresses out of the walk, keeps them live across all stages and spills them (each reload carries a vmcnt(0)
    // that also waits for the weight fragments in flight)
#define F3_LANE() int l3 = threadIdx.x & 63; asm volatile("" : "+v"(l3)); const int g3 = l3 >> 4, c3 = l3 & 15; (void)g3; (void)c3

    auto request_xr = [&](int k) {
        F3_LANE();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const long tok = tok_of(k, 32 * rh + 16 * jj + c3);
            const float* xrow = step_blk(k).x + (tok >= 0 ? tok : 0) * 96 + 48 * mh + 4 * g3;
#pragma unroll
            for (int i = 0; i < 3; ++i) xr[jj][i] = *reinterpret_cast<const f32x4*>(xrow + 16 * i);
        }
    };
    // one K half (round rd's O rows) of the out-projection: C[i = feature][j = row].  Its 24 weight fragments are requested at
    // the END of the interval before (a light one: LN2, or MLP GEMM 2 + LN1 with their registers already dead), so that the
    // barrier wait hides the L2 round trip: requested inside the phase through a ring of four k-steps, every refill sat out a
    // whole L2 latency (6 MFMAs of cover per k-step): 6.6 k cycles for 48 MFMAs.
    frag fw[8][3];
    auto request_fw = [&](int rd, int kstep) {
        const elem* wout = step_blk(kstep).wout;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int i = 0; i < 3; ++i) fw[s8][i] = P::ld_w(wout, inner, (3 * sopaque3(mh) + i) * 16, (8 * rd + s8) * 32);
    };
    auto outproj = [&](int rd) {
        F3_LANE();
        frag fo[4][2];   // O row fragments of a k-step, requested three k-steps ahead
        swpipe<8, 3>(
            [&](int s8) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) fo[s8 % 4][jj] = P::ld_kc(&sm.ob[rd][(2 * rh + jj) * 16][s8 * 32], LDO);
            },
            [&](int s8) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[jj][i] = mma16<HALF>(fw[s8][i], fo[s8 % 4][jj], acc[jj][i]);
            });
    };
    // bias, dropout, residual -> x1 (registers + HBM); partial LN2 statistics of the 48 owned features -> ST
    auto epilogue1 = [&](int k) {
        F3_LANE();
        const float* lnp = lnp_at(k);
        const Drop drop_k = drop_at(k);
        float* const x1p = step_blk(k).x1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const long tok = tok_of(k, 32 * rh + 16 * jj + c3);
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m0 = (3 * mh + i) * 16 + 4 * g3;
                f32x4 o4 = acc[jj][i];
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] += lnp[192 + m0 + r];
                if (DROP && tok >= 0) o4 = drop4(drop_k, 2, (unsigned)(tok * 24 + (m0 >> 2)), o4);
                o4 = o4 + xr[jj][i];
                x1r[jj][i] = o4;
                s1 += (o4[0] + o4[1]) + (o4[2] + o4[3]);
                // saved for the MLP-half backward: fp32 here; the bf16 form (MSST_X1_BF16: a quarter of this kernel's writes less) leaves
                // in ln2(), CENTRED on the row mean that is only known there
                if (x1p && tok >= 0 && !a.x1_bf16) *reinterpret_cast<f32x4*>(x1p + tok * 96 + m0) = o4;
#ifdef MSST_LAB
                if (MSST_LAB_X1OLD && x1p && tok >= 0 && a.x1_bf16) *reinterpret_cast<s16x4*>(reinterpret_cast<unsigned short*>(x1p) + tok * 96 + m0) = f2bf4(o4);   // (round 4's uncentred rows, for timing only)
#endif
            }
            const float mw = colgroup_sum(s1) * (1.f / 48.f);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = x1r[jj][i][r] - mw; m2 += d * d; }
            m2 = colgroup_sum(m2);
            mean_w[jj] = mw; m2_w[jj] = m2;
            if (g3 == 0) sm.st[rw][16 * jj + c3] = make_float2(mw, m2);
        }
    };
    // LN2 of the owned rows (statistics combined with the other feature half) -> XN2
    auto ln2 = [&](int k) {
        F3_LANE();
        const float* lnp = lnp_at(k);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const float2 other = sm.st[rw ^ 1][16 * jj + c3];
            const float mean = 0.5f * (mean_w[jj] + other.x);
            const float dm = mean_w[jj] - other.x;
            const float var = (m2_w[jj] + other.y + dm * dm * 24.f) * (1.f / 96.f);   // Chan's pairwise combination, n = 48 + 48
            const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m0 = (3 * mh + i) * 16 + 4 * g3;
                f32x4 n4;
#pragma unroll
                for (int r = 0; r < 4; ++r) n4[r] = (x1r[jj][i][r] - mean) * rstd * lnp[288 + m0 + r] + lnp[384 + m0 + r];
                *reinterpret_cast<s16x4*>(&sm.xn2[32 * rh + 16 * jj + c3][m0]) = cv4<HALF>(n4);
            }
            mean_r[jj] = mean;
        }
    };
    // MSST_X1_BF16: the backward reads the saved mid-residual rows only through LN2 (statistics, xhat), which does not see a per-row
    // constant -- so the bf16 rows are bf16(x1 - row mean): their rounding error is relative to the row's spread, not to its offset
    // (a trained residual stream with |mean| >> std would otherwise lose its LN2 statistics to the rounding).  The mean is known
    // since ln2() (q1, where the A waves wait for the R waves); the stores leave in q2, where the R waves have slack.
    auto store_x1_bf16 = [&](int k) {
        float* const x1p = step_blk(k).x1;
        if (!(x1p && a.x1_bf16)) return;
#ifdef MSST_LAB
        if (MSST_LAB_X1OLD) return;
#endif
        F3_LANE();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const long tok = tok_of(k, 32 * rh + 16 * jj + c3);
            if (tok < 0) continue;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m0 = (3 * mh + i) * 16 + 4 * g3;
                f32x4 c4;
#pragma unroll
                for (int r = 0; r < 4; ++r) c4[r] = x1r[jj][i][r] - mean_r[jj];
                *reinterpret_cast<s16x4*>(reinterpret_cast<unsigned short*>(x1p) + tok * 96 + m0) = f2bf4(c4);
            }
        }
    };
    // MLP GEMM 1 + GELU: rows 32 rh .., hidden units 32 mh .. + 31 -> HB
    auto mlp1 = [&](int k) {
        F3_LANE();
        const float* lnp = lnp_at(k);
        const Drop drop_k = drop_at(k);
        f32x4 hh[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) { hh[jj][0] = zero4(); hh[jj][1] = zero4(); }
        frag xb[2][3], w1f[2][3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) xb[jj][ks] = P::ld_kc(&sm.xn2[(2 * rh + jj) * 16][ks * 32], LDX);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) w1f[jn][ks] = *reinterpret_cast<const frag*>(sm.wmlp + ((2 * mh + jn) * 3 + ks) * 1024 + l3 * 16);
        }
        MSST_SCHED_FENCE();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) hh[jj][jn] = mma16<HALF>(w1f[jn][ks], xb[jj][ks], hh[jj][jn]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const long tok = tok_of(k, 32 * rh + 16 * jj + c3);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int n0 = (2 * mh + jn) * 16 + 4 * g3;
#pragma unroll
                for (int r = 0; r < 4; ++r) hh[jj][jn][r] = gelu_fast(hh[jj][jn][r] + lnp[576 + n0 + r]);
                if (DROP && tok >= 0) hh[jj][jn] = drop4(drop_k, 3, (unsigned)(tok * 16 + (n0 >> 2)), hh[jj][jn]);
                *reinterpret_cast<s16x4*>(&sm.hb[(2 * rh + jj) * 16 + c3][(2 * mh + jn) * 16 + 4 * g3]) = cv4<HALF>(hh[jj][jn]);   // (P::st_nat's layout)
            }
        }
    };
    // MLP GEMM 2 + bias, dropout, residual -> y
    auto mlp2 = [&](int k) {
        F3_LANE();
        const float* lnp = lnp_at(k);
        const Drop drop_k = drop_at(k);
        float* const yp = step_blk(k).y;
        f32x4 yy[2][3];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int jm = 0; jm < 3; ++jm) yy[jj][jm] = zero4();
        frag hbf[2][2], w2f[3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) hbf[jj][ks] = P::ld_kc(&sm.hb[(2 * rh + jj) * 16][ks * 32], LDH);
#pragma unroll
            for (int jm = 0; jm < 3; ++jm) w2f[jm][ks] = *reinterpret_cast<const frag*>(sm.wmlp + (12 + (3 * mh + jm) * 2 + ks) * 1024 + l3 * 16);
        }
        MSST_SCHED_FENCE();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) yy[jj][jm] = mma16<HALF>(w2f[jm][ks], hbf[jj][ks], yy[jj][jm]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const long tok = tok_of(k, 32 * rh + 16 * jj + c3);
            if (tok >= 0) {
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) {
                    const int m0 = (3 * mh + jm) * 16 + 4 * g3;
                    f32x4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = yy[jj][jm][r] + lnp[480 + m0 + r];
                    if (DROP) o4 = drop4(drop_k, 4, (unsigned)(tok * 24 + (m0 >> 2)), o4);
                    o4 = o4 + x1r[jj][jm];
                    *reinterpret_cast<f32x4*>(yp + tok * 96 + m0) = o4;   // (with sc1 -- write-through, the line dropped from the XCD's L2, which leaves it to the x rows the residual add re-reads: 260.7 vs 255-259 us, no gain)
                }
            }
        }
    };
    // softmax statistics of round rd of walk step k (four heads x 64 rows, left in LDS by the A waves, complete since the barrier that
    // ended the round) -> the block's statistics buffer: 1 KB of consecutive addresses, one float per R thread
    auto copy_lse = [&](int k, int rd) {
        if (!MSST_F3_LSE_STORE) return;
        const int tile_ = tile_at(k);
        float* stats;
        if constexpr (STACK) stats = reinterpret_cast<float*>(const_cast<char*>(sm.blktab[blk_at(k)][BT_X + 4]));
        else stats = a.lse_out;
        if (!stats || tile_ < 0) return;
        int rt_ = (int)threadIdx.x - 256;
        asm volatile("" : "+v"(rt_));
        stats[((long)tile_ * H + 4 * rd) * 64 + rt_] = sm.lseb[rd][rt_];
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[jj][i] = zero4();
    };

    // ---- prologue: LN1 of the first tile ----
    request_ln1(0);
    ln1(0);
    zero_acc();
    request_fw(0, 0);   // (a definition on every path: step 0 has no out-projection in q0)
    __syncthreads();   // (P1)
    if (MSST_F3_RPRIO) __builtin_amdgcn_s_setprio(MSST_F3_RPRIO);   // the R waves are the later-dispatched half: they lose every VALU arbitration to the A wave of their SIMD otherwise
    // walk steps 0 .. nsteps: step k runs q0 / q1 for step k - 1 (k >= 1), q2 for step k (k < nsteps) and step k - 1, q3 for steps k - 1 and k + 1
    bool lnp_switch = false;
    LnpRegs lnp_regs = {};
    for (kc = 0; kc <= nsteps; ++kc) {
        const int k = kc;
        const bool have_prev = k >= 1, have_cur = k < nsteps;
#ifdef MSST_STAMPS
        const bool stamp_on = (a.dbg & 8) && a.stamps && blockIdx.x == 100 && l == 0 && k == nsteps / 2;
#endif
        F3_STAMP(0);
        // ---------------- q0 ----------------
        if (MSST_F3_RPRIO02 != MSST_F3_RPRIO) __builtin_amdgcn_s_setprio(MSST_F3_RPRIO02);
        if constexpr (STACK) {
            // the MLP of step k - 1 (q2 / q3 of this step) belongs to another block than the one before it: its 24 weight fragments
            // replace the old ones now -- the MLP of step k - 2 finished in q3 of the step before -- and are waited for at the end of q0
            if (k >= 2 && epoch_differs(k - 1, k - 2)) {
                const int bk = blk_at(k - 1);
#pragma unroll
                for (int i6 = 0; i6 < 6; ++i6) {
                    const int f = rw * 6 + i6;
                    dma_frag_async(f < 12 ? mlp_w(bk, 0) + f * 1024 : mlp_w(bk, 1) + (f - 12) * 1024, sm.wmlp + f * 1024);
                }
            }
            // the walk moves on to another block with step k + 1: its small vectors go to the other lnp buffer (first read by LN1 of
            // step k + 1 at the end of q3; the buffer's last reader, the MLP of the block before this step's, finished a step ago)
            // -- requested here, stored to LDS at the end of q0: the round trip runs under the out-projection
            lnp_switch = k + 1 < nsteps && epoch_differs(k + 1, k);
            if (lnp_switch && rt < 96) lnp_regs = fetch_lnp(blk_at(k + 1), rt);
        }
        if (have_prev) { request_xr(k - 1); outproj(1); F3_STAMP(1); epilogue1(k - 1); copy_lse(k - 1, 1); }
        F3_STAMP(2);
        // STACK: every memory operation of this wave so far -- the y / x1 rows the steps before stored above all: LN1 of the same tile's
        // NEXT block requests them in q2 of this step at the earliest -- and the MLP weight copy of this interval have completed
        if constexpr (STACK) {
            if (lnp_switch && rt < 96) commit_lnp(slot_at(k + 1), rt, lnp_regs);
            wait_vm0();
        }
        lds_barrier();
        F3_STAMP(3);
        // ---------------- q1 ----------------
        if (MSST_F3_RPRIO02 != MSST_F3_RPRIO) __builtin_amdgcn_s_setprio(MSST_F3_RPRIO);
        if (have_prev) ln2(k - 1);
        if (rt < 64) fill_seq(k + 2, rt);   // (first read in q1 of the next step: LN1 request of step k + 2)
        if (have_cur) zero_acc();
        request_fw(0, have_cur ? k : k - 1);
        F3_STAMP(4);
        lds_barrier();
        F3_STAMP(5);
        // ---------------- q2 ----------------
        if (MSST_F3_RPRIO02 != MSST_F3_RPRIO) __builtin_amdgcn_s_setprio(MSST_F3_RPRIO02);
        if (MSST_F3_LN1Q == 2 && k + 1 < nsteps) request_ln1(k + 1);
        if (have_cur) outproj(0);
        F3_STAMP(6);
        if (have_prev) { store_x1_bf16(k - 1); mlp1(k - 1); }
        if (have_cur) copy_lse(k, 0);
        if (MSST_F3_LN1Q == 2 && k + 1 < nsteps) ln1(k + 1);
        if (MSST_F3_LN1Q == 3 && k + 1 < nsteps) request_ln1(k + 1);   // consumed at the end of q3: the barrier wait and MLP GEMM 2 cover the HBM round trip
        F3_STAMP(7);
        lds_barrier();
        F3_STAMP(8);
        // ---------------- q3 ----------------
        if (MSST_F3_RPRIO02 != MSST_F3_RPRIO) __builtin_amdgcn_s_setprio(MSST_F3_RPRIO);
        if (have_prev) mlp2(k - 1);
        F3_STAMP(9);
        if (MSST_F3_LN1Q == 3 && k + 1 < nsteps) ln1(k + 1);
        request_fw(1, have_cur ? k : k - 1);
        F3_STAMP(10);
        lds_barrier();
        F3_STAMP(11);
        if constexpr (STACK) words_advance();
    }
}

static int fwd3_attrs() {
    static std::atomic<bool> attr_set{false};
    if (attr_set) return 0;
    const void* ks[8] = {reinterpret_cast<const void*>(&block_fwd_rs_kernel<false, false, false>), reinterpret_cast<const void*>(&block_fwd_rs_kernel<true, false, false>),
                         reinterpret_cast<const void*>(&block_fwd_rs_kernel<false, true, false>), reinterpret_cast<const void*>(&block_fwd_rs_kernel<true, true, false>),
                         reinterpret_cast<const void*>(&block_fwd_rs_kernel<false, false, true>), reinterpret_cast<const void*>(&block_fwd_rs_kernel<true, false, true>),
                         reinterpret_cast<const void*>(&block_fwd_rs_kernel<false, true, true>), reinterpret_cast<const void*>(&block_fwd_rs_kernel<true, true, true>)};
    for (int i = 0; i < 8; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Fwd3Smem));
        if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
    return 0;
}

// max tiles per workgroup x blocks a stack launch can walk (its step table lives in LDS)
int block_fwd_stack_max_steps() { return F3_MAX_STEPS; }

int launch_block_fwd_rs_stack(const StackArgs& sa, int grid, hipStream_t st) {
    const BlockArgs& a = sa.base;
    if (a.H != 8 || sa.nblk < 1 || sa.nblk > MSST_MAX_STACK || grid < 1) return MSST_ERR_UNSUPPORTED;
    // steps of the busiest workgroup: its tiles in groups of >= 3 (a lone group of 1 or 2 is padded to 3)
    const int nmine = (a.ntiles + grid - 1) / grid;
    if (sa.nblk * (nmine < 3 ? 3 : nmine) > F3_MAX_STEPS) return MSST_ERR_UNSUPPORTED;
    int rc = fwd3_attrs();
    if (rc) return rc;
    ProfScope ps(K_BLOCK_FWD, st);
    const size_t smem = sizeof(Fwd3Smem);
    if (a.half) {
        if (a.drop.thr) hipLaunchKernelGGL((block_fwd_rs_kernel<true, true, true>), dim3(grid), dim3(512), smem, st, sa);
        else hipLaunchKernelGGL((block_fwd_rs_kernel<false, true, true>), dim3(grid), dim3(512), smem, st, sa);
    } else {
        if (a.drop.thr) hipLaunchKernelGGL((block_fwd_rs_kernel<true, true, false>), dim3(grid), dim3(512), smem, st, sa);
        else hipLaunchKernelGGL((block_fwd_rs_kernel<false, true, false>), dim3(grid), dim3(512), smem, st, sa);
    }
    return (int)hipGetLastError();
}

int launch_block_fwd_rs(const BlockArgs& a, int grid, hipStream_t st) {
    const size_t smem = sizeof(Fwd3Smem);
    if (a.H != 8) return MSST_ERR_UNSUPPORTED;
    int rc = fwd3_attrs();
    if (rc) return rc;
    ProfScope ps(K_BLOCK_FWD, st);
    if (a.half) {
        if (a.drop.thr) hipLaunchKernelGGL((block_fwd_rs_kernel<true, false, true>), dim3(grid), dim3(512), smem, st, a);
        else hipLaunchKernelGGL((block_fwd_rs_kernel<false, false, true>), dim3(grid), dim3(512), smem, st, a);
    } else {
        if (a.drop.thr) hipLaunchKernelGGL((block_fwd_rs_kernel<true, false, false>), dim3(grid), dim3(512), smem, st, a);
        else hipLaunchKernelGGL((block_fwd_rs_kernel<false, false, false>), dim3(grid), dim3(512), smem, st, a);
    }
    return (int)hipGetLastError();
}

}  // namespace msst
