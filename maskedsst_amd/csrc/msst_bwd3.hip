// bf16 attention half of a block, backward (reference vit_spatial_spectral.py:47-78 under PreNorm :22-29; a15 of
// SURVEY.md section 8) -- round-3 kernel.  Same math, same HBM interface (saved bf16 LN1 rows + pre-dropped bf16 da rows
// in, one bf16 d(LN1 out) partial per head and one weight-gradient slab per workgroup out) and the same dropout streams
// as the round-2 kernel (msst_bwd2.hip, retired in round 4); what changes is how the work of a 64-row tile is cut:
//
//   * ONE GEMM = ONE WAVE.  The round-2 kernel split every GEMM four ways (16 x 64 strips), so each wave re-read the
//     whole other operand: 1.4 LDS instructions per MFMA, LDS pipe 71 % busy, MFMA pipes 31 %.  Here the four waves of a
//     workgroup are the roles Q, K, V, O: wave Q owns dWq [64][96] (96 accumulator registers), computes q = LN1(x) Wq^T,
//     dq = dS k, dWq += dq^T LN1(x) and its third of the d(LN1 out) GEMM; K and V likewise; O owns dWout_h^T and computes
//     dO = da Wout_h, o = P v, dWout_h^T += o^T da.  Every operand of a 64 x 64 x 64 (or 64 x 96 x 64) GEMM is then read
//     exactly once, with v_mfma_f32_32x32x16_bf16 (half the operand bytes per FLOP of the 16 x 16 form at the
//     register-blocking these tiles allow).  Only the softmax phase keeps wave <-> 16 query rows and 16 x 16 x 32 MFMAs.
//   * C tiles feed the next GEMM from registers where the layouts allow it: dq / dk / dv / o (C[row][channel], the lane
//     owns a channel) are packed to bf16 and ARE the A operand of the weight-gradient GEMM (contraction over rows, both
//     operands in the same permuted row order); only the d(LN1 out) GEMM needs them through LDS (transposed reads).
//   * LDS tiles are unpadded and XOR-swizzled at 16-byte granularity: conflict-free for the 32-row b128 fragment reads,
//     the 16-row b128 reads of the softmax phase AND ds_read_b64_tr_b16 (tools/peak_microbench.hip measures every
//     pattern: 4.4 / 2.9 cycles per wave-instruction against 8.0 / 4.0 for the padded round-2 tiles).  Every address
//     is lane constant ^ immediate + immediate.
//   * five barriers per tile (eight before).
//
// grid (nchunk, H), 256 threads, 72 KB LDS: two workgroups per CU.
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"
#include <type_traits>

#ifndef MSST_B3_D3A
#define MSST_B3_D3A 2   // software-pipeline depths (steps a fragment is requested ahead of its MFMAs): phase 3 contraction,
#endif
#ifndef MSST_B3_D3B
#define MSST_B3_D3B 5   // weight-gradient GEMM,
#endif
#ifndef MSST_B3_D4
#define MSST_B3_D4 3    // phase 4
#endif
#ifndef MSST_B3_W4
#define MSST_B3_W4 6    // phase-4 weight fragments requested before the weight-gradient GEMM (the other 12 - n during phase 4)
#endif
#ifndef MSST_B3_W1AT
#define MSST_B3_W1AT 6   // phase-4 step behind which the next tile's phase-1 weights are requested (>= 6: behind the last phase-4 weight request)
#endif
#ifndef MSST_B3_PRIO
#define MSST_B3_PRIO 1   // s_setprio level of the MFMA-dense phases (1, 3, 4); the softmax phase and the copy-out run at 0
#endif
#if MSST_B3_PRIO
#define B3_PRIO(n) __builtin_amdgcn_s_setprio((n) ? MSST_B3_PRIO : 0)
#else
#define B3_PRIO(n) do { } while (0)
#endif

namespace msst {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) s16x8 lds_s16x8;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// LDS map (bytes).  64-wide bf16 tiles: 128-byte rows; 96-wide tiles: 192-byte rows.
constexpr int R3_XN = 0, R3_DA = 12288;
constexpr int R3_Q = 24576, R3_K = 32768, R3_DO = 40960, R3_V = 49152, R3_P = 57344, R3_DS = 65536, R3_SMEM = 73728;
constexpr int R3_OUT = R3_P;   // [64][96] staging of the d(LN1 out) rows in P | dS (dead after phase 3; next written in phase 2, behind barrier B1)

// 16-byte slot s of row r lives at slot s ^ fz(r) (128-byte rows) / (s & ~3) | ((s & 3) ^ fz2(r)) (192-byte rows)
__device__ __forceinline__ int fz(int r) { return (((r >> 1) & 1) << 2) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ int fz2(int r) { return (((r >> 3) & 1) << 1) | ((r >> 2) & 1); }

__device__ __forceinline__ s16x8 lds_r128(lds_char* b, unsigned off) { return *reinterpret_cast<const lds_s16x8*>(b + off); }
__device__ __forceinline__ void lds_w64(lds_char* b, unsigned off, s16x4 v) { *reinterpret_cast<lds_s16x4*>(b + off) = v; }
__device__ __forceinline__ s16x8 lds_tr2(lds_char* b, unsigned off0, unsigned off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(b + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(b + off1));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__device__ __forceinline__ f32x16 mma32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, a), __builtin_bit_cast(hbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ s16x4 pk4(const f32x16& c, int q4) {
    f32x4 t = {c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]};
    return f2bf4(t);
}
// registers 8 k0 .. 8 k0 + 7 of a C tile as one bf16 operand fragment (contraction index = tile row, in the order the
// C layout hands it over: rows 16 k0 + 8 (e / 4) + 4 (lane / 32) + e % 4 for element e)
__device__ __forceinline__ s16x8 pk8(const f32x16& c, int k0) {
    const s16x4 a = pk4(c, 2 * k0), b = pk4(c, 2 * k0 + 1);
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// 1 KB fragment f of a fragment-packed (32 rows x 16 k per fragment) bf16 matrix: descriptor and fragment offset are
// wave uniform, the lane offset is shared by all weight loads
__device__ __forceinline__ s16x8 ld_w32(const void* w, int f, int lane16) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, f * 1024, 0));
}
__device__ __forceinline__ void bar3() {
    lds_barrier();
}
__device__ __forceinline__ int launder3(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

}  // namespace

template <bool DROP>
__global__ __launch_bounds__(256, 2) void block_bwd_attn_r3_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char* const sm = (lds_char*)smem_raw;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, inner = H * 64, h = blockIdx.y;
    const TileMap tm = a.tm;
    const int L = tm.L;
    bf16_t* part = reinterpret_cast<bf16_t*>(a.dxn_part) + (long)h * a.ntok * 96;

    // wave roles: 0 Q, 1 K, 2 V, 3 O
    const bool roleO = wave == 3;
    const void* w1p = roleO ? a.w.woutT32 : a.w.wqkv32;                       // phase-1 weights: [rows][96], rows row1_0 ..
    const int f1_0 = ((roleO ? h * 64 : (wave * H + h) * 64) >> 5) * 6;        // fragment (dt, ks) = f1_0 + 6 dt + ks
    const int f4_0 = (wave < 3 ? wave : 0) * ((3 * inner) >> 4) + ((h * 64) >> 4);   // phase 4 (m tile = wave): + (which * inner >> 4) + ks
    const int p1_in = roleO ? R3_DA : R3_XN;
    const int p1_out = wave == 0 ? R3_Q : wave == 1 ? R3_K : wave == 2 ? R3_V : R3_DO;
    const bool pathX = wave == 0 || wave == 3;
    // phase 3: C[row][d] = sum arr1 . arr2 (see below), weight-gradient partner arrX
    const int p3_a1 = (wave == 0 || wave == 1) ? R3_DS : R3_P;
    const int p3_a2 = wave == 0 ? R3_K : wave == 1 ? R3_Q : wave == 2 ? R3_DO : R3_V;
    const int p3_x = roleO ? R3_DA : R3_XN;

    f32x16 G[2][3];   // persistent weight-gradient accumulators: G[d tile][m tile], C[i = head channel][j = model feature]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i][j] = zero16();

    // (sequence slot << 16 | position) of the 64 rows of a tile is tile invariant: a 64-entry table in LDS instead of registers
    // that would live (= be spilled) across the whole tile loop.  slot 0xffff marks a padding row; its position is still
    // r - (r / L) L, so that "first key of my sequence" = row - position holds for every row.
    unsigned* const rowmap = reinterpret_cast<unsigned*>(smem_raw + R3_SMEM);
    if (tid < 64) {
        const int sq = tid / L, ps = tid - sq * L;
        rowmap[tid] = ((unsigned)((sq >= tm.TS ? 0xffff : sq) & 0xffff) << 16) | (unsigned)ps;
    }
    __syncthreads();
    // key tiles (16 keys each) that the 16 queries of this wave can see: those overlapping [first key of the first query's
    // sequence, last key of the last query's sequence]
    unsigned need;
    {
        const int r0 = 16 * wave, r1 = 16 * wave + 15;
        const int klo = r0 - (int)(rowmap[r0] & 0xffffu), khi = min(63, r1 - (int)(rowmap[r1] & 0xffffu) + L - 1);
        unsigned m = 0;
        for (int t = 0; t < 4; ++t) m |= (unsigned)(16 * t <= khi && 16 * t + 15 >= klo) << t;
        need = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }
    // token of tile row `sp` (a rowmap entry) in tile tile_, -1 for padding
    auto tok_sp = [&](int tile_, unsigned sp) -> int {
        const int sx = (int)(sp >> 16), sy = (int)(sp & 0xffffu);
        const int q = tile_ * tm.TS + sx;
        if (tile_ >= a.ntiles || sx == 0xffff || q >= tm.nseq) return -1;
        if (tm.mode == 0) return q * tm.N + sy;
        const int b = tm.nshift >= 0 ? (q >> tm.nshift) : q / tm.N;
        return b * tm.T + sy * tm.N + (q - b * tm.N);
    };
    // row-wise LDS address of this thread's j-th 16-byte slot (row tid / 4, logical slot 3 (tid % 4) + j) in a 96-wide tile
    auto row_slot = [&](int j) -> unsigned {
        const int t_ = launder3(tid);
        const int row = t_ >> 2, s = 3 * (t_ & 3) + j;
        return (unsigned)(row * 192 + (((s & ~3) | ((s & 3) ^ fz2(row))) << 4));
    };
    // The LN1(x) / da rows of the NEXT tile are staged through registers: thread <-> (row tid / 4, 48 bytes), three 16-byte loads
    // of each array.  They are requested at the start of the weight-gradient GEMM (an HBM round trip under load is 2-3 k cycles)
    // and stored into XN / DA -- dead from barrier B3 on -- before barrier B4.  Padding rows: clamped address, zeros stored.
    auto load_rows = [&](int tile_, u32x4 (&xr)[3], u32x4 (&dr)[3]) -> int {
        const int t_ = launder3(tid);
        const int tok = tok_sp(tile_, rowmap[t_ >> 2]);
        const long off = (long)(tok >= 0 ? tok : 0) * 96 + (t_ & 3) * 24;
        const u32x4* sx = reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(a.xn) + off);
        const u32x4* sd = reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(a.dab) + off);
#pragma unroll
        for (int j = 0; j < 3; ++j) { xr[j] = sx[j]; dr[j] = sd[j]; }
        return tok;
    };
    auto store_rows = [&](int tok, const u32x4 (&xr)[3], const u32x4 (&dr)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned o = row_slot(j);
            *reinterpret_cast<lds_u32x4*>(sm + R3_XN + o) = tok >= 0 ? xr[j] : u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<lds_u32x4*>(sm + R3_DA + o) = tok >= 0 ? dr[j] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    {
        u32x4 xr[3], dr[3];
        const int tok0 = load_rows(blockIdx.x, xr, dr);
        store_rows(tok0, xr, dr);
    }
    s16x8 w1[2][4];   // phase-1 weight fragments [d tile][k step % 4]: a ring of four k-steps (tile invariant, re-requested from L2 every tile)
#define W1_(dt, s4) w1[(dt)][(s4)]
    auto load_w1 = [&]() {
        const int l16 = (launder3(tid) & 63) * 16;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) W1_(dt, ks) = ld_w32(w1p, f1_0 + 6 * dt + ks, l16);
    };
    load_w1();
    lds_barrier();   // the first tile's rows are in XN / DA

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#if defined(MSST_STAMPS)
        const bool dump_on = (a.dbg & 0x808) == 0x808 && a.stamps && blockIdx.x == 0 && (int)blockIdx.y == ((a.dbg >> 8) & 7) && tile == (int)blockIdx.x;
        // cycle stamps of lane 0 of wave (dbg >> 8) & 3 of workgroup (7, 3), a mid-walk tile
        const bool stamp_on = (a.dbg & 0x808) == 8 && a.stamps && blockIdx.x == 7 && blockIdx.y == 3 && tid == 64 * ((a.dbg >> 8) & 3) &&
                              tile == (int)blockIdx.x + 20 * (int)gridDim.x;
#define R3_DUMP(stage) do { if (dump_on) { \
            lds_barrier(); \
            for (int i_ = tid; i_ < R3_SMEM / 4; i_ += 256) reinterpret_cast<unsigned*>(a.stamps)[(stage) * (R3_SMEM / 4) + i_] = reinterpret_cast<const unsigned*>(smem_raw)[i_]; \
            lds_barrier(); } } while (0)
#else
#define R3_DUMP(stage) do { } while (0)
#endif
        STAMP(0);
        B3_PRIO(1);
        R3_DUMP(0);
        // ---------------- phase 1: q | k | v | dO = rows . W^T  (C[i = channel][j = row], stored [row][channel]) ----------------
        {
            const int t_ = launder3(tid);
            const int l31 = t_ & 31, hi = (t_ >> 5) & 1;
            const int f2v = fz2(l31);
            unsigned bin[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) bin[p] = p1_in + l31 * 192 + (((hi ^ f2v) << 4) ^ (p << 5));
            f32x16 c[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) c[i][j] = zero16();
            s16x8 fb[3][2];   // LN1(x) / da fragments [slot][row tile], two k-steps ahead of their MFMAs
            swpipe<6, 2>(
                [&](int ks) {
                    fb[ks % 3][0] = lds_r128(sm, bin[ks & 1] + 64 * (ks >> 1));
                    fb[ks % 3][1] = lds_r128(sm, bin[ks & 1] + 64 * (ks >> 1) + 32 * 192);
                },
                [&](int ks) {
                    c[0][0] = mma32(W1_(0, ks & 3), fb[ks % 3][0], c[0][0]);
                    c[0][1] = mma32(W1_(0, ks & 3), fb[ks % 3][1], c[0][1]);
                    c[1][0] = mma32(W1_(1, ks & 3), fb[ks % 3][0], c[1][0]);
                    c[1][1] = mma32(W1_(1, ks & 3), fb[ks % 3][1], c[1][1]);
                    if (ks < 2) {
                        W1_(0, ks) = ld_w32(w1p, f1_0 + ks + 4, (t_ & 63) * 16);
                        W1_(1, ks) = ld_w32(w1p, f1_0 + 6 + ks + 4, (t_ & 63) * 16);
                    }
                });
            const unsigned L7 = p1_out + l31 * 128 + (fz(l31) << 4) + 8 * hi;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        lds_w64(sm, (L7 ^ ((4 * dt + q4) << 4)) + rt * 4096, pk4(c[dt][rt], q4));
        }
        STAMP(2);
        bar3();   // B1
        STAMP(3);
        B3_PRIO(0);
        R3_DUMP(1);
        // ---------------- phase 2: wave <-> 16 query rows; S^T, softmax, P, dP^T, dS (16 x 16 x 32 MFMAs) ----------------
        {
            typedef PBF16 P;
            const int t_ = launder3(tid);
            const int l = t_ & 63, g = l >> 4, c16 = l & 15;
            const int fzc = fz(c16);
            const int qlo = 16 * wave + c16 - (int)(rowmap[16 * wave + c16] & 0xffffu), qhi = qlo + L;   // keys of this query's sequence
            unsigned ak[2], aq[2];   // A operand rows 16 t + c16 (k, v), B operand rows 16 wave + c16 (q, dO); k-step ks2
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                ak[ks2] = c16 * 128 + (((4 * ks2 + g) ^ fzc) << 4);
                aq[ks2] = ak[ks2] + wave * 2048;
            }
            const unsigned L8 = (16 * wave + c16) * 128 + (((g >> 1) ^ fzc) << 4) + 8 * (g & 1);   // ^ (t << 5)
            // Short sequences (spectral blocks): a wave's 16 queries only meet the key tiles that overlap their own sequences -- bit t
            // of `need` (wave uniform, tile invariant).  Every other 16 x 16 score tile is masked anyway and is skipped altogether
            // (operand reads, MFMAs, exps, dropout hashes); its P / dS entries are stored as zeros.
            f32x4 pr[4], dp[4];
            f32x4 dm[4];   // dropout multipliers of site 1 (0 or 1 / (1 - p)): P and dP see the same ones
#pragma unroll
            for (int t = 0; t < 4; ++t) dm[t] = zero4();
            const float cs = a.scale * 1.44269504088896340736f;   // exp(scale (s - max)) = exp2(s c - max c), c = scale log2 e
            // NM: -1 = one 64-token sequence, nothing masked; > 0 = short sequences, the key tiles of this wave known at compile
            // time (straight-line code: the run-time form below breaks the phase into thirty basic blocks and costs 1.5 k cycles);
            // 0 = short sequences, key tiles from `need` at run time (patterns without an instance)
            auto softmax_phase = [&](auto mode_tag) {
                constexpr int NM = decltype(mode_tag)::value;
                constexpr bool MASKED = NM >= 0;
                const unsigned nm = NM > 0 ? (unsigned)NM : NM == 0 ? need : 0xfu;
                auto on = [&](int t) { return NM < 0 || ((nm >> t) & 1u); };
                {
                    s16x8 fq[2], fk[2][4];
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) {
                        fq[ks2] = lds_r128(sm, R3_Q + aq[ks2]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (NM == 0 || on(t)) fk[ks2][t] = lds_r128(sm, R3_K + ak[ks2] + t * 2048);   // (run-time form: all of them, a definition on every path)
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) pr[t] = zero4();
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (on(t)) pr[t] = P::mma(fk[ks2][t], fq[ks2], pr[t]);   // C[i = key][j = query]
                }
                // dP^T = v dO^T is independent of the softmax: its operands are requested and its MFMAs run under the softmax's VALU work
                {
                    s16x8 fdo[2], fv[2][4];
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) {
                        fdo[ks2] = lds_r128(sm, R3_DO + aq[ks2]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (NM == 0 || on(t)) fv[ks2][t] = lds_r128(sm, R3_V + ak[ks2] + t * 2048);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) dp[t] = zero4();
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (on(t)) dp[t] = P::mma(fv[ks2][t], fdo[ks2], dp[t]);   // C[i = key][j = query]
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (MASKED) {
                            const int key = t * 16 + 4 * g + r;
                            pr[t][r] = (key >= qlo && key < qhi) ? pr[t][r] : -INFINITY;
                        }
                        mx = fmaxf(mx, pr[t][r]);
                    }
                }
                mx = colgroup_max(mx);
                const float mc = mx * cs;
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(pr[t][r], cs, -mc)); pr[t][r] = e; sum += e; }
                }
                sum = colgroup_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) { lds_w64(sm, R3_P + (L8 ^ (t << 5)), s16x4{0, 0, 0, 0}); continue; }
                    pr[t] = pr[t] * inv;
                    f32x4 pd = pr[t];   // site 1: O and dV see the dropped probabilities, the softmax backward the raw ones
                    if (DROP) {
                        unsigned ha, hb;
                        drop_bits(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c16) * 16 + t * 4 + g), ha, hb);
                        const unsigned t16 = a.drop.thr << 16;
                        dm[t][0] = (ha << 16) >= t16 ? a.drop.scale : 0.f;
                        dm[t][1] = ha >= t16 ? a.drop.scale : 0.f;
                        dm[t][2] = (hb << 16) >= t16 ? a.drop.scale : 0.f;
                        dm[t][3] = hb >= t16 ? a.drop.scale : 0.f;
                        pd = pd * dm[t];
                    }
                    lds_w64(sm, R3_P + (L8 ^ (t << 5)), f2bf4(pd));   // P[query][key]
                }
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) continue;
                    if (DROP) dp[t] = dp[t] * dm[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) delta += pr[t][r] * dp[t][r];
                }
                delta = colgroup_sum(delta);
                // dS WITHOUT the softmax scale (dim_head^-0.5 = 2^-3, exact in bf16): it is folded into the q / k blocks of the
                // phase-4 weights (msst_prep_weights, pack = 2) and into the dWq / dWk slabs at the end of the kernel
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) { lds_w64(sm, R3_DS + (L8 ^ (t << 5)), s16x4{0, 0, 0, 0}); continue; }
                    f32x4 d4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d4[r] = pr[t][r] * (dp[t][r] - delta);
                    lds_w64(sm, R3_DS + (L8 ^ (t << 5)), f2bf4(d4));   // dS[query][key] / scale
                }
            };
            {
                if (L == 64) softmax_phase(std::integral_constant<int, -1>{});
                else switch (need) {   // (wave uniform, tile invariant)
                    case 0x3: softmax_phase(std::integral_constant<int, 0x3>{}); break;
                    case 0x7: softmax_phase(std::integral_constant<int, 0x7>{}); break;
                    case 0xe: softmax_phase(std::integral_constant<int, 0xe>{}); break;
                    case 0xc: softmax_phase(std::integral_constant<int, 0xc>{}); break;
                    case 0x6: softmax_phase(std::integral_constant<int, 0x6>{}); break;
                    case 0xf: softmax_phase(std::integral_constant<int, 0xf>{}); break;
                    default: softmax_phase(std::integral_constant<int, 0>{}); break;
                }
            }
        }
        STAMP(4);
        // ---------------- phase 3: the four contractions over rows, one per wave ----------------
        //   Q: dq[query][d] = sum_key dS[query][key] k[key][d]      O: o[query][d] = sum_key P[query][key] v[key][d]     (path X)
        //   K: dk[key][d]   = sum_query dS[query][key] q[query][d]  V: dv[key][d]  = sum_query P[query][key] dO[query][d] (path Y)
        // C[i = row][j = d]; then G[d][m] += sum_row C[row][d] . {LN1(x) | da}[row][m] with the packed C tiles as A operand.
        // The second operand (k | q | dO | v) has been complete since barrier B1: its first fragments are requested BEFORE
        // barrier B2, so the phase starts with its MFMAs instead of an LDS round trip.
        {
            s16x8 w4[6];            // phase-4 weight fragments of this wave's m tile: a ring of six, refilled as phase 4 consumes them
            u32x4 xnq[3], daq[3];   // rows of the next tile
            int tokn;
            const int t_ = launder3(tid);
            const int l = t_ & 63, l31 = l & 31, hi = l >> 5, i = l & 15, u = (l >> 4) & 1, b = (i >> 1) & 1, r1 = (i >> 3) & 1;
            // transposed 32-column fragment of a 64-wide tile, natural contraction order: k row = 16 kk + 8 hi + 4 a + i / 4
            const unsigned Lt = (8 * hi + (i >> 2)) * 128 + (((2 * u + b) ^ ((r1 << 2) | (hi << 1) | hi)) << 4) + 8 * (i & 1);
            unsigned tr[2][2];   // [column tile][a]; + 2048 kk + array base
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) tr[ct][aa] = (Lt ^ ((ct << 6) | (aa << 5))) + 512 * aa;
            const unsigned a1 = p3_a1 + l31 * 128 + ((hi ^ fz(l31)) << 4);   // path X first operand: ^ (kk << 5), + 4096 row tile
            f32x16 c[2][2];   // [row tile][d tile]
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j) c[ii][j] = zero16();
            constexpr int D3 = MSST_B3_D3A;
            s16x8 fa[D3 + 1][2], fb[D3 + 1][2];   // [slot][tile], D3 k-steps ahead
            auto issue_b = [&](int kk) {
                fb[kk % (D3 + 1)][0] = lds_tr2(sm, p3_a2 + tr[0][0] + 2048 * kk, p3_a2 + tr[0][1] + 2048 * kk);
                fb[kk % (D3 + 1)][1] = lds_tr2(sm, p3_a2 + tr[1][0] + 2048 * kk, p3_a2 + tr[1][1] + 2048 * kk);
            };
            auto issue_a = [&](int kk) {
                if (pathX) {
                    fa[kk % (D3 + 1)][0] = lds_r128(sm, a1 ^ (kk << 5));
                    fa[kk % (D3 + 1)][1] = lds_r128(sm, (a1 ^ (kk << 5)) + 4096);
                } else {
                    fa[kk % (D3 + 1)][0] = lds_tr2(sm, p3_a1 + tr[0][0] + 2048 * kk, p3_a1 + tr[0][1] + 2048 * kk);
                    fa[kk % (D3 + 1)][1] = lds_tr2(sm, p3_a1 + tr[1][0] + 2048 * kk, p3_a1 + tr[1][1] + 2048 * kk);
                }
            };
#pragma unroll
            for (int kk = 0; kk < D3; ++kk) issue_b(kk);
            bar3();   // B2
            STAMP(5);
            B3_PRIO(1);
            R3_DUMP(2);
#pragma unroll
            for (int kk = 0; kk < D3; ++kk) issue_a(kk);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + D3 < 4) { issue_a(kk + D3); issue_b(kk + D3); }
                MSST_SCHED_FENCE();
                c[0][0] = mma32(fa[kk % (D3 + 1)][0], fb[kk % (D3 + 1)][0], c[0][0]);
                c[0][1] = mma32(fa[kk % (D3 + 1)][0], fb[kk % (D3 + 1)][1], c[0][1]);
                c[1][0] = mma32(fa[kk % (D3 + 1)][1], fb[kk % (D3 + 1)][0], c[1][0]);
                c[1][1] = mma32(fa[kk % (D3 + 1)][1], fb[kk % (D3 + 1)][1], c[1][1]);
                MSST_SCHED_FENCE();
            }
            s16x8 pa[2][4];   // [d tile][k step]: A operand of the weight-gradient GEMM
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pa[dt][kk] = pk8(c[kk >> 1][dt], kk & 1);
            // transposed 32-column fragment of a 96-wide tile in the C-layout row order: k row = 16 kk + 8 a + 4 hi + i / 4
            const unsigned L4 = (4 * hi + (i >> 2)) * 192 + (((2 * u + b) ^ hi) << 4) + 8 * (i & 1);
            unsigned tx[2];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) tx[aa] = p3_x + (L4 ^ (aa << 5)) + 8 * aa * 192;
            auto wgrad = [&]() {
                s16x8 fx[MSST_B3_D3B + 1];   // step s = (kk, mt): transposed row fragment MSST_B3_D3B steps ahead
                swpipe<12, MSST_B3_D3B>(
                    [&](int st) {
                        const int kk = st / 3, mt = st % 3;
                        fx[st % (MSST_B3_D3B + 1)] = lds_tr2(sm, tx[0] + 3072 * kk + 64 * mt, tx[1] + 3072 * kk + 64 * mt);
                    },
                    [&](int st) {
                        const int kk = st / 3, mt = st % 3;
                        G[0][mt] = mma32(pa[0][kk], fx[st % (MSST_B3_D3B + 1)], G[0][mt]);
                        G[1][mt] = mma32(pa[1][kk], fx[st % (MSST_B3_D3B + 1)], G[1][mt]);
                    });
            };
            // dq | dk | dv also go to LDS, transposed ([d][row]), over the tile only this wave read above (k | q | dO)
            if (!roleO) {
                const unsigned L7 = p3_a2 + l31 * 128 + (fz(l31) << 4) + 8 * hi;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
                            lds_w64(sm, (L7 ^ ((4 * ct + q4) << 4)) + dt * 4096, pk4(c[ct][dt], q4));
            }
            // requests of the weight-gradient GEMM's shadow: the first phase-4 weight fragments (waves Q, K, V), the next tile's rows
            const int l16 = l * 16;
            // (wave O, which has no phase 4, requests one hot fragment six times: a definition on every path keeps the register
            // allocator from shuffling the in-flight fragments of the other waves at the join)
#pragma unroll
            for (int k12 = 0; k12 < 6; ++k12)
                w4[k12] = ld_w32(a.w.wqkvT32, roleO ? 0 : f4_0 + (k12 >> 2) * (inner >> 4) + (k12 & 3), l16);
            tokn = load_rows(tile + gridDim.x, xnq, daq);
            wgrad();
            STAMP(6);
            bar3();   // B3
            STAMP(7);
            R3_DUMP(3);
            // ---------------- phase 4: d(LN1 out)[row][m] = dq Wq + dk Wk + dv Wv, wave <-> 32 features (waves Q, K, V) ----------------
            if (roleO) {
                load_w1();   // the next tile's phase-1 weights
            } else {
                f32x16 c4[2];   // [row tile]: C[i = m][j = row]
                c4[0] = zero16(); c4[1] = zero16();
                s16x8 fb4[MSST_B3_D4 + 1][2];   // step k12 = (which, ks): dq^T | dk^T | dv^T fragments MSST_B3_D4 steps ahead
                swpipe<12, MSST_B3_D4>(
                    [&](int k12) {
                        const int which = k12 >> 2, ks = k12 & 3;
                        const int reg = which == 0 ? R3_K : which == 1 ? R3_Q : R3_DO;
                        fb4[k12 % (MSST_B3_D4 + 1)][0] = lds_tr2(sm, reg + tr[0][0] + 2048 * ks, reg + tr[0][1] + 2048 * ks);
                        fb4[k12 % (MSST_B3_D4 + 1)][1] = lds_tr2(sm, reg + tr[1][0] + 2048 * ks, reg + tr[1][1] + 2048 * ks);
                    },
                    [&](int k12) {
                        c4[0] = mma32(w4[k12 % 6], fb4[k12 % (MSST_B3_D4 + 1)][0], c4[0]);
                        c4[1] = mma32(w4[k12 % 6], fb4[k12 % (MSST_B3_D4 + 1)][1], c4[1]);
                        if (k12 < 6)
                            w4[k12] = ld_w32(a.w.wqkvT32, f4_0 + ((k12 + 6) >> 2) * (inner >> 4) + ((k12 + 6) & 3), l16);
                        if (k12 == MSST_B3_W1AT) load_w1();   // the next tile's phase-1 weights, behind this phase's last weight request
                    });
                const unsigned L9 = R3_OUT + l31 * 192 + (fz2(l31) << 4) + 8 * hi + 64 * wave;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) lds_w64(sm, (L9 ^ (q4 << 4)) + rt * 6144, pk4(c4[rt], q4));
            }
            store_rows(tokn, xnq, daq);
        }
        STAMP(8);
        bar3();   // B4
        STAMP(9);
        B3_PRIO(0);
        R3_DUMP(4);
        // copy-out: whole rows of the staged result to this head's partial (buffer stores: a padding row gets an offset outside
        // the descriptor and is dropped).  No barrier follows: phase 1 of the next tile reads XN / DA (published by barrier B4) and
        // writes q | k | v | dO, none of which anybody reads any more; P | dS, where the rows are staged, are next written in phase 2
        {
            const int t_ = launder3(tid);
            const int tok_out = tok_sp(tile, rowmap[t_ >> 2]);
            u32x4 v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const lds_u32x4*>(sm + R3_OUT + row_slot(j));
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)(a.ntok * 192), 0x00020000);
            const unsigned voff = tok_out < 0 ? 0x80000000u : (unsigned)tok_out * 192u + (t_ & 3) * 48;   // (+ 32 must not wrap)
#pragma unroll
            for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rp, voff + 16 * j, 0, 0);
        }
        STAMP(10);
    }

    // ---------------- slab: [dWq | dWk | dWv] [3][64][96], dWout_h [96][64] ----------------
    {
        float* slab = a.slab + ((long)blockIdx.x * H + h) * MSST_ATTN_SLAB_N;
        const int l = tid & 63, l31 = l & 31, hi = l >> 5;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * hi, m = 32 * mt + l31;
                    if (roleO) slab[3 * 6144 + m * 64 + d] = G[dt][mt][r];
                    else slab[wave * 6144 + d * 96 + m] = wave < 2 ? G[dt][mt][r] * a.scale : G[dt][mt][r];   // dq, dk were kept / scale
                }
    }
}

int launch_block_bwd_attn_r3(const AttnBwdArgs& a, int nchunk, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (a.tm.L > 64 || a.tm.L < 1) return MSST_ERR_UNSUPPORTED;
    if (!a.xn || !a.dab || !a.w.wqkv32 || !a.w.woutT32 || !a.w.wqkvT32 || nchunk < 1 || nchunk > a.ntiles) return MSST_ERR_BADARG;
    if (a.ntok * 192 >= 0x7ffffff0L) return MSST_ERR_UNSUPPORTED;   // 32-bit row offsets of the copy-out descriptor
    typedef void (*kern_t)(AttnBwdArgs);
    const kern_t kerns[2] = {&block_bwd_attn_r3_kernel<false>, &block_bwd_attn_r3_kernel<true>};
    if (!attr_set) {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, R3_SMEM + 256);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    ProfScope ps(K_BWD_ATTN, st);
    hipLaunchKernelGGL(kerns[a.drop.thr ? 1 : 0], dim3(nchunk, a.H), dim3(256), R3_SMEM + 256, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
