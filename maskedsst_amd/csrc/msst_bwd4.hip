// bf16 attention half of a block, backward (reference vit_spatial_spectral.py:47-78 under PreNorm :22-29; a15 of
// SURVEY.md section 8) -- round-3 kernel, two heads per workgroup.  The tile program of a head is the one of msst_bwd3.hip
// (one GEMM = one wave, roles Q / K / V / O, 32x32x16 MFMAs, swizzled LDS tiles, four barriers per tile); what changes:
//
//   * a workgroup is EIGHT waves = two heads (2 blockIdx.y, 2 blockIdx.y + 1) x four roles, one workgroup per CU, walking
//     the same tiles.  The second head runs the same code TWO BARRIERS LATE: s_barrier counts arrivals, not program counters,
//     so two extra barriers in its prologue (and two in the first head's epilogue) shift its phases by half a tile for the
//     whole walk.  While head A is in the VALU-bound softmax phase the same SIMDs run head B's phase-4 MFMAs, A's
//     projection MFMAs meet B's contraction phase, and so on -- by construction, not by the drift of two independent
//     workgroups.
//   * the LN1(x) / da rows of a tile are fetched ONCE for both heads (head A's waves, double buffered by tile parity: head B
//     still reads tile k while A is two phases into tile k + 1).
//   * the d(LN1 out) rows of the two heads are summed in LDS (head B adds its C tiles onto the rows head A staged two
//     phases earlier) and leave as ONE bf16 partial per head PAIR: half the partial traffic here and in block_bwd_ln1.
//
// grid (nchunk, H / 2), 512 threads, 156.25 KB LDS.
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
#define MSST_B3_W4 6    // phase-4 weight fragments in registers (a ring: n requested before barrier B3, the other 12 - n as phase 4 frees slots)
#endif
#ifndef MSST_B3_W1AT
#define MSST_B3_W1AT 8   // phase-4 step behind which the next tile's phase-1 weights are requested (>= 6: behind the last phase-4 weight request)
#endif
#if defined(MSST_LAB) && !defined(MSST_LAB_EXP)
#define MSST_LAB_EXP 0
#endif
#ifndef MSST_B4_LAG
#define MSST_B4_LAG 2   // barriers head B runs behind head A (1 or 2; 3 would need a second OUT buffer)
#endif
#ifndef MSST_B4_PRA
#define MSST_B4_PRA 0x0011   // s_setprio level of phase 1 | 2 | 3 | 4 (one hex digit each) of head A's waves
#endif
#ifndef MSST_B4_PRB
#define MSST_B4_PRB 0x0011   // ... of head B's waves
#endif
#define B4_PRIO(ph) do { if (grp) __builtin_amdgcn_s_setprio((MSST_B4_PRB >> (4 * (4 - (ph)))) & 3); \
                         else __builtin_amdgcn_s_setprio((MSST_B4_PRA >> (4 * (4 - (ph)))) & 3); } while (0)

namespace msst {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) s16x8 lds_s16x8;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// LDS map (bytes).  64-wide bf16 tiles: 128-byte rows; 96-wide tiles: 192-byte rows.
// rows: two buffers (tile parity) of XN | DA; then the private tiles of the two heads (offsets below are relative to the
// head's base R4_G0 + grp * R4_GSZ); then the shared [64][96] staging of the summed d(LN1 out) rows and the row map.
constexpr int R4_ROWBUF = 24576, R4_DA = 12288;
constexpr int R4_G0 = 49152, R4_GSZ = 49152;
constexpr int R3_Q = 0, R3_K = 8192, R3_DO = 16384, R3_V = 24576, R3_P = 32768, R3_DS = 40960;
constexpr int R4_OUT = 147456, R4_MAP = 159744, R4_SEQ = 160000, R4_SEQO = 160272, R4_QT = 160272 + 272, R4_LSE = R4_QT + 16, R4_SMEM = R4_LSE + 1024;   // row map [64], sequence bases [65] x 2, tile ring [4], saved softmax statistics [tile parity][head][64] fp32

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
// the same with the fragment offset split into a wave-uniform base (SGPR) and a small constant that rides in the instruction's 12-bit
// offset field (< 4 KB: four fragments per base).  One SGPR per FOUR fragments instead of one per fragment: with twelve phase-1 and
// twelve phase-4 fragment offsets precomputed, the register allocator parked two dozen of them in VGPR lanes and every request paid a
// v_readlane + s_nop 4 to get its offset back (47 spilled SGPRs, ~50 reloads per tile).
__device__ __forceinline__ s16x8 ld_w32b(const void* w, int base_bytes, int imm_bytes, int lane16) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + imm_bytes, base_bytes, 0));
}
__device__ __forceinline__ int launder_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
// 16 bytes per lane, global -> LDS without passing registers: lane i's bytes land at lds_dst + 16 i (lds_dst wave uniform, below
// 64 KB); a lane whose offset lies outside the descriptor writes zeros.  Opaque to the compiler's vmcnt bookkeeping (the builtin
// form makes every later LDS read wait for vmcnt(0)): loads return in order, so the compiler's own waits only become more
// conservative, and the consumer side is ordered by an explicit vmcnt(0) + barrier.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ void bar3() {
    lds_barrier();
}
__device__ __forceinline__ int launder3(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// k-step f of a 32 x 32 bf16 identity as an A operand fragment (32 rows x 16 k; lane = 32 (k / 8 % 2) + row, 8 consecutive k per
// lane): element e of lane (row i, half hi) is 1.0 where 16 f + 8 hi + e == i.  Head B's phase 4 starts from head A's staged rows
// through two such MFMAs per row tile -- C[m][row] = sum_k I[m][k] staged[row][k], exact -- instead of reading, widening and
// adding them on the VALU behind its own MFMAs (read - widen - add - round: 64 VALU + 8 LDS instructions more per head-B wave and tile).  Built on the VALU (a dozen instructions per fragment): as a
// table in memory the two loads sink to their only use and the phase starts with an L2 round trip.
__device__ __forceinline__ s16x8 ident32_frag(int f, int i, int hi) {
    const int e = i - 16 * f - 8 * hi;                                  // 0 .. 7 where this lane holds the 1
    const unsigned one = 0x3F80u << ((e & 1) << 4), d = (unsigned)(e >> 1);   // (e outside 0 .. 7: d matches no dword)
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = d == (unsigned)k ? one : 0u;
    return __builtin_bit_cast(s16x8, w);
}

}  // namespace

// QUEUE (data parallel, opt-in): the tiles of a head pair are not statically partitioned over its workgroups (tile = chunk +
// k nchunk) but drawn from one agent-scope counter per head pair, so a workgroup that starts late -- its CU was held by a
// communication kernel's channel -- simply draws fewer tiles instead of running its whole share BEHIND the others.  Wave O of head
// A draws two walk steps ahead (the atomic's round trip hides under phases 2 and 3) and publishes the tile through a four-entry LDS
// ring; both heads read the same sequence, head B half a tile later.  The partition then depends on timing: gradients are no
// longer bit-reproducible from run to run (summation order), which is why the static form stays the single-GPU default.
// LSE: the forward saved log2 of every query's softmax denominator (BlockArgs.lse_out): the softmax phase computes p = exp2(s c - lse)
// directly -- no row maximum, no row sum, no reciprocal, one cross-lane reduction (delta) instead of three.  The 64 values of a
// (tile, head) ride in with the tile's rows: one 256-byte LDS-DMA per head into a buffer of the tile's parity.
// LSE = 2 (MSST_LSE_RENORM, round 6): the forward that saved the statistics multiplied IEEE-half operands (MSST_FWD_HALF), so its scores
// are not the ones recomputed here from bf16 rows and weights -- on peaky rows (|s c| ~ 40) the difference is several percent of a
// probability.  lse then only serves as the exponent offset (no row maximum needed: the exponentials stay near 1) and the row is
// normalised by its OWN sum: p = softmax of this kernel's scores exactly, as without saved statistics, for one reduction more than LSE = 1.
template <bool DROP, bool QUEUE, int LSE>
__global__ __launch_bounds__(512, 1) void block_bwd_attn_r4_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char* const sm = (lds_char*)smem_raw;

    // waves 0-3: head A = 2 blockIdx.y, waves 4-7: head B = 2 blockIdx.y + 1 (wave w and w + 4 share a SIMD: same role, other head)
    const int tid = threadIdx.x & 255, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = wv >> 2, wave = wv & 3;
    const int H = a.H, inner = H * 64, h = 2 * blockIdx.y + grp;
    const int gb = R4_G0 + grp * R4_GSZ;   // LDS base of this head's tiles
    const TileMap tm = a.tm;
    const int L = tm.L;
    bf16_t* part = reinterpret_cast<bf16_t*>(a.dxn_part) + (long)blockIdx.y * a.ntok * 96;   // one partial per head pair

    // wave roles: 0 Q, 1 K, 2 V, 3 O
    const bool roleO = wave == 3;
    const void* w1p = roleO ? a.w.woutT32 : a.w.wqkv32;                       // phase-1 weights: [rows][96], rows row1_0 ..
    const int f1_0 = ((roleO ? h * 64 : (wave * H + h) * 64) >> 5) * 6;        // fragment (dt, ks) = f1_0 + 6 dt + ks
    const int f4_0 = (wave < 3 ? wave : 0) * ((3 * inner) >> 4) + ((h * 64) >> 4);   // phase 4 (m tile = wave): + (which * inner >> 4) + ks
    const int p1_row = roleO ? R4_DA : 0;   // + row buffer of the tile
    const int p1_out = gb + (wave == 0 ? R3_Q : wave == 1 ? R3_K : wave == 2 ? R3_V : R3_DO);
    const bool pathX = wave == 0 || wave == 3;
    // phase 3: C[row][d] = sum arr1 . arr2 (see below), weight-gradient partner arrX
    const int p3_a1 = gb + ((wave == 0 || wave == 1) ? R3_DS : R3_P);
    const int p3_a2 = gb + (wave == 0 ? R3_K : wave == 1 ? R3_Q : wave == 2 ? R3_DO : R3_V);

    f32x16 G[2][3];   // persistent weight-gradient accumulators: G[d tile][m tile], C[i = head channel][j = model feature]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i][j] = zero16();

    // (sequence slot << 16 | position) of the 64 rows of a tile is tile invariant: a 64-entry table in LDS instead of registers
    // that would live (= be spilled) across the whole tile loop.  slot 0xffff marks a padding row; its position is still
    // r - (r / L) L, so that "first key of my sequence" = row - position holds for every row.
    unsigned* const rowmap = reinterpret_cast<unsigned*>(smem_raw + R4_MAP);
    if (threadIdx.x < 64) {
        const int sq = tid / L, ps = tid - sq * L;
        rowmap[tid] = ((unsigned)((sq >= tm.TS ? 0xffff : sq) & 0xffff) << 16) | (unsigned)ps;
    }
    __syncthreads();
    // key tiles (16 keys each) that the 16 queries of this wave can see: those overlapping [first key of the first query's
    // sequence, last key of the last query's sequence]
    int kvalid = 0;   // (LSE, short sequences) bit 4 t + r: key 16 t + 4 (lane / 16) + r belongs to the sequence of this lane's query 16 wave + lane % 16
    if (LSE && L < 64) {
        const int l_ = tid & 63, qr = 16 * wave + (l_ & 15);
        const int qlo_ = qr - (int)(rowmap[qr] & 0xffffu), qhi_ = qlo_ + L;
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 4; ++r) { const int key = 16 * t + 4 * (l_ >> 4) + r; kvalid |= (int)(key >= qlo_ && key < qhi_) << (4 * t + r); }
    }
    unsigned need;
    {
        const int r0 = 16 * wave, r1 = 16 * wave + 15;
        const int klo = r0 - (int)(rowmap[r0] & 0xffffu), khi = min(63, r1 - (int)(rowmap[r1] & 0xffffu) + L - 1);
        unsigned m = 0;
        for (int t = 0; t < 4; ++t) m |= (unsigned)(16 * t <= khi && 16 * t + 15 >= klo) << t;
        need = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }
    // row-wise LDS address of this thread's j-th 16-byte slot in a 96-wide tile: row tid / 4, logical slot 4 j + tid % 4 -- the four
    // threads of a row cover 64 contiguous bytes per store instruction (3 (tid % 4) + j -- 16-byte pieces 48 bytes apart -- was round 3's)
    auto row_slot = [&](int j) -> unsigned {
        const int t_ = launder3(tid);
        const int row = t_ >> 2, s = 4 * j + (t_ & 3);
        return (unsigned)(row * 192 + (((s & ~3) | ((s & 3) ^ fz2(row))) << 4));
    };
    // The LN1(x) / da rows of the NEXT tile go straight from HBM into the other row buffer (LDS-DMA, head A's waves: 1 KB per
    // instruction, three instructions per wave and array): lane <-> 16-byte LDS position p of the 12 KB array, which holds
    // logical slot s of row p / 12 (the swizzle is an involution), fetched from token(row) * 192 + 16 s; a padding row's
    // offset lies outside the descriptor and is zero filled.  Requested right behind barrier B2 -- the buffer's last reader,
    // head B's weight-gradient GEMM of the tile before, finished one barrier earlier -- and waited for before barrier B4.
    const __amdgpu_buffer_rsrc_t rows_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.xn), 0, (int)(a.ntok * 192), 0x00020000);
    const __amdgpu_buffer_rsrc_t rows_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dab), 0, (int)(a.ntok * 192), 0x00020000);
    // token arithmetic stays off the issuing waves' critical path: wave O of head A leaves the token of position 0 of every
    // sequence slot of the tile being fetched in a 64-entry LDS table (one integer division per lane and tile, during the softmax
    // phase), and what a lane needs besides -- slot of the table, byte offset of its 16 bytes from that token -- is tile invariant
    int* const seqbase = reinterpret_cast<int*>(smem_raw + R4_SEQ);   // [64] = -1: padding rows
    int* const seqout = reinterpret_cast<int*>(smem_raw + R4_SEQO);   // the same for the copy-out: sequence bases of head B's current tile
    auto fill_seqbase = [&](int* table, int tile_) {
        const int sx = launder3(tid) & 63;
        const int q = tile_ * tm.TS + sx;
        int base = -1;
        if (tile_ < a.ntiles && sx < tm.TS && q < tm.nseq) {
            if (tm.mode == 0) base = q * tm.N;
            else { const int b = tm.nshift >= 0 ? (q >> tm.nshift) : q / tm.N; base = b * tm.T + (q - b * tm.N); }
        }
        table[sx] = base;
    };
    int* const qt = reinterpret_cast<int*>(smem_raw + R4_QT);
    int qpend = 0;   // (QUEUE, lane 0 of wave O of head A) the draw in flight
    if (wv == 3) {
        int t0 = (int)blockIdx.x;
        if (QUEUE) {
            int r = 0;
            if ((tid & 63) == 0) { r = __hip_atomic_fetch_add(a.queue + blockIdx.y, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); qt[0] = r; qt[1] = r + 1; }
            t0 = __builtin_amdgcn_readfirstlane(r);
        }
        fill_seqbase(seqbase, t0);
        if ((tid & 63) == 0) { seqbase[64] = -1; seqout[64] = -1; }
    }
    __syncthreads();
    unsigned rinv[3], rsx[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int p = 192 * wave + 64 * jj + (tid & 63);
        const int row = p / 12, sl = p - 12 * row;
        const int s_ = (sl & ~3) | ((sl & 3) ^ fz2(row));
        const unsigned sp = rowmap[row];
        rsx[jj] = (unsigned)(R4_SEQ + 4 * min((int)(sp >> 16), 64));
        rinv[jj] = (sp & 0xffffu) * (unsigned)(tm.mode == 0 ? 192 : 192 * tm.N) + (unsigned)s_ * 16u;
    }
    const __amdgpu_buffer_rsrc_t lse_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lse), 0, LSE ? (int)min((long)a.ntiles * H * 256, 0x7fffffffL) : 0, 0x00020000);
    // (tile_, buf): the statistics of both heads of tile tile_ -> parity buffer of row buffer buf; wave Q of head A, one 4-byte piece per lane and head
    auto dma_lse = [&](int tile_, int buf) {
        if (!LSE || wv != 0) return;
        const unsigned dst = (unsigned)(R4_LSE + (buf ? 512 : 0));
        const unsigned voff = tile_ < a.ntiles ? (unsigned)(((tile_ * H + 2 * (int)blockIdx.y) * 64 + (launder3(tid) & 63)) * 4) : 0x80000000u;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(dst), "v"(voff), "s"(lse_rs) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(dst + 256), "v"(voff + 256u), "s"(lse_rs) : "memory", "m0");   // head B: the next 256 bytes
    };
    auto dma_rows = [&](int buf) {
        int base[3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) base[jj] = *reinterpret_cast<const __attribute__((address_space(3))) int*>(sm + rsx[jj]);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int bs = base[jj];
            const unsigned voff = bs < 0 ? 0x80000000u : (unsigned)bs * 192u + rinv[jj];
            const unsigned dst = (unsigned)(buf + 3072 * wave + 1024 * jj);
            dma16(rows_x, dst, voff);
            dma16(rows_d, dst + R4_DA, voff);
        }
    };
    if (!grp) { dma_rows(0); dma_lse(QUEUE ? __builtin_amdgcn_readfirstlane(qt[0]) : (int)blockIdx.x, 0); wait_vm0(); }
    // copy-out of a finished tile (head B's waves; both heads' rows were summed in OUT by B's phase 4, published by barrier B4):
    // whole rows to the head pair's partial; buffer stores, a padding row gets an offset outside the descriptor and is dropped.
    unsigned cinv, csx;   // copy-out: thread <-> (row tid / 4, 48 bytes)
    {
        const unsigned sp = rowmap[tid >> 2];
        csx = (unsigned)(R4_SEQO + 4 * min((int)(sp >> 16), 64));
        cinv = (sp & 0xffffu) * (unsigned)(tm.mode == 0 ? 192 : 192 * tm.N) + (unsigned)(tid & 3) * 16u;
    }
    auto copy_out = [&]() {
        const int bs = *reinterpret_cast<const __attribute__((address_space(3))) int*>(sm + csx);
        u32x4 v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const lds_u32x4*>(sm + R4_OUT + row_slot(j));
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)(a.ntok * 192), 0x00020000);
        const unsigned voff = bs < 0 ? 0x80000000u : (unsigned)bs * 192u + cinv;   // (+ 128 must not wrap)
#pragma unroll
        for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rp, voff + 64 * j, 0, 0);
    };
    // (Who copies out, measured in round 4: head B behind its phase 1 -- the longest stretch of ITS interval; wave O of head A alone at
    // the end of its phase 3 -- 1.3 k cycles for one wave, it becomes the last arriver; all four waves of head A there, a quarter
    // each: -0.5 %, kept.)
    // phase-1 weight fragments [d tile][k step]: tile invariant, but 48 registers the softmax phase has no room for -- all twelve
    // are re-requested from L2 during phase 4 of the tile before (the rows no longer pass registers: a ring of eight refilled
    // inside phase 1 left its last two k-steps waiting on L2)
    s16x8 w1[2][6];
    auto load_w1 = [&]() {
        const int l16 = (launder3(tid) & 63) * 16;
        const int b1 = launder_s(f1_0 * 1024);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const int fi = 6 * dt + ks;
                w1[dt][ks] = ld_w32b(w1p, b1 + (fi >> 2) * 4096, (fi & 3) * 1024, l16);   // (base + immediate: one precomputed offset per fragment cost registers)
            }
    };
    load_w1();
    lds_barrier();   // the first tile's rows are in row buffer 0
    if (grp) {
#pragma unroll
        for (int i = 0; i < MSST_B4_LAG; ++i) lds_barrier();   // head B: MSST_B4_LAG phases behind head A from here on
    }

    s16x8 idf_h[2];   // head B: the two identity fragments of its phase-4 add, tile invariant (8 registers; rebuilt per tile they were 24 VALU instructions in front of head B's phase 4, the longest phase of the kernel's longest interval: -0.5 %)
    {
        const int t0_ = launder3(tid);
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2) idf_h[f2] = ident32_frag(f2, t0_ & 31, (t0_ >> 5) & 1);
    }
    int xb = 0;   // row buffer of the tile
    int ks = 0;   // walk step
    for (int tile = QUEUE ? __builtin_amdgcn_readfirstlane(qt[0]) : (int)blockIdx.x; tile < a.ntiles;
         xb = R4_ROWBUF - xb, ++ks, tile = QUEUE ? __builtin_amdgcn_readfirstlane(qt[ks & 3]) : tile + (int)gridDim.x) {
        const int p1_in = xb + p1_row, p3_x = p1_in;
#if defined(MSST_STAMPS)
        // cycle stamps of lane 0 of every wave of workgroup (7, 1), a mid-walk tile: stamps[16 wave + i]
        const bool stamp_on = (a.dbg & 8) && a.stamps && blockIdx.x == 7 && blockIdx.y == 1 && (threadIdx.x & 63) == 0 && tile == (int)blockIdx.x + 20 * (int)gridDim.x;
#define R4_STAMP(i) do { if (stamp_on) a.stamps[16 * wv + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define R4_STAMP(i) do { } while (0)
#endif
        R4_STAMP(0);
        B4_PRIO(1);
        unsigned p2a[4];   // the softmax phase's lane addresses and first key of the lane's sequence, computed under phase 1's MFMAs as well
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
            // Row tile 0 first, then row tile 1 (twelve MFMAs each): the conversion + LDS stores of row tile 0's two C tiles and the
            // softmax phase's lane addresses issue BETWEEN the MFMAs of row tile 1 (a 32x32x16 MFMA hides four single-issue
            // instructions of its own wave) instead of all 64 values of the four C tiles behind the last MFMA, where both waves of
            // the SIMD sit in their epilogues with the matrix pipe idle.
            const unsigned L7 = p1_out + l31 * 128 + (fz(l31) << 4) + 8 * hi;
            auto ep1 = [&](int rt, int i) {
                const int dt = i >> 2, q4 = i & 3;
                lds_w64(sm, (L7 ^ ((4 * dt + q4) << 4)) + rt * 4096, pk4(c[dt][rt], q4));
            };
            auto p2a_part = [&](int part) {
                const int l = t_ & 63, g = l >> 4, c16 = l & 15, fzc = fz(c16);
                if (part == 0) {
                    p2a[0] = gb + c16 * 128 + ((g ^ fzc) << 4);          // (gb: no bits below 16 K, commutes with the XORs)
                    p2a[1] = gb + c16 * 128 + (((4 + g) ^ fzc) << 4);
                    asm volatile("" : "+v"(p2a[0]));
                    asm volatile("" : "+v"(p2a[1]));
                } else {
                    p2a[2] = gb + (16 * wave + c16) * 128 + (((g >> 1) ^ fzc) << 4) + 8 * (g & 1);
                    p2a[3] = (unsigned)(16 * wave + c16 - (int)(rowmap[16 * wave + c16] & 0xffffu));
                    asm volatile("" : "+v"(p2a[2]));
                    asm volatile("" : "+v"(p2a[3]));
                }
            };
            s16x8 fb[4];   // LN1(x) / da fragments of step s = 6 rt + ks, three steps ahead of their MFMAs
            auto rd1 = [&](int s_) {
                const int rt = s_ / 6, ks = s_ % 6;
                fb[s_ & 3] = lds_r128(sm, bin[ks & 1] + 64 * (ks >> 1) + rt * 32 * 192);
            };
#ifdef MSST_LAB
            // kernel-study build only (msst_version() < 0, refused by maskedsst_amd/_lib.py): MSST_LAB_EXP & 1 = the q / k / v waves skip
            // their projections (WRONG results: stale tiles) -- what any scheme that hands q / k / v to the backward could gain at most
            if ((MSST_LAB_EXP & 1) && !roleO) { p2a_part(0); p2a_part(1); } else
#ifdef MSST_LAB_QKV
            // (tools/gate_qkv.py) ... and with a scratch given, the q / k / v waves fetch their 8 KB tile from it by LDS-DMA instead
            // (garbage values; never waited for by itself: what a fetch placed a tile ahead would cost in issue slots and HBM traffic)
            if (a.stamps && !roleO) {
                p2a_part(0); p2a_part(1);
                const char* src = reinterpret_cast<const char*>(a.stamps) + ((long)(tile * H + h) * 3 + wave) * 8192 + (t_ & 63) * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned dst = (unsigned)(size_t)sm + p1_out + i * 1024;
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src + i * 1024) : "memory", "m0");
                }
            } else
#endif
#endif
            {
            rd1(0); rd1(1); rd1(2);
#pragma unroll
            for (int s_ = 0; s_ < 12; ++s_) {
                const int rt = s_ / 6, ks = s_ % 6;
                if (s_ + 3 < 12) rd1(s_ + 3);
                MSST_SCHED_FENCE();
                c[0][rt] = mma32(w1[0][ks], fb[s_ & 3], c[0][rt]);
                if (s_ == 2) p2a_part(0);
                if (s_ >= 7 && s_ <= 10) ep1(0, 2 * (s_ - 7));
                MSST_SCHED_FENCE();
                c[1][rt] = mma32(w1[1][ks], fb[s_ & 3], c[1][rt]);
                if (s_ == 3) p2a_part(1);
                if (s_ >= 7 && s_ <= 10) ep1(0, 2 * (s_ - 7) + 1);
                MSST_SCHED_FENCE();
            }
            R4_STAMP(12);
#pragma unroll
            for (int i = 0; i < 8; ++i) ep1(1, i);
            }
        }
        // (copy-out placed behind phase 1's MFMAs: in front of them the stores sat in vmcnt order before the phase's weight requests)
        R4_STAMP(1);
        bar3();   // B1
        R4_STAMP(2);
        B4_PRIO(2);
        if (wv == 3) {   // (read by head A's row requests behind barrier B2)
            fill_seqbase(seqbase, QUEUE ? qt[(ks + 1) & 3] : tile + (int)gridDim.x);
            if (QUEUE && (tid & 63) == 0) qpend = __hip_atomic_fetch_add(a.queue + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // walk step ks + 2
        }
        if (wv == 7) fill_seqbase(seqout, tile);                        // (read by head B's copy-out of this tile, three barriers on)
        // ---------------- phase 2: wave <-> 16 query rows; S^T, softmax, P, dP^T, dS (16 x 16 x 32 MFMAs) ----------------
        {
            typedef PBF16 P;
            const int t_ = launder3(tid);
            const int l = t_ & 63, g = l >> 4, c16 = l & 15;
            const int qlo = (int)p2a[3], qhi = qlo + L;   // keys of this query's sequence
            unsigned ak[2], aq[2];   // A operand rows 16 t + c16 (k, v), B operand rows 16 wave + c16 (q, dO); k-step ks2
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                ak[ks2] = p2a[ks2];
                aq[ks2] = ak[ks2] + wave * 2048;
            }
            const unsigned L8 = p2a[2];   // ^ (t << 5)
            // Short sequences (spectral blocks): a wave's 16 queries only meet the key tiles that overlap their own sequences -- bit t
            // of `need` (wave uniform, tile invariant).  Every other 16 x 16 score tile is masked anyway and is skipped altogether
            // (operand reads, MFMAs, exps, dropout hashes); its P / dS entries are stored as zeros.
            f32x4 pr[4], dp[4];
            f32x4 dm[4];   // dropout multipliers of site 1 (0 or 1 / (1 - p)): P and dP see the same ones
#pragma unroll
            for (int t = 0; t < 4; ++t) dm[t] = zero4();
            const float cs = a.scale * 1.44269504088896340736f;   // exp(scale (s - max)) = exp2(s c - max c), c = scale log2 e
            int kv_ = kvalid;
            asm volatile("" : "+v"(kv_));   // (opaque per tile: left visible, the sixteen bit tests are hoisted out of the walk as lane masks in 32 SGPRs, spilled, and re-read with two v_readlane per score)
            float lse2 = 0.f;
            if (LSE) lse2 = *reinterpret_cast<const __attribute__((address_space(3))) float*>(sm + R4_LSE + (xb ? 512 : 0) + grp * 256 + (16 * wave + c16) * 4);
            // The dropout hash needs no data: all four key tiles' keep bits first, while the phase's operand reads are in flight,
            // instead of 12 quarter-rate multiplies inside the dependent chain max -> exp -> sum -> P.  (Issued under phase 1's MFMAs
            // -- the VALU is idle there too -- it cost 2 %: phase 1 is on the critical path of its barrier interval.)
            if (DROP) {
                const unsigned t16 = a.drop.thr << 16;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    unsigned ha, hb;
                    drop_bits(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c16) * 16 + t * 4 + g), ha, hb);
                    // the multipliers themselves (0 or 1 / (1 - p)), one compare + one select per score, instead of keep bits packed into
                    // a word here and unpacked (and + compare + select per score) behind the softmax: ~70 instructions per wave and tile
                    dm[t][0] = (ha << 16) >= t16 ? a.drop.scale : 0.f;
                    dm[t][1] = ha >= t16 ? a.drop.scale : 0.f;
                    dm[t][2] = (hb << 16) >= t16 ? a.drop.scale : 0.f;
                    dm[t][3] = hb >= t16 ? a.drop.scale : 0.f;
                }
            }
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
                if (LSE) {
                    // p = exp2(s c - lse): the statistics of the forward's own softmax of this (tile, head, query) -- the rows, weights and
                    // rounding points of q / k are the forward's, so the recomputed scores differ from the ones it normalised by fp32
                    // summation order only (a flipped bf16 rounding of q or k now and then: sum p = 1 to ~1e-3, the kernels' bf16 level)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (!on(t)) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(pr[t][r], cs, -lse2));
                            // keys outside the query's own sequence: bit 4 t + r of the lane's (tile invariant) validity mask, as 0 / ~0
                            if (MASKED) {
                                int m_;   // (as inline asm: from the builtin the compiler makes v_and + v_cmp_ne + v_cndmask, three instructions per score)
                                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m_) : "v"(kv_), "n"(4 * t + r));
                                pr[t][r] = __int_as_float(__float_as_int(e) & m_);
                            } else pr[t][r] = e;
                        }
                    }
                    if (LSE == 2) {
                        // (the row sums on the matrix cores instead -- ones x bf16(e), no cross-lane VALU -- measured 531 against 519 us: LABNOTES round 6, row 16)
                        f32x4 es = zero4();
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (on(t)) es = es + pr[t];
                        const float inv = __builtin_amdgcn_rcpf(colgroup_sum((es[0] + es[1]) + (es[2] + es[3])));
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (on(t)) pr[t] = pr[t] * inv;
                    }
                } else {
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
                for (int t = 0; t < 4; ++t) if (on(t)) pr[t] = pr[t] * inv;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) { lds_w64(sm, R3_P + (L8 ^ (t << 5)), s16x4{0, 0, 0, 0}); continue; }
                    f32x4 pd = pr[t];   // site 1: O and dV see the dropped probabilities, the softmax backward the raw ones
                    if (DROP) {
                        pd = pd * dm[t];
                    }
                    lds_w64(sm, R3_P + (L8 ^ (t << 5)), f2bf4(pd));   // P[query][key]
                    dp[t] = dp[t] * pd;            // Pd o dPd (see below)
                }
                // dS = P o (dP - delta), dP = dPd o dm, delta = sum_key P o dP  ==  Pd o dPd - P delta, delta = sum_key Pd o dPd  (Pd = P o dm:
                // the dropped probabilities just stored): the products Pd o dPd serve both, dPd is never multiplied by dm -- three
                // instructions per score instead of four
                // (a tree over the key tiles, two packed adds per level: summed one by one the sixteen adds are ONE dependent chain, 8.5
                // cycles each with nothing else left to issue at the end of the phase)
                f32x4 dsum = zero4();
                {
                    f32x4 lvl[4];
                    int n = 0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (on(t)) lvl[n++] = dp[t];
                    if (NM == 0) {   // (run-time tile set: absent tiles hold zeros -- dp was cleared and never multiplied)
                        dsum = (dp[0] + dp[1]) + (dp[2] + dp[3]);
                    } else {
                        dsum = n == 4 ? (lvl[0] + lvl[1]) + (lvl[2] + lvl[3]) : n == 3 ? (lvl[0] + lvl[1]) + lvl[2] : n == 2 ? lvl[0] + lvl[1] : lvl[0];
                    }
                }
                float delta = -colgroup_sum((dsum[0] + dsum[1]) + (dsum[2] + dsum[3]));
                // dS WITHOUT the softmax scale (dim_head^-0.5 = 2^-3, exact in bf16): it is folded into the q / k blocks of the
                // phase-4 weights (msst_prep_weights, pack = 2) and into the dWq / dWk slabs at the end of the kernel
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!on(t)) { lds_w64(sm, R3_DS + (L8 ^ (t << 5)), s16x4{0, 0, 0, 0}); continue; }
                    f32x4 d4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d4[r] = fmaf(pr[t][r], delta, dp[t][r]);
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
        // ---------------- phase 3: the four contractions over rows, one per wave ----------------
        //   Q: dq[query][d] = sum_key dS[query][key] k[key][d]      O: o[query][d] = sum_key P[query][key] v[key][d]     (path X)
        //   K: dk[key][d]   = sum_query dS[query][key] q[query][d]  V: dv[key][d]  = sum_query P[query][key] dO[query][d] (path Y)
        // C[i = row][j = d]; then G[d][m] += sum_row C[row][d] . {LN1(x) | da}[row][m] with the packed C tiles as A operand.
        // The second operand (k | q | dO | v) has been complete since barrier B1: its first fragments are requested BEFORE
        // barrier B2, so the phase starts with its MFMAs instead of an LDS round trip.
        {
            s16x8 w4[MSST_B3_W4];   // phase-4 weight fragments of this wave's m tile: a ring, refilled as phase 4 consumes them
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
            s16x8 pa[2][4];   // [d tile][k step]: A operand of the weight-gradient GEMM
            // d tile 0 first, then d tile 1 (eight MFMAs each): the conversions of d tile 0's C tiles -- the weight-gradient GEMM's A
            // operand and the dq | dk | dv rows of phase 4 -- and their LDS stores issue between the MFMAs of d tile 1.  The stores go
            // over the tile this wave reads as its second operand: every read of it is issued (pass 0) before the first store; a
            // wave's LDS instructions execute in order.  One straight-line instance per role (Q: path X + stores, O: path X, K / V:
            // path Y + stores) instead of a role branch per k-step.
            s16x8 fb0[4];   // second operand, d tile 0: requested before barrier B2 (complete since barrier B1)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fb0[kk] = lds_tr2(sm, p3_a2 + tr[0][0] + 2048 * kk, p3_a2 + tr[0][1] + 2048 * kk);
            R4_STAMP(3);
            bar3();   // B2
            R4_STAMP(4);
            if (!grp) { dma_rows(R4_ROWBUF - xb); dma_lse(QUEUE ? __builtin_amdgcn_readfirstlane(qt[(ks + 1) & 3]) : tile + (int)gridDim.x, R4_ROWBUF - xb); }
            B4_PRIO(3);
            auto phase3 = [&](auto role_tag) {
                constexpr int ROLE = decltype(role_tag)::value;   // 0: Q, 1: O, 2: K / V
                constexpr bool PX = ROLE < 2, WR = ROLE != 1;
                s16x8 fa[4][2], fb1[4];
                auto issue_a = [&](int kk) {
                    if (PX) {
                        fa[kk][0] = lds_r128(sm, a1 ^ (kk << 5));
                        fa[kk][1] = lds_r128(sm, (a1 ^ (kk << 5)) + 4096);
                    } else {
                        fa[kk][0] = lds_tr2(sm, p3_a1 + tr[0][0] + 2048 * kk, p3_a1 + tr[0][1] + 2048 * kk);
                        fa[kk][1] = lds_tr2(sm, p3_a1 + tr[1][0] + 2048 * kk, p3_a1 + tr[1][1] + 2048 * kk);
                    }
                };
                const unsigned L7 = p3_a2 + l31 * 128 + (fz(l31) << 4) + 8 * hi;
                s16x4 q0[2][4];   // bf16 of d tile 0's C tiles [row tile][4 registers]
                auto ep3 = [&](int ct, int q4) {
                    q0[ct][q4] = pk4(c[ct][0], q4);
                    asm volatile("" : "+v"(q0[ct][q4]));   // (converted HERE, between the MFMAs: without stores -- wave O -- the compiler sinks the conversions behind the last one)
                    if (WR) lds_w64(sm, (L7 ^ ((4 * ct + q4) << 4)), q0[ct][q4]);
                };
                // C tile by C tile -- (row tile 0, d 0), (1, d 0), (0, d 1), (1, d 1), four MFMAs each -- with the conversion + LDS
                // stores of a tile under the MFMAs of the next (three instructions per MFMA): only the last tile's twelve are left
                // behind the last MFMA, where the matrix pipe has nothing to hide them under (d-tile-major it was two tiles' worth)
                s16x4 q1[4];      // bf16 of C tile (row tile 0, d tile 1)
                auto ep3b = [&](int q4) {
                    q1[q4] = pk4(c[0][1], q4);
                    asm volatile("" : "+v"(q1[q4]));
                    if (WR) lds_w64(sm, (L7 ^ (q4 << 4)) + 4096, q1[q4]);
                };
                issue_a(0); issue_a(1);
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const int tl = st >> 2, kk = st & 3, ct = tl & 1, dt = tl >> 1;   // tile order: (ct 0, dt 0), (1, 0), (0, 1), (1, 1)
                    if (st + 2 < 4) issue_a(st + 2);
                    if (st < 4) fb1[st] = lds_tr2(sm, p3_a2 + tr[1][0] + 2048 * st, p3_a2 + tr[1][1] + 2048 * st);
                    MSST_SCHED_FENCE();
                    c[ct][dt] = mma32(fa[kk][ct], dt ? fb1[kk] : fb0[kk], c[ct][dt]);
                    if (tl == 1) ep3(0, kk);          // (row tile 0, d 0) under (1, d 0)
                    if (tl == 2) ep3(1, kk);          // (1, d 0) under (0, d 1)
                    if (tl == 3) ep3b(kk);            // (0, d 1) under (1, d 1)
                    MSST_SCHED_FENCE();
                }
                R4_STAMP(11);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s16x8 r;
                    const s16x4 lo = q0[kk >> 1][2 * (kk & 1)], hi4 = q0[kk >> 1][2 * (kk & 1) + 1];
                    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
                    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
                    pa[0][kk] = r;
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s16x8 r;
                    const s16x4 lo = q1[2 * kk], hi4 = q1[2 * kk + 1];
                    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
                    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
                    pa[1][kk] = r;
                    pa[1][2 + kk] = pk8(c[1][1], kk);
                }
                if (WR) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) lds_w64(sm, (L7 ^ ((4 + q4) << 4)) + 4096, pk4(c[1][1], q4));
                }
            };
            if (wave == 0) phase3(std::integral_constant<int, 0>{});
            else if (wave == 3) phase3(std::integral_constant<int, 1>{});
            else phase3(std::integral_constant<int, 2>{});
            // transposed 32-column fragment of a 96-wide tile in the C-layout row order: k row = 16 kk + 8 a + 4 hi + i / 4
            const unsigned L4 = (4 * hi + (i >> 2)) * 192 + (((2 * u + b) ^ hi) << 4) + 8 * (i & 1);
            unsigned tx[2];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) tx[aa] = p3_x + (L4 ^ (aa << 5)) + 8 * aa * 192;
            // steps [LO, HI) of the weight-gradient GEMM's twelve (kk, mt) steps
            auto wgrad = [&](auto lo_c, auto hi_c) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
                if constexpr (HI > LO) {
                    s16x8 fx[MSST_B3_D3B + 1];   // step s = (kk, mt): transposed row fragment MSST_B3_D3B steps ahead
                    swpipe<HI - LO, MSST_B3_D3B>(
                        [&](int s0) {
                            const int st = s0 + LO, kk = st / 3, mt = st % 3;
                            fx[s0 % (MSST_B3_D3B + 1)] = lds_tr2(sm, tx[0] + 3072 * kk + 64 * mt, tx[1] + 3072 * kk + 64 * mt);
                        },
                        [&](int s0) {
                            const int st = s0 + LO, kk = st / 3, mt = st % 3;
                            G[0][mt] = mma32(pa[0][kk], fx[s0 % (MSST_B3_D3B + 1)], G[0][mt]);
                            G[1][mt] = mma32(pa[1][kk], fx[s0 % (MSST_B3_D3B + 1)], G[1][mt]);
                        });
                }
            };
            // requests of the weight-gradient GEMM's shadow: the first phase-4 weight fragments (waves Q, K, V), the next tile's rows
            const int l16 = l * 16;
            // (wave O, which has no phase 4, requests one hot fragment six times: a definition on every path keeps the register
            // allocator from shuffling the in-flight fragments of the other waves at the join)
            // phase-4 fragment k12 = (which, ks): base of `which` (one SGPR each) + 1024 ks in the offset field
            const int b4 = launder_s(roleO ? 0 : f4_0 * 1024), b4s = launder_s(roleO ? 0 : (inner >> 4) * 1024);
            auto ld_w4 = [&](int k12) -> s16x8 {
                return ld_w32b(a.w.wqkvT32, b4 + (k12 >> 2) * b4s, (k12 & 3) * 1024, l16);
            };
#pragma unroll
            for (int k12 = 0; k12 < MSST_B3_W4; ++k12) w4[k12] = ld_w4(k12);
            // copy-out of the tile of the walk step before (complete since barrier B2: head B's phase 4 ran two intervals behind)
            if (!grp && ks != 0) copy_out();   // (all four waves of head A, a quarter each)
            R4_STAMP(5);
            bar3();   // B3
            R4_STAMP(6);
            B4_PRIO(4);
            // the weight-gradient GEMM runs BEHIND barrier B3 (the rows it reads stay put: the next tile's go to the other row
            // buffer): phases 1 | 3 and 2 | 4 of the two heads, which share the barrier intervals, are then of equal length.  (Six or
            // all twelve of its steps in FRONT of the barrier, tried again in round 5 with the shorter softmax phase: +1.7 %.)
            wgrad(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
            R4_STAMP(9);
            // ---------------- phase 4: d(LN1 out)[row][m] = dq Wq + dk Wk + dv Wv, wave <-> 32 features (waves Q, K, V) ----------------
            if (roleO) {
                if (QUEUE && !grp && (tid & 63) == 0) qt[(ks + 2) & 3] = qpend;   // published by barrier B4; first read at the top of walk step ks + 1 (wave O: next tile's bases)
                load_w1();   // the next tile's phase-1 weights
            } else {
                f32x16 c4[2];   // [row tile]: C[i = m][j = row]
                c4[0] = zero16(); c4[1] = zero16();
                if (grp) {   // head B: accumulate ONTO head A's rows (staged two intervals ago), exact: 1.0 x bf16 in fp32
                    s16x8 fs[2][2], idf[2];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int f2 = 0; f2 < 2; ++f2)
                            fs[rt][f2] = lds_r128(sm, R4_OUT + (32 * rt + l31) * 192 + ((4 * wave + ((2 * f2 + hi) ^ fz2(l31))) << 4));
#pragma unroll
                    for (int f2 = 0; f2 < 2; ++f2) idf[f2] = idf_h[f2];
#pragma unroll
                    for (int f2 = 0; f2 < 2; ++f2) {
                        c4[0] = mma32(idf[f2], fs[0][f2], c4[0]);
                        c4[1] = mma32(idf[f2], fs[1][f2], c4[1]);
                    }
                }
                s16x8 fb4[MSST_B3_D4 + 1][2];   // step k12 = (which, ks): dq^T | dk^T | dv^T fragments MSST_B3_D4 steps ahead
                swpipe<12, MSST_B3_D4>(
                    [&](int k12) {
                        const int which = k12 >> 2, ks = k12 & 3;
                        const int reg = gb + (which == 0 ? R3_K : which == 1 ? R3_Q : R3_DO);
                        fb4[k12 % (MSST_B3_D4 + 1)][0] = lds_tr2(sm, reg + tr[0][0] + 2048 * ks, reg + tr[0][1] + 2048 * ks);
                        fb4[k12 % (MSST_B3_D4 + 1)][1] = lds_tr2(sm, reg + tr[1][0] + 2048 * ks, reg + tr[1][1] + 2048 * ks);
                    },
                    [&](int k12) {
                        c4[0] = mma32(w4[k12 % MSST_B3_W4], fb4[k12 % (MSST_B3_D4 + 1)][0], c4[0]);
                        c4[1] = mma32(w4[k12 % MSST_B3_W4], fb4[k12 % (MSST_B3_D4 + 1)][1], c4[1]);
                        if (k12 + MSST_B3_W4 < 12) w4[k12 % MSST_B3_W4] = ld_w4(k12 + MSST_B3_W4);
                        if (k12 == MSST_B3_W1AT) load_w1();   // the next tile's phase-1 weights, behind this phase's last weight request
                    });
                R4_STAMP(10);
                // head A stages its rows; head B, two phases later, adds its own onto them (fp32 add of the bf16 values, one rounding)
                const unsigned L9 = R4_OUT + l31 * 192 + (fz2(l31) << 4) + 8 * hi + 64 * wave;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const unsigned o = (L9 ^ (q4 << 4)) + rt * 6144;
                        f32x4 t4 = {c4[rt][4 * q4], c4[rt][4 * q4 + 1], c4[rt][4 * q4 + 2], c4[rt][4 * q4 + 3]};
                        lds_w64(sm, o, f2bf4(t4));
                    }
            }
            // the next tile's rows are in LDS.  The twelve phase-1 weight fragments requested above are this wave's twelve YOUNGEST memory
            // operations (head A: rows at barrier B2, then the phase-4 weights, then load_w1; requests return in order): vmcnt(12) would cover
            // the rows without sitting out the weights' L2 round trip in front of barrier B4 -- measured: nothing to gain (526.9 / 527.4 vs 526.9 / 531.2 us), and the
            // count would be an invariant to maintain; everything is waited for
            if (!grp) wait_vm0();
        }
        R4_STAMP(7);
        bar3();   // B4
        R4_STAMP(8);
    }
    if (!grp) {
#pragma unroll
        for (int i = 0; i < MSST_B4_LAG; ++i) lds_barrier();   // head A: the barriers head B is behind
    } else if (ks > 0) {   // (a workgroup without a tile never filled seqout: nothing to copy out)
        copy_out();
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

int launch_block_bwd_attn_r4(const AttnBwdArgs& a, int nchunk, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (a.tm.L > 64 || a.tm.L < 1 || (a.H & 1)) return MSST_ERR_UNSUPPORTED;
    if (!a.xn || !a.dab || !a.w.wqkv32 || !a.w.woutT32 || !a.w.wqkvT32 || nchunk < 1 || nchunk > a.ntiles) return MSST_ERR_BADARG;
    if (a.ntok * 192 >= 0x7ffffff0L) return MSST_ERR_UNSUPPORTED;   // 32-bit row offsets of the copy-out descriptor
    typedef void (*kern_t)(AttnBwdArgs);
    const kern_t kerns[12] = {&block_bwd_attn_r4_kernel<false, false, 0>, &block_bwd_attn_r4_kernel<true, false, 0>,
                              &block_bwd_attn_r4_kernel<false, true, 0>, &block_bwd_attn_r4_kernel<true, true, 0>,
                              &block_bwd_attn_r4_kernel<false, false, 1>, &block_bwd_attn_r4_kernel<true, false, 1>,
                              &block_bwd_attn_r4_kernel<false, true, 1>, &block_bwd_attn_r4_kernel<true, true, 1>,
                              &block_bwd_attn_r4_kernel<false, false, 2>, &block_bwd_attn_r4_kernel<true, false, 2>,
                              &block_bwd_attn_r4_kernel<false, true, 2>, &block_bwd_attn_r4_kernel<true, true, 2>};
    if (a.lse && (long)a.ntiles * a.H * 256 >= 0x7ffffff0L) return MSST_ERR_UNSUPPORTED;   // 32-bit offsets of the statistics' descriptor
    if (!attr_set) {
        for (int i = 0; i < 12; ++i) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, R4_SMEM);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    ProfScope ps(K_BWD_ATTN, st);
    hipLaunchKernelGGL(kerns[(a.drop.thr ? 1 : 0) + (a.queue ? 2 : 0) + (a.lse ? (a.lse_renorm ? 8 : 4) : 0)], dim3(nchunk, a.H / 2), dim3(512), R4_SMEM, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
