// bf16 attention half of a block, backward (reference vit_spatial_spectral.py:47-78 under PreNorm :22-29;
// a15 of SURVEY.md section 8) -- the tuned throughput kernel.  Same math, same HBM interface, same dropout
// streams and the same slab / partial layouts as block_bwd_attn_kernel<PBF16> (msst_bwd.hip, which stays the
// fp32 parity kernel and the bf16 reference for this one); what changes is the schedule inside a tile:
//   * every LDS-fed GEMM requests its operand fragments a few steps ahead of the MFMAs that consume them
//     (swpipe, msst_dev.h).  The compiler's own schedule at 250 registers is read / s_waitcnt lgkmcnt(0) / MFMA:
//     one exposed LDS round trip per one to three MFMAs;
//   * nothing at the top of a tile waits for memory: the rows of the NEXT tile are requested during phase C2 and
//     normalised (LN1, packed to bf16) between the two phase-D GEMMs; its da rows are requested before the
//     d(LN1 out) GEMM and dropped / packed after the copy-out.  A tile starts with six LDS stores per thread.
//     (The template kernel waits vmcnt(0) there, which on gfx9 also waits for the previous tile's partial-row
//     stores to be acknowledged.)
//   * LN1 row statistics use DPP quad moves instead of ds_bpermute round trips;
//   * xd rows are 14 sixteen-byte slots apart (2 mod 4): conflict-free b128 and transposed fragment reads;
//   * token rows and per-thread LDS addresses are re-derived from laundered ids where they are used, so no
//     tile-start value stays live (= spilled) across the register-hungry phases.
// grid (nchunk, H): workgroup (chunk, h) walks the 64-row tiles chunk, chunk + nchunk, ... for ONE head and keeps
// the head's weight gradients (dWq | dWk | dWv [3][64][96], dWout_h [96][64]) in 96 registers per lane.
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"

#ifndef MSST_B2_PERM
#define MSST_B2_PERM 1   // k-permuted transposed fragment reads (rows 4 g + i and 16 + 4 g + i instead of 8 g + i, 8 g + 4 + i: the 32 lanes
                         // of a ds_read_b64_tr_b16 lane group then touch 8 consecutive rows).  12 % fewer bank-conflict cycles; measured
                         // +4 % time on the round-1 kernel before the saved-row rework, -1.6 % now (659 -> 648 us, three alternating builds)
#endif
#if MSST_B2_PERM
#define B2_LDKS ld_ks_perm
#define B2_LDKC ld_kc_perm
#else
#define B2_LDKS ld_ks
#define B2_LDKC ld_kc
#endif

#ifndef MSST_B2_PRIO
#define MSST_B2_PRIO 1   // s_setprio level of the MFMA-dense phases (A, C, D); the row-local VALU phases run at 0: -3 %
#endif
#if MSST_B2_PRIO
#define B2_PRIO(n) __builtin_amdgcn_s_setprio((n) ? MSST_B2_PRIO : 0)
#else
#define B2_PRIO(n) do { } while (0)
#endif
#ifndef MSST_B2_PRIOB
#define MSST_B2_PRIOB 0   // also raise the priority for the O / dO / dP MFMA bursts of phase B
#endif

namespace msst {

namespace {

typedef PBF16 P;
typedef bf16_t elem;
typedef s16x8 frag;

struct Bwd2Smem {
    static constexpr int LDX = 96 + 16;
    static constexpr int LDH = 64 + 8;
    elem xd[64][LDX];   // LN1(x) -> da -> LN1(x) -> staging of the d(LN1 out) rows
    // q | k | dO | ds are contiguous: dead after phase C, they receive the staged wqkvT fragments (36 KB)
    elem q[64][LDH];    // q[row][d]
    elem k[64][LDH];    // k[row][d]
    elem dO[64][LDH];   // dO[query][d]
    elem ds[64][LDH];   // ds[query][key]
    elem vt[64][LDH];   // vt[d][row]     -> later dv[row][d]
    elem p[64][LDH];    // p[query][key]  -> later dk[row][d]
    elem o[64][LDH];    // o[query][d]    -> later dq[row][d]
};

__device__ __forceinline__ int launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// LN1 of this thread's 24 features (4 threads per row), packed to bf16.  All-zero input (padding row) gives beta.
__device__ __forceinline__ void ln_pack(const f32x4 (&xv)[6], const float* lnp, int lpart, s16x4 (&out)[6]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    const float mean = quad_sum(s) * (1.f / 96.f);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; vs += d * d; }
    const float rstd = rsqrtf(quad_sum(vs) * (1.f / 96.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(lnp + lpart * 24 + 4 * i);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(lnp + 96 + lpart * 24 + 4 * i);
        f32x4 n4;
#pragma unroll
        for (int e = 0; e < 4; ++e) n4[e] = (xv[i][e] - mean) * rstd * gm[e] + bt[e];
        out[i] = f2bf4(n4);
    }
}

// da rows: dropout site 2 backward (same element groups as the forward) and bf16 packing
template <bool DROP>
__device__ __forceinline__ void da_pack(const f32x4 (&dav)[6], const Drop& drop, long tok, int lpart, s16x4 (&out)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 t4 = tok >= 0 ? dav[i] : zero4();
        if (DROP && tok >= 0) t4 = drop4(drop, 2, (unsigned)(tok * 24 + lpart * 6 + i), t4);
        out[i] = f2bf4(t4);
    }
}

}  // namespace

// DROP: dropout compiled in / out (uniform run-time tests would split the instruction stream into basic blocks)
// XN: the forward saved LN1(x) as bf16 rows (a.xn) and the MLP half left the dropped bf16 da rows (a.dab): both are loaded
// straight into the packed registers; x / da are never read, nothing is normalised, hashed or converted here
template <bool DROP, bool XN>
__global__ __launch_bounds__(256, 2) void block_bwd_attn_bf16_kernel(AttnBwdArgs a) {
    typedef Bwd2Smem SM;
    constexpr int LDX = SM::LDX, LDH = SM::LDH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const int H = a.H, inner = H * 64, h = blockIdx.y;
    const elem* wqkv = reinterpret_cast<const elem*>(a.w.wqkv);
    const elem* wqkvT = reinterpret_cast<const elem*>(a.w.wqkvT);
    const elem* woutT = reinterpret_cast<const elem*>(a.w.woutT);
    const TileMap tm = a.tm;
    const int L = tm.L;
    elem* part = reinterpret_cast<elem*>(a.dxn_part) + (long)h * a.ntok * 96;

    // persistent weight-grad accumulators: dWqkv: C[i = d in tile wave][j = m tile], for q, k, v;
    // dWout: C[i = m tile][j = d in tile wave]
    f32x4 gq[6], gk[6], gv[6], go[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { gq[i] = zero4(); gk[i] = zero4(); gv[i] = zero4(); go[i] = zero4(); }

    // (sequence slot, position) of this thread's row (tid / 4) in a tile, packed into one register; slot 0xffff = padding row
    const int2 sp_ln0 = tm.row_sp(tid >> 2);
    const unsigned sp_pack = ((unsigned)(sp_ln0.x & 0xffff) << 16) | (unsigned)sp_ln0.y;
    const int qlo = ((wave * 16 + c) / L) * L, qhi = qlo + L;
    constexpr int CPR = 12;   // 16-byte chunks per bf16 row of 96

#ifdef MSST_STAMPS
    const bool stamp_wg = (a.dbg & 8) && blockIdx.x == 7 && blockIdx.y == 3 && tid == 0;
#endif
    float* lnp = reinterpret_cast<float*>(smem_raw + sizeof(SM));   // LN1 gamma | beta (tile invariant)
    if (tid < 96) { lnp[tid] = a.w.ln1_g[tid]; lnp[96 + tid] = a.w.ln1_b[tid]; }
    __syncthreads();
    // token row of this thread (row tid / 4, features 24 (tid % 4) ..) in a tile; the sequence slot is tile invariant
    auto tok_of = [&](int tile_) -> long {
        const unsigned sp = (unsigned)launder((int)sp_pack);   // opaque: the result is recomputed at every use, never kept
        const int sx = (int)(sp >> 16), sy = (int)(sp & 0xffffu);
        const int q = tile_ * tm.TS + sx;
        if (tile_ >= a.ntiles || sx == 0xffff || q >= tm.nseq) return -1;
        if (tm.mode == 0) return (long)(q * tm.N + sy);
        const int b = tm.nshift >= 0 ? (q >> tm.nshift) : q / tm.N, n = q - b * tm.N;
        return (long)(b * tm.T + sy * tm.N + n);
    };
    auto load_rows = [&](const float* base, long tok, f32x4 (&dst)[6]) {
        // requested unconditionally from a clamped address (a branch per row serialises the round trips)
        const f32x4* src = reinterpret_cast<const f32x4*>(base + (tok >= 0 ? tok : 0) * 96 + (launder(tid) & 3) * 24);
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = src[i];
    };

    auto load_bf = [&](const void* base, long tok, s16x4 (&dst)[6]) {   // 48 bytes of the thread's bf16 row slice, clamped address
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const elem*>(base) + (tok >= 0 ? tok : 0) * 96 + (launder(tid) & 3) * 24);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const u32x4 v = src[i];
            dst[2 * i] = __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 0, 1));
            dst[2 * i + 1] = __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 2, 3));
        }
    };
    auto zero_xn = [&](s16x4 (&dst)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = s16x4{0, 0, 0, 0};
    };
    s16x4 xnk[6];   // LN1(x) rows of the tile to process, packed (stored at the top of the tile, again before phase D)
    s16x4 dak[6];   // its da rows, dropped and packed (stored after phase A)
    {
        const long tok0 = tok_of(blockIdx.x);
        f32x4 xv[6], dav[6];
        if constexpr (XN) load_bf(a.xn, tok0, xnk); else load_rows(a.x, tok0, xv);
        if constexpr (XN) load_bf(a.dab, tok0, dak); else load_rows(a.da, tok0, dav);
        if constexpr (XN) {
            if (tok0 < 0) { zero_xn(xnk); zero_xn(dak); }
        } else {
            if (tok0 < 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) xv[i] = zero4();
            }
            ln_pack(xv, lnp, tid & 3, xnk);
        }
        if constexpr (!XN) da_pack<DROP>(dav, a.drop, tok0, tid & 3, dak);
    }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef MSST_STAMPS
        const bool stamp_on = stamp_wg && tile == blockIdx.x + 20 * (int)gridDim.x;   // a mid-walk tile
#endif
        STAMP(0);
        const int rq = (0 * H + h) * 64 + wave * 16, rk = (1 * H + h) * 64 + wave * 16, rv = (2 * H + h) * 64 + wave * 16;
        frag wk[3][3];   // phase-A weight fragments [k-step][q | k | v]
        wk[0][0] = P::ld_w(wqkv, 96, rq, 0);
        wk[0][1] = P::ld_w(wqkv, 96, rk, 0);
        wk[0][2] = P::ld_w(wqkv, 96, rv, 0);
        {
            const int t0 = launder(tid);
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<s16x4*>(&sm.xd[t0 >> 2][(t0 & 3) * 24 + 4 * i]) = xnk[i];
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);
        B2_PRIO(1);
        // ---------------- phase A: q, k, v^T (wave <-> 16 head channels) ----------------
        {
            f32x4 cq[4], ck[4], cv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { cq[t] = zero4(); ck[t] = zero4(); cv[t] = zero4(); }
            // step s = (ks, t): LN1(x) fragment two steps ahead; the next k-step's weight fragments are requested at
            // the start of each k-step
            frag xs[3];
            swpipe<12, 2>(
                [&](int s) { xs[s % 3] = P::ld_kc(&sm.xd[(s & 3) * 16][(s >> 2) * 32], LDX); },
                [&](int s) {
                    const int ks = s >> 2, t = s & 3;
                    if (t == 0 && ks < 2) {
                        wk[ks + 1][0] = P::ld_w(wqkv, 96, rq, (ks + 1) * 32);
                        wk[ks + 1][1] = P::ld_w(wqkv, 96, rk, (ks + 1) * 32);
                        wk[ks + 1][2] = P::ld_w(wqkv, 96, rv, (ks + 1) * 32);
                    }
                    cq[t] = P::mma(wk[ks][0], xs[s % 3], cq[t]);
                    ck[t] = P::mma(wk[ks][1], xs[s % 3], ck[t]);
                    cv[t] = P::mma(xs[s % 3], wk[ks][2], cv[t]);
                });
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                P::st_nat(&sm.q[t * 16][wave * 16], LDH, cq[t]);
                P::st_nat(&sm.k[t * 16][wave * 16], LDH, ck[t]);
                P::st_nat(&sm.vt[wave * 16][t * 16], LDH, cv[t]);
            }
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);
        B2_PRIO(0);
        // ---------------- phase B: wave <-> 16 query rows ----------------
        f32x4 pr[4];          // raw probabilities (C layout [key][query])
        frag pb[2];           // dropped probabilities, packed as the B operand of key chunk m (permuted key order)
        frag wd[2][4];        // Wout_h^T fragments for the dO GEMM, two k-steps in flight
        unsigned keep1 = 0;
        {
            frag fq[2], fk[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[ks] = P::ld_kc(&sm.q[wave * 16][ks * 32], LDH);
#pragma unroll
                for (int t = 0; t < 4; ++t) fk[ks][t] = P::ld_kc(&sm.k[t * 16][ks * 32], LDH);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) wd[0][t] = P::ld_w(woutT, 96, h * 64 + t * 16, 0);
            MSST_SCHED_FENCE();
            {   // da rows -> xd (overwrites LN1(x); rows of this wave only)
                const int t1 = launder(tid);
#pragma unroll
                for (int i = 0; i < 6; ++i) *reinterpret_cast<s16x4*>(&sm.xd[t1 >> 2][(t1 & 3) * 24 + 4 * i]) = dak[i];
            }
            MSST_SCHED_FENCE();
#pragma unroll
            for (int t = 0; t < 4; ++t) pr[t] = zero4();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) pr[t] = P::mma(fk[ks][t], fq[ks], pr[t]);   // C[i = key][j = query]
        }
        // operands of o = P v (v^T fragments, permuted key order): first key chunk requested under the softmax
        frag fv[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fv[0][t] = P::ld_kc_perm(&sm.vt[t * 16][0], LDH);
        MSST_SCHED_FENCE();
        {
            // same arithmetic as block_fwd_hw_kernel: exp2(s c - max c), c = scale log2 e; nothing to mask when L == 64
            const float cs = a.scale * 1.44269504088896340736f;
            float mx = -INFINITY;
            if (L == 64) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, pr[t][r]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + 4 * g + r;
                        const float v = (key >= qlo && key < qhi) ? pr[t][r] : -INFINITY;
                        pr[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            mx = colgroup_max(mx);
            const float mc = mx * cs;
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(pr[t][r], cs, -mc)); pr[t][r] = e; sum += e; }
            sum = colgroup_sum(sum);
            const float inv = 1.f / sum;
            f32x4 pdr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pr[t] = pr[t] * inv;
                f32x4 pd = pr[t];   // site 1: O and dV see the dropped probabilities, the softmax backward the raw ones
                if (DROP) {
                    unsigned kb;
                    pd = drop4_keep(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c) * 16 + t * 4 + g), pd, kb);
                    keep1 |= kb << (4 * t);   // the 16 keep decisions of this lane, reused for dP below
                }
                P::st_nat(&sm.p[wave * 16][t * 16], LDH, pd);  // p[query][key]
                pdr[t] = pd;
            }
            pb[0] = P::pack2(pdr[0], pdr[1]);
            pb[1] = P::pack2(pdr[2], pdr[3]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) fv[1][t] = P::ld_kc_perm(&sm.vt[t * 16][32], LDH);
        MSST_SCHED_FENCE();
        STAMP(5);
        if (MSST_B2_PRIOB) B2_PRIO(1);
        frag dob[2];   // dO^T of this wave's queries, packed as the B operand of channel chunk m (permuted order)
        {
            // o = P v  (C[i = d][j = query]) and dO = Wout_h^T da (C[i = d][j = query])
            f32x4 dov[4];
            frag fd[3];   // own da rows
            {
                f32x4 o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = zero4();
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = P::mma(fv[m][t], pb[m], o[t]);
                MSST_SCHED_FENCE();
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) fd[ks] = P::ld_kc(&sm.xd[wave * 16][ks * 32], LDX);
#pragma unroll
                for (int t = 0; t < 4; ++t) wd[1][t] = P::ld_w(woutT, 96, h * 64 + t * 16, 32);
#pragma unroll
                for (int t = 0; t < 4; ++t) P::st_nat(&sm.o[wave * 16][t * 16], LDH, o[t]);      // o[query][d]
                MSST_SCHED_FENCE();
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) dov[t] = P::mma(wd[0][t], fd[0], zero4());
#pragma unroll
            for (int t = 0; t < 4; ++t) wd[0][t] = P::ld_w(woutT, 96, h * 64 + t * 16, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) dov[t] = P::mma(wd[1][t], fd[1], dov[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) dov[t] = P::mma(wd[0][t], fd[2], dov[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) P::st_nat(&sm.dO[wave * 16][t * 16], LDH, dov[t]);   // dO[query][d]
            dob[0] = P::pack2(dov[0], dov[1]);
            dob[1] = P::pack2(dov[2], dov[3]);
        }
        STAMP(6);
        {
            // dP^T[key][query] = sum_d v[key][d] dO[query][d]: A = v (k-strided read of vt), B = dO^T from registers
            f32x4 dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) dp[t] = zero4();
            frag fz[2][4];
            swpipe<2, 1>(
                [&](int s) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) fz[s][t] = P::ld_ks_perm(&sm.vt[s * 32][t * 16], LDH);
                },
                [&](int s) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dp[t] = P::mma(fz[s][t], dob[s], dp[t]);
                });
            if (MSST_B2_PRIOB) B2_PRIO(0);
            if (DROP) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dp[t] = drop4_bits(a.drop, keep1 >> (4 * t), dp[t]);
            }
            float delta = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) delta += pr[t][r] * dp[t][r];
            delta = colgroup_sum(delta);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 d4;
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = pr[t][r] * (dp[t][r] - delta) * a.scale;
                P::st_nat(&sm.ds[wave * 16][t * 16], LDH, d4);  // ds[query][key]
            }
        }
        STAMP(7);
        lds_barrier();
        STAMP(8);
        B2_PRIO(1);
        // XN: the NEXT tile's saved rows are requested here, four GEMM phases (~5 k cycles) before they are needed: a row
        // request takes 2-3 k cycles under load.  This tile's da rows went to xd at the head of phase B (dak is free); its
        // LN1 rows are still needed for the restore after C2, so the next ones land in a second register set
        s16x4 xnn[6];
        if constexpr (XN) {
            const long tokn = tok_of(tile + gridDim.x);
            load_bf(a.xn, tokn, xnn);
            load_bf(a.dab, tokn, dak);
        }
        // ---------------- phase C: contractions over all 64 queries / keys ----------------
        // C1: dWout_h and dv (reads xd = da, o, p, dO); dv -> vt (dead since phase B)
        {
            f32x4 dv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) dv[t] = zero4();
            // per k-step four steps: da^T tiles 0-2 (+ o^T), 3-5, dO^T tiles 0-1 (+ p^T), 2-3; operands two steps ahead
            //   dWout_h[m][d] += sum_q da[q][m] o[q][d]      C[i = m tile][j = d tile wave]
            //   dv[key][d]     = sum_q p[q][key] dO[q][d]     C[i = d tile][j = key tile wave]
            frag fo[2], fp[2], fz[3][3];
            swpipe<8, 2>(
                [&](int s) {
                    const int ks = s >> 2, j = s & 3;
                    if (j == 0) fo[ks] = P::B2_LDKS(&sm.o[ks * 32][wave * 16], LDH);
                    if (j == 2) fp[ks] = P::B2_LDKS(&sm.p[ks * 32][wave * 16], LDH);
                    if (j < 2) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) fz[s % 3][i] = P::B2_LDKS(&sm.xd[ks * 32][(3 * j + i) * 16], LDX);
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i) fz[s % 3][i] = P::B2_LDKS(&sm.dO[ks * 32][(2 * (j - 2) + i) * 16], LDH);
                    }
                },
                [&](int s) {
                    const int ks = s >> 2, j = s & 3;
                    if (j < 2) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) go[3 * j + i] = P::mma(fz[s % 3][i], fo[ks], go[3 * j + i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i) dv[2 * (j - 2) + i] = P::mma(fz[s % 3][i], fp[ks], dv[2 * (j - 2) + i]);
                    }
                });
#pragma unroll
            for (int t = 0; t < 4; ++t) P::st_nat(&sm.vt[wave * 16][t * 16], LDH, dv[t]);   // dv[key][d]
        }
        STAMP(9);
        lds_barrier();
        STAMP(10);
        // C2: dk -> p, dq -> o (both dead now); LN1(x) rows back into xd for phase D
        {
            f32x4 dq[4], dk[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { dq[t] = zero4(); dk[t] = zero4(); }
            // per k-step four steps: q^T tiles 0-1 (+ ds^T), 2-3, k^T tiles 0-1 (+ ds rows), 2-3; operands three steps ahead
            //   dk[key][d]   = sum_q ds[q][key] q[q][d]            C[i = d tile][j = key tile wave]
            //   dq[query][d] = sum_key ds[query][key] k[key][d]    C[i = d tile][j = query tile wave]
            frag fs[2], fr[2], fz[4][2];
            swpipe<8, 3>(
                [&](int s) {
                    const int ks = s >> 2, j = s & 3;
                    if (j == 0) fs[ks] = P::B2_LDKS(&sm.ds[ks * 32][wave * 16], LDH);
                    if (j == 2) fr[ks] = P::B2_LDKC(&sm.ds[wave * 16][ks * 32], LDH);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        fz[s % 4][i] = j < 2 ? P::B2_LDKS(&sm.q[ks * 32][(2 * j + i) * 16], LDH)
                                             : P::B2_LDKS(&sm.k[ks * 32][(2 * (j - 2) + i) * 16], LDH);
                },
                [&](int s) {
                    const int ks = s >> 2, j = s & 3;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (j < 2) dk[2 * j + i] = P::mma(fz[s % 4][i], fs[ks], dk[2 * j + i]);
                        else dq[2 * (j - 2) + i] = P::mma(fz[s % 4][i], fr[ks], dq[2 * (j - 2) + i]);
                    }
                });
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                P::st_nat(&sm.p[wave * 16][t * 16], LDH, dk[t]);   // dk[key][d]
                P::st_nat(&sm.o[wave * 16][t * 16], LDH, dq[t]);   // dq[query][d]
            }
        }
        {
            const int t2 = launder(tid);
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<s16x4*>(&sm.xd[t2 >> 2][(t2 & 3) * 24 + 4 * i]) = xnk[i];
        }
        STAMP(11);
        lds_barrier();
        STAMP(12);
        {
            // stage this head's 36 wqkvT fragments (all four waves need all of them in the d(LN1 out) GEMM) into the
            // dead q | k | dO | ds region with async global->LDS copies, 9 per wave
            char* stage = reinterpret_cast<char*>(&sm.q[0][0]);
            const int l16 = (launder(tid) & 63) * 16;
#pragma unroll
            for (int i9 = 0; i9 < 9; ++i9) {
                const int idx = wave * 9 + i9;                    // idx = (t * 3 + which) * 2 + ks
                const int t = idx / 6, which = (idx >> 1) % 3, ks = idx & 1;
                const int f = t * ((3 * inner) >> 5) + ((which * inner + h * 64 + ks * 32) >> 5);
                dma_frag_async_s(wqkvT + (long)f * 512, stage + idx * 1024, l16);
            }
        }
        // rows of the NEXT tile of this workgroup: requested now, normalised between the two phase-D GEMMs
        f32x4 xv[6];
        if constexpr (!XN) load_rows(a.x, tok_of(tile + gridDim.x), xv);
        // ---------------- phase D: qkv weight grads and the head's d(LN1 out) partial ----------------
        {
            // dW{q,k,v}[d][m] += sum_row d{q,k,v}[row][d] xn[row][m]    C[i = d tile wave][j = m tile t]
            // step s = (ks, t): three MFMAs on three accumulators; xn^T fragment requested three steps ahead
            frag fa[2][3], fx[4];
            swpipe<12, 3>(
                [&](int s) {
                    const int ks = s / 6, t = s % 6;
                    if (t == 0) {
                        fa[ks][0] = P::B2_LDKS(&sm.o[ks * 32][wave * 16], LDH);
                        fa[ks][1] = P::B2_LDKS(&sm.p[ks * 32][wave * 16], LDH);
                        fa[ks][2] = P::B2_LDKS(&sm.vt[ks * 32][wave * 16], LDH);
                    }
                    fx[s % 4] = P::B2_LDKS(&sm.xd[ks * 32][t * 16], LDX);
                },
                [&](int s) {
                    const int ks = s / 6, t = s % 6;
                    gq[t] = P::mma(fa[ks][0], fx[s % 4], gq[t]);
                    gk[t] = P::mma(fa[ks][1], fx[s % 4], gk[t]);
                    gv[t] = P::mma(fa[ks][2], fx[s % 4], gv[t]);
                });
        }
        STAMP(13);
        // this wave's staged fragments have landed
        wait_vm0();
        lds_barrier();
        STAMP(14);    // every wave is done reading xd (weight-grad GEMM above); staged weights visible
        f32x4 dav[6];   // da rows of the next tile: requested under the d(LN1 out) GEMM, packed after the copy-out
        if constexpr (!XN) load_rows(a.da, tok_of(tile + gridDim.x), dav);
        {
            // dxn_h[row][m] = sum_d dq Wq + dk Wk + dv Wv         C[i = m tile][j = row tile wave]
            f32x4 dx[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) dx[t] = zero4();
            // step s = (ks, which, half): three MFMAs on three accumulators (m tiles 3 half ..); staged weight fragments
            // requested two steps ahead
            const char* stage = reinterpret_cast<const char*>(&sm.q[0][0]) + (launder(tid) & 63) * 16;
            frag fb[2][3], fw[3][3];
            swpipe<12, 2>(
                [&](int s) {
                    const int ks = s / 6, which = (s % 6) >> 1, half = s & 1;
                    if (s % 6 == 0) {
                        fb[ks][0] = P::ld_kc(&sm.o[wave * 16][ks * 32], LDH);
                        fb[ks][1] = P::ld_kc(&sm.p[wave * 16][ks * 32], LDH);
                        fb[ks][2] = P::ld_kc(&sm.vt[wave * 16][ks * 32], LDH);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        fw[s % 3][j] = *reinterpret_cast<const frag*>(stage + (((3 * half + j) * 3 + which) * 2 + ks) * 1024);
                },
                [&](int s) {
                    const int ks = s / 6, which = (s % 6) >> 1, half = s & 1;
#pragma unroll
                    for (int j = 0; j < 3; ++j) dx[3 * half + j] = P::mma(fw[s % 3][j], fb[ks][which], dx[3 * half + j]);
                });
            STAMP(15);
            if constexpr (XN) {
                // pin the arrival of the next tile's saved rows HERE, before the copy-out stores are issued: a wait placed after
                // them (or left to the store at the top of the next iteration, across the loop edge) becomes vmcnt(0) and
                // sits out the acknowledgement of those stores -- 1.3 k cycles per tile
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(xnn[i]));
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(dak[i]));
#pragma unroll
                for (int i = 0; i < 6; ++i) xnk[i] = xnn[i];   // this tile's LN1 rows were last used by the restore after C2
                if (tok_of(tile + gridDim.x) < 0) { zero_xn(xnk); zero_xn(dak); }   // padding rows must carry zero da
            }
            // stage the [16 x 96] result rows of this wave in xd (dead now) and write whole rows
#pragma unroll
            for (int t = 0; t < 6; ++t) P::st_nat(&sm.xd[wave * 16][t * 16], LDX, dx[t]);
        }
        B2_PRIO(0);
        __builtin_amdgcn_wave_barrier();
        {
            const int t3 = launder(tid);
            const int lr3 = t3 >> 2, lpart3 = t3 & 3;
            const long tok_out = tok_of(tile);
            f32x4 v[CPR / 4];
#pragma unroll
            for (int i = 0; i < CPR / 4; ++i)
                v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(&sm.xd[lr3][0]) + (lpart3 * (CPR / 4) + i) * 16);
            if (tok_out >= 0) {
#pragma unroll
                for (int i = 0; i < CPR / 4; ++i)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(part + tok_out * 96) + (lpart3 * (CPR / 4) + i) * 16) = v[i];
            }
        }
        // LN1 of the next tile's rows (requested before the weight-grad GEMM); padding rows normalise zeros
        if constexpr (XN) {
            // (the saved rows were pinned before the copy-out, see above)
        } else {
            if (tok_of(tile + gridDim.x) < 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) xv[i] = zero4();
            }
            ln_pack(xv, lnp, launder(tid) & 3, xnk);
        }
        if constexpr (!XN) da_pack<DROP>(dav, a.drop, tok_of(tile + gridDim.x), launder(tid) & 3, dak);
        STAMP(16);
        // no block barrier here: the next tile's first stores only touch this wave's own xd rows, and q / k / vt
        // are not written before the barrier that follows them
    }

    // ---------------- slab: [dWq | dWk | dWv] [3][64][96], dWout_h [96][64] ----------------
    float* slab = a.slab + ((long)blockIdx.x * H + h) * MSST_ATTN_SLAB_N;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = wave * 16 + 4 * g + r, m = t * 16 + c;
            slab[0 * 6144 + d * 96 + m] = gq[t][r];
            slab[1 * 6144 + d * 96 + m] = gk[t][r];
            slab[2 * 6144 + d * 96 + m] = gv[t][r];
            slab[3 * 6144 + (t * 16 + 4 * g + r) * 64 + wave * 16 + c] = go[t][r];  // dWout_h[m][d]
        }
}

int launch_block_bwd_attn_bf16(const AttnBwdArgs& a, int nchunk, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (a.tm.L > 64 || a.tm.L < 1) return MSST_ERR_UNSUPPORTED;
    const size_t smem = sizeof(Bwd2Smem) + 192 * sizeof(float);
    typedef void (*kern_t)(AttnBwdArgs);
    const kern_t kerns[4] = {&block_bwd_attn_bf16_kernel<false, false>, &block_bwd_attn_bf16_kernel<true, false>,
                             &block_bwd_attn_bf16_kernel<false, true>, &block_bwd_attn_bf16_kernel<true, true>};
    if (!attr_set) {
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    ProfScope ps(K_BWD_ATTN, st);
    hipLaunchKernelGGL(kerns[(a.drop.thr ? 1 : 0) + ((a.xn && a.dab) ? 2 : 0)], dim3(nchunk, a.H), dim3(256), smem, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
