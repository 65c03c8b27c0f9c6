// Forward kernels of the MaskedSST masked-pretraining hot path for gfx950 (MI355X).
//
//   tokenize_fwd   a1+a2+a3+a5  cube tile -> LN(P) -> per-block Linear(P->96) -> LN(96) -> +pos -> mask select
//   block_fwd      a7-a10       fused pre-norm transformer block on a 64-row tile of whole sequences
//                               (spatial and strided-spectral variants share the kernel via TileMap)
//   head_fwd       a12-a14      gather masked tokens -> per-spectral-block Linear(96->P) -> masked L1
//
// Reference semantics (file:line under the reference repo) are cited at each kernel.
#include <atomic>
#include "msst_dev.h"
#include "msst_kernels.h"

namespace msst {

// ==========================================================================================
// tokenizer: reference vit_spatial_spectral.py:197-222 (to_patch + embed),
// vit_simmim_original.py:236-249,285 (pos add, mask-token select).
// grid (S, B), 256 threads: 4 threads per spatial token, 24 output features each.
// img [B][C][N] (N = H*W, patch 1x1), spectral block c holds bands c*P .. c*P+P-1: the tile
// img[b, cP:(c+1)P, :] is P*N contiguous floats -> coalesced load into LDS.
// ==========================================================================================
// PC: pixels per patch as a compile-time constant (10 = the reference's spectral patch, configs/config.yaml: band_patch_size)
// so that the small loops over it unroll and their LDS reads are batched; 0 = run-time value (any P <= 16)
template <int PC>
__global__ __launch_bounds__(256) void tokenize_fwd_kernel(TokArgs a) {
    __shared__ float patch[16][64];
    __shared__ float W[96][17];
    __shared__ float bias[96];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int P = PC ? PC : a.P, N = a.N;
    const float* src = a.img + ((long)b * a.S + c) * P * N;
    for (int i = tid; i < P * N; i += 256) patch[i / N][i % N] = src[i];
    for (int i = tid; i < 96 * P; i += 256) W[i / P][i % P] = a.w_emb[(long)c * 96 * P + i];
    if (tid < 96) bias[tid] = a.b_emb[c * 96 + tid];
    __syncthreads();
    const int n = tid >> 2, part = tid & 3;
    if (n >= N) return;
    // LN over the P raw pixel values (pre_norm, eps 1e-5)
    float xn[16];
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) mean += patch[k][n];
    mean /= P;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) { const float d = patch[k][n] - mean; var += d * d; }
    const float rstd = rsqrtf(var / P + 1e-5f);
#pragma unroll
    for (int k = 0; k < P; ++k) xn[k] = (patch[k][n] - mean) * rstd * a.pre_g[k] + a.pre_b[k];
    // per-block Linear(P -> 96): this thread's 24 features 16 (i / 4) + 4 part + i % 4 -- the four threads of a token then store 64
    // contiguous bytes per instruction (24 consecutive features per thread made every store instruction hit four 16-byte pieces
    // 96 bytes apart per token)
    auto feat = [&](int i) { return 16 * (i >> 2) + 4 * part + (i & 3); };
    float e[24];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int d = feat(i);
        float acc = bias[d];
#pragma unroll
        for (int k = 0; k < P; ++k) acc += W[d][k] * xn[k];
        e[i] = acc;
        s += acc;
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float m2 = s * (1.f / 96.f);
    float v2 = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { const float d = e[i] - m2; v2 += d * d; }
    v2 += __shfl_xor(v2, 1); v2 += __shfl_xor(v2, 2);
    const float rstd2 = rsqrtf(v2 * (1.f / 96.f) + 1e-5f);
    const int t = c * N + n;
    const bool masked = a.mask[(long)b * a.T + t] != 0;
    float* dst = a.out + ((long)b * a.T + t) * 96 + part * 4;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int d = feat(i);
        float pos;
        if (a.pos_split) pos = d < a.pos_split ? a.pos_a[n * a.pos_split + d] : a.pos_b[c * (96 - a.pos_split) + d - a.pos_split];
        else pos = a.pos_a[(long)t * 96 + d];
        const float tok = (e[i] - m2) * rstd2 * a.post_g[d] + a.post_b[d];
        e[i] = (masked ? a.mask_token[d] : tok) + pos;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 v = {e[4*i], e[4*i+1], e[4*i+2], e[4*i+3]};
        if (a.drop.thr) v = drop4(a.drop, 0, (unsigned)(((long)b * a.T + t) * 24 + 4 * i + part), v);   // emb dropout (group = feature / 4)
        *reinterpret_cast<f32x4*>(dst + 16 * i) = v;
    }
}


// ------------------------------------------------------------------------------------------
// The same tokenizer for the reference's shapes (P = 10 pixels per patch, N = 64 spatial tokens) on the fp32 matrix cores
// (round 4).  The kernel above reads its [96][P] weight from LDS once per multiply-add (240 four-byte LDS reads per thread and
// token) and re-stages that weight for every sample: 100 us for 141 MB.  Here the per-block Linear(P -> 96) of 16 tokens is
// 18 v_mfma_f32_16x16x4_f32 (exact fp32, the fmaf chain of the reference's addmm): weights as the A operand -- 18 registers
// per lane, loaded once per workgroup and kept for its whole walk over the batch -- the LN(10)-normalised pixels as the B
// operand straight from the lane that loaded them (lane (j, g) holds pixels 4 ks + g of token j: exactly the 16x16x4 B
// layout), and the C layout (4 consecutive features of one token per lane) is the 16-byte store of the token row.
// grid (S, nchunk), 256 threads: wave w <-> tokens 16 w .. + 15 of spectral block c, samples chunk, chunk + nchunk, ...
// Position rows, bias, both LayerNorms' vectors and the mask token are tile invariant for a wave: registers / LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void tokenize_fwd_mfma_kernel(TokArgs a) {
    constexpr int P = 10, N = 64;
    __shared__ __attribute__((aligned(16))) float vec[3][96];   // post_g | post_b | mask_token
    const int c = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, j = l & 15;
    if (tid < 96) { vec[0][tid] = a.post_g[tid]; vec[1][tid] = a.post_b[tid]; vec[2][tid] = a.mask_token[tid]; }
    // A fragments of W_c [96][10]: lane (i = l & 15, kq = l >> 4) holds W[16 mt + i][4 ks + kq] (zero beyond k = 9)
    float wf[6][3];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int k = 4 * ks + g;
            wf[mt][ks] = k < P ? a.w_emb[((long)c * 96 + 16 * mt + j) * P + k] : 0.f;
        }
    float pg[3], pb[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; pg[ks] = k < P ? a.pre_g[k] : 0.f; pb[ks] = k < P ? a.pre_b[k] : 0.f; }
    // C layout: lane (token j, g) holds features 16 mt + 4 g + r
    const int n = 16 * w + j, t = c * N + n;
    f32x4 bias4[6], pos4[6];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) {
        const int d0 = 16 * mt + 4 * g;
        bias4[mt] = *reinterpret_cast<const f32x4*>(a.b_emb + c * 96 + d0);
        if (a.pos_split) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = d0 + r;
                pos4[mt][r] = d < a.pos_split ? a.pos_a[n * a.pos_split + d] : a.pos_b[c * (96 - a.pos_split) + d - a.pos_split];
            }
        } else {
            pos4[mt] = *reinterpret_cast<const f32x4*>(a.pos_a + (long)t * 96 + d0);
        }
    }
    __syncthreads();
    const int nb = (int)gridDim.y;
    float px[3];
    unsigned char mk;
    auto request = [&](int b) {
        const int bc = b < a.B ? b : a.B - 1;
        const float* src = a.img + ((long)bc * a.S + c) * P * N + n;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; px[ks] = src[(k < P ? k : 0) * N]; }
        mk = a.mask[(long)bc * a.T + t];
    };
    request(blockIdx.y);
    for (int b = blockIdx.y; b < a.B; b += nb) {
        float x[3];
        const bool masked = mk != 0;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) x[ks] = (4 * ks + g < P) ? px[ks] : 0.f;
        request(b + nb);
        // LN over the 10 pixels of the token (pre_norm, eps 1e-5): 3 (2) per lane, summed over the four lanes l, l ^ 16, l ^ 32, l ^ 48
        const float mean = colgroup_sum(x[0] + x[1] + x[2]) / P;
        float var = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const float d = (4 * ks + g < P) ? x[ks] - mean : 0.f; var += d * d; }
        const float rstd = rsqrtf(colgroup_sum(var) / P + 1e-5f);
        float xn[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) xn[ks] = (4 * ks + g < P) ? (x[ks] - mean) * rstd * pg[ks] + pb[ks] : 0.f;
        // per-block Linear(10 -> 96) + bias: C[i = feature][j = token]
        f32x4 e[6];
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            f32x4 acc = bias4[mt];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[mt][ks], xn[ks], acc, 0, 0, 0);
            e[mt] = acc;
            s += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        const float m2 = colgroup_sum(s) * (1.f / 96.f);
        float v2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = e[mt][r] - m2; v2 += d * d; }
        const float rstd2 = rsqrtf(colgroup_sum(v2) * (1.f / 96.f) + 1e-5f);
        float* dst = a.out + ((long)b * a.T + t) * 96 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int d0 = 16 * mt + 4 * g;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(&vec[0][d0]), b4 = *reinterpret_cast<const f32x4*>(&vec[1][d0]),
                        m4 = *reinterpret_cast<const f32x4*>(&vec[2][d0]);
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (masked ? m4[r] : (e[mt][r] - m2) * rstd2 * g4[r] + b4[r]) + pos4[mt][r];
            if (a.drop.thr) v = drop4(a.drop, 0, (unsigned)(((long)b * a.T + t) * 24 + 4 * mt + g), v);   // emb dropout (group = feature / 4)
            *reinterpret_cast<f32x4*>(dst + 16 * mt) = v;
        }
    }
}

// ==========================================================================================
// fused transformer block, forward.  Reference vit_spatial_spectral.py:22-29 (PreNorm),
// :47-78 (Attention: bias-free qkv, q|k|v chunks, head-major (h d), softmax(q k^T * dh^-0.5) v,
// out-projection with bias), :32-44 (FeedForward, exact-erf GELU), :100-104 (residuals, no
// final norm).  Dropout sites are identity (p = 0 / eval).
//
// One workgroup (4 waves) owns a 64-row tile of whole sequences; x stays on chip for the whole
// block.  Per head:  phase A (wave <-> 16 of the 64 head channels): q,k,v^T = LN1(x) W^T into LDS;
// phase B (wave <-> 16 query rows): S^T = k q^T, masked softmax over keys in registers, P -> LDS,
// O = P v, out-projection accumulated in registers across heads.  Then residual, LN2, MLP, residual.
// ==========================================================================================
template <class P>
struct FwdSmem {
    typedef typename P::elem elem;
    static constexpr int LDX = 96 + P::PADE;
    static constexpr int LDH = 64 + P::PADE;
    elem xn[64][LDX];
    elem q[64][LDH];
    elem k[64][LDH];
    elem vt[64][LDH];
    elem p[64][LDH];
};

template <class P>
__global__ __launch_bounds__(256, P::WAVES_PER_SIMD) void block_fwd_kernel(BlockArgs a) {
    typedef typename P::elem elem;
    typedef typename P::frag frag;
    typedef FwdSmem<P> SM;
    constexpr int KS = P::KS, LDX = SM::LDX, LDH = SM::LDH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const int H = a.H, inner = H * 64;
    const elem* wqkv = reinterpret_cast<const elem*>(a.w.wqkv);
    const elem* wout = reinterpret_cast<const elem*>(a.w.wout);
    const elem* w1 = reinterpret_cast<const elem*>(a.w.w1);
    const elem* w2 = reinterpret_cast<const elem*>(a.w.w2);
    const TileMap tm = a.tm;
    const int L = tm.L;
    const int2 sp_ln = tm.row_sp(tid >> 2);          // row handled in the LN1 phase
    const int2 sp_ep = tm.row_sp(wave * 16 + c);     // row handled in the epilogue
    const int qlo = ((wave * 16 + c) / L) * L, qhi = qlo + L;   // keys of the query row's own sequence

#ifdef MSST_STAMPS
    const bool stamp_on = (a.dbg & 8) && blockIdx.x == (unsigned)(a.ntiles / 2) && tid == 0;
#endif

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        STAMP(0);
        // ---------------- LN1 (4 threads per row, 24 features each) ----------------
        {
            const int r = tid >> 2, part = tid & 3;
            const long tok = tm.token_sp(tile, sp_ln);
            float v[24];
            if (tok >= 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(a.x + tok * 96 + part * 24);
#pragma unroll
                for (int i = 0; i < 6; ++i) { f32x4 t4 = src[i]; v[4*i] = t4[0]; v[4*i+1] = t4[1]; v[4*i+2] = t4[2]; v[4*i+3] = t4[3]; }
            } else {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = 0.f;
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) s += v[i];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
            const float mean = s * (1.f / 96.f);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) { const float d = v[i] - mean; vs += d * d; }
            vs += __shfl_xor(vs, 1); vs += __shfl_xor(vs, 2);
            const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int d = part * 24 + i;
                sm.xn[r][d] = P::cvt((v[i] - mean) * rstd * a.w.ln1_g[d] + a.w.ln1_b[d]);
            }
        }
        STAMP(1);
        __syncthreads();
        STAMP(2);

        f32x4 oacc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) oacc[i] = zero4();

        for (int h = 0; h < H; ++h) {
            STAMP(3 + h * 8);
            // ---------------- phase A: q, k, v^T for head h ----------------
            {
                f32x4 cq[4], ck[4], cv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { cq[t] = zero4(); ck[t] = zero4(); cv[t] = zero4(); }
                const int hw = (a.dbg & 1) ? 0 : h;   // ablation: all heads read the same (L1-resident) weights
                const int rq = (0 * H + hw) * 64 + wave * 16, rk = (1 * H + hw) * 64 + wave * 16, rv = (2 * H + hw) * 64 + wave * 16;
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 96; k0 += KS) {
                    const frag aq = P::ld_w(wqkv, 96, rq, k0);
                    const frag ak = P::ld_w(wqkv, 96, rk, k0);
                    const frag av = P::ld_w(wqkv, 96, rv, k0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const frag xb = P::ld_kc(&sm.xn[t * 16][k0], LDX);
                        cq[t] = P::mma(aq, xb, cq[t]);  // C[i = d][j = row]
                        ck[t] = P::mma(ak, xb, ck[t]);
                        cv[t] = P::mma(xb, av, cv[t]);  // C[i = row][j = d]
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    P::st_nat(&sm.q[t * 16][wave * 16], LDH, cq[t]);   // q[row][d]
                    P::st_nat(&sm.k[t * 16][wave * 16], LDH, ck[t]);   // k[row][d]
                    P::st_nat(&sm.vt[wave * 16][t * 16], LDH, cv[t]);  // vt[d][row]
                }
            }
            STAMP(4 + h * 8);
            if (!(a.dbg & 2)) __syncthreads();
            STAMP(5 + h * 8);
            // ---------------- phase B: attention for query rows wave*16 .. +15 ----------------
            {
                f32x4 s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) s[t] = zero4();
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 64; k0 += KS) {
                    const frag qb = P::ld_kc(&sm.q[wave * 16][k0], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) s[t] = P::mma(P::ld_kc(&sm.k[t * 16][k0], LDH), qb, s[t]);  // C[i = key][j = query]
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + 4 * g + r;
                        const float v = (key >= qlo && key < qhi) ? s[t][r] * a.scale : -INFINITY;
                        s[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = colgroup_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = (a.dbg & 4) ? (s[t][r] > -1e30f ? 1.f : 0.f) : P::exp(s[t][r] - mx);  // exp(-inf) = 0 for masked keys
                        s[t][r] = e;
                        sum += e;
                    }
                sum = colgroup_sum(sum);
                const float inv = 1.f / sum;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 pv = s[t] * inv;
                    if (a.drop.thr)   // site 1: group = ((tile*H + h)*64 + query row)*16 + key/4
                        pv = drop4(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c) * 16 + t * 4 + g), pv);
                    P::st_nat(&sm.p[wave * 16][t * 16], LDH, pv);  // p[query][key]
                }
                STAMP(6 + h * 8);
                __builtin_amdgcn_wave_barrier();
                f32x4 o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = zero4();
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 64; k0 += KS) {
                    const frag pb = P::ld_kc(&sm.p[wave * 16][k0], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = P::mma(P::ld_kc(&sm.vt[t * 16][k0], LDH), pb, o[t]);  // C[i = d][j = query]
                }
                // O over this wave's (now dead) q rows: o[query][d]
#pragma unroll
                for (int t = 0; t < 4; ++t) P::st_nat(&sm.q[wave * 16][t * 16], LDH, o[t]);
                STAMP(7 + h * 8);
                __builtin_amdgcn_wave_barrier();
                // out-projection, accumulated over heads: C[i = m][j = query], A = Wout[m][h*64 + d]
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 64; k0 += KS) {
                    const frag ob = P::ld_kc(&sm.q[wave * 16][k0], LDH);
#pragma unroll
                    for (int mt = 0; mt < 6; ++mt)
                        oacc[mt] = P::mma(P::ld_w(wout, inner, mt * 16, ((a.dbg & 1) ? 0 : h * 64) + k0), ob, oacc[mt]);
                }
            }
            STAMP(8 + h * 8);
            if (!(a.dbg & 2)) __syncthreads();
            STAMP(9 + h * 8);
        }
        STAMP(3 + H * 8);

        // ---------------- residual + LN2 + MLP + residual (wave owns 16 rows) ----------------
        const long tok = tm.token_sp(tile, sp_ep);
        float x1[6][4];
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 xr = zero4();
            if (tok >= 0) xr = *reinterpret_cast<const f32x4*>(a.x + tok * 96 + m0);
            f32x4 av;
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = oacc[mt][r] + a.w.bo[m0 + r];
            if (a.drop.thr && tok >= 0) av = drop4(a.drop, 2, (unsigned)(tok * 24 + (m0 >> 2)), av);   // site 2
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x1[mt][r] = av[r] + xr[r];
                s1 += x1[mt][r];
            }
            if (a.x1 && tok >= 0) {
                f32x4 o4 = {x1[mt][0], x1[mt][1], x1[mt][2], x1[mt][3]};
                *reinterpret_cast<f32x4*>(a.x1 + tok * 96 + m0) = o4;
            }
        }
        s1 = colgroup_sum(s1);
        const float mean = s1 * (1.f / 96.f);
        float vs = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x1[mt][r] - mean; vs += d * d; }
        vs = colgroup_sum(vs);
        const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 n4;
#pragma unroll
            for (int r = 0; r < 4; ++r) n4[r] = (x1[mt][r] - mean) * rstd * a.w.ln2_g[m0 + r] + a.w.ln2_b[m0 + r];
            P::st_nat(&sm.xn[wave * 16][mt * 16], LDX, n4);  // xn2[row][m] (wave-private rows)
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 hh[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hh[nt] = zero4();
#pragma unroll P::UNROLL
        for (int k0 = 0; k0 < 96; k0 += KS) {
            const frag xb = P::ld_kc(&sm.xn[wave * 16][k0], LDX);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) hh[nt] = P::mma(P::ld_w(w1, 96, nt * 16, k0), xb, hh[nt]);  // C[i = n][j = row]
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = nt * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[nt][r] = P::gelu(hh[nt][r] + a.w.b1[n0 + r]);
            if (a.drop.thr && tok >= 0) hh[nt] = drop4(a.drop, 3, (unsigned)(tok * 16 + (n0 >> 2)), hh[nt]);   // site 3
            P::st_nat(&sm.p[wave * 16][nt * 16], LDH, hh[nt]);  // h[row][n]
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 yy[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) yy[mt] = zero4();
#pragma unroll P::UNROLL
        for (int k0 = 0; k0 < 64; k0 += KS) {
            const frag hb = P::ld_kc(&sm.p[wave * 16][k0], LDH);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) yy[mt] = P::mma(P::ld_w(w2, 64, mt * 16, k0), hb, yy[mt]);  // C[i = m][j = row]
        }
        if (tok >= 0) {
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const int m0 = mt * 16 + 4 * g;
                f32x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = yy[mt][r] + a.w.b2[m0 + r];
                if (a.drop.thr) o4 = drop4(a.drop, 4, (unsigned)(tok * 24 + (m0 >> 2)), o4);   // site 4
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] += x1[mt][r];
                *reinterpret_cast<f32x4*>(a.y + tok * 96 + m0) = o4;
            }
        }
        STAMP(4 + H * 8);
        __syncthreads();
        STAMP(5 + H * 8);
    }
}


// ==========================================================================================
// bf16 throughput variant of block_fwd: same math and LDS layout as block_fwd_kernel<PBF16>, but
// (1) every global operand of a phase is fetched into registers one phase AHEAD (the measured
// global-load round trip is ~2k cycles under load: per-head weights for phase A are requested when
// the previous head's phase A retires them, the out-projection slice at the start of phase B, the
// MLP weights at the start of the epilogue, the next tile's rows during the current tile), and
// (2) the grid is persistent (2 workgroups per CU walk the tiles).
// ==========================================================================================
__global__ __launch_bounds__(256, 2) void block_fwd_bf16_kernel(BlockArgs a) {
    typedef PBF16 P;
    typedef bf16_t elem;
    typedef s16x8 frag;
    typedef FwdSmem<P> SM;
    constexpr int LDX = SM::LDX, LDH = SM::LDH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const int H = a.H, inner = H * 64;
    const elem* wqkv = reinterpret_cast<const elem*>(a.w.wqkv);
    const elem* wout = reinterpret_cast<const elem*>(a.w.wout);
    const elem* w1 = reinterpret_cast<const elem*>(a.w.w1);
    const elem* w2 = reinterpret_cast<const elem*>(a.w.w2);
    const TileMap tm = a.tm;
    const int L = tm.L;
    const int2 sp_ln = tm.row_sp(tid >> 2);
    const int2 sp_ep = tm.row_sp(wave * 16 + c);
    const int qlo = ((wave * 16 + c) / L) * L, qhi = qlo + L;
    const int part = tid & 3, lr = tid >> 2;

    // LN1 gamma/beta live in LDS (tile invariant; keeps 48 registers free)
    // small parameter vectors live in LDS (tile invariant; no dependent global loads in the tile loop)
    float* lnp = reinterpret_cast<float*>(smem_raw + sizeof(SM));   // ln1_g | ln1_b | bo | ln2_g | ln2_b | b2 | b1
    if (tid < 96) {
        lnp[tid] = a.w.ln1_g[tid]; lnp[96 + tid] = a.w.ln1_b[tid]; lnp[192 + tid] = a.w.bo[tid];
        lnp[288 + tid] = a.w.ln2_g[tid]; lnp[384 + tid] = a.w.ln2_b[tid]; lnp[480 + tid] = a.w.b2[tid];
        if (tid < 64) lnp[576 + tid] = a.w.b1[tid];
    }
    // MLP weights (tile invariant, 24 fragment-packed KB) stay in LDS for the life of the workgroup
    char* wmlp = smem_raw + sizeof(SM) + 640 * sizeof(float);   // [w1: 12 frags | w2: 12 frags]
#pragma unroll
    for (int i6 = 0; i6 < 6; ++i6) {
        const int f = wave * 6 + i6;
        dma_frag(f < 12 ? reinterpret_cast<const char*>(w1) + f * 1024 : reinterpret_cast<const char*>(w2) + (f - 12) * 1024,
                 wmlp + f * 1024);
    }
    wait_vm0();
    __syncthreads();

    // rows of the first tile
    f32x4 xv[6];
    {
        const long tok = tm.token_sp(blockIdx.x, sp_ln);
#pragma unroll
        for (int i = 0; i < 6; ++i) xv[i] = tok >= 0 ? reinterpret_cast<const f32x4*>(a.x + tok * 96 + part * 24)[i] : zero4();
    }

#ifdef MSST_STAMPS
    const bool stamp_on = (a.dbg & 8) && blockIdx.x == 100 && tid == 0;
#endif
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        STAMP(0);
        // phase-A weights of head 0 (q, k, v rows of this wave's 16 channels; 3 k-steps each)
        frag wa[3][3];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                wa[m][ks] = P::ld_w(wqkv, 96, (m * H + 0) * 64 + wave * 16, ks * 32);
        // ---------------- LN1 from the prefetched rows ----------------
        {
            const long tok_ln1 = a.xn_out ? tm.token_sp(tile, sp_ln) : -1;
            float v[24];
#pragma unroll
            for (int i = 0; i < 6; ++i) { v[4*i] = xv[i][0]; v[4*i+1] = xv[i][1]; v[4*i+2] = xv[i][2]; v[4*i+3] = xv[i][3]; }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) s += v[i];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
            const float mean = s * (1.f / 96.f);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) { const float d = v[i] - mean; vs += d * d; }
            vs += __shfl_xor(vs, 1); vs += __shfl_xor(vs, 2);
            const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x4 n4;
#pragma unroll
                for (int e = 0; e < 4; ++e) n4[e] = (v[4*i+e] - mean) * rstd * lnp[part * 24 + 4*i+e] + lnp[96 + part * 24 + 4*i+e];
                const s16x4 nb = f2bf4(n4);
                *reinterpret_cast<s16x4*>(&sm.xn[lr][part * 24 + 4 * i]) = nb;
                // the same bf16 rows go to HBM for the tuned attention backward kernels (as in block_fwd_hw_kernel)
                if (a.xn_out && tok_ln1 >= 0) *reinterpret_cast<s16x4*>(reinterpret_cast<bf16_t*>(a.xn_out) + tok_ln1 * 96 + part * 24 + 4 * i) = nb;
            }
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);

        f32x4 oacc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) oacc[i] = zero4();
        const long tok = tm.token_sp(tile, sp_ep);
        f32x4 xres[6];   // residual rows in C layout, requested during the last head

        for (int h = 0; h < H; ++h) {
            STAMP(3 + h * 8);
            // ---------------- phase A ----------------
            {
                f32x4 cq[4], ck[4], cv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { cq[t] = zero4(); ck[t] = zero4(); cv[t] = zero4(); }
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const frag xb = P::ld_kc(&sm.xn[t * 16][ks * 32], LDX);
                        cq[t] = P::mma(wa[0][ks], xb, cq[t]);
                        ck[t] = P::mma(wa[1][ks], xb, ck[t]);
                        cv[t] = P::mma(xb, wa[2][ks], cv[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    P::st_nat(&sm.q[t * 16][wave * 16], LDH, cq[t]);
                    P::st_nat(&sm.k[t * 16][wave * 16], LDH, ck[t]);
                    P::st_nat(&sm.vt[wave * 16][t * 16], LDH, cv[t]);
                }
            }
            STAMP(4 + h * 8);
            // requests for later phases: this head's out-projection slice, the next head's phase A
            frag wo[6][2];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wo[mt][ks] = P::ld_w(wout, inner, mt * 16, h * 64 + ks * 32);
            if (h + 1 < H) {
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks)
                        wa[m][ks] = P::ld_w(wqkv, 96, (m * H + h + 1) * 64 + wave * 16, ks * 32);
            } else {
                // last head: wa is free -> request the residual rows (C layout) and the NEXT tile's rows
#pragma unroll
                for (int mt = 0; mt < 6; ++mt)
                    xres[mt] = tok >= 0 ? *reinterpret_cast<const f32x4*>(a.x + tok * 96 + mt * 16 + 4 * g) : zero4();
                const int nt = tile + gridDim.x;
                const long tokn = nt < a.ntiles ? tm.token_sp(nt, sp_ln) : -1;
#pragma unroll
                for (int i = 0; i < 6; ++i) xv[i] = tokn >= 0 ? reinterpret_cast<const f32x4*>(a.x + tokn * 96 + part * 24)[i] : zero4();
            }
            lds_barrier();
            STAMP(5 + h * 8);
            // ---------------- phase B ----------------
            {
                f32x4 s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) s[t] = zero4();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const frag qb = P::ld_kc(&sm.q[wave * 16][ks * 32], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) s[t] = P::mma(P::ld_kc(&sm.k[t * 16][ks * 32], LDH), qb, s[t]);
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + 4 * g + r;
                        const float v = (key >= qlo && key < qhi) ? s[t][r] * a.scale : -INFINITY;
                        s[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = colgroup_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __expf(s[t][r] - mx); s[t][r] = e; sum += e; }
                sum = colgroup_sum(sum);
                const float inv = 1.f / sum;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 pv = s[t] * inv;
                    if (a.drop.thr)
                        pv = drop4(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c) * 16 + t * 4 + g), pv);
                    P::st_nat(&sm.p[wave * 16][t * 16], LDH, pv);
                }
                STAMP(6 + h * 8);
                __builtin_amdgcn_wave_barrier();
                f32x4 o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = zero4();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const frag pb = P::ld_kc(&sm.p[wave * 16][ks * 32], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = P::mma(P::ld_kc(&sm.vt[t * 16][ks * 32], LDH), pb, o[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) P::st_nat(&sm.q[wave * 16][t * 16], LDH, o[t]);
                STAMP(7 + h * 8);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const frag ob = P::ld_kc(&sm.q[wave * 16][ks * 32], LDH);
#pragma unroll
                    for (int mt = 0; mt < 6; ++mt) oacc[mt] = P::mma(wo[mt][ks], ob, oacc[mt]);
                }
            }
            STAMP(8 + h * 8);
            lds_barrier();
            STAMP(9 + h * 8);
        }
        STAMP(3 + H * 8);

        // ---------------- epilogue: residual, LN2, MLP, residual ----------------
        STAMP(80);
        float x1[6][4];
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = oacc[mt][r] + lnp[192 + m0 + r];
            if (a.drop.thr && tok >= 0) o4 = drop4(a.drop, 2, (unsigned)(tok * 24 + (m0 >> 2)), o4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x1[mt][r] = o4[r] + xres[mt][r];
                o4[r] = x1[mt][r];
                s1 += x1[mt][r];
            }
            if (a.x1 && tok >= 0) *reinterpret_cast<f32x4*>(a.x1 + tok * 96 + m0) = o4;
        }
        STAMP(81);
        s1 = colgroup_sum(s1);
        const float mean = s1 * (1.f / 96.f);
        float vs = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x1[mt][r] - mean; vs += d * d; }
        vs = colgroup_sum(vs);
        const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 n4;
#pragma unroll
            for (int r = 0; r < 4; ++r) n4[r] = (x1[mt][r] - mean) * rstd * lnp[288 + m0 + r] + lnp[384 + m0 + r];
            P::st_nat(&sm.xn[wave * 16][mt * 16], LDX, n4);
        }
        STAMP(82);
        __builtin_amdgcn_wave_barrier();
        f32x4 hh[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hh[nt] = zero4();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const frag xb = P::ld_kc(&sm.xn[wave * 16][ks * 32], LDX);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) hh[nt] = P::mma(*reinterpret_cast<const frag*>(wmlp + (nt * 3 + ks) * 1024 + l * 16), xb, hh[nt]);
        }
        STAMP(83);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = nt * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[nt][r] = gelu_fast(hh[nt][r] + lnp[576 + n0 + r]);
            if (a.drop.thr && tok >= 0) hh[nt] = drop4(a.drop, 3, (unsigned)(tok * 16 + (n0 >> 2)), hh[nt]);
            P::st_nat(&sm.p[wave * 16][nt * 16], LDH, hh[nt]);
        }
        STAMP(84);
        __builtin_amdgcn_wave_barrier();
        f32x4 yy[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) yy[mt] = zero4();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const frag hb = P::ld_kc(&sm.p[wave * 16][ks * 32], LDH);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) yy[mt] = P::mma(*reinterpret_cast<const frag*>(wmlp + (12 + mt * 2 + ks) * 1024 + l * 16), hb, yy[mt]);
        }
        STAMP(85);
        if (tok >= 0) {
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const int m0 = mt * 16 + 4 * g;
                f32x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = yy[mt][r] + lnp[480 + m0 + r];
                if (a.drop.thr) o4 = drop4(a.drop, 4, (unsigned)(tok * 24 + (m0 >> 2)), o4);
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] += x1[mt][r];
                *reinterpret_cast<f32x4*>(a.y + tok * 96 + m0) = o4;
            }
        }
        STAMP(4 + H * 8);
        lds_barrier();
        STAMP(5 + H * 8);
    }
}

template __global__ void block_fwd_kernel<PF32>(BlockArgs);
template __global__ void block_fwd_kernel<PBF16>(BlockArgs);

// ==========================================================================================
// head: reference vit_simmim_original.py:314 (gather), :21-40 + :317-330 (BlockwiseToPixels,
// block id = idx // N), :335 (target = raw pixels of the masked patch), :338 (mean |.| / K).
// grid (ceil(K/64), B), 256 threads: 4 threads per masked entry, 24 features each.
// Writes per-workgroup partial |.| sums (deterministic two-stage reduction) and
// dpred = sign(pred - target) for the backward.
// ==========================================================================================
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, e = tid >> 2, part = tid & 3;
    const int b = blockIdx.y, k = blockIdx.x * 64 + e;
    const int P = a.P;
    float lsum = 0.f;
    if (k < a.K) {
        const int t = a.idx[(long)b * a.K + k];
        const int c = t / a.N, n = t - c * a.N;
        const float* enc = a.y + ((long)b * a.T + t) * 96 + part * 24;
        float ev[24];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 t4 = reinterpret_cast<const f32x4*>(enc)[i];
            ev[4*i] = t4[0]; ev[4*i+1] = t4[1]; ev[4*i+2] = t4[2]; ev[4*i+3] = t4[3];
        }
        const int wc = a.per_block ? c : 0;
        const float* Wp = a.w_pix + (long)wc * P * 96 + part * 24;
        for (int p = 0; p < P; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) acc += Wp[p * 96 + i] * ev[i];
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
            if (part == 0) {
                const float pred = acc + a.b_pix[wc * P + p];
                const float target = a.img[((long)b * a.S * P + (long)c * P + p) * a.N + n];
                const float d = pred - target;
                lsum += fabsf(d);
                a.dpred[((long)b * a.K + k) * P + p] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                if (a.pred) a.pred[((long)b * a.K + k) * P + p] = pred;
            }
        }
    }
    // workgroup reduction (fixed order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) a.partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// final loss: sum partials in a fixed order, scale by 1 / (B*K*P) / K
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* partial, int n, float scale, float* loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)scale);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_tokenize_fwd(const TokArgs& a, hipStream_t st) {
    if (a.P > 16 || a.N > 64) return MSST_ERR_UNSUPPORTED;
    ProfScope ps(K_TOK_FWD, st);
    if (a.P == 10 && a.N == 64) {   // the reference's shapes: fp32 matrix cores, persistent over the batch
        int nchunk = 1024 / (a.S > 0 ? a.S : 1);
        if (nchunk < 1) nchunk = 1;
        if (nchunk > a.B) nchunk = a.B;
        hipLaunchKernelGGL(tokenize_fwd_mfma_kernel, dim3(a.S, nchunk), dim3(256), 0, st, a);
    } else if (a.P == 10) hipLaunchKernelGGL(tokenize_fwd_kernel<10>, dim3(a.S, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tokenize_fwd_kernel<0>, dim3(a.S, a.B), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

template <class P>
static int launch_block_fwd_t(const BlockArgs& a, int grid, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    const size_t smem = sizeof(FwdSmem<P>);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_kernel<P>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ProfScope ps(K_BLOCK_FWD, st);
    hipLaunchKernelGGL(block_fwd_kernel<P>, dim3(grid), dim3(256), smem, st, a);
    return (int)hipGetLastError();
}

static int launch_block_fwd_bf16(const BlockArgs& a, int grid, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    const size_t smem = sizeof(FwdSmem<PBF16>) + 640 * sizeof(float) + 24 * 1024;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_bf16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ProfScope ps(K_BLOCK_FWD, st);
    hipLaunchKernelGGL(block_fwd_bf16_kernel, dim3(grid), dim3(256), smem, st, a);
    return (int)hipGetLastError();
}

int launch_block_fwd(const BlockArgs& a, int prec, hipStream_t st) {
    if (a.tm.L > 64 || a.tm.L < 1) return MSST_ERR_UNSUPPORTED;
    if (a.ntiles < 1) return 0;
    if (prec == MSST_PREC_F32 || (a.dbg & 16)) {   // dbg 16: generic template also for bf16 (A/B studies)
        const int grid = a.ntiles < a.max_grid ? a.ntiles : a.max_grid;
        return prec == MSST_PREC_F32 ? launch_block_fwd_t<PF32>(a, grid, st) : launch_block_fwd_t<PBF16>(a, grid, st);
    }
    if (a.H == 8 && !(a.dbg & 64)) {   // head-per-wave kernel: one 512-thread workgroup per CU walks the tiles
        static std::atomic<int> ncu_cached{0};   // idempotent once-value (every gfx950 part this library targets has the same count per process)
        int ncu = ncu_cached.load(std::memory_order_relaxed);
        if (!ncu) {
            int dev = 0;
            hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
            ncu_cached.store(ncu, std::memory_order_relaxed);
        }
        int grid = a.max_grid < a.ntiles ? a.max_grid : a.ntiles;
        if (grid > ncu) grid = ncu;
        // the role-split kernel (msst_fwd3.hip: attention waves + row-local waves, phases of adjacent tiles overlapped)
        return launch_block_fwd_rs(a, grid, st);
    }
    // persistent grid: 2 workgroups per CU (LDS 50 KB, <= 256 VGPRs)
    int grid = a.max_grid < a.ntiles ? a.max_grid : a.ntiles;
    if (grid > 512) grid = 512;
    return launch_block_fwd_bf16(a, grid, st);
}

bool block_fwd_writes_lse(const BlockArgs& a, int prec) {
    // exactly the condition under which launch_block_fwd picks launch_block_fwd_rs -- and a statistics buffer the 31-bit range of a
    // buffer descriptor covers (the attention backward refuses a larger one, msst_bwd4.hip: the forward must not claim to have written it)
    return prec == MSST_PREC_BF16 && a.H == 8 && !(a.dbg & (16 | 64)) && (long)a.ntiles * a.H * 256 < 0x7fffffffL;
}

bool block_fwd_writes_xn(const BlockArgs& a, int prec) {
    return prec == MSST_PREC_BF16 && !(a.dbg & 16);   // both tuned bf16 kernels (head-per-wave for 8 heads, 4-wave otherwise) store their LN1 rows
}

int launch_head_fwd(const HeadArgs& a, float* loss, hipStream_t st) {
    if (a.P > 16) return MSST_ERR_UNSUPPORTED;
    dim3 grid((a.K + 63) / 64, a.B);
    { ProfScope ps(K_HEAD_FWD, st);
    hipLaunchKernelGGL(head_fwd_kernel, grid, dim3(256), 0, st, a); }
    const int np = grid.x * grid.y;
    const float scale = 1.0f / ((float)a.B * (float)a.K * (float)a.P) / (float)a.K;
    ProfScope ps(K_LOSS_REDUCE, st);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, a.partial, np, scale, loss);
    return (int)hipGetLastError();
}

}  // namespace msst

namespace msst {

// ==========================================================================================
// classification head (row a17): reference vit_spatial_spectral.py:536-564 + :481-493 --
// 'b (c h w) d -> b c h w d', mean over the spectral axis c, LayerNorm(96), Linear(96 -> n_classes),
// output [B, n_classes, H*W].  grid (B), 256 threads = 4 threads per spatial position, 24 features each.
// ==========================================================================================
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(ClsArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, n = tid >> 2, part = tid & 3;
    if (n >= a.N) return;
    float m[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) m[i] = 0.f;
    for (int c = 0; c < a.S; ++c) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.y + ((long)b * a.T + c * a.N + n) * 96 + part * 24);
#pragma unroll
        for (int i = 0; i < 6; ++i) { const f32x4 t4 = src[i]; m[4*i] += t4[0]; m[4*i+1] += t4[1]; m[4*i+2] += t4[2]; m[4*i+3] += t4[3]; }
    }
    const float invS = 1.f / a.S;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { m[i] *= invS; s += m[i]; }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s * (1.f / 96.f);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { const float d = m[i] - mean; vs += d * d; }
    vs += __shfl_xor(vs, 1); vs += __shfl_xor(vs, 2);
    const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 24; ++i) m[i] = (m[i] - mean) * rstd * a.ln_g[part * 24 + i] + a.ln_b[part * 24 + i];
    for (int k = 0; k < a.NC; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) acc += a.w[k * 96 + part * 24 + i] * m[i];
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (part == 0) a.logits[((long)b * a.NC + k) * a.N + n] = acc + a.b[k];
    }
}

int launch_cls_head_fwd(const ClsArgs& a, hipStream_t st) {
    if (a.N > 64) return MSST_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cls_head_fwd_kernel, dim3(a.B), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
