// Backward kernels of the MaskedSST masked-pretraining hot path for gfx950 (row a15 of the scope
// table: what PyTorch autograd does for the reference at pretrain.py:116).
//
//   head_bwd        d(loss)/d(encoder out) through the masked gather + BlockwiseToPixels, to_pixels grads
//   block_bwd_mlp   FeedForward + LN2 + residual of one block (recompute from the saved x1)
//   block_bwd_attn  attention of one block, one workgroup per (tile chunk, head): recompute q,k,v,P,
//                   all six attention GEMM gradients, per-head weight grads in registers
//   block_bwd_ln1   sum of the per-head d(LN1 out) partials, LN1 backward, residual
//   tokenize_bwd    grads of the patch embedding, its two LayerNorms, position table and mask token
//   reduce_slabs    deterministic reduction of the per-workgroup partial-gradient slabs
#include <atomic>
#include <type_traits>
#include "msst_dev.h"
#include "msst_kernels.h"

namespace msst {

// ==========================================================================================
// head backward.  Forward (vit_simmim_original.py:314-338): pred[b,k] = W_c enc[b, idx[b,k]] + b_c,
// loss = sum |pred - target| * gscale.  dpred holds sign(pred - target).
// grid (S, nchunk), 384 threads = 4 token groups x 96 features.
// ==========================================================================================
__global__ __launch_bounds__(384) void head_bwd_kernel(HeadBwdArgs a) {
    __shared__ float gsh[64][17];
    __shared__ float red[4][16][96];
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int qg = tid / 96, d = tid - qg * 96;
    const int P = a.P, N = a.N, T = a.T, K = a.K;
    const int wc = a.per_block ? c : 0;
    float Wreg[16], accW[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) { Wreg[p] = p < P ? a.w_pix[((long)wc * P + p) * 96 + d] : 0.f; accW[p] = 0.f; }
    float accB = 0.f;
    const float gs = a.gout ? a.gscale * a.gout[0] : a.gscale;
    for (int b = chunk; b < a.B; b += gridDim.y) {
        for (int i = tid; i < N * P; i += 384) {
            const int n = i / P, p = i - n * P;
            const int t = c * N + n;
            const int e0 = a.csr_ptr[(long)b * (T + 1) + t], e1 = a.csr_ptr[(long)b * (T + 1) + t + 1];
            float s = 0.f;
            for (int e = e0; e < e1; ++e) s += a.dpred[((long)b * K + a.csr_pos[(long)b * K + e]) * P + p];
            gsh[n][p] = s * gs;
        }
        __syncthreads();
        if (tid < P) for (int n = 0; n < N; ++n) accB += gsh[n][tid];
        for (int n0 = qg; n0 < N; n0 += 32) {   // eight rows requested together (was one memory round trip per row)
            float yv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + 4 * j;
                yv[j] = n < N ? a.y[((long)b * T + c * N + n) * 96 + d] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + 4 * j;
                if (n < N) {
                    float dyv = 0.f;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        if (p < P) { const float gp = gsh[n][p]; dyv += gp * Wreg[p]; accW[p] += gp * yv[j]; }
                    }
                    a.dy[((long)b * T + c * N + n) * 96 + d] = dyv;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) red[qg][p][d] = accW[p];
    __syncthreads();
    float* slab = a.slab + ((long)c * gridDim.y + chunk) * (P * 96 + P);
    if (qg == 0) {
        for (int p = 0; p < P; ++p) slab[p * 96 + d] = (red[0][p][d] + red[1][p][d]) + (red[2][p][d] + red[3][p][d]);
    }
    if (tid < P) slab[P * 96 + tid] = accB;
}


// ------------------------------------------------------------------------------------------
// The same head backward for the reference's shapes (P = 10, N = 64) on the fp32 matrix cores (round 4).  The kernel above reads
// its gathered gradient rows from LDS once per multiply-add (ten broadcast reads per row and thread), moves y / dy as scalars
// and walks the CSR lists (three dependent loads) at the top of every sample: 131 us for 266 MB.  Here
//   * wave w <-> tokens 16 w .. + 15 of spectral block c, persistent over the samples of its chunk;
//   * the CSR walk of sample k + 3 / k + 2 / k + 1 (list bounds / first position / gradient values) is in flight while sample k
//     is computed: each stage's request is one iteration old when its result is needed;
//   * dy[token][d] = sum_p g[token][p] W[p][d]: 18 MFMAs (W^T as A operand, 18 registers for the whole walk; the gathered g is
//     the B operand in the lane that gathered it); dW[p][d] += sum_token g[token][p] y[token][d]: 24 MFMAs over the wave's 16
//     tokens, both operands transposed through a wave-private LDS tile.
// Slab layout as above.  grid (S, nchunk), 256 threads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void head_bwd_mfma_kernel(HeadBwdArgs a) {
    constexpr int P = 10, N = 64;
    __shared__ float y_raw[4 * 96 * 17];   // wave-private y[feature][token] tiles in the walk; the epilogue's reduction array afterwards
    __shared__ float g_t[4][16][17];      // wave-private g[p][token]
    float (*y_t)[96][17] = reinterpret_cast<float (*)[96][17]>(y_raw);
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, j = l & 15;
    const int T = a.T, K = a.K, nb = (int)gridDim.y;
    const int wc = a.per_block ? c : 0;
    // A fragments of W^T: lane (i = l & 15, kq = l >> 4) holds W[p = 4 ks + kq][16 mt + i] (zero beyond p = 9)
    float wf[6][3];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int pp = 4 * ks + g;
            wf[mt][ks] = pp < P ? a.w_pix[((long)wc * P + pp) * 96 + 16 * mt + j] : 0.f;
        }
    const float gs = a.gout ? a.gscale * a.gout[0] : a.gscale;
    const int n = 16 * w + j, t = c * N + n;
    f32x4 dW[6];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) dW[mt] = zero4();
    float accB[3] = {0.f, 0.f, 0.f};
    // the three in-flight stages of the CSR walk (values of samples b + nb, b + 2 nb, b + 3 nb at the top of iteration b)
    int e0_2 = 0, e1_2 = 0, e0_3 = 0, e1_3 = 0;   // list bounds of the sample two / three iterations ahead
    int e0_1 = 0, e1_1 = 0, pos_1 = 0;             // bounds + first position of the next sample
    int e0_0 = 0, e1_0 = 0;                         // bounds of the current sample
    float gv[3] = {0.f, 0.f, 0.f};                 // gradient values (first list entry) of the current sample
    f32x4 yrow[6];
    auto load_bounds = [&](int b, int& e0, int& e1) {
        const int bc = b < a.B ? b : a.B - 1;
        e0 = a.csr_ptr[(long)bc * (T + 1) + t];
        e1 = a.csr_ptr[(long)bc * (T + 1) + t + 1];
    };
    auto load_pos = [&](int b, int e0, int e1) -> int {
        const int bc = b < a.B ? b : a.B - 1;
        return a.csr_pos[(long)bc * K + (e1 > e0 ? e0 : 0)];
    };
    auto load_vals = [&](int b, int pos, float (&v)[3]) {
        const int bc = b < a.B ? b : a.B - 1;
        const float* src = a.dpred + ((long)bc * K + pos) * P;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int pp = 4 * ks + g; v[ks] = src[pp < P ? pp : 0]; }
    };
    auto request_y = [&](int b) {
        const int bc = b < a.B ? b : a.B - 1;
        const float* src = a.y + ((long)bc * T + t) * 96 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) yrow[mt] = *reinterpret_cast<const f32x4*>(src + 16 * mt);
    };
    // prologue: fill the pipeline for the first sample
    const int b0 = chunk;
    load_bounds(b0, e0_0, e1_0);
    load_bounds(b0 + nb, e0_1, e1_1);
    load_bounds(b0 + 2 * nb, e0_2, e1_2);
    load_bounds(b0 + 3 * nb, e0_3, e1_3);
    { const int p0 = load_pos(b0, e0_0, e1_0); load_vals(b0, p0, gv); }
    pos_1 = load_pos(b0 + nb, e0_1, e1_1);
    request_y(b0);
    for (int b = b0; b < a.B; b += nb) {
        int ll = threadIdx.x & 63;
        asm volatile("" : "+v"(ll));
        const int gl = ll >> 4, jl = ll & 15;
        // ---- this sample's gathered gradient g[token][p = 4 ks + g] ----
        float gg[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) gg[ks] = (e1_0 > e0_0 && 4 * ks + g < P) ? gv[ks] : 0.f;
        for (int e = e0_0 + 1; e < e1_0; ++e) {   // duplicates (the reference's misaligned index slicing produces them): rare
            const int pos = a.csr_pos[(long)b * K + e];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) if (4 * ks + g < P) gg[ks] += a.dpred[((long)b * K + pos) * P + 4 * ks + g];
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { gg[ks] *= gs; accB[ks] += gg[ks]; }
        f32x4 yv[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) yv[mt] = yrow[mt];
        // ---- advance the CSR pipeline: values of b + nb, position of b + 2 nb, bounds of b + 4 nb ----
        load_vals(b + nb, pos_1, gv);
        e0_0 = e0_1; e1_0 = e1_1;
        pos_1 = load_pos(b + 2 * nb, e0_2, e1_2);
        e0_1 = e0_2; e1_1 = e1_2;
        e0_2 = e0_3; e1_2 = e1_3;
        load_bounds(b + 4 * nb, e0_3, e1_3);
        // ---- dy = W^T g on the matrix cores; y and g to the wave's LDS tile for the weight gradient ----
        float* dst = a.dy + ((long)b * T + t) * 96 + 4 * g;
        float* yt = &y_t[w][4 * gl][jl];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            f32x4 acc = zero4();
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[mt][ks], gg[ks], acc, 0, 0, 0);
            *reinterpret_cast<f32x4*>(dst + 16 * mt) = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) yt[(16 * mt + r) * 17] = yv[mt][r];
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) g_t[w][4 * ks + gl][jl] = gg[ks];
        request_y(b + nb);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dW[p][d] += sum over the wave's 16 tokens: A[i = feature][kk = token], B[j = p][kk = token]
        const float* ya = &y_t[w][jl][gl];
        const float* ga = &g_t[w][jl][gl];
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) {
            const float bx = ga[4 * ts];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) dW[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[16 * mt * 17 + 4 * ts], bx, dW[mt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---------------- slab: dW[p][d] (C[i = feature 16 mt + 4 g + r][j = p]) summed over the four waves, db[p] ----------------
    float* slab = a.slab + ((long)c * gridDim.y + chunk) * (P * 96 + P);
    float* red = y_raw;   // [96][17]
    __syncthreads();
    for (int which = 0; which < 4; ++which) {
        if (w == which) {
#pragma unroll
            for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* cell = red + (16 * mt + 4 * g + r) * 17 + j;
                    *cell = which == 0 ? dW[mt][r] : *cell + dW[mt][r];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < P * 96; i += 256) { const int pp = i / 96, d = i - pp * 96; slab[i] = red[d * 17 + pp]; }
    __syncthreads();
    // db[p]: lane (j, g) holds the sums of p = 4 ks + g over its token's samples; sum over the 64 tokens
    float* redb = y_raw;   // [64 tokens][16]
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) redb[n * 16 + 4 * ks + g] = accB[ks];
    __syncthreads();
    if (tid < P) {
        float sres = 0.f;
        for (int rr = 0; rr < 64; ++rr) sres += redb[rr * 16 + tid];
        slab[P * 96 + tid] = sres;
    }
}

// ==========================================================================================
// reduce_segs: deterministic reduction of per-workgroup partial-gradient slabs, all segments of one
// backward stage in ONE launch.  A 256-thread block owns 32 (or, 16-byte path, 128) consecutive outputs of one
// segment: 8 thread groups stride over the slabs, partials are combined through LDS in a fixed order.
// ==========================================================================================
__global__ __launch_bounds__(256) void reduce_segs_kernel(RSegs r) {
    __shared__ f32x4 part[8][33];
    int si = 0;
    while (si + 1 < r.nseg && (int)blockIdx.x >= r.s[si + 1].blk0) ++si;
    RSeg g = r.s[si];
    if ((int)blockIdx.y >= g.ny) return;   // (block uniform)
    g.src += (long)blockIdx.y * r.src_ystride;
    g.dst += (long)blockIdx.y * r.dst_ystride;
    const int io = threadIdx.x & 31, sg = threadIdx.x >> 5;
    if (g.vec4) {
        // 128 consecutive outputs per block, 4 per thread; the slab loop keeps four 16-byte requests in flight
        const int i = (((int)blockIdx.x - g.blk0) * 32 + io) * 4;
        f32x4 s = zero4();
        if (i < g.n) {
            const float* p = g.src + i;
            int k = sg;
            // long slab lists (the 512 slabs of the two-workgroups-per-CU MLP backward) with sixteen requests in flight: with four, a
            // block walked 16 serial HBM round trips and the launch took as long as its slowest block
            for (; k + 120 < g.nslab; k += 128) {
                f32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long)(k + 8 * u) * g.slab_stride);
#pragma unroll
                for (int u = 0; u < 16; ++u) s = s + v[u];
            }
            // the 64 slabs of the attention backward (and the 256 of the fused LN1 + MLP launch): eight requests in flight, ONE round
            // trip per block instead of two serial rounds of four
            for (; k + 56 < g.nslab; k += 64) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long)(k + 8 * u) * g.slab_stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) s = s + v[u];
            }
            for (; k + 24 < g.nslab; k += 32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long)k * g.slab_stride);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long)(k + 8) * g.slab_stride);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long)(k + 16) * g.slab_stride);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long)(k + 24) * g.slab_stride);
                s = (((s + v0) + v1) + v2) + v3;
            }
            for (; k < g.nslab; k += 8) s = s + *reinterpret_cast<const f32x4*>(p + (long)k * g.slab_stride);
        }
        part[sg][io] = s;
        __syncthreads();
        if (sg == 0 && i < g.n) {
            const f32x4 t = ((part[0][io] + part[1][io]) + (part[2][io] + part[3][io])) +
                            ((part[4][io] + part[5][io]) + (part[6][io] + part[7][io]));
            const int row = i / g.row_len, col = i - row * g.row_len;
            *reinterpret_cast<f32x4*>(g.dst + (long)row * g.row_stride + col) = t;
        }
        return;
    }
    const int i = ((int)blockIdx.x - g.blk0) * 32 + io;
    float s = 0.f;
    if (i < g.n)
        for (int k = sg; k < g.nslab; k += 8) s += g.src[(long)k * g.slab_stride + i];
    float* ps = reinterpret_cast<float*>(&part[0][0]);
    ps[sg * 33 + io] = s;
    __syncthreads();
    if (sg == 0 && i < g.n) {
        const float t = ((ps[0 * 33 + io] + ps[1 * 33 + io]) + (ps[2 * 33 + io] + ps[3 * 33 + io])) +
                        ((ps[4 * 33 + io] + ps[5 * 33 + io]) + (ps[6 * 33 + io] + ps[7 * 33 + io]));
        const int row = i / g.row_len, col = i - row * g.row_len;
        g.dst[(long)row * g.row_stride + col] = t;
    }
}

// ==========================================================================================
// MLP half of a block, backward.  y = x1 + W2 gelu(W1 LN2(x1) + b1) + b2
// (vit_spatial_spectral.py:32-44, :22-29, :103).  Contiguous 64-token tiles, persistent grid.
// Phase 1 (wave <-> 16 rows): recompute, dh, dh_pre, d(LN2 out), LN2 backward, dx1 store.
// Phase 2 (wave <-> 16 output rows of each weight-grad): dW1 += dhp^T xn2, dW2 += dy^T h over the
// 64 rows of the tile; bias grads as LDS column sums.
// ==========================================================================================
template <class P>
struct MlpBwdSmem {
    typedef typename P::elem elem;
    static constexpr int LDX = 96 + P::PADE;
    static constexpr int LDH = 64 + P::PADE;
    elem xn2[64][LDX];
    elem dy[64][LDX];
    elem h[64][LDH];
    elem dhp[64][LDH];
};

template <class P, bool X1B = false>
__global__ __launch_bounds__(256, P::WAVES_BWD_MLP) void block_bwd_mlp_kernel(MlpBwdArgs a) {
    typedef typename P::elem elem;
    typedef typename P::frag frag;
    typedef MlpBwdSmem<P> SM;
    constexpr int KS = P::KS, LDX = SM::LDX, LDH = SM::LDH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SM& sm = *reinterpret_cast<SM*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, c = l & 15;
    const elem* w1 = reinterpret_cast<const elem*>(a.w.w1);
    const elem* w1T = reinterpret_cast<const elem*>(a.w.w1T);
    const elem* w2T = reinterpret_cast<const elem*>(a.w.w2T);

    f32x4 dW1[6], dW2[6];   // dW1: C[i = n in tile wave][j = m tile jt]; dW2: C[i = m tile it][j = n in tile wave]
    float dgam[6][4], dbet[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        dW1[i] = zero4(); dW2[i] = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) { dgam[i][r] = 0.f; dbet[i][r] = 0.f; }
    }
    // bias gradients are column sums over the rows of a tile: one more MFMA against a fragment of ones in the weight-grad
    // k-loop (every column of the C tile then holds the sum).  db1: this wave's 16 hidden units; db2: feature tiles wave, wave + 4
    f32x4 db1a = zero4(), db2a = zero4(), db2b = zero4();

    // tile-invariant small vectors in LDS: ln2_g | ln2_b | b1
    float* lnp = reinterpret_cast<float*>(smem_raw + sizeof(SM));
    if (tid < 96) { lnp[tid] = a.w.ln2_g[tid]; lnp[96 + tid] = a.w.ln2_b[tid]; if (tid < 64) lnp[192 + tid] = a.w.b1[tid]; }
    // bf16: the three weight matrices (36 fragment-packed KB) stay in LDS for the life of the workgroup and the
    // next tile's rows are requested one tile ahead (this kernel runs 1 wave/SIMD: registers are plentiful)
    constexpr bool BF = sizeof(elem) == 2;
    constexpr bool TWO = BF && P::WAVES_BWD_MLP == 2;   // two workgroups per CU: w1T from L2 (LDS 71 KB), rows not prefetched
    constexpr bool PREF = BF && !TWO;
    constexpr int NST = TWO ? 6 : 9;
    char* wl = smem_raw + sizeof(SM) + 256 * sizeof(float);   // [w1 12 | w2T 12 | w1T 12] fragments of 1 KB
    if constexpr (BF) {
#pragma unroll
        for (int i9 = 0; i9 < NST; ++i9) {
            const int f = wave * NST + i9;
            const char* src = f < 12 ? reinterpret_cast<const char*>(w1) + f * 1024
                            : f < 24 ? reinterpret_cast<const char*>(w2T) + (f - 12) * 1024
                                     : reinterpret_cast<const char*>(w1T) + (f - 24) * 1024;
            dma_frag(src, wl + f * 1024);
        }
        wait_vm0();
    }
    __syncthreads();
    auto wfrag = [&](int which, const elem* wg, int K, int row0, int k0) -> frag {
        if (TWO && which == 2) return P::ld_w(wg, K, row0, k0);
        if constexpr (BF) {
            const int f = which * 12 + (row0 >> 4) * (K >> 5) + (k0 >> 5);
            return *reinterpret_cast<const frag*>(wl + f * 1024 + l * 16);
        } else {
            return P::ld_w(wg, K, row0, k0);
        }
    };

    const int ntiles = (a.ntok + 63) / 64;
    f32x4 xpre[6], dpre[6];
    if constexpr (PREF) {
        const long tok0 = (long)blockIdx.x * 64 + wave * 16 + c;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            xpre[mt] = tok0 < a.ntok ? ld_x1_4<X1B>(a.x1, tok0, mt * 16 + 4 * g) : zero4();
            dpre[mt] = tok0 < a.ntok ? *reinterpret_cast<const f32x4*>(a.dy + tok0 * 96 + mt * 16 + 4 * g) : zero4();
        }
    }
    // Two workgroups per CU (no register room for a whole-tile-ahead prefetch): the rows of a tile are requested
    // unconditionally from a clamped address, all twelve at once (a load inside `if (valid)` is compiled as request /
    // s_waitcnt vmcnt(0) / use: six serialised HBM round trips per tile), at the START OF THE WEIGHT-GRAD PHASE of the previous
    // tile -- the 48 registers of this tile's rows are dead by then and that phase only reads LDS.
    f32x4 xrow[6], drow[6];
    auto request_rows = [&](int tile_) {
        const long t_ = (long)tile_ * 64 + wave * 16 + c;
        const long tokc = (tile_ < ntiles && t_ < a.ntok) ? t_ : 0;
        const float* ds_ = a.dy + tokc * 96 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) xrow[mt] = ld_x1_4<X1B>(a.x1, tokc, mt * 16 + 4 * g);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) drow[mt] = *reinterpret_cast<const f32x4*>(ds_ + mt * 16);
    };
    if constexpr (!PREF) request_rows(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long tok = (long)tile * 64 + wave * 16 + c;
        const bool valid = tok < a.ntok;
        float xhat[6][4], dyv[6][4];
        float s1 = 0.f;
        // rows of this tile, requested unconditionally from a clamped address, all twelve before the first is used: a load
        // inside `if (valid)` is compiled as request / s_waitcnt vmcnt(0) / use -- six serialised HBM round trips per tile
        // (the rows of this tile are in xrow / drow: requested during the weight-grad phase of the previous tile)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const int m0 = mt * 16 + 4 * g;
            f32x4 xr = zero4(), dr = zero4();
            if constexpr (PREF) {
                xr = xpre[mt]; dr = dpre[mt];
                const long tokn = tok + (long)gridDim.x * 64;   // same rows of this workgroup's next tile
                const bool vn = tile + (int)gridDim.x < ntiles && tokn < a.ntok;
                xpre[mt] = vn ? ld_x1_4<X1B>(a.x1, tokn, m0) : zero4();
                dpre[mt] = vn ? *reinterpret_cast<const f32x4*>(a.dy + tokn * 96 + m0) : zero4();
            } else {
                xr = valid ? xrow[mt] : zero4();
                dr = valid ? drow[mt] : zero4();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { xhat[mt][r] = xr[r]; dyv[mt][r] = dr[r]; s1 += xr[r]; }
            // site 4 (MLP-out dropout): the FeedForward branch sees the masked gradient, the residual the raw one
            if (a.drop.thr && valid) dr = drop4(a.drop, 4, (unsigned)(tok * 24 + (m0 >> 2)), dr);
            P::st_nat(&sm.dy[wave * 16][mt * 16], LDX, dr);
        }
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
        // h_pre = W1 xn2 + b1, dh = W2^T dy   (C[i = n][j = row])
        f32x4 hp[4], dh[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { hp[nt] = zero4(); dh[nt] = zero4(); }
#pragma unroll P::UNROLL
        for (int k0 = 0; k0 < 96; k0 += KS) {
            const frag xb = P::ld_kc(&sm.xn2[wave * 16][k0], LDX);
            const frag db = P::ld_kc(&sm.dy[wave * 16][k0], LDX);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                hp[nt] = P::mma(wfrag(0, w1, 96, nt * 16, k0), xb, hp[nt]);
                dh[nt] = P::mma(wfrag(1, w2T, 96, nt * 16, k0), db, dh[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = nt * 16 + 4 * g;
            f32x4 hv, dv;
            f32x4 dhm = dh[nt];
            unsigned keep3 = 0xfu;   // site 3: one hash for the backward mask here and the forward mask below
            if (a.drop.thr && valid) dhm = drop4_keep(a.drop, 3, (unsigned)(tok * 16 + (n0 >> 2)), dhm, keep3);   // site 3 backward
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pre = hp[nt][r] + lnp[192 + n0 + r];
                float gv, gg;
                P::gelu_both(pre, gv, gg);
                hv[r] = gv;
                dv[r] = dhm[r] * gg;
            }
            if (a.drop.thr && valid) hv = drop4_bits(a.drop, keep3, hv);                                  // site 3 forward
            P::st_nat(&sm.h[wave * 16][nt * 16], LDH, hv);
            P::st_nat(&sm.dhp[wave * 16][nt * 16], LDH, dv);
        }
        __builtin_amdgcn_wave_barrier();
        // d(xn2) = W1^T dh_pre   (C[i = m][j = row])
        f32x4 dxn[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) dxn[mt] = zero4();
        if constexpr (TWO) {
            // two workgroups per CU: the w1^T fragments come from L2.  All twelve are requested before the first MFMA (hp / dh
            // are dead, 48 registers are free) -- left to the compiler each one was request / s_waitcnt vmcnt(0) / MFMA:
            // twelve serialised L2 round trips per tile
            frag w1t[2][6];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) w1t[ks][mt] = P::ld_w(w1T, 64, mt * 16, ks * KS);
            const frag hb0 = P::ld_kc(&sm.dhp[wave * 16][0], LDH);
            const frag hb1 = P::ld_kc(&sm.dhp[wave * 16][KS], LDH);
            MSST_SCHED_FENCE();
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) dxn[mt] = P::mma(w1t[0][mt], hb0, dxn[mt]);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) dxn[mt] = P::mma(w1t[1][mt], hb1, dxn[mt]);
        } else {
#pragma unroll P::UNROLL
            for (int k0 = 0; k0 < 64; k0 += KS) {
                const frag hb = P::ld_kc(&sm.dhp[wave * 16][k0], LDH);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) dxn[mt] = P::mma(wfrag(2, w1T, 64, mt * 16, k0), hb, dxn[mt]);
            }
        }
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
                const float dg = dn * lnp[m0 + r];
                dxn[mt][r] = dg;
                g1 += dg;
                g2 += dg * xhat[mt][r];
            }
        }
        g1 = colgroup_sum(g1) * (1.f / 96.f);
        g2 = colgroup_sum(g2) * (1.f / 96.f);
        if (valid) {
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const int m0 = mt * 16 + 4 * g;
                f32x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = dyv[mt][r] + rstd * (dxn[mt][r] - g1 - xhat[mt][r] * g2);
                *reinterpret_cast<f32x4*>(a.dx1 + tok * 96 + m0) = o4;
                if constexpr (sizeof(elem) == 2) {
                    if (a.dab) {   // the attention half's operand: same rows, to_out dropout (site 2) applied, bf16
                        f32x4 d4 = o4;
                        if (a.drop.thr) d4 = drop4(a.drop, 2, (unsigned)(tok * 24 + (m0 >> 2)), d4);
                        *reinterpret_cast<s16x4*>(reinterpret_cast<bf16_t*>(a.dab) + tok * 96 + m0) = f2bf4(d4);
                    }
                }
            }
        }
        lds_barrier();
        if constexpr (!PREF) request_rows(tile + (int)gridDim.x);
        // ---------------- phase 2: weight grads over the 64 rows of the tile ----------------
#pragma unroll P::UNROLL
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
        lds_barrier();
    }

    // ---------------- write this workgroup's slab ----------------
    float* slab = a.slab + (long)blockIdx.x * MSST_MLP_SLAB_N;
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
    // LN2 gamma/beta: sum over the 16 rows of the wave, then over waves through LDS
    float* red = reinterpret_cast<float*>(smem_raw);  // [4 waves][2][96]
    __syncthreads();
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
    __syncthreads();
    if (tid < 192) {
        const int which = tid / 96, m = tid - which * 96;
        slab[12288 + 160 + which * 96 + m] = (red[(0 * 2 + which) * 96 + m] + red[(1 * 2 + which) * 96 + m]) +
                                             (red[(2 * 2 + which) * 96 + m] + red[(3 * 2 + which) * 96 + m]);
    }
}

template __global__ void block_bwd_mlp_kernel<PF32>(MlpBwdArgs);
template __global__ void block_bwd_mlp_kernel<PBF16>(MlpBwdArgs);

// ==========================================================================================
// attention half of a block, backward (vit_spatial_spectral.py:47-78 under PreNorm :22-29).
// grid (nchunk, H): workgroup (chunk, h) walks the 64-row tiles chunk, chunk+nchunk, ... for ONE
// head, so that the head's weight gradients (dWq|dWk|dWv [3][64][96], dWout_h [96][64]) stay in
// registers (96 per lane) for the whole walk and are written once as a slab.
// Per tile: LN1(x) -> q,k,v^T -> S^T,P -> O;  dO = da Wout_h;  dP -> dS;  dV, dQ, dK;  weight grads;
// the head's partial d(LN1 out) = [dq|dk|dv] Wqkv_h is stored for block_bwd_ln1 to sum over heads.
// LDS holds every intermediate once; operands needed in the other orientation are fetched with
// k-strided fragment loads (ds_read_b64_tr_b16 in bf16).  xd holds LN1(x), then da, then LN1(x) again.
// ==========================================================================================
template <class P>
struct AttnBwdSmem {
    typedef typename P::elem elem;
    static constexpr int LDX = 96 + P::PADE;
    static constexpr int LDH = 64 + P::PADE;
    elem xd[64][LDX];
    // q | k | dO | ds are contiguous: dead after phase C, they receive the staged wqkvT fragments (bf16)
    elem q[64][LDH];    // q[row][d]
    elem k[64][LDH];    // k[row][d]
    elem dO[64][LDH];   // dO[query][d]
    elem ds[64][LDH];   // ds[query][key]
    elem vt[64][LDH];   // vt[d][row]     -> later dv[row][d]
    elem p[64][LDH];    // p[query][key]  -> later dk[row][d]
    elem o[64][LDH];    // o[query][d]    -> later dq[row][d]
};

template <class P>
__device__ __forceinline__ void ln1_rows_to_lds(const float* x, const float* gam, const float* bet, long tok,
                                                typename P::elem (*dst)[96 + P::PADE]) {
    const int tid = threadIdx.x;
    const int r = tid >> 2, part = tid & 3;
    float v[24];
    if (tok >= 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x + tok * 96 + part * 24);
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
        dst[r][d] = P::cvt((v[i] - mean) * rstd * gam[d] + bet[d]);
    }
}

template <class P>
__global__ __launch_bounds__(256, P::WAVES_BWD_ATTN) void block_bwd_attn_kernel(AttnBwdArgs a) {
    typedef typename P::elem elem;
    typedef typename P::frag frag;
    typedef AttnBwdSmem<P> SM;
    constexpr int KS = P::KS, LDX = SM::LDX, LDH = SM::LDH;
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

    // persistent weight-grad accumulators: dWqkv: C[i = d in tile wave][j = m tile], for q,k,v;
    // dWout: C[i = m tile][j = d in tile wave]
    f32x4 gq[6], gk[6], gv[6], go[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { gq[i] = zero4(); gk[i] = zero4(); gv[i] = zero4(); go[i] = zero4(); }

    const int2 sp_ln = tm.row_sp(tid >> 2);
    const int qlo = ((wave * 16 + c) / L) * L, qhi = qlo + L;
    constexpr int CPR = 96 * (int)sizeof(elem) / 16;   // 16-byte chunks per row (copy-out of the partial)

#ifdef MSST_STAMPS
    const bool stamp_wg = (a.dbg & 8) && blockIdx.x == 7 && blockIdx.y == 3 && tid == 0;
#endif
    // bf16: software prefetch (global round trips are ~2k cycles under load); fp32 keeps the simple loads
    // (measured: prefetching rows + da + both weight sets needs ~400 registers -> 1 workgroup/CU, slower than
    // 2 workgroups/CU without it; only the cheap parts are enabled)
    constexpr bool BF = sizeof(elem) == 2;
#ifndef MSST_PF_X
#define MSST_PF_X 1
#endif
#ifndef MSST_PF_DA
#define MSST_PF_DA 0
#endif
#ifndef MSST_PF_WA
#define MSST_PF_WA 0
#endif
#ifndef MSST_PF_WDO
#define MSST_PF_WDO 0
#endif
    constexpr bool PF_X = BF && MSST_PF_X;      // next tile's rows requested during phase C      (+24 registers)
    constexpr bool PF_DA = BF && MSST_PF_DA;    // da rows requested at tile start                (+24)
    constexpr bool PF_WA = BF && MSST_PF_WA;    // phase-A weight fragments requested at tile start (+36)
    constexpr bool PF_WDO = BF && MSST_PF_WDO;  // Wout^T fragments requested before S / softmax   (+48)
    constexpr bool KEEP_XN = BF;                // keep LN1(x) packed (12 registers) instead of recomputing it for phase D
    constexpr int NKX = PF_WA ? 3 : 1, NKD = PF_WDO ? 3 : 1;
    // LN1 gamma / beta in LDS (tile invariant)
    float* lnp = reinterpret_cast<float*>(smem_raw + sizeof(SM));
    void* touch_pad = lnp + 192;   // 256 B nobody reads (l2_touch target)
    if (tid < 96) { lnp[tid] = a.w.ln1_g[tid]; lnp[96 + tid] = a.w.ln1_b[tid]; }
    __syncthreads();
    const int lr = tid >> 2, lpart = tid & 3;
    f32x4 xv[6];          // this thread's 24 row values of the tile to process (prefetched one tile ahead)
    if constexpr (PF_X) {
        const long tok0 = tm.token_sp(blockIdx.x, sp_ln);
#pragma unroll
        for (int i = 0; i < 6; ++i) xv[i] = tok0 >= 0 ? reinterpret_cast<const f32x4*>(a.x + tok0 * 96 + lpart * 24)[i] : zero4();
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef MSST_STAMPS
        const bool stamp_on = stamp_wg && tile == blockIdx.x + 20 * (int)gridDim.x;   // a mid-walk tile (the last ones run on a half-empty chip)
#endif
        STAMP(0);
        const long tok_ln = tm.token_sp(tile, sp_ln);
        f32x4 dav[6];         // da rows of this tile (consumed after phase A)
        frag wa[3][NKX];      // phase-A weight fragments
        s16x4 xnk[6];         // LN1(x) of this thread's 24 features, packed bf16 (re-stored before phase D)
        if constexpr (PF_DA) {
#pragma unroll
            for (int i = 0; i < 6; ++i) dav[i] = tok_ln >= 0 ? reinterpret_cast<const f32x4*>(a.da + tok_ln * 96 + lpart * 24)[i] : zero4();
        }
        if constexpr (PF_WA) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int ks = 0; ks < NKX; ++ks) wa[m][ks] = P::ld_w(wqkv, 96, (m * H + h) * 64 + wave * 16, ks * 32);
        }
        frag wa0[3];          // bf16: first k-step of the phase-A weights, requested under the row loads
        if constexpr (BF && !PF_WA) {
            if constexpr (!PF_DA) l2_touch(a.da + (tok_ln >= 0 ? tok_ln : 0) * 96 + lpart * 24, touch_pad);
#pragma unroll
            for (int m = 0; m < 3; ++m) wa0[m] = P::ld_w(wqkv, 96, (m * H + h) * 64 + wave * 16, 0);
        }
        if constexpr (KEEP_XN) {
            float v[24];
            if constexpr (PF_X) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { v[4*i] = xv[i][0]; v[4*i+1] = xv[i][1]; v[4*i+2] = xv[i][2]; v[4*i+3] = xv[i][3]; }
            } else if (tok_ln >= 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(a.x + tok_ln * 96 + lpart * 24);
#pragma unroll
                for (int i = 0; i < 6; ++i) { f32x4 t4 = src[i]; v[4*i] = t4[0]; v[4*i+1] = t4[1]; v[4*i+2] = t4[2]; v[4*i+3] = t4[3]; }
            } else {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = 0.f;
            }
            float sm1 = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) sm1 += v[i];
            sm1 += __shfl_xor(sm1, 1); sm1 += __shfl_xor(sm1, 2);
            const float mean = sm1 * (1.f / 96.f);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) { const float d = v[i] - mean; vs += d * d; }
            vs += __shfl_xor(vs, 1); vs += __shfl_xor(vs, 2);
            const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x4 n4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = lpart * 24 + 4 * i + e;
                    n4[e] = (v[4*i+e] - mean) * rstd * lnp[d] + lnp[96 + d];
                }
                xnk[i] = f2bf4(n4);
                *reinterpret_cast<s16x4*>(&sm.xd[lr][lpart * 24 + 4 * i]) = xnk[i];
            }
        } else {
            ln1_rows_to_lds<P>(a.x, lnp, lnp + 96, tok_ln, sm.xd);
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);
        // ---------------- phase A: q, k, v^T (wave <-> 16 head channels) ----------------
        {
            f32x4 cq[4], ck[4], cv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { cq[t] = zero4(); ck[t] = zero4(); cv[t] = zero4(); }
            const int rq = (0 * H + h) * 64 + wave * 16, rk = (1 * H + h) * 64 + wave * 16, rv = (2 * H + h) * 64 + wave * 16;
            if constexpr (PF_WA) {
#pragma unroll
                for (int ks = 0; ks < NKX; ++ks) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const frag xb = P::ld_kc(&sm.xd[t * 16][ks * 32], LDX);
                        cq[t] = P::mma(wa[0][ks], xb, cq[t]);
                        ck[t] = P::mma(wa[1][ks], xb, ck[t]);
                        cv[t] = P::mma(xb, wa[2][ks], cv[t]);
                    }
                }
            } else if constexpr (BF) {
                frag cur[3] = {wa0[0], wa0[1], wa0[2]};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {   // next k-step's fragments are in flight under this one's MFMAs
                    frag nxt[3];
                    if (ks < 2) {
                        nxt[0] = P::ld_w(wqkv, 96, rq, (ks + 1) * 32);
                        nxt[1] = P::ld_w(wqkv, 96, rk, (ks + 1) * 32);
                        nxt[2] = P::ld_w(wqkv, 96, rv, (ks + 1) * 32);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const frag xb = P::ld_kc(&sm.xd[t * 16][ks * 32], LDX);
                        cq[t] = P::mma(cur[0], xb, cq[t]);
                        ck[t] = P::mma(cur[1], xb, ck[t]);
                        cv[t] = P::mma(xb, cur[2], cv[t]);
                    }
                    if (ks < 2) { cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; }
                }
            } else {
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 96; k0 += KS) {
                    const frag aq = P::ld_w(wqkv, 96, rq, k0);
                    const frag ak = P::ld_w(wqkv, 96, rk, k0);
                    const frag av = P::ld_w(wqkv, 96, rv, k0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const frag xb = P::ld_kc(&sm.xd[t * 16][k0], LDX);
                        cq[t] = P::mma(aq, xb, cq[t]);
                        ck[t] = P::mma(ak, xb, ck[t]);
                        cv[t] = P::mma(xb, av, cv[t]);
                    }
                }
            }
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
        // ---------------- da rows -> xd (overwrites LN1(x); rows of this wave only) ----------------
        {
            const int r = tid >> 2, pt = tid & 3;
            const long tok = tok_ln;
            f32x4 dl[6];   // all six requests in flight before the first use (one memory round trip, not six)
            if constexpr (PF_DA) {
#pragma unroll
                for (int i = 0; i < 6; ++i) dl[i] = dav[i];
            } else {
                const f32x4* src = reinterpret_cast<const f32x4*>(a.da + (tok >= 0 ? tok : 0) * 96 + pt * 24);
#pragma unroll
                for (int i = 0; i < 6; ++i) dl[i] = src[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x4 t4 = tok >= 0 ? dl[i] : zero4();
                if (a.drop.thr && tok >= 0) t4 = drop4(a.drop, 2, (unsigned)(tok * 24 + pt * 6 + i), t4);   // site 2 backward
#pragma unroll
                for (int e = 0; e < 4; ++e) sm.xd[r][pt * 24 + 4 * i + e] = P::cvt(t4[e]);
            }
        }
        frag wdo[4][NKD];   // Wout_h^T fragments for the dO GEMM (requested now, used after S / softmax / O)
        if constexpr (PF_WDO) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < NKD; ++ks) wdo[t][ks] = P::ld_w(woutT, 96, h * 64 + t * 16, ks * 32);
        }
        // ---------------- phase B: wave <-> 16 query rows ----------------
        f32x4 pr[4], pdr[4];   // raw / dropped probabilities (C layout [key][query])
        unsigned keep1 = 0;
        {
#pragma unroll
            for (int t = 0; t < 4; ++t) pr[t] = zero4();
#pragma unroll P::UNROLL
            for (int k0 = 0; k0 < 64; k0 += KS) {
                const frag qb = P::ld_kc(&sm.q[wave * 16][k0], LDH);
#pragma unroll
                for (int t = 0; t < 4; ++t) pr[t] = P::mma(P::ld_kc(&sm.k[t * 16][k0], LDH), qb, pr[t]);  // C[i = key][j = query]
            }
            float mx = -INFINITY;
            float sum = 0.f;
            if constexpr (BF) {
                // same arithmetic as block_fwd_hw_kernel: exp2(s c - max c), c = scale log2 e; nothing to mask when L == 64
                const float cs = a.scale * 1.44269504088896340736f;
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
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(pr[t][r], cs, -mc)); pr[t][r] = e; sum += e; }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + 4 * g + r;
                        const float v = (key >= qlo && key < qhi) ? pr[t][r] * a.scale : -INFINITY;
                        pr[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = colgroup_max(mx);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = P::exp(pr[t][r] - mx); pr[t][r] = e; sum += e; }
            }
            sum = colgroup_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pr[t] = pr[t] * inv;
                f32x4 pd = pr[t];   // site 1: O and dV see the dropped probabilities, the softmax backward the raw ones
                if (a.drop.thr) {
                    unsigned kb;
                    pd = drop4_keep(a.drop, 1, (unsigned)(((tile * H + h) * 64 + wave * 16 + c) * 16 + t * 4 + g), pd, kb);
                    keep1 |= kb << (4 * t);   // the 16 keep decisions of this lane, reused for dP below
                }
                P::st_nat(&sm.p[wave * 16][t * 16], LDH, pd);  // p[query][key]
                pdr[t] = pd;
            }
        }
        STAMP(5);
        if constexpr (!BF) __builtin_amdgcn_wave_barrier();
        f32x4 dor[4];   // bf16: dO^T of this wave's queries stays in registers for dP
        {
            // o = P v  (C[i = d][j = query]) and dO = Wout_h^T da (C[i = d][j = query])
            f32x4 o[4], dov[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { o[t] = zero4(); dov[t] = zero4(); }
            frag wd0[4];   // bf16: first k-step of Wout_h^T, requested under o = P v
            if constexpr (BF && !PF_WDO) {
#pragma unroll
                for (int t = 0; t < 4; ++t) wd0[t] = P::ld_w(woutT, 96, h * 64 + t * 16, 0);
            }
            if constexpr (BF) {
                // P stays in registers: C tiles (2m, 2m+1) of S^T packed = B operand of key chunk m (permuted key order)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const frag pb = PBF16::pack2(pdr[2 * m], pdr[2 * m + 1]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = P::mma(PBF16::ld_kc_perm(reinterpret_cast<const bf16_t*>(&sm.vt[t * 16][m * 32]), LDH), pb, o[t]);
                }
            } else {
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 64; k0 += KS) {
                    const frag pb = P::ld_kc(&sm.p[wave * 16][k0], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = P::mma(P::ld_kc(&sm.vt[t * 16][k0], LDH), pb, o[t]);
                }
            }
            if constexpr (PF_WDO) {
#pragma unroll
                for (int ks = 0; ks < NKD; ++ks) {
                    const frag db = P::ld_kc(&sm.xd[wave * 16][ks * 32], LDX);
#pragma unroll
                    for (int t = 0; t < 4; ++t) dov[t] = P::mma(wdo[t][ks], db, dov[t]);
                }
            } else {
                if constexpr (BF) {
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {   // next k-step's four fragments in flight under this one's MFMAs
                        frag nx[4];
                        if (ks < 2) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) nx[t] = P::ld_w(woutT, 96, h * 64 + t * 16, (ks + 1) * 32);
                        }
                        const frag db = P::ld_kc(&sm.xd[wave * 16][ks * 32], LDX);
#pragma unroll
                        for (int t = 0; t < 4; ++t) dov[t] = P::mma(wd0[t], db, dov[t]);
                        if (ks < 2) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) wd0[t] = nx[t];
                        }
                    }
                } else {
#pragma unroll P::UNROLL
                    for (int k0 = 0; k0 < 96; k0 += KS) {
                        const frag db = P::ld_kc(&sm.xd[wave * 16][k0], LDX);
                        frag wf[4];   // the four requests go out together
#pragma unroll
                        for (int t = 0; t < 4; ++t) wf[t] = P::ld_w(woutT, 96, h * 64 + t * 16, k0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) dov[t] = P::mma(wf[t], db, dov[t]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                P::st_nat(&sm.o[wave * 16][t * 16], LDH, o[t]);      // o[query][d]
                P::st_nat(&sm.dO[wave * 16][t * 16], LDH, dov[t]);   // dO[query][d]
                dor[t] = dov[t];
            }
        }
        STAMP(6);
        if constexpr (!BF) __builtin_amdgcn_wave_barrier();
        {
            // dP^T[key][query] = sum_d v[key][d] dO[query][d]: A = v (k-strided read of vt), B = dO rows
            f32x4 dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) dp[t] = zero4();
            if constexpr (BF) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const frag db = PBF16::pack2(dor[2 * m], dor[2 * m + 1]);   // channel chunk m, permuted order
#pragma unroll
                    for (int t = 0; t < 4; ++t) dp[t] = P::mma(PBF16::ld_ks_perm(reinterpret_cast<const bf16_t*>(&sm.vt[m * 32][t * 16]), LDH), db, dp[t]);
                }
            } else {
#pragma unroll P::UNROLL
                for (int k0 = 0; k0 < 64; k0 += KS) {
                    const frag db = P::ld_kc(&sm.dO[wave * 16][k0], LDH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) dp[t] = P::mma(P::ld_ks(&sm.vt[k0][t * 16], LDH), db, dp[t]);
                }
            }
            if (a.drop.thr) {
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
        // ---------------- phase C: contractions over all 64 queries / keys ----------------
        if constexpr (PF_X) {   // rows of the NEXT tile of this workgroup
            const int nt = tile + gridDim.x;
            const long tokn = nt < a.ntiles ? tm.token_sp(nt, sp_ln) : -1;
#pragma unroll
            for (int i = 0; i < 6; ++i) xv[i] = tokn >= 0 ? reinterpret_cast<const f32x4*>(a.x + tokn * 96 + lpart * 24)[i] : zero4();
        }
        if constexpr (BF && !PF_X) {   // warm the L2 with the NEXT tile's rows (the LN1 loads are the longest stall)
            const int nt = tile + gridDim.x;
            const long tokn = nt < a.ntiles ? tm.token_sp(nt, sp_ln) : -1;
            l2_touch(a.x + (tokn >= 0 ? tokn : 0) * 96 + lpart * 24, touch_pad);
        }
        // C1: dWout_h and dv (reads xd = da, o, p, dO); dv -> vt (dead since phase B)
        {
            f32x4 dv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) dv[t] = zero4();
#pragma unroll P::UNROLL
            for (int k0 = 0; k0 < 64; k0 += KS) {
                // dWout_h[m][d] += sum_q da[q][m] o[q][d]      C[i = m tile t][j = d tile wave]
                const frag ob = P::ld_ks(&sm.o[k0][wave * 16], LDH);
#pragma unroll
                for (int t = 0; t < 6; ++t) go[t] = P::mma(P::ld_ks(&sm.xd[k0][t * 16], LDX), ob, go[t]);
                // dv[key][d] = sum_q p[q][key] dO[q][d]         C[i = d tile t][j = key tile wave]
                const frag pb = P::ld_ks(&sm.p[k0][wave * 16], LDH);
#pragma unroll
                for (int t = 0; t < 4; ++t) dv[t] = P::mma(P::ld_ks(&sm.dO[k0][t * 16], LDH), pb, dv[t]);
            }
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
#pragma unroll P::UNROLL
            for (int k0 = 0; k0 < 64; k0 += KS) {
                // dk[key][d] = sum_q ds[q][key] q[q][d]         C[i = d tile t][j = key tile wave]
                const frag sb = P::ld_ks(&sm.ds[k0][wave * 16], LDH);
                // dq[query][d] = sum_key ds[query][key] k[key][d]   C[i = d tile t][j = query tile wave]
                const frag sq = P::ld_kc(&sm.ds[wave * 16][k0], LDH);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dk[t] = P::mma(P::ld_ks(&sm.q[k0][t * 16], LDH), sb, dk[t]);
                    dq[t] = P::mma(P::ld_ks(&sm.k[k0][t * 16], LDH), sq, dq[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                P::st_nat(&sm.p[wave * 16][t * 16], LDH, dk[t]);   // dk[key][d]
                P::st_nat(&sm.o[wave * 16][t * 16], LDH, dq[t]);   // dq[query][d]
            }
        }
        if constexpr (KEEP_XN) {
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<s16x4*>(&sm.xd[lr][lpart * 24 + 4 * i]) = xnk[i];
        } else {
            ln1_rows_to_lds<P>(a.x, lnp, lnp + 96, tok_ln, sm.xd);
        }
        STAMP(11);
        lds_barrier();
        STAMP(12);
        if constexpr (sizeof(elem) == 2) {
            // bf16: stage this head's 36 wqkvT fragments (all four waves need all of them in the d(LN1 out)
            // GEMM) into the dead q|k|dO|ds region with async global->LDS copies, 9 per wave
            char* stage = reinterpret_cast<char*>(&sm.q[0][0]);
#pragma unroll
            for (int i9 = 0; i9 < 9; ++i9) {
                const int idx = wave * 9 + i9;                    // idx = (t * 3 + which) * 2 + ks
                const int t = idx / 6, which = (idx >> 1) % 3, ks = idx & 1;
                const int f = t * ((3 * inner) >> 5) + ((which * inner + h * 64 + ks * 32) >> 5);
                dma_frag_async(wqkvT + (long)f * 512, stage + idx * 1024);
            }
        }
        // ---------------- phase D: qkv weight grads and the head's d(LN1 out) partial ----------------
#pragma unroll P::UNROLL
        for (int k0 = 0; k0 < 64; k0 += KS) {
            // dW{q,k,v}[d][m] += sum_row d{q,k,v}[row][d] xn[row][m]    C[i = d tile wave][j = m tile t]
            const frag aq = P::ld_ks(&sm.o[k0][wave * 16], LDH);
            const frag ak = P::ld_ks(&sm.p[k0][wave * 16], LDH);
            const frag av = P::ld_ks(&sm.vt[k0][wave * 16], LDH);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const frag xb = P::ld_ks(&sm.xd[k0][t * 16], LDX);
                gq[t] = P::mma(aq, xb, gq[t]);
                gk[t] = P::mma(ak, xb, gk[t]);
                gv[t] = P::mma(av, xb, gv[t]);
            }
        }
        {
            // dxn_h[row][m] = sum_d dq Wq + dk Wk + dv Wv         C[i = m tile t][j = row tile wave]
            f32x4 dx[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) dx[t] = zero4();
            STAMP(13);
            if constexpr (sizeof(elem) == 2) wait_vm0();   // this wave's staged fragments have landed
            lds_barrier();
            STAMP(14);   // every wave is done reading xd (weight-grad loop above); staged weights visible
#pragma unroll P::UNROLL
            for (int k0 = 0; k0 < 64; k0 += KS) {
                const frag bq = P::ld_kc(&sm.o[wave * 16][k0], LDH);
                const frag bk = P::ld_kc(&sm.p[wave * 16][k0], LDH);
                const frag bv = P::ld_kc(&sm.vt[wave * 16][k0], LDH);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    if constexpr (sizeof(elem) == 2) {
                        const char* stage = reinterpret_cast<const char*>(&sm.q[0][0]) + l * 16;
                        const int ks = k0 >> 5;
                        dx[t] = P::mma(*reinterpret_cast<const frag*>(stage + ((t * 3 + 0) * 2 + ks) * 1024), bq, dx[t]);
                        dx[t] = P::mma(*reinterpret_cast<const frag*>(stage + ((t * 3 + 1) * 2 + ks) * 1024), bk, dx[t]);
                        dx[t] = P::mma(*reinterpret_cast<const frag*>(stage + ((t * 3 + 2) * 2 + ks) * 1024), bv, dx[t]);
                    } else {
                        dx[t] = P::mma(P::ld_w(wqkvT, 3 * inner, t * 16, h * 64 + k0), bq, dx[t]);
                        dx[t] = P::mma(P::ld_w(wqkvT, 3 * inner, t * 16, inner + h * 64 + k0), bk, dx[t]);
                        dx[t] = P::mma(P::ld_w(wqkvT, 3 * inner, t * 16, 2 * inner + h * 64 + k0), bv, dx[t]);
                    }
                }
            }
            STAMP(15);
            // stage the [16 x 96] result rows of this wave in xd (dead now) and write whole rows
#pragma unroll
            for (int t = 0; t < 6; ++t) P::st_nat(&sm.xd[wave * 16][t * 16], LDX, dx[t]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < CPR / 4; ++i) {   // thread <-> (row tid / 4, quarter tid % 4) as in LN1: address already known
                if (tok_ln >= 0) {
                    const int ch = lpart * (CPR / 4) + i;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(&sm.xd[lr][0]) + ch * 16);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(part + tok_ln * 96) + ch * 16) = v;
                }
            }
        }
        STAMP(16);
        // no block barrier here: the next tile's LN1 only writes this wave's own xd rows, and q/k/vt
        // are not touched before the barrier that follows it
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

template __global__ void block_bwd_attn_kernel<PF32>(AttnBwdArgs);
template __global__ void block_bwd_attn_kernel<PBF16>(AttnBwdArgs);

// ==========================================================================================
// LN1 backward + residual:  dx = dx1 + LN1_bwd(sum_h part[h]; x)   (vit_spatial_spectral.py:22-29,:102)
// persistent grid over row tiles; 8 threads x 12 features per row (bf16 mode) or 4 x 24 (fp32 mode).
// ==========================================================================================
template <class E>
__global__ __launch_bounds__(256) void block_bwd_ln1_kernel(Ln1BwdArgs a) {
    // features per thread: 12 in bf16 mode (8 threads per row, 32 rows per pass: half the registers, twice the waves
    // and bytes in flight of the 24-feature mapping -- this kernel only moves data), 24 in the fp32 parity mode
#ifndef MSST_LN1_FPT
#define MSST_LN1_FPT 12
#endif
    constexpr int FPT = sizeof(E) == 2 ? MSST_LN1_FPT : 24, TPR = 96 / FPT, ROWS = 256 / TPR, NV = FPT / 4;
    __shared__ float red[ROWS][97];
    const int tid = threadIdx.x, r = tid / TPR, part = tid % TPR;
    const E* parts = reinterpret_cast<const E*>(a.dxn_part);
    float dg[FPT], db[FPT], dbo[FPT], gam[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) { dg[i] = 0.f; db[i] = 0.f; dbo[i] = 0.f; gam[i] = a.ln1_g[part * FPT + i]; }
    const int ntiles = (a.ntok + ROWS - 1) / ROWS;
#ifndef MSST_LN1_REV
#define MSST_LN1_REV 1
#endif
    // the walk runs from the last tile to the first: the attention kernel before this one wrote its partials (and the MLP
    // kernel dx1) in ascending tile order, so the rows most likely to be still in the memory-side cache are read first
    const int tile_last = (int)blockIdx.x + ((ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x) * (int)gridDim.x;
    for (int tile_ = blockIdx.x; tile_ < ntiles; tile_ += gridDim.x) {
        const int tile = MSST_LN1_REV ? tile_last - (tile_ - (int)blockIdx.x) : tile_;
        const long tok = (long)tile * ROWS + r;
        if (tok < a.ntok) {   // the threads of a row are in/out together
            float v[FPT], dn[FPT];
            const f32x4* src = reinterpret_cast<const f32x4*>(a.x + tok * 96 + part * FPT);
#pragma unroll
            for (int i = 0; i < NV; ++i) { f32x4 t4 = src[i]; v[4*i] = t4[0]; v[4*i+1] = t4[1]; v[4*i+2] = t4[2]; v[4*i+3] = t4[3]; }
#pragma unroll
            for (int i = 0; i < FPT; ++i) dn[i] = 0.f;
            // per-head partials: four heads' row slices are requested together (one memory round trip per four
            // heads, not one per head), summed in head order
            const f32x4* d1 = reinterpret_cast<const f32x4*>(a.dx1 + tok * 96 + part * FPT);
            f32x4 d1v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) d1v[i] = d1[i];
            // widest aligned vector a head's row slice allows: 16 bytes when the slice is a multiple of 16 bytes, else 8
            constexpr bool W16 = (FPT * (int)sizeof(E)) % 16 == 0;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef typename std::conditional<W16, f32x4, f32x2>::type pvec;
            constexpr int NPV = FPT * (int)sizeof(E) / (int)sizeof(pvec);
            int h = 0;
            for (; h + 4 <= a.H; h += 4) {
                pvec raw[4][NPV];
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const pvec* p = reinterpret_cast<const pvec*>(parts + ((long)(h + hh) * a.ntok + tok) * 96 + part * FPT);
#pragma unroll
                    for (int q = 0; q < NPV; ++q) raw[hh][q] = p[q];
                }
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const E* e = reinterpret_cast<const E*>(&raw[hh][0]);
#pragma unroll
                    for (int i = 0; i < FPT; ++i) {
                        if constexpr (sizeof(E) == 4) dn[i] += e[i]; else dn[i] += bf2f(e[i]);
                    }
                }
            }
            for (; h < a.H; ++h) {
                const E* p = parts + ((long)h * a.ntok + tok) * 96 + part * FPT;
#pragma unroll
                for (int i = 0; i < FPT; ++i) {
                    if constexpr (sizeof(E) == 4) dn[i] += p[i]; else dn[i] += bf2f(p[i]);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < FPT; ++i) s += v[i];
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) s += __shfl_xor(s, m);
            const float mean = s * (1.f / 96.f);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < FPT; ++i) { const float d = v[i] - mean; vs += d * d; }
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) vs += __shfl_xor(vs, m);
            const float rstd = rsqrtf(vs * (1.f / 96.f) + 1e-5f);
            float g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int i = 0; i < FPT; ++i) {
                const float xh = (v[i] - mean) * rstd;
                v[i] = xh;
                dg[i] += dn[i] * xh;
                db[i] += dn[i];
                dn[i] *= gam[i];
                g1 += dn[i];
                g2 += dn[i] * xh;
            }
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) { g1 += __shfl_xor(g1, m); g2 += __shfl_xor(g2, m); }
            g1 *= (1.f / 96.f); g2 *= (1.f / 96.f);
            f32x4* dst = reinterpret_cast<f32x4*>(a.dx + tok * 96 + part * FPT);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 t4 = d1v[i], o4;
                f32x4 tm4 = t4;
                if (a.drop.thr) tm4 = drop4(a.drop, 2, (unsigned)(tok * 24 + part * NV + i), tm4);   // site 2 backward
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o4[e] = t4[e] + rstd * (dn[4*i+e] - g1 - v[4*i+e] * g2);
                    dbo[4*i+e] += tm4[e];   // to_out bias grad = column sum of d(attention output before dropout)
                }
                dst[i] = o4;
            }
        }
    }
    // reduce dgamma / dbeta / d(to_out bias) over the row-threads
    float* slab = a.slab + (long)blockIdx.x * 288;
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FPT; ++i) red[r][part * FPT + i] = which == 0 ? dg[i] : which == 1 ? db[i] : dbo[i];
        __syncthreads();
        if (tid < 96) {
            float s = 0.f;
            for (int rr = 0; rr < ROWS; ++rr) s += red[rr][tid];
            slab[which * 96 + tid] = s;
        }
    }
}

// ==========================================================================================
// tokenizer backward (forward: tokenize_fwd_kernel).  grid (S, nchunk), 256 threads.
// slab per (c, chunk): [dpos N*96 | dW 96*P | db 96 | dpost_g 96 | dpost_b 96 | dmask 96 | dpre_g 16 | dpre_b 16]
// ==========================================================================================
// PC: pixels per patch as a compile-time constant (10; 0 = run-time value), see tokenize_fwd_kernel
template <int PC>
__global__ __launch_bounds__(256) void tokenize_bwd_kernel(TokBwdArgs a) {
    __shared__ float patch[16][64];
    __shared__ float W[96][17];
    __shared__ float de_s[64][97];
    __shared__ float xn_s[64][17];
    __shared__ float bias[96];
    __shared__ float postg[96], preg[16], preb[16];   // small parameter vectors read inside the batch walk: from LDS, not through L2
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int P = PC ? PC : a.P, N = a.N, T = a.T;
    for (int i = tid; i < 96 * P; i += 256) W[i / P][i % P] = a.w_emb[(long)c * 96 * P + i];
    if (tid < 96) { bias[tid] = a.b_emb[c * 96 + tid]; postg[tid] = a.post_g[tid]; }
    if (tid < 16) { preg[tid] = tid < P ? a.pre_g[tid] : 0.f; preb[tid] = tid < P ? a.pre_b[tid] : 0.f; }
    const int n = tid >> 2, part = tid & 3;
    // this thread's 24 features, as in tokenize_fwd_kernel: the four threads of a token read 64 contiguous bytes per instruction
    auto feat = [&](int i) { return 16 * (i >> 2) + 4 * part + (i & 3); };
    const bool active = n < N;
    // (the mask-token gradient of this thread's slots is dpos - dpb: every token adds its dt to dpos, the unmasked ones also
    // to dpb -- one accumulator set less, the kernel sits at the 256-register limit)
    float dpos[24], dpg[24], dpb[24], dbc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) { dpos[i] = 0.f; dpg[i] = 0.f; dpb[i] = 0.f; dbc[i] = 0.f; }
    float dpre_g[16], dpre_b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { dpre_g[k] = 0.f; dpre_b[k] = 0.f; }
    // dW accumulators: thread owns pairs (d, k) = linear index tid + 256*j < 96*P
    float dWacc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) dWacc[j] = 0.f;

    for (int b = chunk; b < a.B; b += gridDim.y) {
        __syncthreads();
        const float* src = a.img + ((long)b * a.S + c) * P * N;
        for (int i = tid; i < P * N; i += 256) patch[i / N][i % N] = src[i];
        __syncthreads();
        // this thread's dx0 slice, requested unconditionally (clamped row) and all six at once: inside `if (active)` every
        // load was followed by its own s_waitcnt vmcnt(0)
        f32x4 drow[6];
        {
            const int tc = c * N + (active ? n : 0);
            const float* dsrc = a.dx0 + ((long)b * T + tc) * 96 + part * 4;
#pragma unroll
            for (int i = 0; i < 6; ++i) drow[i] = *reinterpret_cast<const f32x4*>(dsrc + 16 * i);
        }
        if (active) {
            const int t = c * N + n;
            const bool masked = a.mask[(long)b * T + t] != 0;
            float dt[24];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x4 t4 = drow[i];
                if (a.drop.thr) t4 = drop4(a.drop, 0, (unsigned)(((long)b * T + t) * 24 + 4 * i + part), t4);   // emb dropout backward (group = feature / 4)
                dt[4*i] = t4[0]; dt[4*i+1] = t4[1]; dt[4*i+2] = t4[2]; dt[4*i+3] = t4[3];
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) dpos[i] += dt[i];
            // recompute
            float xh0[16], xn[16];
            float mean = 0.f;
#pragma unroll
            for (int k = 0; k < P; ++k) mean += patch[k][n];
            mean /= P;
            float var = 0.f;
#pragma unroll
            for (int k = 0; k < P; ++k) { const float d = patch[k][n] - mean; var += d * d; }
            const float rstd = rsqrtf(var / P + 1e-5f);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < P) { xh0[k] = (patch[k][n] - mean) * rstd; xn[k] = xh0[k] * preg[k] + preb[k]; }
                else { xh0[k] = 0.f; xn[k] = 0.f; }
            }
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
            // masked tokens: gradient goes to the mask token only (and the position table, above)
            float g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int d = feat(i);
                const float eh = (e[i] - m2) * rstd2;
                e[i] = eh;
                if (masked) dt[i] = 0.f;
                dpg[i] += dt[i] * eh;
                dpb[i] += dt[i];
                dt[i] *= postg[d];
                g1 += dt[i];
                g2 += dt[i] * eh;
            }
            g1 += __shfl_xor(g1, 1); g1 += __shfl_xor(g1, 2);
            g2 += __shfl_xor(g2, 1); g2 += __shfl_xor(g2, 2);
            g1 *= (1.f / 96.f); g2 *= (1.f / 96.f);
            float dxn[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) dxn[k] = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int d = feat(i);
                const float de = rstd2 * (dt[i] - g1 - e[i] * g2);
                dbc[i] += de;
                de_s[n][d] = de;
#pragma unroll
                for (int k = 0; k < P; ++k) dxn[k] += W[d][k] * de;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float v = dxn[k];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
                if (part == 0 && k < P) { dpre_g[k] += v * xh0[k]; dpre_b[k] += v; xn_s[n][k] = xn[k]; }
            }
        }
        __syncthreads();
        // dW[d][k] += sum_n de[n][d] * xn[n][k]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int idx = tid + 256 * j;
            if (idx < 96 * P) {
                const int d = idx / P, k = idx - d * P;
                float s = 0.f;
                for (int nn = 0; nn < N; ++nn) s += de_s[nn][d] * xn_s[nn][k];
                dWacc[j] += s;
            }
        }
    }
    // ---------------- slab ----------------
    const int slab_n = N * 96 + 96 * P + 96 * 4 + 32;
    float* slab = a.slab + ((long)c * gridDim.y + chunk) * slab_n;
    if (active) {
#pragma unroll
        for (int i = 0; i < 24; ++i) slab[n * 96 + feat(i)] = dpos[i];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int idx = tid + 256 * j;
        if (idx < 96 * P) slab[N * 96 + idx] = dWacc[j];
    }
    float* vec = slab + N * 96 + 96 * P;
    for (int which = 0; which < 4; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const float v = which == 0 ? dbc[i] : which == 1 ? dpg[i] : which == 2 ? dpb[i] : dpos[i] - dpb[i];
            de_s[n][feat(i)] = active ? v : 0.f;
        }
        __syncthreads();
        if (tid < 96) {
            float s = 0.f;
            for (int rr = 0; rr < 64; ++rr) s += de_s[rr][tid];
            vec[which * 96 + tid] = s;
        }
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { de_s[n][k] = active ? dpre_g[k] : 0.f; de_s[n][16 + k] = active ? dpre_b[k] : 0.f; }
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int rr = 0; rr < 64; ++rr) s += de_s[rr][tid];
        vec[4 * 96 + tid] = s;
    }
}


// ------------------------------------------------------------------------------------------
// The same backward for the reference's shapes (P = 10, N = 64) on the fp32 matrix cores (round 4; forward:
// tokenize_fwd_mfma_kernel).  The kernel above spends its time in LDS reads of the [96][P] weight (one per multiply-add, in
// the forward recompute and in d(xn) = W^T de) and in a 64-deep serial loop per dW element: 230 us for 184 MB.  Here
//   * wave w <-> tokens 16 w .. + 15 of spectral block c, persistent over the samples of its chunk: the position-gradient
//     rows of those tokens accumulate in registers with no reduction at all;
//   * forward recompute e = W xn + b: 18 MFMAs (weights as A operand from LDS-resident fragments);
//   * d(xn)[k][token] = sum_d W[d][k] de[d][token]: 24 MFMAs whose B operand is the C-layout de as it stands (the contraction
//     index runs over the features in the order the C layout hands them over; the A operand W^T is fetched in the same order);
//   * dW[d][k] += sum_token de[d][token] xn[token][k]: 24 MFMAs over the wave's 16 tokens, both operands transposed through a
//     wave-private LDS tile; a column of ones appended to xn makes dW[d][10] the bias gradient.
// Slab layout as above.  grid (S, nchunk), 256 threads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void tokenize_bwd_mfma_kernel(TokBwdArgs a) {
    constexpr int P = 10, N = 64;
    __shared__ float wA[6][3][64];        // forward A fragments: [mt][ks][lane] = W[16 mt + (lane & 15)][4 ks + (lane >> 4)]
    __shared__ float wT[6][4][64];        // d(xn) A fragments: [mt][r][lane] = W[16 mt + 4 (lane >> 4) + r][lane & 15] (0 for k >= 10)
    __shared__ float vecs[2][96];         // bias | post_g
    __shared__ float de_raw[4 * 96 * 17];  // wave-private de[feature][token] tiles in the walk; the epilogue's reduction array afterwards
    __shared__ float xn_t[4][16][17];     // wave-private: xn[k][token] (row 10 = ones)
    float (*de_t)[96][17] = reinterpret_cast<float (*)[96][17]>(de_raw);
    float (*red)[97] = reinterpret_cast<float (*)[97]>(de_raw);   // [64][97] = 6208 floats <= 6528
    const int c = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, j = l & 15;
    const int T = a.T;
    const float* Wc = a.w_emb + (long)c * 96 * P;
    for (int i = tid; i < 6 * 3 * 64; i += 256) {
        const int ln = i & 63, ks = (i >> 6) % 3, mt = i / 192, k = 4 * ks + (ln >> 4);
        (&wA[0][0][0])[i] = k < P ? Wc[(16 * mt + (ln & 15)) * P + k] : 0.f;
    }
    for (int i = tid; i < 6 * 4 * 64; i += 256) {
        const int ln = i & 63, r = (i >> 6) & 3, mt = i >> 8, k = ln & 15;
        (&wT[0][0][0])[i] = k < P ? Wc[(16 * mt + 4 * (ln >> 4) + r) * P + k] : 0.f;
    }
    if (tid < 96) { vecs[0][tid] = a.b_emb[c * 96 + tid]; vecs[1][tid] = a.post_g[tid]; }
    float pg[3], pb[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; pg[ks] = k < P ? a.pre_g[k] : 0.f; pb[ks] = k < P ? a.pre_b[k] : 0.f; }
    __syncthreads();
    const int n = 16 * w + j, t = c * N + n;
    // persistent accumulators (C layout: lane (token j, g) <-> features 16 mt + 4 g + r)
    f32x4 dpos[6], dpg[6], dpb[6], dW[6];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) { dpos[mt] = zero4(); dpg[mt] = zero4(); dpb[mt] = zero4(); dW[mt] = zero4(); }
    f32x4 dprg = zero4(), dprb = zero4();   // pre_norm gamma / beta gradients of pixels k = 4 g + r (this lane's token)
    const int nb = (int)gridDim.y;
    float px[3], px2[4];
    f32x4 drow[6];
    unsigned char mk;
    auto request = [&](int b) {
        const int bc = b < a.B ? b : a.B - 1;
        const float* src = a.img + ((long)bc * a.S + c) * P * N + n;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; px[ks] = src[(k < P ? k : 0) * N]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int k = 4 * g + r; px2[r] = src[(k < P ? k : 0) * N]; }
        mk = a.mask[(long)bc * T + t];
    };
    // the dx0 rows of the next sample are requested late in the iteration (behind the LayerNorm backward: 24 registers that are
    // free only then), the pixels and the mask byte early
    auto request_rows = [&](int b) {
        const int bc = b < a.B ? b : a.B - 1;
        const float* dsrc = a.dx0 + ((long)bc * T + t) * 96 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) drow[mt] = *reinterpret_cast<const f32x4*>(dsrc + 16 * mt);
    };
    request(chunk);
    request_rows(chunk);
    for (int b = chunk; b < a.B; b += nb) {
        // lane indices re-derived from a laundered id every iteration: the weight fragments, the bias and post_norm's gamma are
        // loop-invariant LDS reads -- with an invariant address they are hoisted out of the walk (90 registers) and spilled
        int ll = threadIdx.x & 63;
        asm volatile("" : "+v"(ll));
        const int gl = ll >> 4, jl = ll & 15;
        float x[3], x2[4];
        f32x4 dt[6];
        const bool masked = mk != 0;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) x[ks] = (4 * ks + g < P) ? px[ks] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) x2[r] = px2[r];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            dt[mt] = drow[mt];
            if (a.drop.thr) dt[mt] = drop4(a.drop, 0, (unsigned)(((long)b * T + t) * 24 + 4 * mt + g), dt[mt]);   // emb dropout backward
        }
        request(b + nb);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) dpos[mt] = dpos[mt] + dt[mt];
        // ---- recompute: LN(10), Linear(10 -> 96), LN(96) statistics ----
        const float mean = colgroup_sum(x[0] + x[1] + x[2]) / P;
        float var = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const float d = (4 * ks + g < P) ? x[ks] - mean : 0.f; var += d * d; }
        const float rstd = rsqrtf(colgroup_sum(var) / P + 1e-5f);
        float xn[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) xn[ks] = (4 * ks + g < P) ? (x[ks] - mean) * rstd * pg[ks] + pb[ks] : 0.f;
        f32x4 e[6];
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(&vecs[0][16 * mt + 4 * gl]);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[mt][ks][ll], xn[ks], acc, 0, 0, 0);
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
        // ---- LN(96) backward; masked tokens: the gradient goes to the mask token only (and the position table, above) ----
        float g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const f32x4 pg4 = *reinterpret_cast<const f32x4*>(&vecs[1][16 * mt + 4 * gl]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float eh = (e[mt][r] - m2) * rstd2;
                e[mt][r] = eh;
                const float d = masked ? 0.f : dt[mt][r];
                dpg[mt][r] += d * eh;
                dpb[mt][r] += d;
                const float dg = d * pg4[r];
                dt[mt][r] = dg;
                g1 += dg;
                g2 += dg * eh;
            }
        }
        g1 = colgroup_sum(g1) * (1.f / 96.f);
        g2 = colgroup_sum(g2) * (1.f / 96.f);
        // de[feature][token] (C layout) -> the wave's LDS tile; d(xn) = W^T de on the matrix cores
        float* det = &de_t[w][4 * gl][jl];
        f32x4 dxn = zero4();   // C[i = pixel k = 4 g + r][j = token]
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float de = rstd2 * (dt[mt][r] - g1 - e[mt][r] * g2);
                det[(16 * mt + r) * 17] = de;
                dxn = __builtin_amdgcn_mfma_f32_16x16x4f32(wT[mt][r][ll], de, dxn, 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) xn_t[w][4 * ks + gl][jl] = (4 * ks + gl == P) ? 1.0f : xn[ks];   // row 10: ones -> bias gradient
        request_rows(b + nb);
        // pre_norm gamma / beta: pixels k = 4 g + r of this lane's token
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = 4 * g + r < P;
            const float xh = ok ? (x2[r] - mean) * rstd : 0.f;
            dprg[r] += ok ? dxn[r] * xh : 0.f;
            dprb[r] += ok ? dxn[r] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dW[d][k] += sum over the wave's 16 tokens: A[i = feature][kk = token], B[j = pixel k][kk = token]
        const float* dea = &de_t[w][jl][gl];
        const float* xna = &xn_t[w][jl][gl];
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) {
            const float bx = xna[4 * ts];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) dW[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dea[16 * mt * 17 + 4 * ts], bx, dW[mt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---------------- slab ----------------
    const int slab_n = N * 96 + 96 * P + 96 * 4 + 32;
    float* slab = a.slab + ((long)c * gridDim.y + chunk) * slab_n;
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) *reinterpret_cast<f32x4*>(slab + n * 96 + 16 * mt + 4 * g) = dpos[mt];
    float* vec = slab + N * 96 + 96 * P;
    // dW: C[i = feature 16 mt + 4 g + r][j = pixel k]; the four waves' tiles are summed through LDS (column k = 10 is the bias gradient)
    __syncthreads();
    for (int which = 0; which < 4; ++which) {   // the four waves take turns adding their dW tile into red[feature][k]
        if (w == which) {
#pragma unroll
            for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* cell = &red[0][0] + (16 * mt + 4 * g + r) * 17 + j;
                    *cell = which == 0 ? dW[mt][r] : *cell + dW[mt][r];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < 96 * P; i += 256) { const int d = i / P, k = i - d * P; slab[N * 96 + i] = (&red[0][0])[d * 17 + k]; }
    if (tid < 96) vec[tid] = (&red[0][0])[tid * 17 + P];   // db
    // per-feature column sums over the 64 tokens: dpost_g, dpost_b, dmask = sum(dpos) - dpost_b
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[n][16 * mt + 4 * g + r] = which == 0 ? dpg[mt][r] : which == 1 ? dpb[mt][r] : dpos[mt][r] - dpb[mt][r];
        __syncthreads();
        if (tid < 96) {
            float sres = 0.f;
            for (int rr = 0; rr < 64; ++rr) sres += red[rr][tid];
            vec[(which + 1) * 96 + tid] = sres;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[n][4 * g + r] = dprg[r]; red[n][16 + 4 * g + r] = dprb[r]; }
    __syncthreads();
    if (tid < 32) {
        float sres = 0.f;
        for (int rr = 0; rr < 64; ++rr) sres += red[rr][tid];
        vec[4 * 96 + tid] = sres;
    }
}

// position-table gradient for the factorised (spectral_pos_embed) table:
// dpos_embed[n][0:split] = sum_c dpos[c][n][0:split]; dchannel_embed[c][0:96-split] = sum_n dpos[c][n][split:96]
__global__ __launch_bounds__(256) void pos_split_kernel(const float* dpos /*[S][N][96]*/, int S, int N, int split,
                                                        float* dpe, float* dce) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int npe = N * split, nce = S * (96 - split);
    if (i < npe) {
        const int n = i / split, d = i - n * split;
        float s = 0.f;
        for (int c = 0; c < S; ++c) s += dpos[((long)c * N + n) * 96 + d];
        dpe[i] = s;
    } else if (i < npe + nce) {
        const int j = i - npe, c = j / (96 - split), d = j - c * (96 - split);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dpos[((long)c * N + n) * 96 + split + d];
        dce[j] = s;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_head_bwd(const HeadBwdArgs& a, int nchunk, hipStream_t st) {
    if (a.P > 16 || a.N > 64) return MSST_ERR_UNSUPPORTED;
    ProfScope ps(K_HEAD_BWD, st);
    if (a.P == 10 && a.N == 64) hipLaunchKernelGGL(head_bwd_mfma_kernel, dim3(a.S, nchunk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(head_bwd_kernel, dim3(a.S, nchunk), dim3(384), 0, st, a);
    return (int)hipGetLastError();
}

int launch_reduce_segs(const RSegs& r, hipStream_t st) {
    if (r.nseg <= 0 || r.nblocks <= 0) return 0;
    ProfScope ps(K_REDUCE, st);
    hipLaunchKernelGGL(reduce_segs_kernel, dim3(r.nblocks, r.ny_max > 1 ? r.ny_max : 1), dim3(256), 0, st, r);
    return (int)hipGetLastError();
}

template <class K>
static int set_smem(K kernel, size_t smem, std::atomic<bool>& done) {
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

int launch_block_bwd_mlp(const MlpBwdArgs& a, int grid, int prec, hipStream_t st) {
    static std::atomic<bool> d0{false}, d1{false}, d2{false};
    if (prec == MSST_PREC_F32) {
        if (a.x1_bf16) return MSST_ERR_UNSUPPORTED;
        const size_t smem = sizeof(MlpBwdSmem<PF32>) + 256 * sizeof(float);
        int rc = set_smem(&block_bwd_mlp_kernel<PF32>, smem, d0);
        if (rc) return rc;
        ProfScope ps(K_BWD_MLP, st);
        hipLaunchKernelGGL(block_bwd_mlp_kernel<PF32>, dim3(grid), dim3(256), smem, st, a);
    } else {
        const size_t smem = sizeof(MlpBwdSmem<PBF16>) + 256 * sizeof(float) + (PBF16::WAVES_BWD_MLP == 2 ? 24 : 36) * 1024;
        int rc = set_smem(&block_bwd_mlp_kernel<PBF16, false>, smem, d1);
        if (!rc) rc = set_smem(&block_bwd_mlp_kernel<PBF16, true>, smem, d2);
        if (rc) return rc;
        ProfScope ps(K_BWD_MLP, st);
        if (a.x1_bf16) hipLaunchKernelGGL((block_bwd_mlp_kernel<PBF16, true>), dim3(grid), dim3(256), smem, st, a);
        else hipLaunchKernelGGL((block_bwd_mlp_kernel<PBF16, false>), dim3(grid), dim3(256), smem, st, a);
    }
    return (int)hipGetLastError();
}

int launch_block_bwd_attn(const AttnBwdArgs& a, int nchunk, int prec, hipStream_t st, int* nparts) {
    static std::atomic<bool> d0{false}, d1{false};
    if (a.tm.L > 64 || a.tm.L < 1) return MSST_ERR_UNSUPPORTED;
    *nparts = a.H;   // d(LN1 out) partials written: one per head, or one per head pair (two-head kernel)
    dim3 grid(nchunk, a.H);
    if (prec == MSST_PREC_F32) {
        const size_t smem = sizeof(AttnBwdSmem<PF32>) + 192 * sizeof(float) + 256;   // LN vectors + the l2_touch pad
        int rc = set_smem(&block_bwd_attn_kernel<PF32>, smem, d0);
        if (rc) return rc;
        ProfScope ps(K_BWD_ATTN, st);
        hipLaunchKernelGGL(block_bwd_attn_kernel<PF32>, grid, dim3(256), smem, st, a);
    } else if (!(a.dbg & 16) && a.xn && a.dab && a.w.wqkv32 && a.w.woutT32 && a.w.wqkvT32 && a.ntok * 192 < 0x7ffffff0L) {
        // the tuned kernels (they need the LN1 rows saved by the forward and the pre-dropped bf16 da rows of the MLP half): two
        // heads per workgroup (msst_bwd4.hip) by default, one head per workgroup (msst_bwd3.hip) for an odd head count or with
        // MSST_DBG=128.  MSST_DBG=16, or a caller without saved rows: the template kernel below (the reference they are tested
        // against).  (The round-2 kernel msst_bwd2.hip was retired in round 4.)
        if (!(a.dbg & 128) && !(a.H & 1)) { *nparts = a.H / 2; return launch_block_bwd_attn_r4(a, nchunk, st); }
        return launch_block_bwd_attn_r3(a, nchunk, st);
    } else {
        const size_t smem = sizeof(AttnBwdSmem<PBF16>) + 192 * sizeof(float) + 256;   // LN vectors + the l2_touch pad
        int rc = set_smem(&block_bwd_attn_kernel<PBF16>, smem, d1);
        if (rc) return rc;
        ProfScope ps(K_BWD_ATTN, st);
        hipLaunchKernelGGL(block_bwd_attn_kernel<PBF16>, grid, dim3(256), smem, st, a);
    }
    return (int)hipGetLastError();
}

int launch_block_bwd_ln1(const Ln1BwdArgs& a, int grid, int prec, hipStream_t st) {
    ProfScope ps(K_BWD_LN1, st);
    if (prec == MSST_PREC_F32) hipLaunchKernelGGL(block_bwd_ln1_kernel<float>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(block_bwd_ln1_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int launch_tokenize_bwd(const TokBwdArgs& a, int nchunk, hipStream_t st) {
    if (a.P > 16 || a.N > 64) return MSST_ERR_UNSUPPORTED;
    ProfScope ps(K_TOK_BWD, st);
    if (a.P == 10 && a.N == 64) hipLaunchKernelGGL(tokenize_bwd_mfma_kernel, dim3(a.S, nchunk), dim3(256), 0, st, a);
    else if (a.P == 10) hipLaunchKernelGGL(tokenize_bwd_kernel<10>, dim3(a.S, nchunk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tokenize_bwd_kernel<0>, dim3(a.S, nchunk), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int launch_pos_split(const float* dpos, int S, int N, int split, float* dpe, float* dce, hipStream_t st) {
    const int n = N * split + S * (96 - split);
    ProfScope ps(K_POS_SPLIT, st);
    hipLaunchKernelGGL(pos_split_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dpos, S, N, split, dpe, dce);
    return (int)hipGetLastError();
}

}  // namespace msst

namespace msst {

// ==========================================================================================
// classification head backward (forward: cls_head_fwd_kernel).  grid (B), 256 threads.
// slab per sample: [dW NC*96 | db NC | dgamma 96 | dbeta 96]; dy [B][T][96] fully written
// (every spectral block of a position receives d(mean) / S).
// ==========================================================================================
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(ClsBwdArgs a) {
    __shared__ float xn_s[64][97];
    __shared__ float dl_s[64][33];
    const int b = blockIdx.x, tid = threadIdx.x, n = tid >> 2, part = tid & 3;
    const int NC = a.NC;
    const bool active = n < a.N;
    float dg[24], db[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    if (active) {
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
        float dxn[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) { m[i] = (m[i] - mean) * rstd; dxn[i] = 0.f; }   // m = xhat
        for (int k = 0; k < NC; ++k) {
            const float dl = a.dlogits[((long)b * NC + k) * a.N + n];
            if (part == 0) dl_s[n][k] = dl;
#pragma unroll
            for (int i = 0; i < 24; ++i) dxn[i] += dl * a.w[k * 96 + part * 24 + i];
        }
        float g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int d = part * 24 + i;
            xn_s[n][d] = m[i] * a.ln_g[d] + a.ln_b[d];
            dg[i] = dxn[i] * m[i];
            db[i] = dxn[i];
            dxn[i] *= a.ln_g[d];
            g1 += dxn[i];
            g2 += dxn[i] * m[i];
        }
        g1 += __shfl_xor(g1, 1); g1 += __shfl_xor(g1, 2);
        g2 += __shfl_xor(g2, 1); g2 += __shfl_xor(g2, 2);
        g1 *= (1.f / 96.f); g2 *= (1.f / 96.f);
        f32x4 o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[i][e] = rstd * (dxn[4*i+e] - g1 - m[4*i+e] * g2) * invS;
        for (int c = 0; c < a.S; ++c) {
            f32x4* dst = reinterpret_cast<f32x4*>(a.dy + ((long)b * a.T + c * a.N + n) * 96 + part * 24);
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[i] = o[i];
        }
    }
    __syncthreads();
    float* slab = a.slab + (long)b * (NC * 96 + NC + 192);
    for (int idx = tid; idx < NC * 96; idx += 256) {
        const int k = idx / 96, d = idx - k * 96;
        float s = 0.f;
        for (int nn = 0; nn < a.N; ++nn) s += dl_s[nn][k] * xn_s[nn][d];
        slab[idx] = s;
    }
    if (tid < NC) {
        float s = 0.f;
        for (int nn = 0; nn < a.N; ++nn) s += dl_s[nn][tid];
        slab[NC * 96 + tid] = s;
    }
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 24; ++i) xn_s[n][part * 24 + i] = active ? (which ? db[i] : dg[i]) : 0.f;
        __syncthreads();
        if (tid < 96) {
            float s = 0.f;
            for (int nn = 0; nn < 64; ++nn) s += xn_s[nn][tid];
            slab[NC * 96 + NC + which * 96 + tid] = s;
        }
    }
}

int launch_cls_head_bwd(const ClsBwdArgs& a, hipStream_t st) {
    if (a.N > 64 || a.NC > 32) return MSST_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cls_head_bwd_kernel, dim3(a.B), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace msst
