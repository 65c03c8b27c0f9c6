// Internal kernel argument blocks + launcher prototypes (C++ side of the C-ABI in include/msst.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msst_dev.h"

#define MSST_PREC_F32 0
#define MSST_PREC_BF16 1
#define MSST_ERR_UNSUPPORTED (-2)
#define MSST_ERR_BADARG (-3)

namespace msst {

// Per-block weights.  The big matrices are "prepped" copies in the operand element type
// (fp32 or bf16), each also in transposed form for the backward GEMMs; the small vectors are fp32.
struct BlockWeights {
    const void* wqkv;   // [3*H*64][96]   rows: q heads | k heads | v heads (reference chunk order)
    const void* wout;   // [96][H*64]
    const void* w1;     // [64][96]
    const void* w2;     // [96][64]
    const void* wqkvT;  // [96][3*H*64]
    const void* woutT;  // [H*64][96]
    const void* w1T;    // [96][64]
    const void* w2T;    // [64][96]
    const float* ln1_g; const float* ln1_b; const float* bo;
    const float* ln2_g; const float* ln2_b; const float* b1; const float* b2;
    const void* wqkv32; const void* woutT32; const void* wqkvT32;   // bf16, 32 x 16 fragment packing (optional)
};

struct BlockArgs {
    BlockWeights w;
    const float* x;   // [tokens][96] block input (residual stream, fp32)
    float* y;         // [tokens][96] block output
    float* x1;        // [tokens][96] mid-block residual (x + attn), saved for the backward; may be null
    int x1_bf16;      // MSST_X1_BF16: x1 holds bf16 (role-split forward only)
    int half;         // MSST_FWD_HALF: the forward's GEMM operands are IEEE half (w.wqkv / wout / w1 / w2 then point at the half copies); role-split forward only
    void* xn_out;     // optional [tokens][96] bf16: LN1(x) exactly as the block used it, saved for the attention backward (head-per-wave kernel only)
    float* lse_out;   // optional [ntiles][H][64] fp32: per (tile, head, row) log2 of the softmax denominator of the scaled scores, max folded in
                      // (p = exp2(s scale log2 e - lse)): the attention backward then skips max / sum / 1 / x (role-split kernel only)
    TileMap tm;
    int ntiles, max_grid, H;
    float scale;      // dim_head^-0.5
    int dbg;          // ablation switches for kernel studies (0 in production)
    unsigned long long* stamps;  // dbg & 8: s_memtime stamps of one wave (kernel studies)
    Drop drop;
};

// A run of blocks of ONE stack (same mode) as one launch of the role-split forward (msst_fwd3.hip, STACK): per-block operands
#define MSST_MAX_STACK 16
struct StackBlk {
    const void* wqkv; const void* wout; const void* w1; const void* w2;
    const float* ln1_g; const float* ln1_b; const float* bo; const float* ln2_g; const float* ln2_b; const float* b1; const float* b2;
    const float* x; float* y; float* x1; void* xn_out; float* lse_out;
    int layer, pad_;
};
struct StackStride {   // byte distance of every per-block operand from one block of the run to the next (the caller's arrays are affine in the block index)
    int wqkv, wout, w1, w2, ln1_g, ln1_b, bo, ln2_g, ln2_b, b1, b2, x, y, x1, xn_out, lse_out;
};
struct StackArgs {
    BlockArgs base;   // what does not depend on the block: tile map, heads, scale, dropout stream (its `layer` is per block below), x1_bf16
    int nblk;
    StackBlk b0;      // block 0 of the run
    StackStride st;
    const float* x_rest;   // y of block 0 minus one y stride: block j >= 1 reads x_rest + j x st.y
};
static_assert(sizeof(StackArgs) <= 4096, "StackArgs travels as a kernel argument");

struct TokArgs {
    const float* img;        // [B][S*P][N]
    const float* pre_g; const float* pre_b;     // [P]
    const float* w_emb;      // [S][96][P]
    const float* b_emb;      // [S][96]
    const float* post_g; const float* post_b;   // [96]
    const float* pos_a;      // pos_split == 0: learned table [T][96]; else spatial table [N][pos_split]
    const float* pos_b;      // pos_split != 0: spectral table [S][96 - pos_split]
    const float* mask_token; // [96]
    const uint8_t* mask;     // [B][T] (1 = masked); all-zero for the classification path
    float* out;              // [B][T][96]
    int B, S, N, T, P, pos_split;
    Drop drop;               // embedding dropout on (token + pos) (vit_spatial_spectral.py:530), classification path only
};

struct HeadArgs {
    const float* y;      // [B][T][96] encoder output
    const float* img;    // [B][S*P][N]
    const int* idx;      // [B][K] masked token indices
    const float* w_pix;  // [S or 1][P][96]
    const float* b_pix;  // [S or 1][P]
    float* dpred;        // [B][K][P] sign(pred - target)
    float* pred;         // optional [B][K][P]
    float* partial;      // [B * ceil(K/64)]
    int B, S, N, T, P, K, per_block;
};

struct HeadBwdArgs {
    const float* y; const float* dpred; const int* csr_ptr; const int* csr_pos; const float* w_pix;
    float* dy; float* slab; float gscale;
    const float* gout;   // optional device scalar: d(final)/d(loss), multiplied into gscale
    int B, S, N, T, P, K, per_block;
};

#define MSST_MLP_SLAB_N (64 * 96 + 96 * 64 + 64 + 96 + 96 + 96)
#define MSST_ATTN_SLAB_N (3 * 64 * 96 + 96 * 64)

struct MlpBwdArgs {
    BlockWeights w;
    const float* x1; const float* dy; float* dx1; float* slab;
    void* dab;   // optional [tokens][96] bf16: dx1 with the to_out dropout (site 2) applied, packed -- what the bf16 attention backward feeds its MFMAs
    long ntok;
    Drop drop;
    int x1_bf16;  // MSST_X1_BF16: x1 holds bf16 (bf16 kernel only)
};

struct AttnBwdArgs {
    BlockWeights w;
    const float* x; const float* da; void* dxn_part; float* slab;
    const void* dab;  // optional [tokens][96] bf16 pre-dropped da rows written by the MLP half (given together with xn)
    const void* xn;   // optional [tokens][96] bf16 LN1(x) saved by the forward: the bf16 kernel then neither re-reads x nor renormalises
    TileMap tm;
    int ntiles, H;
    long ntok;
    float scale;
    int dbg;
    unsigned long long* stamps;
    Drop drop;
    int* queue;       // optional (two-head kernel): one zeroed counter per head pair -> dynamic tile queue instead of the static partition
    const float* lse; // optional (two-head kernel): [ntiles][H][64] saved by the forward (BlockArgs.lse_out)
    int lse_renorm;   // MSST_LSE_RENORM: the forward's scores are not this kernel's (half-operand forward): exp2(s c - lse) rows are renormalised by their own sum
};

struct Ln1BwdArgs {
    const float* x; const float* dx1; const void* dxn_part; const float* ln1_g; float* dx; float* slab;
    long ntok;
    int H;
    Drop drop;
};

// LN1 backward of block i fused with the MLP-half backward of block i - 1 (msst_bwd5.hip)
struct LnMlpArgs {
    BlockWeights w;          // block i - 1 (MLP half: w1, w1T, w2T, ln2_g, ln2_b, b1)
    const float* ln1_g;      // block i
    const float* ln1_b;      // block i (xn path only)
    const void* xn;          // optional: [tokens][96] bf16 LN1 rows of block i as its forward used them, with
    const float* rstd;       //           [tokens] rstd of that LN1: xhat = (xn - ln1_b) / ln1_g replaces the read + renormalisation of x
    const float* x;          // [tokens][96] input of block i (= output of block i - 1)
    float* dx1;              // in: dx1 of block i (gradient at its mid residual); out: dx1 of block i - 1 (same rows, in place)
    const void* dxn_part;    // [nparts][tokens][96] bf16 partial d(LN1 out) of block i
    const float* x1;         // [tokens][96] mid residual of block i - 1 (fp32, or bf16 with x1_bf16)
    int x1_bf16;
    void* dab;               // out: [tokens][96] bf16, dx1 of block i - 1 with the to_out dropout applied (attention half's operand)
    float* slab_mlp;         // [grid][MSST_MLP_SLAB_N]
    float* slab_ln1;         // [grid][288]
    long ntok;
    int nparts;
    Drop drop_i, drop_p;     // dropout streams of block i (site 2) and of block i - 1 (sites 2, 3, 4)
    unsigned long long* stamps;   // -DMSST_STAMPS builds: cycle stamps of the eight waves of one workgroup (tools/stamps_bwd5.py)
    int* queue;                   // optional: one zeroed counter -> dynamic tile queue instead of the static partition
};

struct TokBwdArgs {
    const float* img; const float* pre_g; const float* pre_b; const float* w_emb; const float* b_emb;
    const float* post_g; const float* post_b; const uint8_t* mask; const float* dx0; float* slab;
    int B, S, N, T, P;
    Drop drop;
};

// ---- opt-in per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----
enum KernelId {
    K_PREP = 0, K_TOK_FWD, K_BLOCK_FWD, K_HEAD_FWD, K_LOSS_REDUCE, K_HEAD_BWD, K_REDUCE, K_BWD_MLP, K_BWD_ATTN,
    K_ATTN_REDUCE, K_BWD_LN1, K_TOK_BWD, K_POS_SPLIT, K_ADAMW, K_BWD_LN1MLP, K_LAYERNORM, K_COUNT
};
void prof_begin(int id, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st;
    ProfScope(int id, hipStream_t s) : st(s) { prof_begin(id, s); }
    ~ProfScope() { prof_end(st); }
};

struct ClsArgs {
    const float* y; const float* ln_g; const float* ln_b; const float* w; const float* b; float* logits;
    int B, S, N, T, NC;
};
struct ClsBwdArgs {
    const float* y; const float* dlogits; const float* ln_g; const float* ln_b; const float* w;
    float* dy; float* slab;
    int B, S, N, T, NC;
};
int launch_cls_head_fwd(const ClsArgs& a, hipStream_t st);
int launch_cls_head_bwd(const ClsBwdArgs& a, hipStream_t st);

int launch_tokenize_fwd(const TokArgs& a, hipStream_t st);
int launch_head_bwd(const HeadBwdArgs& a, int nchunk, hipStream_t st);
// One reduction segment: dst[(i / row_len) * row_stride + i % row_len] = sum_{k < nslab} src[k * slab_stride + i], i < n
struct RSeg {
    const float* src; float* dst;
    long slab_stride;
    int nslab, n, row_len, row_stride, blk0, vec4;
    int ny;   // batched launches (blockIdx.y = y): the segment exists for y < ny
};
#define MSST_MAX_RSEG 72
// A batched launch (ny_max > 1) reduces the same segment table for ny_max slab sets src_ystride floats apart into destinations
// dst_ystride floats apart (the deferred reduction of a run of msst_block_bwd_chain calls: one slab set and one gradient block per call).
struct RSegs { RSeg s[MSST_MAX_RSEG]; int nseg; int nblocks; int ny_max; long src_ystride, dst_ystride; };
struct RSegBuilder {
    RSegs r;
    RSegBuilder() { r.nseg = 0; r.nblocks = 0; r.ny_max = 1; r.src_ystride = 0; r.dst_ystride = 0; }
    bool add(const float* src, long slab_stride, int nslab, float* dst, int n, int row_len = 0, int row_stride = 0, int ny = 1) {
        if (r.nseg >= MSST_MAX_RSEG) return false;
        if (ny < 1) return true;   // a segment no launch of the batch wrote
        RSeg& g = r.s[r.nseg++];
        g.ny = ny;
        if (ny > r.ny_max) r.ny_max = ny;
        g.src = src; g.dst = dst; g.slab_stride = slab_stride; g.nslab = nslab; g.n = n;
        g.row_len = row_len > 0 ? row_len : n; g.row_stride = row_stride > 0 ? row_stride : n;
        g.blk0 = r.nblocks;
        // 16-byte path: a thread owns 4 consecutive outputs (one dwordx4 per slab) when every address involved is aligned
        // (batched: the y strides must keep that alignment too -- checked by the caller that sets them)
        g.vec4 = (n % 4 == 0 && g.row_len % 4 == 0 && g.row_stride % 4 == 0 && slab_stride % 4 == 0 &&
                  ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) ? 1 : 0;
        const int per_block = g.vec4 ? 128 : 32;
        r.nblocks += (n + per_block - 1) / per_block;
        return true;
    }
};
static_assert(sizeof(RSegs) <= 4096, "RSegs travels as a kernel argument");
int launch_reduce_segs(const RSegs& r, hipStream_t st);
int launch_block_bwd_mlp(const MlpBwdArgs& a, int grid, int prec, hipStream_t st);
int launch_block_bwd_attn(const AttnBwdArgs& a, int nchunk, int prec, hipStream_t st, int* nparts);
int launch_block_bwd_attn_r3(const AttnBwdArgs& a, int nchunk, hipStream_t st);     // msst_bwd3.hip (bf16 throughput kernel: one GEMM per wave, 32x32x16 tiles)
int launch_block_bwd_attn_r4(const AttnBwdArgs& a, int nchunk, hipStream_t st);     // msst_bwd4.hip (the same, two heads per workgroup half a tile apart)
int launch_block_bwd_ln1(const Ln1BwdArgs& a, int grid, int prec, hipStream_t st);
int launch_block_bwd_ln1mlp(const LnMlpArgs& a, int grid, hipStream_t st);   // msst_bwd5.hip (bf16: LN1 backward of block i + MLP backward of block i - 1)
int launch_tokenize_bwd(const TokBwdArgs& a, int nchunk, hipStream_t st);
int launch_pos_split(const float* dpos, int S, int N, int split, float* dpe, float* dce, hipStream_t st);
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                 float wd, int step, float clamp, float gscale, hipStream_t st);
int launch_cu_thief(int nblocks, int us, unsigned* sink, hipStream_t st);   // msst_opt.hip (occupancy probe)
int launch_box_probe(double* out4, void* scratch, long bytes, hipStream_t st);   // msst_opt.hip (MFMA rate / shader clock / HBM read rate of this box)
int launch_block_fwd_rs(const BlockArgs& a, int grid, hipStream_t st);   // msst_fwd3.hip (bf16, 8 heads; role split: the default)
int launch_block_fwd_rs_stack(const StackArgs& a, int grid, hipStream_t st);   // the same for a run of blocks of one stack, ONE launch
int block_fwd_stack_max_steps();
int launch_block_fwd(const BlockArgs& a, int prec, hipStream_t st);
bool block_fwd_writes_xn(const BlockArgs& a, int prec);   // does the kernel launch_block_fwd selects honour a.xn_out?
bool block_fwd_writes_lse(const BlockArgs& a, int prec);  // ... a.lse_out?  (the role-split kernel: bf16, 8 heads, no selection flags)
int launch_head_fwd(const HeadArgs& a, float* loss, hipStream_t st);
// msst_ln.hip: stand-alone LayerNorm over the last axis (D <= 128; D = 96 vectorised)
int launch_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, long rows, int D,
                         float eps, hipStream_t st);
int layernorm_bwd_grid(long rows, int D);   // workgroups (= slab rows of 2 D floats) the backward launch uses
int launch_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* slab, int grid, long rows, int D,
                         float eps, hipStream_t st);

}  // namespace msst
