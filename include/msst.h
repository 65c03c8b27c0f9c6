/* libmsst -- C-ABI of the MI355X-native MaskedSST masked-pretraining hot path.
 *
 * The reference (HSG-AIML/MaskedSST) is pure Python/PyTorch and has NO FFI of its own; its
 * boundary for this path is the nn.Module surface of
 *     src/vit_spatial_spectral.py:256-564  (ViTSpatialSpectral)
 *     src/vit_simmim_original.py:139-340   (SimMIMSpatialSpectral)
 * which maskedsst_amd/ mirrors in Python.  This header is the build-defined C boundary underneath
 * that mirror: every entry point states which reference lines it replaces, and
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Rules: plain pointers and sizes only (no torch types); every pointer is DEVICE memory owned by
 * the caller unless marked "host"; calls only enqueue work on `stream` (a hipStream_t passed as
 * void*), never synchronise, never allocate; returns 0 or a negative MSST_ERR_* / positive
 * hipError_t code, never throws; re-entrant (no mutable global state besides a thread-local error
 * string).  All activations are fp32 [tokens][96] in the reference token order 'b (c h w) d'.
 * The kernels are specialised for dim = 96, dim_head = 64, mlp_dim = 64 (configs/config.yaml:19-22
 * of the reference); heads, depth, bands, batch are runtime.
 */
#ifndef MSST_H
#define MSST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSST_VERSION 100
#define MSST_DIM 96
#define MSST_DIM_HEAD 64
#define MSST_MLP 64

#define MSST_PREC_F32 0  /* exact fp32 MFMA (parity mode)               */
#define MSST_PREC_BF16 1 /* bf16 MFMA operands, fp32 accumulate/residual */

#define MSST_MODE_SPATIAL 0  /* sequences = (b, c), N tokens each, contiguous            */
#define MSST_MODE_SPECTRAL 1 /* sequences = (b, n), S tokens each, stride N*96 floats    */

#define MSST_ERR_UNSUPPORTED (-2)
#define MSST_ERR_BADARG (-3)

int msst_version(void);
const char* msst_last_error(void);

/* One weight-prep job: dst = (elem)src, optionally transposed.  elem = float (F32) or bf16. */
typedef struct MsstPrepJob {
    const float* src; /* [rows][cols] fp32 master weight          */
    void* dst;        /* [rows][cols] or [cols][rows] (transpose) */
    int32_t rows, cols, transpose, _pad;
} MsstPrepJob;

/* Converts / transposes all matrices of the model into operand layout in ONE launch.
 * `jobs` is a DEVICE array. max_elems = max(rows*cols) over jobs. */
int msst_prep_weights(const MsstPrepJob* jobs, int njobs, int max_elems, int prec, void* stream);

/* Operand-layout weights of one transformer block (device pointers).
 * Replaces the parameters of reference vit_spatial_spectral.py:85-97 (one Transformer layer). */
typedef struct MsstBlockWeights {
    const void* wqkv;  /* [3*H*64][96]  to_qkv.weight, rows q|k|v, head-major */
    const void* wout;  /* [96][H*64]    to_out.0.weight                        */
    const void* w1;    /* [64][96]      net.0.weight                           */
    const void* w2;    /* [96][64]      net.3.weight                           */
    const void* wqkvT; /* [96][3*H*64]  (backward)                             */
    const void* woutT; /* [H*64][96]                                           */
    const void* w1T;   /* [96][64]                                             */
    const void* w2T;   /* [64][96]                                             */
    const float* ln1_g; const float* ln1_b; const float* bo;
    const float* ln2_g; const float* ln2_b; const float* b1; const float* b2;
} MsstBlockWeights;

/* a1+a2+a3+a5: BlockwisePatchEmbedding.to_patch/.embed (vit_spatial_spectral.py:197-222), position
 * add and mask-token select (vit_simmim_original.py:236-249,285).
 * img [B][S*P][N]; mask [B][T] bytes (all zero for the classification path);
 * pos_split == 0: pos_a = learned table [T][96] (pos_embedding[0,:T]);
 * pos_split  > 0: pos_a = pos_embed [N][pos_split], pos_b = channel_embed [S][96-pos_split]
 *                 (get_pos_embeddings, vit_spatial_spectral.py:501-516).  out [B][T][96]. */
int msst_tokenize_fwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const float* pos_a,
                      const float* pos_b, int pos_split, const float* mask_token, const uint8_t* mask,
                      float* out, int B, int S, int N, int P, void* stream);

/* a7-a10: one fused pre-norm transformer block (PreNorm+Attention+FeedForward+residuals,
 * vit_spatial_spectral.py:22-104) over all B*S*N tokens; mode selects the spatial or spectral
 * sequence grouping of vit_spatial_spectral.py:410-431 (no transposes are materialised).
 * x -> y (y != x); x1 (optional) receives x + attn(LN(x)) for the backward. */
int msst_block_fwd(const MsstBlockWeights* w /*host*/, const float* x, float* y, float* x1, int mode,
                   int B, int S, int N, int heads, int prec, int max_grid, void* stream);

/* a12-a14: gather of masked tokens, BlockwiseToPixels (vit_simmim_original.py:9-40,314-330),
 * target gather from the raw cube (:335) and mean-L1 / K (:338).
 * idx [B][K] int32; w_pix [S or 1][P][96], b_pix [S or 1][P]; per_block = to_pixels_per_spectral_block.
 * dpred [B][K][P] receives sign(pred-target); pred optional; partial: >= B*ceil(K/64) floats scratch.
 * loss: 1 float (device). */
int msst_head_fwd(const float* y, const float* img, const int32_t* idx, const float* w_pix,
                  const float* b_pix, int per_block, float* dpred, float* pred, float* partial,
                  float* loss, int B, int S, int N, int P, int K, void* stream);

/* ---- backward (a15) ---- */

/* d(loss)/d(encoder output) and to_pixels grads.  csr_ptr [B][T+1], csr_pos [B][K]: for each token
 * the positions k with idx[b][k] == token (duplicates allowed -- the reference's misaligned index
 * slicing produces them).  dy [B][T][96] is fully written.
 * slab: [S][nchunk][P*96 + P] partial to_pixels grads, reduced by msst_reduce_slabs. */
int msst_head_bwd(const float* y, const float* dpred, const int32_t* csr_ptr, const int32_t* csr_pos,
                  const float* w_pix, int per_block, float gscale, float* dy, float* slab, int nchunk,
                  int B, int S, int N, int P, int K, void* stream);

/* MLP half of a block: dy -> dx1 (= d/d(x + attn)), weight-grad partial slabs [grid][MSST_MLP_SLAB]. */
#define MSST_MLP_SLAB (64 * 96 + 96 * 64 + 64 + 96 + 96 + 96)
int msst_block_bwd_mlp(const MsstBlockWeights* w /*host*/, const float* x1, const float* dy, float* dx1,
                       float* slab, int grid, int mode, int B, int S, int N, int prec, void* stream);

/* attention half: recompute q,k,v,P from x, consume da = dx1; per (chunk, head) workgroups.
 * dxn_part [heads][tokens][96] (elem type) receives per-head partial d/d(LN1(x));
 * slab [nchunk][heads][MSST_ATTN_SLAB] receives dWq|dWk|dWv [3][64][96] and dWout_h [96][64]. */
#define MSST_ATTN_SLAB (3 * 64 * 96 + 96 * 64)
int msst_block_bwd_attn(const MsstBlockWeights* w /*host*/, const float* x, const float* da, void* dxn_part,
                        float* slab, int nchunk, int mode, int B, int S, int N, int heads, int prec,
                        void* stream);

/* dx = dx1 + LN1_bwd(sum_h dxn_part[h]; x); LN1 gamma/beta grad partials slab [grid][192]. */
int msst_block_bwd_ln1(const MsstBlockWeights* w /*host*/, const float* x, const float* dx1,
                       const void* dxn_part, float* dx, float* slab, int grid, int B, int S, int N,
                       int heads, int prec, void* stream);

/* out[i] (+)= sum_{s<nslab} slab[s*stride + i], i < n  (fixed order, deterministic). */
int msst_reduce_slabs(const float* slab, int nslab, long stride, float* out, int n, int accumulate,
                      void* stream);

/* Tokenizer backward: grads of blockwise_embed, pre/post norm, position table(s), mask token.
 * slab [S][nchunk][MSST_TOK_SLAB(N,P)] partials over batch chunks. */
int msst_tokenize_bwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const uint8_t* mask,
                      const float* dx0, float* slab, int nchunk, int B, int S, int N, int P, void* stream);

/* Fused AdamW over a flat fp32 buffer (torch.optim.AdamW semantics, src/utils.py:36-45), with the
 * reference's value clamp of the gradient (pretrain.py:71-73) when clamp > 0. */
int msst_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float clamp, float gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
