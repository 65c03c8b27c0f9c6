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
 * hipError_t code, never throws; re-entrant: the compute entry points keep no state between calls and
 * read no environment (a thread-local error string; idempotent once-flags for kernel attributes; the
 * opt-in msst_profile_* state is mutex-guarded and, when disabled, costs one relaxed atomic load).  All activations are fp32 [tokens][96] in the reference token order 'b (c h w) d'.
 * The kernels are specialised for dim = 96, dim_head = 64, mlp_dim = 64 (configs/config.yaml:19-22
 * of the reference); heads, depth, bands, batch are runtime.
 */
#ifndef MSST_H
#define MSST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSST_VERSION 104
#define MSST_DIM 96
#define MSST_DIM_HEAD 64
#define MSST_MLP 64

#define MSST_PREC_F32 0  /* exact fp32 MFMA (parity mode)               */
#define MSST_PREC_BF16 1 /* bf16 MFMA operands, fp32 accumulate/residual */

/* Kernel-selection flags, OR-ed into the `prec` argument of msst_block_fwd / msst_block_bwd (bits 8..23).  0 selects the
 * tuned kernels; the others exist for the cross-kernel agreement tests and A/B studies.  They are explicit arguments:
 * the library never reads the environment and keeps no per-call state. */
#define MSST_KERNEL_GENERIC (16 << 8)    /* generic template kernels also in bf16 (fwd and attention bwd)   */
#define MSST_KERNEL_FWD_4WAVE (64 << 8)  /* bf16 forward: tuned 4-wave kernel instead of head-per-wave      */
#define MSST_X1_BF16 (1024 << 8)         /* the saved mid-residual rows x1 are bf16 instead of fp32: msst_block_fwd writes them so (role-split bf16 forward only:
                                            8 heads, no MSST_KERNEL_* flag; MSST_ERR_UNSUPPORTED otherwise), msst_block_bwd / _chain read x1 and x1_prev so (bf16
                                            kernels only).  A quarter of the forward's writes and 8 % of the fused row-local backward's reads less; the LN2 statistics
                                            of the backward are then those of the rounded rows (parity: tests/test_gpu_backward.py::test_bf16_x1_rows) */
#define MSST_FWD_HALF (4096 << 8)         /* msst_block_fwd / msst_block_fwd_stack, role-split bf16 forward only (MSST_VERSION 104): the forward's GEMM operands -- LN rows,
                                            weights (MsstBlockWeights.wqkv_h ...), q / k / v, probabilities, attention output, GELU output -- are rounded to IEEE
                                            half (11 significant bits) instead of bf16 (8) and multiplied by v_mfma_f32_16x16x32_f16 (same rate).  Everything
                                            else is unchanged: fp32 accumulation / softmax / LayerNorm / residual stream, bf16 rows saved for the backward, bf16
                                            backward.  Why: the bf16 forward's loss error against the fp32 reference (2.6e-4 on the Houston-shape anchor) is
                                            systematic and owned by the rounding of the WEIGHTS (tools/bf16_error_table.py); with half operands it is 7e-6.  Every
                                            operand of the forward is a LayerNorm row, a weight, a probability or a bounded activation: inside half's range */
#define MSST_LSE_RENORM (8192 << 8)       /* msst_block_bwd / _chain (MSST_VERSION 104): lse_saved comes from a forward whose scores are not the backward's own -- the
                                            half-operand forward (MSST_FWD_HALF) against the bf16 recomputation here.  The two-head attention backward then uses lse
                                            as the exponent offset only and normalises every row by its own sum (p = softmax of ITS scores exactly, one reduction
                                            more); without the flag p = exp2(s c - lse) as saved.  Always pass it for blocks run with MSST_FWD_HALF */
#define MSST_LN1_FROM_XN (2048 << 8)      /* msst_block_bwd_chain (MSST_VERSION 104): the fused LN1 + MLP launch takes xhat of LN1 from the saved bf16 LN1 rows and
                                            the saved rstd -- xhat = (xn_saved - ln1_b) / ln1_g, rstd = the tail of lse_saved (MSST_SAVED_RSTD) -- instead of
                                            re-reading and re-normalising the fp32 block input x: 192 bytes per token less of 2304.  The caller sets it only when
                                            every ln1_g is safely away from 0 and |ln1_b / ln1_g| is moderate (the division amplifies the rows' bf16 rounding by
                                            1 + |b / g| / |xhat|; maskedsst_amd/engine.py: max |b / g| <= 12); needs xn_saved and lse_saved */
#define MSST_BWD_DEFER_REDUCE (512 << 8) /* msst_block_bwd_chain: leave the partial-gradient slabs of this call unreduced (msst_block_bwd_reduce does a run of calls in one launch) */
#define MSST_KERNEL_ATTN_R3 (128 << 8)   /* bf16 attention backward: one head per workgroup (msst_bwd3.hip) instead of two (msst_bwd4.hip) */

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
    int32_t rows, cols, transpose;
    int32_t pack;     /* bf16 only: 0 = 16-row x 32-k operand fragments (16x16x32 MFMA), 1 = 32-row x 16-k fragments (32x32x16 MFMA;
                         destination rows % 32 == 0 and k % 16 == 0); + MSST_PREP_HALF (256, MSST_VERSION 104): the destination
                         elements are IEEE half instead of bf16 (same fragment layout; the fp16-operand forward, MSST_FWD_HALF).
                         Any other value: the job is skipped. */
    int32_t scale_rows; /* the first scale_rows SOURCE rows are multiplied by `scale` (0: none).  The round-3 attention backward   */
    float scale;        /* wants the q and k blocks of to_qkv^T pre-multiplied by dim_head^-0.5 (2^-3: exact in bf16).             */
} MsstPrepJob;

#define MSST_PREP_HALF 256
/* Converts / transposes all matrices of the model into operand layout in ONE launch.
 * `jobs` is a DEVICE array. max_elems = max(rows*cols) over jobs.  job_bytes = sizeof(MsstPrepJob) of the CALLER's header: a
 * table laid out by another revision is refused (MSST_ERR_BADARG) instead of being read mis-strided.  err_flag (optional, one
 * zeroed int32 in device memory): the kernel ORs in 1 for a job with a bad pack / shape and 2 for a pack = 1 job that is not
 * whole fragments -- such jobs are skipped, their destinations stay unwritten; the host cannot see the table, so a caller
 * that builds it should read the word back once after its first call. */
int msst_prep_weights(const MsstPrepJob* jobs, int njobs, int job_bytes, int max_elems, int prec, int32_t* err_flag, void* stream);

/* Operand-layout weights of one transformer block (device pointers).
 * Replaces the parameters of reference vit_spatial_spectral.py:85-97 (one Transformer layer). */
typedef struct MsstBlockWeights {
    uint64_t struct_bytes; /* = sizeof(MsstBlockWeights): a caller built against another revision of this header is refused
                              (MSST_ERR_BADARG) instead of being read past the end of its struct */
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
    /* bf16 only, optional (null: msst_block_bwd runs the template attention backward): the three matrices the round-3
     * attention backward feeds to 32x32x16 MFMAs, fragment-packed with MsstPrepJob.pack = 1 */
    const void* wqkv32;  /* [3*H*64][96]  */
    const void* woutT32; /* [H*64][96]    */
    const void* wqkvT32; /* [96][3*H*64], pack = 1, scale_rows = 2*H*64, scale = dim_head^-0.5 (q and k blocks carry the softmax scale) */
    /* MSST_VERSION 104, bf16 only, optional (null: MSST_FWD_HALF is refused): the four forward matrices as IEEE half, pack = 0 | MSST_PREP_HALF
     * -- operands of the fp16-operand forward (same layout as wqkv / wout / w1 / w2) */
    const void* wqkv_h; const void* wout_h; const void* w1_h; const void* w2_h;
} MsstBlockWeights;

/* a1+a2+a3+a5: BlockwisePatchEmbedding.to_patch/.embed (vit_spatial_spectral.py:197-222), position
 * add and mask-token select (vit_simmim_original.py:236-249,285).
 * img [B][S*P][N]; mask [B][T] bytes (all zero for the classification path);
 * pos_split == 0: pos_a = learned table [T][96] (pos_embedding[0,:T]);
 * pos_split  > 0: pos_a = pos_embed [N][pos_split], pos_b = channel_embed [S][96-pos_split]
 *                 (get_pos_embeddings, vit_spatial_spectral.py:501-516).  out [B][T][96].
 * emb_dropout_p > 0: embedding dropout on (token + pos) (forward_features, :530; classification path). */
int msst_tokenize_fwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const float* pos_a,
                      const float* pos_b, int pos_split, const float* mask_token, const uint8_t* mask,
                      float* out, int B, int S, int N, int P, float emb_dropout_p, uint32_t seed, void* stream);

/* a7-a10: one fused pre-norm transformer block (PreNorm+Attention+FeedForward+residuals,
 * vit_spatial_spectral.py:22-104) over all B*S*N tokens; mode selects the spatial or spectral
 * sequence grouping of vit_spatial_spectral.py:410-431 (no transposes are materialised).
 * x -> y (y != x); x1 (optional) receives x + attn(LN(x)) for the backward.
 * dropout_p > 0 enables the reference's four dropout sites (:38,40,57,62) with a stateless counter-based
 * mask keyed by (seed, layer, site, element); msst_block_bwd regenerates it from the same three values.
 * xn_out (optional, [tokens][96] bf16): receives LN1(x) exactly as the block used it, so that msst_block_bwd neither
 * re-reads x nor renormalises it.
 * lse_out (optional, msst_block_lse_floats(...) fp32): the block's saved STATISTICS, two arrays back to back --
 *   [msst_block_tiles(...)][heads][64]: per (64-row tile, head, row), lse = log2 of the row's softmax denominator in the exponent
 *     domain of the kernels (p = exp2(s * dim_head^-0.5 * log2 e - lse), vit_spatial_spectral.py:71-73): the attention backward
 *     then computes the probabilities directly -- no row maximum, no row sum, no reciprocal; the tile order is the forward's own
 *     (same mode / shapes on both sides);
 *   [tokens] (MSST_VERSION 104): rstd of LN1 of every token (vit_spatial_spectral.py:22-29), token order: with it and the bf16 LN1
 *     rows (xn_out) the fused row-local backward rebuilds xhat = (LN1 row - beta) / gamma and does not read the fp32 block input
 *     at all (MSST_LN1_FROM_XN).
 *   10-12 MB per block at the bench shape.
 * *saved (host, optional) tells which of them the selected kernel wrote: MSST_SAVED_XN (the two tuned bf16 kernels -- role
 * split for 8 heads, 4-wave otherwise; not fp32, not MSST_KERNEL_GENERIC), MSST_SAVED_LSE and MSST_SAVED_RSTD (the role-split
 * kernel only).  Pass xn_saved / lse_saved to msst_block_bwd only when the bit is set. */
#define MSST_SAVED_XN 1
#define MSST_SAVED_LSE 2
#define MSST_SAVED_RSTD 4
long msst_block_lse_floats(int mode, int B, int S, int N, int heads);   /* floats of the whole statistics buffer: tiles * heads * 64 + tokens */
long msst_block_tiles(int mode, int B, int S, int N);                   /* 64-row tiles of a block launch (the forward's own tiling) */
int msst_block_fwd(const MsstBlockWeights* w /*host*/, const float* x, float* y, float* x1, int mode,
                   int B, int S, int N, int heads, int prec, int max_grid, float dropout_p, uint32_t seed,
                   int layer, void* xn_out, float* lse_out, int* saved /*host*/, void* stream);

/* The same forward for a RUN of blocks of ONE stack (same mode) as ONE launch (round 5): the blocks of a stack never mix the rows
 * of different 64-row tiles (vit_spatial_spectral.py:410-431: the regroupings sit between the stacks), so a workgroup takes a tile
 * through block after block -- block j reads what block j - 1 wrote, two blocks of one tile at least three pipeline steps apart --
 * and the prologue + pipeline fill / drain of all launches but one go away (26 us each).  Bit-identical to nblk calls of
 * msst_block_fwd with layer = layer0 .. layer0 + nblk - 1: block 0 reads x0, block j > 0 reads y[j - 1].
 * w, y, x1, xn_out, lse_out: HOST arrays of nblk pointers (x1 / xn_out / lse_out may be NULL as a whole: nothing saved).
 * Every per-block operand (each member of w[j], y[j], x1[j], xn_out[j], lse_out[j]) must lie a constant byte stride (|stride| < 2 GiB,
 * one stride per operand) from block to block -- the kernel addresses block j as block 0 + j x stride instead of fetching pointers.
 * The role-split bf16 forward only (8 heads; of the prec flags only MSST_X1_BF16), at most 16 blocks, and at most 1024 (tile, block)
 * steps per workgroup: MSST_ERR_UNSUPPORTED otherwise (nothing launched) -- call msst_block_fwd per block then. */
int msst_block_fwd_stack(const MsstBlockWeights* const* w /*host*/, int nblk, const float* x0, float* const* y /*host*/,
                         float* const* x1 /*host*/, void* const* xn_out /*host*/, float* const* lse_out /*host*/, int mode,
                         int B, int S, int N, int heads, int prec, int max_grid, float dropout_p, uint32_t seed, int layer0,
                         int* saved /*host, optional*/, void* stream);

/* a12-a14: gather of masked tokens, BlockwiseToPixels (vit_simmim_original.py:9-40,314-330),
 * target gather from the raw cube (:335) and mean-L1 / K (:338).
 * idx [B][K] int32; w_pix [S or 1][P][96], b_pix [S or 1][P]; per_block = to_pixels_per_spectral_block.
 * dpred [B][K][P] receives sign(pred-target); pred optional; partial: >= B*ceil(K/64) floats scratch.
 * loss: 1 float (device). */
int msst_head_fwd(const float* y, const float* img, const int32_t* idx, const float* w_pix,
                  const float* b_pix, int per_block, float* dpred, float* pred, float* partial,
                  float* loss, int B, int S, int N, int P, int K, void* stream);

/* ---- backward (a15: what autograd does for the reference at pretrain.py:116) ---- */

/* d(loss)/d(encoder output) through the gather + to_pixels, and the to_pixels grads.
 * csr_ptr [B][T+1], csr_pos [B][K]: for each token the positions k with idx[b][k] == token
 * (duplicates allowed -- the reference's misaligned index slicing produces them).
 * gscale = 1 / (B*K*P) / K; gout = optional device scalar d(final)/d(loss) multiplied in.
 * dy [B][T][96] is fully written.
 * slab: S * nchunk * (P*96 + P) floats of scratch.  dw_pix / db_pix laid out like w_pix / b_pix. */
int msst_head_bwd(const float* y, const float* dpred, const int32_t* csr_ptr, const int32_t* csr_pos,
                  const float* w_pix, int per_block, float gscale, const float* gout, float* dy, float* slab,
                  int nchunk, float* dw_pix, float* db_pix, int B, int S, int N, int P, int K, void* stream);

/* Gradient destinations of one transformer block (fp32, shapes of the reference parameters). */
typedef struct MsstBlockGrads {
    float* ln1_g; float* ln1_b; float* wqkv; float* wout; float* bo;
    float* ln2_g; float* ln2_b; float* w1; float* b1; float* w2; float* b2;
} MsstBlockGrads;

/* Scratch sizes (floats) for msst_block_bwd */
#define MSST_MLP_SLAB (64 * 96 + 96 * 64 + 64 + 96 + 96 + 96)
#define MSST_ATTN_SLAB (3 * 64 * 96 + 96 * 64)
#define MSST_LN1_SLAB 288

/* Backward of one fused block: given the saved block input x, the saved mid residual x1 and dy,
 * writes dx and every parameter gradient of the block.  Internally: MLP half (recompute from x1)
 * -> attention half, one workgroup per (tile chunk, head) -- or per (tile chunk, head PAIR): the default bf16 kernel for an
 * even head count -- weight grads in registers -> LN1 backward + residual; partial-gradient slabs are reduced in a fixed
 * order (deterministic).
 * Workspace (caller-owned, device): dx1 [tokens][96] f32; dxn_part heads*tokens*96 elems
 * (f32 or bf16 by prec; the head-pair kernel uses the first half); slab grid_rows*(2*MSST_MLP_SLAB + MSST_LN1_SLAB) + nchunk*heads*MSST_ATTN_SLAB
 * floats (the bf16 MLP half runs up to 2*grid_rows workgroups, one slab each). */
int msst_block_bwd(const MsstBlockWeights* w /*host*/, const MsstBlockGrads* g /*host*/, const float* x,
                   const float* x1, const float* dy, float* dx, float* dx1, void* dxn_part, float* slab,
                   int grid_rows, int nchunk, int mode, int B, int S, int N, int heads, int prec,
                   float dropout_p, uint32_t seed, int layer, const void* xn_saved /*optional, see msst_block_fwd*/,
                   const float* lse_saved /*optional, see msst_block_fwd; used together with xn_saved by the two-head attention backward*/,
                   void* dab_ws /*optional workspace [tokens][96] bf16, used together with xn_saved: the MLP half leaves the
                                  dropped bf16 copy of dx1 there for the attention half*/,
                   void* stream);

/* The same backward for a RUN of blocks (bf16 throughput path): called once per block in reverse order, it fuses the row-local
 * seam BETWEEN consecutive blocks -- the LN1 backward of block i and the MLP-half backward of block i - 1
 * (vit_spatial_spectral.py:22-44,102-103 are row-local) run as ONE launch, so dx of block i never reaches HBM:
 *   first != 0 (the last block of the model = first call): its MLP half runs on dy first (dy -> dx1, dab_ws);
 *   every call: attention half of block i on the dx1 / dab_ws rows left by the step above or by the previous call;
 *   w_prev != NULL: LN1 backward of block i + MLP half of block i - 1 (weights w_prev, saved mid residual x1_prev; its
 *       parameter gradients go to g_prev): dx1 and dab_ws are overwritten IN PLACE with block i - 1's; dx is not touched;
 *   w_prev == NULL (block 0): the LN1 backward runs alone and writes dx.
 * Gradients of block i are complete when the call for block i returns (its MLP-half gradients were written by the call
 * before).  Requires prec = MSST_PREC_BF16 (tuned kernels), xn_saved and dab_ws, and at most four d(LN1 out) partials
 * (heads <= 8 even, or heads <= 4): MSST_ERR_BADARG / MSST_ERR_UNSUPPORTED otherwise -- use msst_block_bwd then.
 * slab: grid_rows*(3*MSST_MLP_SLAB + MSST_LN1_SLAB) + nchunk*heads*MSST_ATTN_SLAB floats.  x1 is read only when first != 0.
 * tile_queue != NULL (data parallel): the attention backward and the fused LN1 + MLP launch draw their tiles from agent-scope
 * counters in this scratch (zeroed on `stream` by the call) instead of the static partition tile = workgroup + k * grid: a
 * workgroup that starts late because a communication kernel holds its CU draws fewer tiles instead of running its whole
 * share behind the others.  The partition then depends on timing, so the gradients differ from run to run in fp32 summation
 * order; NULL keeps the static, bit-reproducible partition (the single-GPU default). */
#define MSST_TILE_QUEUE_WORDS 64
int msst_block_bwd_chain(const MsstBlockWeights* w /*host*/, const MsstBlockGrads* g /*host*/,
                         const MsstBlockWeights* w_prev /*host, block i - 1 or NULL*/, const MsstBlockGrads* g_prev /*host*/,
                         const float* x, const float* x1, const float* x1_prev, const float* dy, float* dx, float* dx1,
                         void* dxn_part, float* slab, int grid_rows, int nchunk, int mode, int B, int S, int N, int heads,
                         int prec, float dropout_p, uint32_t seed, int layer, const void* xn_saved, const float* lse_saved /*optional*/,
                         void* dab_ws, int first, int32_t* tile_queue /*optional, MSST_TILE_QUEUE_WORDS int32 of device scratch*/, void* stream);

/* Deferred slab reduction for a RUN of msst_block_bwd_chain calls made with MSST_BWD_DEFER_REDUCE in `prec`: one launch instead of
 * one per block (the reduction is the only part of a block backward whose cost does not shrink with the problem: 62 MB of
 * slabs per block at the bench grid, whatever the batch).  Call y = 0 .. count - 1 of the run used the workspace slab + y *
 * slab_stride floats (same grid_rows / nchunk / mode / shapes for all of them) and reduced-gradient destinations that lie
 * grad_stride floats apart: g + y * grad_stride for the attention half and LN1 of its block, g_prev + y * grad_stride for the
 * MLP half of the block before it that its fused launch ran (the first count_prev calls of the run had one; count_prev = count
 * except for the run that ends with block 0).  first != 0: call 0 of the run was made with first != 0 (standalone MLP half of
 * its own block, reduced into g).  slab_stride and grad_stride must be multiples of 4 floats.  Deterministic: same
 * summation order as the per-call reduction. */
int msst_block_bwd_reduce(const MsstBlockGrads* g /*host*/, const MsstBlockGrads* g_prev /*host, may be NULL when count_prev == 0*/,
                          float* slab, long slab_stride, long grad_stride, int count, int count_prev, int first,
                          int grid_rows, int nchunk, int mode, int B, int S, int N, int heads, int prec, void* stream);

/* Tokenizer backward: grads of blockwise_embed, pre/post norm, position table(s), mask token.
 * slab: S * nchunk * (N*96 + 96*P + 4*96 + 32) floats + S*N*96 floats (position staging).
 * dpos_a / dpos_b follow pos_a / pos_b of msst_tokenize_fwd; dmask_token may be null. */
int msst_tokenize_bwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const uint8_t* mask,
                      const float* dx0, float* slab, int nchunk, float* dpre_g, float* dpre_b,
                      float* dw_emb, float* db_emb, float* dpost_g, float* dpost_b, float* dpos_a,
                      float* dpos_b, int pos_split, float* dmask_token, int B, int S, int N, int P,
                      float emb_dropout_p, uint32_t seed, void* stream);

/* a17: classification head of ViTSpatialSpectral.forward (vit_spatial_spectral.py:536-564, :481-493):
 * mean over the spectral axis -> LayerNorm(96) -> Linear(96 -> n_classes); logits [B][n_classes][N].
 * _bwd: dy [B][T][96] fully written; slab B*(n_classes*97 + 192) floats. */
int msst_cls_head_fwd(const float* y, const float* ln_g, const float* ln_b, const float* w, const float* b,
                      float* logits, int B, int S, int N, int n_classes, void* stream);
int msst_cls_head_bwd(const float* y, const float* dlogits, const float* ln_g, const float* ln_b, const float* w,
                      float* dy, float* slab, float* dln_g, float* dln_b, float* dw, float* db, int B, int S,
                      int N, int n_classes, void* stream);

/* a7: LayerNorm over the last axis as an op of its own (nn.LayerNorm of PreNorm, vit_spatial_spectral.py:25, and of the
 * tokenizer, :194-195: eps 1e-5, affine, biased variance), fp32, rows of D <= 128 contiguous floats (D = 96 vectorised; D = 10 =
 * the tokenizer's pixel rows).  On the hot path the same arithmetic runs fused into msst_tokenize_* / msst_block_*; these
 * entry points serve a caller that needs a lone LayerNorm.  Row statistics are wave-shuffle (DPP) reductions.
 * _fwd: mean / rstd [rows] are optional outputs.  _bwd: recomputes the statistics from x; dgamma / dbeta [D] are fully
 * written (fixed summation order: bit-reproducible); slab: msst_layernorm_bwd_slab(rows, D) floats of scratch. */
int msst_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean /*optional*/,
                       float* rstd /*optional*/, long rows, int D, float eps, void* stream);
long msst_layernorm_bwd_slab(long rows, int D);
int msst_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta,
                       float* slab, long rows, int D, float eps, void* stream);

/* Fused AdamW over a flat fp32 buffer (torch.optim.AdamW semantics, src/utils.py:36-45), with the
 * reference's value clamp of the gradient (pretrain.py:71-73) when clamp > 0. */
int msst_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float clamp, float gscale, void* stream);

/* Opt-in per-kernel timing: when enabled every kernel launch of the library is bracketed by a pair
 * of HIP events recorded on the launch stream.  msst_profile_collect synchronises on the recorded
 * events and returns, per kernel id (0 .. msst_profile_kernels()-1), the summed duration in ms and
 * the launch count since enable / the previous collect.  Thread-safe (mutex); meant for bench.py.
 * msst_debug_stamps: kernel-study builds (-DMSST_STAMPS) only; returns MSST_ERR_UNSUPPORTED otherwise. */
int msst_debug_stamps(void* device_buf /* >= 256 u64; kernel-study aid, see tools/stamps.py */);
/* Occupancy probe (diagnostic, bench.py --cu-thief): nblocks workgroups that only hold a CU each -- 256 threads and ALL 160 KB
 * of the CU's LDS, so that no workgroup that uses LDS (every MFMA kernel of this library does) fits beside one -- for `microseconds`,
 * enqueued on `stream`; sink: 4 bytes of device scratch.  Stands in for the channel workgroups of an RCCL collective when
 * the overlap of the gradient all-reduce with the backward is studied on ONE GPU. */
int msst_debug_cu_thief(int nblocks, int microseconds, void* sink, void* stream);
/* Box probe (diagnostic, bench.py `box_probe`; MSST_VERSION 104): what THIS GPU sustains right now -- out[0] = TFLOP/s of back-to-back
 * v_mfma_f32_32x32x16_bf16 on hashed full-range operands (two waves per SIMD on every CU; the chip clocks to its power budget, so this
 * differs from box to box and from the 2.5 PFLOP/s nominal), out[1] = the shader clock in MHz that stream held, out[2] = GB/s of a
 * read-only stream over the scratch buffer, out[3] = seconds spent.  scratch: device memory, >= 1 MiB (>= 1 GiB for an HBM number:
 * the memory-side cache holds 256 MB).  The one entry point that SYNCHRONISES the stream (it times with HIP events); <= 0.3 s. */
int msst_debug_box_probe(double* out4, void* scratch, long scratch_bytes, void* stream);
int msst_profile_enable(int on);
/* restrict the event pairs to the kernel ids whose bit is set (default: all); each pair costs ~10 us of stream time */
int msst_profile_select(unsigned long long mask);
/* bracket only every n-th launch of each selected kernel (default 1 = all): keeps the event pairs inside a timed region cheap */
int msst_profile_sample(int every);
int msst_profile_kernels(void);
const char* msst_profile_name(int id);
int msst_profile_collect(double* total_ms /*host*/, long* count /*host*/);

#ifdef __cplusplus
}
#endif
#endif
