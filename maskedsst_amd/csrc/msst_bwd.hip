// TEMPORARY stubs (replaced by the backward kernels).
#include "../../include/msst.h"
extern "C" {
int msst_head_bwd(const float*, const float*, const int32_t*, const int32_t*, const float*, int, float, float*, float*, int, int, int, int, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_block_bwd_mlp(const MsstBlockWeights*, const float*, const float*, float*, float*, int, int, int, int, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_block_bwd_attn(const MsstBlockWeights*, const float*, const float*, void*, float*, int, int, int, int, int, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_block_bwd_ln1(const MsstBlockWeights*, const float*, const float*, const void*, float*, float*, int, int, int, int, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_reduce_slabs(const float*, int, long, float*, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_tokenize_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, const float*, float*, int, int, int, int, int, void*) { return MSST_ERR_UNSUPPORTED; }
int msst_adamw(float*, const float*, float*, float*, long, float, float, float, float, float, int, float, float, void*) { return MSST_ERR_UNSUPPORTED; }
}
