// extern "C" boundary of libmsst (see include/msst.h).  Thin argument marshalling only.
#include "../../include/msst.h"
#include "msst_kernels.h"
#include <stdio.h>
#include <string.h>

namespace msst {

static thread_local char g_err[256] = "";

static int fail(int code, const char* what) {
    if (code > 0) snprintf(g_err, sizeof(g_err), "%s: hip error %d (%s)", what, code, hipGetErrorString((hipError_t)code));
    else if (code < 0) snprintf(g_err, sizeof(g_err), "%s: msst error %d", what, code);
    return code;
}

// ------------------------------------------------------------------------------------------
// weight prep: fp32 master -> operand element type, optional transpose.  One launch for all jobs.
// ------------------------------------------------------------------------------------------
template <class E>
__global__ __launch_bounds__(256) void prep_weights_kernel(const MsstPrepJob* jobs) {
    const MsstPrepJob j = jobs[blockIdx.y];
    const int n = j.rows * j.cols;
    E* dst = reinterpret_cast<E*>(j.dst);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // i indexes the destination (coalesced writes)
        int src_i = i;
        if (j.transpose) {
            const int c = i / j.rows, r = i - c * j.rows;  // dst[c][r] = src[r][c]
            src_i = r * j.cols + c;
        }
        const float v = j.src[src_i];
        if constexpr (sizeof(E) == 4) dst[i] = v; else dst[i] = f2bf(v);
    }
}

static TileMap make_tilemap(int mode, int B, int S, int N) {
    TileMap tm;
    tm.mode = mode;
    tm.N = N;
    tm.T = S * N;
    tm.L = mode == MSST_MODE_SPATIAL ? N : S;
    tm.TS = 64 / tm.L;
    tm.nseq = mode == MSST_MODE_SPATIAL ? B * S : B * N;
    return tm;
}

static int ntiles_of(const TileMap& tm) { return (tm.nseq + tm.TS - 1) / tm.TS; }

static BlockWeights to_bw(const MsstBlockWeights* w) {
    BlockWeights b;
    b.wqkv = w->wqkv; b.wout = w->wout; b.w1 = w->w1; b.w2 = w->w2;
    b.wqkvT = w->wqkvT; b.woutT = w->woutT; b.w1T = w->w1T; b.w2T = w->w2T;
    b.ln1_g = w->ln1_g; b.ln1_b = w->ln1_b; b.bo = w->bo;
    b.ln2_g = w->ln2_g; b.ln2_b = w->ln2_b; b.b1 = w->b1; b.b2 = w->b2;
    return b;
}

}  // namespace msst

using namespace msst;

extern "C" {

int msst_version(void) { return MSST_VERSION; }
const char* msst_last_error(void) { return g_err; }

int msst_prep_weights(const MsstPrepJob* jobs, int njobs, int max_elems, int prec, void* stream) {
    if (njobs <= 0) return 0;
    int gx = (max_elems + 256 * 4 - 1) / (256 * 4);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    dim3 grid(gx, njobs);
    if (prec == MSST_PREC_F32) hipLaunchKernelGGL(prep_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, jobs);
    else hipLaunchKernelGGL(prep_weights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, jobs);
    return fail((int)hipGetLastError(), "msst_prep_weights");
}

int msst_tokenize_fwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const float* pos_a,
                      const float* pos_b, int pos_split, const float* mask_token, const uint8_t* mask,
                      float* out, int B, int S, int N, int P, void* stream) {
    TokArgs a;
    a.img = img; a.pre_g = pre_g; a.pre_b = pre_b; a.w_emb = w_emb; a.b_emb = b_emb;
    a.post_g = post_g; a.post_b = post_b; a.pos_a = pos_a; a.pos_b = pos_b; a.mask_token = mask_token;
    a.mask = mask; a.out = out; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P; a.pos_split = pos_split;
    return fail(launch_tokenize_fwd(a, (hipStream_t)stream), "msst_tokenize_fwd");
}

int msst_block_fwd(const MsstBlockWeights* w, const float* x, float* y, float* x1, int mode, int B, int S,
                   int N, int heads, int prec, int max_grid, void* stream) {
    if (!w || !x || !y || x == y) return fail(MSST_ERR_BADARG, "msst_block_fwd");
    if (N > 64 || S > 64) return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd (sequence length > 64)");
    BlockArgs a;
    a.w = to_bw(w);
    a.x = x; a.y = y; a.x1 = x1;
    a.tm = make_tilemap(mode, B, S, N);
    a.ntiles = ntiles_of(a.tm);
    a.max_grid = max_grid > 0 ? max_grid : a.ntiles;
    a.H = heads;
    a.scale = 0.125f;  // dim_head ** -0.5, dim_head = 64 (vit_spatial_spectral.py:54)
    return fail(launch_block_fwd(a, prec, (hipStream_t)stream), "msst_block_fwd");
}

int msst_head_fwd(const float* y, const float* img, const int32_t* idx, const float* w_pix,
                  const float* b_pix, int per_block, float* dpred, float* pred, float* partial,
                  float* loss, int B, int S, int N, int P, int K, void* stream) {
    HeadArgs a;
    a.y = y; a.img = img; a.idx = idx; a.w_pix = w_pix; a.b_pix = b_pix; a.dpred = dpred; a.pred = pred;
    a.partial = partial; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P; a.K = K; a.per_block = per_block;
    return fail(launch_head_fwd(a, loss, (hipStream_t)stream), "msst_head_fwd");
}

}  // extern "C"
