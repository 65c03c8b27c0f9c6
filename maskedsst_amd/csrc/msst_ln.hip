// Stand-alone LayerNorm over the last axis, forward and backward, fp32 (SURVEY.md 8b export list; reference
// vit_spatial_spectral.py:25 -- PreNorm's nn.LayerNorm(96) -- and :194-195 -- the tokenizer's LayerNorm(10) / LayerNorm(96);
// eps 1e-5, affine, biased variance as torch.nn.functional.layer_norm).  On the hot path LayerNorm only exists fused
// into its consumers (tokenizer, block forward, block backward); these two entry points are the same arithmetic as an op
// of its own for callers that want a lone LayerNorm (and the unit the fused ones are checked against).
//
// Mapping: a row is owned by LPR consecutive lanes of a wave (8 lanes x three 16-byte loads for D = 96: every load
// instruction covers whole 128-byte segments of a row; 16 lanes x scalar loads for any other D <= 128, D = 10 included),
// so a wave holds 64 / LPR rows and the row statistics are wave-shuffle reductions over LPR lanes (DPP quad permutes and
// row mirrors: no LDS round trip).  HBM-bound: x in, y out (forward); x, dy in, dx out (backward).
// Backward: dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); d gamma = sum_rows dy xhat, d beta = sum_rows dy are kept
// per lane in registers over the workgroup's rows, combined through LDS once per workgroup and left as one slab per
// workgroup, which launch_reduce_segs adds up in a fixed order (bit-reproducible).
#include "msst_dev.h"
#include "msst_kernels.h"

namespace msst {

namespace {

// sum over the LPR (8 or 16) consecutive lanes that own a row, result on every one of them -- VALU only
template <int LPR>
__device__ __forceinline__ float lanes_sum(float v) {
    v = quad_sum(v);
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));                    // row_half_mirror: lane i <-> 7 - i of its half row
    if (LPR == 16) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror: lane i <-> 15 - i
    return v;
}

// D == 96: VEC = true, LPR = 8, lane j of a row holds the float4 pieces j, j + 8, j + 16;  else: LPR = 16, lane j holds elements j + 16 t
template <bool VEC>
struct RowTraits {
    static constexpr int LPR = VEC ? 8 : 16;
    static constexpr int NV = VEC ? 12 : 8;   // values per lane
    static constexpr int ROWS = 64 / LPR;      // rows per wave
};

template <bool VEC>
__device__ __forceinline__ void ld_row(const float* p, int j, int D, float (&v)[RowTraits<VEC>::NV]) {
    if constexpr (VEC) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + 4 * (j + 8 * t));
            v[4 * t] = q[0]; v[4 * t + 1] = q[1]; v[4 * t + 2] = q[2]; v[4 * t + 3] = q[3];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (j + 16 * t < D) ? p[j + 16 * t] : 0.f;
    }
}
template <bool VEC>
__device__ __forceinline__ void st_row(float* p, int j, int D, const float (&v)[RowTraits<VEC>::NV]) {
    if constexpr (VEC) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f32x4 q = {v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
            *reinterpret_cast<f32x4*>(p + 4 * (j + 8 * t)) = q;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) if (j + 16 * t < D) p[j + 16 * t] = v[t];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, int D, float eps) {
    typedef RowTraits<VEC> T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane % T::LPR, rsub = lane / T::LPR;
    float gv[T::NV], bv[T::NV];
    ld_row<VEC>(g, j, D, gv);
    ld_row<VEC>(b, j, D, bv);
    const float invD = 1.0f / (float)D;
    for (long r0 = ((long)blockIdx.x * 4 + wv) * T::ROWS; r0 < rows; r0 += (long)gridDim.x * 4 * T::ROWS) {
        const long r = r0 + rsub;
        const bool on = r < rows;
        const long rr = on ? r : rows - 1;   // clamped address: the load is unconditional, the store is not
        float v[T::NV];
        ld_row<VEC>(x + rr * D, j, D, v);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) s += v[t];
        const float mean = lanes_sum<T::LPR>(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) {
            const float d = (VEC || j + 16 * t < D) ? v[t] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(lanes_sum<T::LPR>(q) * invD + eps);
#pragma unroll
        for (int t = 0; t < T::NV; ++t) v[t] = (v[t] - mean) * rstd * gv[t] + bv[t];
        if (on) {
            st_row<VEC>(y + r * D, j, D, v);
            if (j == 0) {
                if (mean_out) mean_out[r] = mean;
                if (rstd_out) rstd_out[r] = rstd;
            }
        }
    }
}

// slab per workgroup: [d gamma (D) | d beta (D)]
template <bool VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ slab, long rows, int D, float eps) {
    typedef RowTraits<VEC> T;
    __shared__ float red[2][4 * T::ROWS][T::LPR * T::NV];   // [d gamma | d beta][row slot of the workgroup][column image of a row]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane % T::LPR, rsub = lane / T::LPR;
    float gv[T::NV], ag[T::NV], ab[T::NV];
    ld_row<VEC>(g, j, D, gv);
#pragma unroll
    for (int t = 0; t < T::NV; ++t) { ag[t] = 0.f; ab[t] = 0.f; }
    const float invD = 1.0f / (float)D;
    for (long r0 = ((long)blockIdx.x * 4 + wv) * T::ROWS; r0 < rows; r0 += (long)gridDim.x * 4 * T::ROWS) {
        const long r = r0 + rsub;
        const bool on = r < rows;
        const long rr = on ? r : rows - 1;
        float v[T::NV], d[T::NV];
        ld_row<VEC>(x + rr * D, j, D, v);
        ld_row<VEC>(dy + rr * D, j, D, d);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) s += v[t];
        const float mean = lanes_sum<T::LPR>(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) {
            v[t] = (VEC || j + 16 * t < D) ? v[t] - mean : 0.f;
            q += v[t] * v[t];
        }
        const float rstd = rsqrtf(lanes_sum<T::LPR>(q) * invD + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) {
            v[t] *= rstd;                 // xhat
            const float dg = d[t] * gv[t];
            s1 += dg;
            s2 += dg * v[t];
            if (on) { ag[t] += d[t] * v[t]; ab[t] += d[t]; }
        }
        const float m1 = lanes_sum<T::LPR>(s1) * invD, m2 = lanes_sum<T::LPR>(s2) * invD;
#pragma unroll
        for (int t = 0; t < T::NV; ++t) d[t] = rstd * (d[t] * gv[t] - m1 - v[t] * m2);
        if (on) st_row<VEC>(dx + r * D, j, D, d);
    }
    // the workgroup's 4 * ROWS row slots -> one slab row, fixed order
    const int slot = wv * T::ROWS + rsub;
#pragma unroll
    for (int t = 0; t < T::NV; ++t) {
        red[0][slot][j * T::NV + t] = ag[t];
        red[1][slot][j * T::NV + t] = ab[t];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, col = i - which * D;
        // column col lives with lane jj, register tt of a row image
        int jj, tt;
        if (VEC) { const int piece = col >> 2; jj = piece & 7; tt = 4 * (piece >> 3) + (col & 3); }
        else { jj = col & 15; tt = col >> 4; }
        float sacc = 0.f;
        for (int s_ = 0; s_ < 4 * T::ROWS; ++s_) sacc += red[which][s_][jj * T::NV + tt];
        slab[(long)blockIdx.x * 2 * D + i] = sacc;
    }
}

int ln_grid(long rows, int rows_per_wg) {
    long g = (rows + rows_per_wg - 1) / rows_per_wg;
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;   // persistent: four workgroups per CU
    return (int)g;
}

}  // namespace

int launch_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, long rows, int D,
                         float eps, hipStream_t st) {
    if (!x || !g || !b || !y || rows < 0 || D < 1 || D > 128) return MSST_ERR_BADARG;
    if (rows == 0) return 0;
    ProfScope ps(K_LAYERNORM, st);
    if (D == 96) hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(ln_grid(rows, 32)), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, D, eps);
    else hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(ln_grid(rows, 16)), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, D, eps);
    return (int)hipGetLastError();
}

int layernorm_bwd_grid(long rows, int D) { return ln_grid(rows, D == 96 ? 32 : 16); }

int launch_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* slab, int grid, long rows, int D,
                         float eps, hipStream_t st) {
    if (!x || !g || !dy || !dx || !slab || rows < 1 || D < 1 || D > 128 || grid < 1) return MSST_ERR_BADARG;
    ProfScope ps(K_LAYERNORM, st);
    if (D == 96) hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(grid), dim3(256), 0, st, x, g, dy, dx, slab, rows, D, eps);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(grid), dim3(256), 0, st, x, g, dy, dx, slab, rows, D, eps);
    return (int)hipGetLastError();
}

}  // namespace msst
