// extern "C" boundary of libmsst (see include/msst.h).  Thin argument marshalling only.
#include "../../include/msst.h"
#include "msst_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

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
__global__ __launch_bounds__(256) void prep_weights_kernel(const MsstPrepJob* jobs, int* err_flag) {
    const MsstPrepJob j = jobs[blockIdx.y];
    // malformed jobs are skipped (the job table lives in device memory: the host entry point cannot validate it) and reported
    // through the caller's error word: bit 0 = bad pack / shape, bit 1 = pack 1 on a shape that is not whole 32 x 16 fragments
    const bool half = sizeof(E) == 2 && (j.pack & MSST_PREP_HALF);   // destination elements IEEE half instead of bf16 (the fp16-operand forward)
    if (j.pack < 0 || (j.pack & ~MSST_PREP_HALF) > 1 || (sizeof(E) != 2 && (j.pack & MSST_PREP_HALF)) || j.rows < 1 || j.cols < 1) {
        if (err_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err_flag, 1);
        return;
    }
    const int pack = j.pack & 1;
    if (sizeof(E) == 2 && pack == 1 && (((j.transpose ? j.cols : j.rows) & 31) || ((j.transpose ? j.rows : j.cols) & 15))) {
        if (err_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err_flag, 2);
        return;
    }
    const int n = j.rows * j.cols;
    E* dst = reinterpret_cast<E*>(j.dst);
    if constexpr (sizeof(E) == 2) {
        // bf16: a thread writes the 8 consecutive k of one lane of one fragment -- 16 bytes of the fragment-packed destination
        // (see PBF16::ld_w; round 4: one 16-byte store instead of eight 2-byte ones, 55 -> see DESIGN us for the launch)
        const int R = j.transpose ? j.cols : j.rows, K = j.transpose ? j.rows : j.cols;   // logical [R][K] destination
        if ((K & 7) == 0) {
            const int RB = pack ? 32 : 16, KB = pack ? 16 : 32;   // fragment = RB rows x KB k, lane = (k / 8) * RB + row
            if ((R % RB) == 0 && (K % KB) == 0) {
                for (int i8 = blockIdx.x * 256 + threadIdx.x; i8 < n / 8; i8 += gridDim.x * 256) {
                    const int f = i8 / 64, lane = i8 - f * 64;
                    const int fr = f / (K / KB), fk = f - fr * (K / KB);
                    const int r = fr * RB + (lane % RB), c0 = fk * KB + (lane / RB) * 8;
                    s16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = c0 + e;
                        const int src_i = j.transpose ? c * j.cols + r : r * j.cols + c;   // dst[r][c] = src[c][r] when transposed
                        float v = j.src[src_i];
                        if (src_i / j.cols < j.scale_rows) v *= j.scale;
                        o[e] = (short)(half ? f2h(v) : f2bf(v));
                    }
                    *reinterpret_cast<s16x8*>(dst + (long)i8 * 8) = o;
                }
                return;
            }
        }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // i indexes the destination (coalesced writes)
        int src_i = i;
        if (j.transpose) {
            const int c = i / j.rows, r = i - c * j.rows;  // dst[c][r] = src[r][c]
            src_i = r * j.cols + c;
        }
        float v = j.src[src_i];
        // to_qkv^T for the round-3 attention backward: the q and k blocks (the first scale_rows source rows) carry the softmax scale
        // dim_head^-0.5 = 2^-3, exact in bf16 -- that kernel keeps dq / dk unscaled (reference vit_spatial_spectral.py:54,71)
        if (src_i / j.cols < j.scale_rows) v *= j.scale;
        if constexpr (sizeof(E) == 4) {
            dst[i] = v;
        } else {
            // fragment-packed destination (see PBF16::ld_w): logical (r, c) of the [R][K] destination matrix
            const int K = j.transpose ? j.rows : j.cols;
            const int r = i / K, c = i - r * K;
            if (pack >= 1) {   // 32 rows x 16 k per fragment (v_mfma_f32_32x32x16_bf16 operand: lane = row % 32 + 32 (k % 16 / 8))
                const int f = (r >> 5) * (K >> 4) + (c >> 4);
                const int lane = ((c & 15) >> 3) * 32 + (r & 31);
                dst[((long)f * 64 + lane) * 8 + (c & 7)] = half ? f2h(v) : f2bf(v);
            } else {
                const int f = (r >> 4) * (K >> 5) + (c >> 5);
                const int lane = ((c & 31) >> 3) * 16 + (r & 15);
                dst[((long)f * 64 + lane) * 8 + (c & 7)] = half ? f2h(v) : f2bf(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// opt-in profiler: a pair of HIP events around each kernel launch, on the launch stream
// ------------------------------------------------------------------------------------------
// The compute entry points keep no state: with the profiler off (the default) a call reads ONE relaxed atomic flag
// and nothing else that is shared.  The opt-in profiler state below is guarded by a mutex (records) and is
// per-thread where a launch is bracketed (the open record), so enabling it never makes a compute call unsafe.
struct ProfRec { int id; hipEvent_t a, b; };
#if defined(MSST_STAMPS) || defined(MSST_LAB)
static unsigned long long* g_stamps = nullptr;   // kernel-study builds only (python -m maskedsst_amd.build --stamps; -DMSST_LAB: the scratch of tools/gate_qkv.py)
#endif
static std::atomic<bool> g_prof_on{false};
static std::atomic<unsigned long long> g_prof_mask{~0ull};   // kernels (bit = id) that get event pairs
static std::atomic<int> g_prof_every{1};                      // ... every n-th launch of each (msst_profile_sample)
static std::atomic<unsigned> g_prof_seen[K_COUNT];
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;
static thread_local hipEvent_t g_open_end = nullptr;   // end event of the launch this thread is bracketing

static hipEvent_t pool_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}

void prof_begin(int id, hipStream_t st) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    if (!((g_prof_mask.load(std::memory_order_relaxed) >> id) & 1ull)) return;
    const int every = g_prof_every.load(std::memory_order_relaxed);
    if (every > 1 && g_prof_seen[id].fetch_add(1u, std::memory_order_relaxed) % (unsigned)every != 0u) return;
    ProfRec r;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        r.id = id; r.a = pool_event(); r.b = pool_event();
        g_prof.push_back(r);
    }
    hipEventRecord(r.a, st);
    g_open_end = r.b;
}
void prof_end(hipStream_t st) {
    if (!g_open_end) return;
    hipEventRecord(g_open_end, st);
    g_open_end = nullptr;
}

static TileMap make_tilemap(int mode, int B, int S, int N) {
    TileMap tm;
    tm.mode = mode;
    tm.N = N;
    tm.nshift = -1;
    for (int sh = 0; sh < 31; ++sh) if ((1 << sh) == N) tm.nshift = sh;
    tm.T = S * N;
    tm.L = mode == MSST_MODE_SPATIAL ? N : S;
    tm.TS = 64 / tm.L;
    tm.nseq = mode == MSST_MODE_SPATIAL ? B * S : B * N;
    return tm;
}

static Drop make_drop(float p, uint32_t seed, int layer) {
    Drop d;
    d.seed = seed; d.layer = layer;
    if (p > 0.f) {
        d.thr = (unsigned)(p * 65536.0f + 0.5f);
        if (d.thr > 65535u) d.thr = 65535u;   // the kernels test whole words against thr << 16 (p >= 1 is degenerate anyway)
        d.scale = 1.0f / (1.0f - (float)d.thr / 65536.0f);   // exact inverse of the realised keep probability
    } else {
        d.thr = 0; d.scale = 1.0f;
    }
    return d;
}

static int ntiles_of(const TileMap& tm) { return (tm.nseq + tm.TS - 1) / tm.TS; }

static bool bw_ok(const MsstBlockWeights* w) { return w && w->struct_bytes == sizeof(MsstBlockWeights); }

static BlockWeights to_bw(const MsstBlockWeights* w) {
    BlockWeights b;
    b.wqkv = w->wqkv; b.wout = w->wout; b.w1 = w->w1; b.w2 = w->w2;
    b.wqkvT = w->wqkvT; b.woutT = w->woutT; b.w1T = w->w1T; b.w2T = w->w2T;
    b.ln1_g = w->ln1_g; b.ln1_b = w->ln1_b; b.bo = w->bo;
    b.ln2_g = w->ln2_g; b.ln2_b = w->ln2_b; b.b1 = w->b1; b.b2 = w->b2;
    b.wqkv32 = w->wqkv32; b.woutT32 = w->woutT32; b.wqkvT32 = w->wqkvT32;
    return b;
}

}  // namespace msst

using namespace msst;

extern "C" {

// a kernel-study build (-DMSST_LAB: timing modes that compute wrong results can be compiled in) identifies itself with a
// negative version; maskedsst_amd/_lib.py refuses to load it
#ifdef MSST_LAB
int msst_version(void) { return -MSST_VERSION; }
#else
int msst_version(void) { return MSST_VERSION; }
#endif
const char* msst_last_error(void) { return g_err; }

int msst_debug_stamps(void* buf) {
#if defined(MSST_STAMPS) || defined(MSST_LAB)
    g_stamps = (unsigned long long*)buf;
    return 0;
#else
    (void)buf;
    return fail(MSST_ERR_UNSUPPORTED, "msst_debug_stamps (library built without -DMSST_STAMPS)");
#endif
}

int msst_debug_cu_thief(int nblocks, int microseconds, void* sink, void* stream) {
    return fail(launch_cu_thief(nblocks, microseconds, (unsigned*)sink, (hipStream_t)stream), "msst_debug_cu_thief");
}

int msst_debug_box_probe(double* out4, void* scratch, long scratch_bytes, void* stream) {
    return fail(launch_box_probe(out4, scratch, scratch_bytes, (hipStream_t)stream), "msst_debug_box_probe");
}

int msst_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) { g_prof.clear(); g_pool_next = 0; }
    g_prof_on.store(on != 0);
    return 0;
}

int msst_profile_select(unsigned long long mask) { g_prof_mask.store(mask); return 0; }

int msst_profile_sample(int every) {
    if (every < 1) return fail(MSST_ERR_BADARG, "msst_profile_sample");
    g_prof_every.store(every);
    for (int i = 0; i < K_COUNT; ++i) g_prof_seen[i].store(0u);
    return 0;
}

int msst_profile_kernels(void) { return K_COUNT; }

const char* msst_profile_name(int id) {
    static const char* names[K_COUNT] = {"prep_weights", "tokenize_fwd", "block_fwd", "head_fwd", "loss_reduce",
                                         "head_bwd", "reduce_slabs", "block_bwd_mlp", "block_bwd_attn",
                                         "attn_slab_reduce", "block_bwd_ln1", "tokenize_bwd", "pos_split", "adamw", "block_bwd_ln1mlp", "layernorm"};
    return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}

int msst_profile_collect(double* total_ms, long* count) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0.0; count[i] = 0; }
    for (const ProfRec& r : g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) return fail(MSST_ERR_BADARG, "msst_profile_collect");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        total_ms[r.id] += ms;
        count[r.id] += 1;
    }
    g_prof.clear();
    g_pool_next = 0;
    return 0;
}

int msst_prep_weights(const MsstPrepJob* jobs, int njobs, int job_bytes, int max_elems, int prec, int32_t* err_flag, void* stream) {
    // a caller built against another revision of MsstPrepJob would hand over a mis-strided table: refused on the host
    if (job_bytes != (int)sizeof(MsstPrepJob)) return fail(MSST_ERR_BADARG, "msst_prep_weights (MsstPrepJob of another header revision)");
    if (njobs <= 0) return 0;
    if (!jobs) return fail(MSST_ERR_BADARG, "msst_prep_weights");
    ProfScope ps(K_PREP, (hipStream_t)stream);
    int gx = (max_elems + 256 * 4 - 1) / (256 * 4);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    dim3 grid(gx, njobs);
    if (prec == MSST_PREC_F32) hipLaunchKernelGGL(prep_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, jobs, (int*)err_flag);
    else hipLaunchKernelGGL(prep_weights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, jobs, (int*)err_flag);
    return fail((int)hipGetLastError(), "msst_prep_weights");
}

int msst_tokenize_fwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const float* pos_a,
                      const float* pos_b, int pos_split, const float* mask_token, const uint8_t* mask,
                      float* out, int B, int S, int N, int P, float emb_dropout_p, uint32_t seed, void* stream) {
    TokArgs a;
    a.drop = make_drop(emb_dropout_p, seed, 255);
    a.img = img; a.pre_g = pre_g; a.pre_b = pre_b; a.w_emb = w_emb; a.b_emb = b_emb;
    a.post_g = post_g; a.post_b = post_b; a.pos_a = pos_a; a.pos_b = pos_b; a.mask_token = mask_token;
    a.mask = mask; a.out = out; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P; a.pos_split = pos_split;
    return fail(launch_tokenize_fwd(a, (hipStream_t)stream), "msst_tokenize_fwd");
}

long msst_block_lse_floats(int mode, int B, int S, int N, int heads) {
    if (B < 1 || S < 1 || N < 1 || heads < 1 || N > 64 || S > 64) return 0;
    return (long)ntiles_of(make_tilemap(mode, B, S, N)) * heads * 64 + (long)B * S * N;   // [tiles][heads][64] lse | [tokens] rstd of LN1
}

long msst_block_tiles(int mode, int B, int S, int N) {
    if (B < 1 || S < 1 || N < 1 || N > 64 || S > 64) return 0;
    return (long)ntiles_of(make_tilemap(mode, B, S, N));
}

int msst_block_fwd(const MsstBlockWeights* w, const float* x, float* y, float* x1, int mode, int B, int S,
                   int N, int heads, int prec, int max_grid, float dropout_p, uint32_t seed, int layer,
                   void* xn_out, float* lse_out, int* saved, void* stream) {
    if (!bw_ok(w) || !x || !y || x == y) return fail(MSST_ERR_BADARG, "msst_block_fwd (null argument, or MsstBlockWeights of another header revision)");
    if (N > 64 || S > 64) return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd (sequence length > 64)");
    const int dbg = (prec >> 8) & 0xffff;   // MSST_KERNEL_* selection flags ride in the upper bits of `prec`
    prec &= 0xff;
    BlockArgs a;
    a.w = to_bw(w);
    a.x = x; a.y = y; a.x1 = x1;
    a.tm = make_tilemap(mode, B, S, N);
    a.ntiles = ntiles_of(a.tm);
    a.max_grid = max_grid > 0 ? max_grid : a.ntiles;
    a.H = heads;
    a.scale = 0.125f;  // dim_head ** -0.5, dim_head = 64 (vit_spatial_spectral.py:54)
    a.dbg = dbg & ~8;
    a.stamps = nullptr;
#ifdef MSST_STAMPS
    a.stamps = g_stamps;
    if (g_stamps) a.dbg = dbg;
#elif defined(MSST_LAB)
    a.stamps = g_stamps;
#endif
    // MSST_X1_BF16: only the role-split forward (bf16, 8 heads, no kernel selection flags) writes bf16 x1 rows
    a.x1_bf16 = (dbg & 1024) ? 1 : 0;
    if (a.x1_bf16 && !(prec == MSST_PREC_BF16 && heads == 8 && !(dbg & (16 | 64))))
        return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd (MSST_X1_BF16 needs the role-split bf16 forward: 8 heads, no MSST_KERNEL_* flags)");
    // MSST_FWD_HALF: fp16 operands -- the role-split forward only, and the caller must have prepared the half copies of its four matrices
    a.half = (dbg & 4096) ? 1 : 0;
    if (a.half) {
        if (!(prec == MSST_PREC_BF16 && heads == 8 && !(dbg & (16 | 64))))
            return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd (MSST_FWD_HALF needs the role-split bf16 forward: 8 heads, no MSST_KERNEL_* flags)");
        if (!w->wqkv_h || !w->wout_h || !w->w1_h || !w->w2_h) return fail(MSST_ERR_BADARG, "msst_block_fwd (MSST_FWD_HALF without the half weight copies)");
        a.w.wqkv = w->wqkv_h; a.w.wout = w->wout_h; a.w.w1 = w->w1_h; a.w.w2 = w->w2_h;
    }
    a.drop = make_drop(dropout_p, seed, layer);
    a.xn_out = (xn_out && block_fwd_writes_xn(a, prec)) ? xn_out : nullptr;
    a.lse_out = (lse_out && block_fwd_writes_lse(a, prec)) ? lse_out : nullptr;
    if (saved) *saved = (a.xn_out ? MSST_SAVED_XN : 0) | (a.lse_out ? (MSST_SAVED_LSE | MSST_SAVED_RSTD) : 0);
    return fail(launch_block_fwd(a, prec, (hipStream_t)stream), "msst_block_fwd");
}

int msst_block_fwd_stack(const MsstBlockWeights* const* w, int nblk, const float* x0, float* const* y, float* const* x1,
                         void* const* xn_out, float* const* lse_out, int mode, int B, int S, int N, int heads, int prec, int max_grid,
                         float dropout_p, uint32_t seed, int layer0, int* saved, void* stream) {
    if (!w || !x0 || !y || nblk < 1) return fail(MSST_ERR_BADARG, "msst_block_fwd_stack");
    if (nblk > MSST_MAX_STACK || N > 64 || S > 64) return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd_stack (more than 16 blocks, or sequence length > 64)");
    const int dbg = (prec >> 8) & 0xffff;
    prec &= 0xff;
    // the role-split bf16 forward only: 8 heads, no kernel selection flags (MSST_X1_BF16 is the one flag it takes)
    if (prec != MSST_PREC_BF16 || heads != 8 || (dbg & ~(1024 | 4096))) return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd_stack (bf16, 8 heads, no MSST_KERNEL_* flags)");
    const bool half = (dbg & 4096) != 0;
    for (int j = 0; j < nblk; ++j) {
        if (!bw_ok(w[j])) return fail(MSST_ERR_BADARG, "msst_block_fwd_stack (null block, or MsstBlockWeights of another header revision)");
        if (half && (!w[j]->wqkv_h || !w[j]->wout_h || !w[j]->w1_h || !w[j]->w2_h)) return fail(MSST_ERR_BADARG, "msst_block_fwd_stack (MSST_FWD_HALF without the half weight copies)");
    }
    StackArgs sa;
    BlockArgs& a = sa.base;
    memset(&a.w, 0, sizeof(a.w));
    a.x = x0; a.y = nullptr; a.x1 = nullptr; a.xn_out = nullptr; a.lse_out = nullptr;
    a.tm = make_tilemap(mode, B, S, N);
    a.ntiles = ntiles_of(a.tm);
    a.max_grid = max_grid > 0 ? max_grid : a.ntiles;
    a.H = heads;
    a.scale = 0.125f;
    a.dbg = 0;
    a.stamps = nullptr;
    a.x1_bf16 = (dbg & 1024) ? 1 : 0;
    a.half = half ? 1 : 0;
    a.drop = make_drop(dropout_p, seed, layer0);
    if (lse_out && !block_fwd_writes_lse(a, prec)) lse_out = nullptr;   // same rule as msst_block_fwd: a statistics buffer past the 31-bit descriptor range is declined, not clipped
    sa.nblk = nblk;
    // the kernel addresses block j's operands as block 0's + j x a byte stride: every array of the call must be affine in the block
    // index (maskedsst_amd lays its weight copies, parameters and activations out that way); anything else is refused
    auto blk_of = [&](int j, const float* xin) {
        StackBlk sb;
        sb.wqkv = half ? w[j]->wqkv_h : w[j]->wqkv; sb.wout = half ? w[j]->wout_h : w[j]->wout;
        sb.w1 = half ? w[j]->w1_h : w[j]->w1; sb.w2 = half ? w[j]->w2_h : w[j]->w2;
        sb.ln1_g = w[j]->ln1_g; sb.ln1_b = w[j]->ln1_b; sb.bo = w[j]->bo; sb.ln2_g = w[j]->ln2_g; sb.ln2_b = w[j]->ln2_b; sb.b1 = w[j]->b1; sb.b2 = w[j]->b2;
        sb.x = xin; sb.y = y[j]; sb.x1 = x1 ? x1[j] : nullptr; sb.xn_out = xn_out ? xn_out[j] : nullptr; sb.lse_out = lse_out ? lse_out[j] : nullptr;
        sb.layer = layer0 + j; sb.pad_ = 0;
        return sb;
    };
    for (int j = 0; j < nblk; ++j)
        if (!bw_ok(w[j]) || !y[j] || y[j] == (j ? y[j - 1] : x0))
            return fail(MSST_ERR_BADARG, "msst_block_fwd_stack (null argument, or MsstBlockWeights of another header revision)");
    sa.b0 = blk_of(0, x0);
    memset(&sa.st, 0, sizeof(sa.st));
    bool affine = true;
    if (nblk > 1) {
        const StackBlk b1 = blk_of(1, y[0]);
        auto dist = [&](const void* p1, const void* p0, int& out) {
            const long d = (const char*)p1 - (const char*)p0;
            if (d > 0x7fffffffL || d < -0x7fffffffL) affine = false;
            out = (int)d;
        };
        dist(b1.wqkv, sa.b0.wqkv, sa.st.wqkv); dist(b1.wout, sa.b0.wout, sa.st.wout); dist(b1.w1, sa.b0.w1, sa.st.w1); dist(b1.w2, sa.b0.w2, sa.st.w2);
        dist(b1.ln1_g, sa.b0.ln1_g, sa.st.ln1_g); dist(b1.ln1_b, sa.b0.ln1_b, sa.st.ln1_b); dist(b1.bo, sa.b0.bo, sa.st.bo);
        dist(b1.ln2_g, sa.b0.ln2_g, sa.st.ln2_g); dist(b1.ln2_b, sa.b0.ln2_b, sa.st.ln2_b); dist(b1.b1, sa.b0.b1, sa.st.b1); dist(b1.b2, sa.b0.b2, sa.st.b2);
        dist(b1.y, sa.b0.y, sa.st.y); dist(b1.x1, sa.b0.x1, sa.st.x1); dist(b1.xn_out, sa.b0.xn_out, sa.st.xn_out); dist(b1.lse_out, sa.b0.lse_out, sa.st.lse_out);
        // block j > 0 reads what block j - 1 wrote: x_j = y_(j-1) = y_0 + (j - 1) stride_y -- affine from block 1 on; block 0 reads x0.  The
        // kernel takes x_j = x_base + j stride_y with x_base = y_0 - stride_y for j >= 1 and x0 for j = 0: keep both
        sa.st.x = sa.st.y;
        const float* xin = y[0];
        for (int j = 1; j < nblk && affine; ++j) {
            const StackBlk bj = blk_of(j, xin);
            auto same = [&](const void* pj, const void* p0, int stride) { return !p0 ? !pj : (const char*)pj == (const char*)p0 + (long)j * stride; };
            affine = same(bj.wqkv, sa.b0.wqkv, sa.st.wqkv) && same(bj.wout, sa.b0.wout, sa.st.wout) && same(bj.w1, sa.b0.w1, sa.st.w1) &&
                     same(bj.w2, sa.b0.w2, sa.st.w2) && same(bj.ln1_g, sa.b0.ln1_g, sa.st.ln1_g) && same(bj.ln1_b, sa.b0.ln1_b, sa.st.ln1_b) &&
                     same(bj.bo, sa.b0.bo, sa.st.bo) && same(bj.ln2_g, sa.b0.ln2_g, sa.st.ln2_g) && same(bj.ln2_b, sa.b0.ln2_b, sa.st.ln2_b) &&
                     same(bj.b1, sa.b0.b1, sa.st.b1) && same(bj.b2, sa.b0.b2, sa.st.b2) && same(bj.y, sa.b0.y, sa.st.y) &&
                     same(bj.x1, sa.b0.x1, sa.st.x1) && same(bj.xn_out, sa.b0.xn_out, sa.st.xn_out) && same(bj.lse_out, sa.b0.lse_out, sa.st.lse_out);
            xin = y[j];
        }
    }
    if (!affine) return fail(MSST_ERR_UNSUPPORTED, "msst_block_fwd_stack (per-block operands not a constant stride apart)");
    sa.x_rest = nblk > 1 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(y[0]) - (long)sa.st.y) : x0;   // x of block j >= 1 = x_rest + j stride_y
    if (saved) *saved = (xn_out ? MSST_SAVED_XN : 0) | (lse_out ? (MSST_SAVED_LSE | MSST_SAVED_RSTD) : 0);
    if (a.ntiles < 1) return 0;
    int ncu = 256, dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
    int grid = a.max_grid < a.ntiles ? a.max_grid : a.ntiles;
    if (grid > ncu) grid = ncu;
    return fail(launch_block_fwd_rs_stack(sa, grid, (hipStream_t)stream), "msst_block_fwd_stack");
}

int msst_head_fwd(const float* y, const float* img, const int32_t* idx, const float* w_pix,
                  const float* b_pix, int per_block, float* dpred, float* pred, float* partial,
                  float* loss, int B, int S, int N, int P, int K, void* stream) {
    HeadArgs a;
    a.y = y; a.img = img; a.idx = idx; a.w_pix = w_pix; a.b_pix = b_pix; a.dpred = dpred; a.pred = pred;
    a.partial = partial; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P; a.K = K; a.per_block = per_block;
    return fail(launch_head_fwd(a, loss, (hipStream_t)stream), "msst_head_fwd");
}

int msst_head_bwd(const float* y, const float* dpred, const int32_t* csr_ptr, const int32_t* csr_pos,
                  const float* w_pix, int per_block, float gscale, const float* gout, float* dy, float* slab,
                  int nchunk, float* dw_pix, float* db_pix, int B, int S, int N, int P, int K, void* stream) {
    if (nchunk < 1) return fail(MSST_ERR_BADARG, "msst_head_bwd");
    hipStream_t st = (hipStream_t)stream;
    HeadBwdArgs a;
    a.y = y; a.dpred = dpred; a.csr_ptr = csr_ptr; a.csr_pos = csr_pos; a.w_pix = w_pix; a.dy = dy; a.slab = slab;
    a.gscale = gscale; a.gout = gout; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P; a.K = K; a.per_block = per_block;
    int rc = launch_head_bwd(a, nchunk, st);
    if (rc) return fail(rc, "msst_head_bwd");
    const long ss = (long)P * 96 + P;
    RSegBuilder rb;
    bool ok = true;
    if (per_block) {
        for (int c = 0; c < S && ok; ++c) {
            ok = rb.add(slab + (long)c * nchunk * ss, ss, nchunk, dw_pix + (long)c * P * 96, P * 96);
            ok = ok && rb.add(slab + (long)c * nchunk * ss + P * 96, ss, nchunk, db_pix + (long)c * P, P);
            if (rb.r.nseg > MSST_MAX_RSEG - 2 && c + 1 < S) {   // table full (S > 36): flush and start a new one
                rc = launch_reduce_segs(rb.r, st);
                if (rc) return fail(rc, "msst_head_bwd(reduce)");
                rb = RSegBuilder();
            }
        }
    } else {
        ok = rb.add(slab, ss, S * nchunk, dw_pix, P * 96);
        ok = ok && rb.add(slab + P * 96, ss, S * nchunk, db_pix, P);
    }
    if (!ok) return fail(MSST_ERR_UNSUPPORTED, "msst_head_bwd(reduce table)");
    rc = launch_reduce_segs(rb.r, st);
    return fail(rc, "msst_head_bwd(reduce)");
}

// where the parts of one block backward keep their partial-gradient slabs inside the caller's `slab` workspace
struct SlabLayout {
    int grid, grid_mlp, nc;
    float *mlp, *attn, *ln1, *mlp_prev;
};
static SlabLayout slab_layout(float* slab, long ntok, int grid_rows, int nchunk, int ntiles_attn, int heads, int prec) {
    SlabLayout l;
    const int ntiles_rows = (int)((ntok + 63) / 64);
    l.grid = grid_rows < ntiles_rows ? grid_rows : ntiles_rows;   // LN1 backward: HBM bound at one workgroup per CU
    // the bf16 MLP backward is latency bound and fits two workgroups per CU: twice the persistent grid (and slabs)
    const int gmlp2 = (prec == MSST_PREC_BF16 && PBF16::WAVES_BWD_MLP == 2) ? 2 * grid_rows : grid_rows;
    l.grid_mlp = gmlp2 < ntiles_rows ? gmlp2 : ntiles_rows;
    l.nc = nchunk < ntiles_attn ? nchunk : ntiles_attn;
    l.mlp = slab;
    l.attn = l.mlp + (long)l.grid_mlp * MSST_MLP_SLAB_N;
    l.ln1 = l.attn + (long)l.nc * heads * MSST_ATTN_SLAB_N;
    l.mlp_prev = l.ln1 + (long)l.grid * MSST_LN1_SLAB;
    return l;
}
// the reduction table of one block backward (or, ny_* > 1, of a run of chained ones: see msst_block_bwd_reduce).
// g_mlp: gradients the standalone MLP half wrote slabs for (null: it did not run); g_prev: the same for the fused launch's MLP half
static int reduce_block_slabs(const SlabLayout& lay, int heads, const MsstBlockGrads* g, const MsstBlockGrads* g_mlp, int ny_mlp,
                              const MsstBlockGrads* g_prev, int ny_prev, int ny, long src_ystride, long dst_ystride, hipStream_t st) {
    RSegBuilder rb;
    auto fresh = [&]() { rb = RSegBuilder(); rb.r.src_ystride = src_ystride; rb.r.dst_ystride = dst_ystride; };
    fresh();
    const long ms = MSST_MLP_SLAB_N;
    bool ok = true;
    auto add_mlp = [&](const float* sm_, int nslab, const MsstBlockGrads* gg, int nyy) {
        ok = ok && rb.add(sm_, ms, nslab, gg->w1, 6144, 0, 0, nyy);
        ok = ok && rb.add(sm_ + 6144, ms, nslab, gg->w2, 6144, 0, 0, nyy);
        ok = ok && rb.add(sm_ + 12288, ms, nslab, gg->b1, 64, 0, 0, nyy);
        ok = ok && rb.add(sm_ + 12288 + 64, ms, nslab, gg->b2, 96, 0, 0, nyy);
        ok = ok && rb.add(sm_ + 12288 + 160, ms, nslab, gg->ln2_g, 96, 0, 0, nyy);
        ok = ok && rb.add(sm_ + 12288 + 256, ms, nslab, gg->ln2_b, 96, 0, 0, nyy);
    };
    if (g_mlp) add_mlp(lay.mlp, lay.grid_mlp, g_mlp, ny_mlp);
    if (g_prev) add_mlp(lay.mlp_prev, lay.grid, g_prev, ny_prev);
    if (g) {
        ok = ok && rb.add(lay.ln1, 288, lay.grid, g->ln1_g, 96, 0, 0, ny);
        ok = ok && rb.add(lay.ln1 + 96, 288, lay.grid, g->ln1_b, 96, 0, 0, ny);
        ok = ok && rb.add(lay.ln1 + 192, 288, lay.grid, g->bo, 96, 0, 0, ny);
        const long as = (long)heads * MSST_ATTN_SLAB_N;
        const int inner = heads * 64;
        for (int h = 0; h < heads && ok; ++h) {
            const float* sh = lay.attn + (long)h * MSST_ATTN_SLAB_N;
            for (int which = 0; which < 3 && ok; ++which)
                ok = rb.add(sh + which * 6144, as, lay.nc, g->wqkv + ((long)(which * heads + h) * 64) * 96, 6144, 0, 0, ny);
            ok = ok && rb.add(sh + 3 * 6144, as, lay.nc, g->wout + h * 64, 6144, 64, inner, ny);
            if (rb.r.nseg > MSST_MAX_RSEG - 4 && h + 1 < heads) {   // flush when the table is nearly full
                int rc = launch_reduce_segs(rb.r, st);
                if (rc) return rc;
                fresh();
            }
        }
    }
    if (!ok) return MSST_ERR_UNSUPPORTED;
    return launch_reduce_segs(rb.r, st);
}

// chain == 0: the three halves of ONE block (MLP half -> attention half -> LN1 backward), msst_block_bwd.
// chain != 0: msst_block_bwd_chain (see include/msst.h): [MLP half of block i when `first`] -> attention half of block i ->
//             LN1 backward of block i fused with the MLP half of block i - 1 (w_prev), or alone (block 0).
static int block_bwd_impl(const MsstBlockWeights* w, const MsstBlockGrads* g, const MsstBlockWeights* w_prev,
                          const MsstBlockGrads* g_prev, const float* x, const float* x1, const float* x1_prev,
                          const float* dy, float* dx, float* dx1, void* dxn_part, float* slab, int grid_rows,
                          int nchunk, int mode, int B, int S, int N, int heads, int prec, float dropout_p,
                          uint32_t seed, int layer, const void* xn_saved, const float* lse_saved, void* dab_ws, int chain, int first,
                          int32_t* tile_queue, hipStream_t st) {
    if (!bw_ok(w) || (w_prev && !bw_ok(w_prev)) || !g || grid_rows < 1 || nchunk < 1)
        return fail(MSST_ERR_BADARG, "msst_block_bwd (null argument, or MsstBlockWeights of another header revision)");
    if (N > 64 || S > 64) return fail(MSST_ERR_UNSUPPORTED, "msst_block_bwd (sequence length > 64)");
    const int dbg = (prec >> 8) & 0xffff;   // MSST_KERNEL_* selection flags ride in the upper bits of `prec`
    prec &= 0xff;
    const long ntok = (long)B * S * N;
    const BlockWeights bw = to_bw(w);
    const Drop drop = make_drop(dropout_p, seed, layer);
    AttnBwdArgs aa;
    aa.tm = make_tilemap(mode, B, S, N);
    aa.ntiles = ntiles_of(aa.tm);
    const SlabLayout lay = slab_layout(slab, ntok, grid_rows, nchunk, aa.ntiles, heads, prec);
    const int grid = lay.grid, grid_mlp = lay.grid_mlp, nc = lay.nc;
    float* slab_mlp = lay.mlp;
    float* slab_attn = lay.attn;
    float* slab_ln1 = lay.ln1;
    float* slab_mlp_prev = lay.mlp_prev;   // chain: the fused launch's MLP slabs (block i - 1), one per workgroup
    // saved LN1 rows + pre-dropped bf16 da rows: both or neither, and only for the tuned bf16 attention kernel
    const bool fast_rows = xn_saved && dab_ws && prec == MSST_PREC_BF16 && !(dbg & 16);
    if (chain && (!fast_rows || (w_prev && (!g_prev || !x1_prev)) || (first && !dy) || (!w_prev && !dx)))
        return fail(MSST_ERR_BADARG, "msst_block_bwd_chain (bf16 with saved LN1 rows and the dab workspace only)");
    const bool run_mlp = !chain || first;
    // MSST_X1_BF16: the saved mid-residual rows (x1, x1_prev) are bf16 -- the bf16 MLP-half kernels read either kind
    const int x1b = (dbg & 1024) ? 1 : 0;
    if (x1b && prec != MSST_PREC_BF16)
        return fail(MSST_ERR_UNSUPPORTED, "msst_block_bwd (MSST_X1_BF16 needs the bf16 kernels)");
    // dynamic tile queues (data parallel): counters [0, heads / 2) for the two-head attention backward, [32] for the fused launch
    if (tile_queue) {
        hipError_t e = hipMemsetAsync(tile_queue, 0, MSST_TILE_QUEUE_WORDS * sizeof(int32_t), st);
        if (e != hipSuccess) return fail((int)e, "msst_block_bwd_chain(tile queue)");
    }
    // 1. MLP half: dy -> dx1
    if (run_mlp) {
        MlpBwdArgs a;
        a.w = bw; a.x1 = x1; a.dy = dy; a.dx1 = dx1; a.slab = slab_mlp; a.ntok = ntok; a.drop = drop;
        a.dab = fast_rows ? dab_ws : nullptr;
        a.x1_bf16 = x1b;
        int rc = launch_block_bwd_mlp(a, grid_mlp, prec, st);
        if (rc) return fail(rc, "msst_block_bwd(mlp)");
    }
    // 2. attention half, per (chunk, head)
    int nparts = heads;
    {
        aa.w = bw; aa.x = x; aa.da = dx1; aa.dxn_part = dxn_part; aa.slab = slab_attn;
        aa.xn = fast_rows ? xn_saved : nullptr; aa.dab = fast_rows ? dab_ws : nullptr;
        aa.H = heads; aa.ntok = ntok; aa.scale = 0.125f; aa.drop = drop;
        aa.queue = (tile_queue && heads / 2 <= 32) ? tile_queue : nullptr;
        aa.lse = fast_rows ? lse_saved : nullptr;   // (only the two-head kernel reads it; launch_block_bwd_attn drops it for the others)
        aa.lse_renorm = (dbg & 8192) ? 1 : 0;       // MSST_LSE_RENORM: statistics of a half-operand forward
        aa.dbg = dbg & ~8;
        aa.stamps = nullptr;
#ifdef MSST_STAMPS
        aa.stamps = g_stamps;
        if (g_stamps) aa.dbg = dbg;
#elif defined(MSST_LAB)
        aa.stamps = g_stamps;
#endif
        int rc = launch_block_bwd_attn(aa, nc, prec, st, &nparts);
        if (rc) return fail(rc, "msst_block_bwd(attn)");
    }
    // 3. LN1 backward + residual -- alone, or fused with the MLP half of the block before (dx stays on chip)
    const bool fused = chain && w_prev;
    if (fused) {
        if (nparts > 4) return fail(MSST_ERR_UNSUPPORTED, "msst_block_bwd_chain (more than four d(LN1 out) partials)");
        LnMlpArgs a;
        a.w = to_bw(w_prev); a.ln1_g = w->ln1_g; a.x = x; a.dx1 = dx1; a.dxn_part = dxn_part; a.x1 = x1_prev; a.dab = dab_ws;
        a.x1_bf16 = x1b;
        // MSST_LN1_FROM_XN: xhat of LN1 from the saved bf16 LN1 rows + the saved rstd (the tail of the statistics buffer)
        a.xn = nullptr; a.rstd = nullptr; a.ln1_b = w->ln1_b;
        if (dbg & 2048) {
            if (!xn_saved || !lse_saved) return fail(MSST_ERR_BADARG, "msst_block_bwd_chain (MSST_LN1_FROM_XN needs xn_saved and lse_saved)");
            a.xn = xn_saved; a.rstd = lse_saved + (long)aa.ntiles * heads * 64;
        }
        a.slab_mlp = slab_mlp_prev; a.slab_ln1 = slab_ln1; a.ntok = ntok; a.nparts = nparts;
        a.drop_i = drop; a.drop_p = make_drop(dropout_p, seed, layer - 1);
        a.stamps = nullptr;
#ifdef MSST_STAMPS
        a.stamps = g_stamps;
#endif
        a.queue = tile_queue ? tile_queue + 32 : nullptr;
        int rc = launch_block_bwd_ln1mlp(a, grid, st);
        if (rc) return fail(rc, "msst_block_bwd_chain(ln1 + mlp)");
    } else {
        Ln1BwdArgs a;
        a.x = x; a.dx1 = dx1; a.dxn_part = dxn_part; a.ln1_g = w->ln1_g; a.dx = dx; a.slab = slab_ln1; a.ntok = ntok; a.H = nparts; a.drop = drop;
        int rc = launch_block_bwd_ln1(a, grid, prec, st);
        if (rc) return fail(rc, "msst_block_bwd(ln1)");
    }
    // 4. one deterministic reduction of all partial-gradient slabs written above -- unless the caller collects the slab sets of a
    //    run of chained calls and reduces them in one launch (MSST_BWD_DEFER_REDUCE, msst_block_bwd_reduce)
    if (!(chain && (dbg & 512))) {
        int rc = reduce_block_slabs(lay, heads, g, run_mlp ? g : nullptr, 1, fused ? g_prev : nullptr, 1, 1, 0, 0, st);
        if (rc) return fail(rc, "msst_block_bwd(reduce)");
    }
    return 0;
}

int msst_block_bwd(const MsstBlockWeights* w, const MsstBlockGrads* g, const float* x, const float* x1,
                   const float* dy, float* dx, float* dx1, void* dxn_part, float* slab, int grid_rows,
                   int nchunk, int mode, int B, int S, int N, int heads, int prec, float dropout_p,
                   uint32_t seed, int layer, const void* xn_saved, const float* lse_saved, void* dab_ws, void* stream) {
    return block_bwd_impl(w, g, nullptr, nullptr, x, x1, nullptr, dy, dx, dx1, dxn_part, slab, grid_rows, nchunk, mode, B, S, N,
                          heads, prec, dropout_p, seed, layer, xn_saved, lse_saved, dab_ws, 0, 0, nullptr, (hipStream_t)stream);
}

int msst_block_bwd_chain(const MsstBlockWeights* w, const MsstBlockGrads* g, const MsstBlockWeights* w_prev,
                         const MsstBlockGrads* g_prev, const float* x, const float* x1, const float* x1_prev,
                         const float* dy, float* dx, float* dx1, void* dxn_part, float* slab, int grid_rows,
                         int nchunk, int mode, int B, int S, int N, int heads, int prec, float dropout_p,
                         uint32_t seed, int layer, const void* xn_saved, const float* lse_saved, void* dab_ws, int first,
                         int32_t* tile_queue, void* stream) {
    return block_bwd_impl(w, g, w_prev, g_prev, x, x1, x1_prev, dy, dx, dx1, dxn_part, slab, grid_rows, nchunk, mode, B, S, N,
                          heads, prec, dropout_p, seed, layer, xn_saved, lse_saved, dab_ws, 1, first, tile_queue, (hipStream_t)stream);
}

int msst_block_bwd_reduce(const MsstBlockGrads* g, const MsstBlockGrads* g_prev, float* slab, long slab_stride, long grad_stride,
                          int count, int count_prev, int first, int grid_rows, int nchunk, int mode, int B, int S, int N, int heads,
                          int prec, void* stream) {
    if (!g || count < 1 || count_prev < 0 || count_prev > count || (count_prev && !g_prev) || !slab || grid_rows < 1 || nchunk < 1 ||
        (count > 1 && ((slab_stride & 3) || (grad_stride & 3))))
        return fail(MSST_ERR_BADARG, "msst_block_bwd_reduce");
    prec &= 0xff;
    const long ntok = (long)B * S * N;
    const TileMap tm = make_tilemap(mode, B, S, N);
    const SlabLayout lay = slab_layout(slab, ntok, grid_rows, nchunk, ntiles_of(tm), heads, prec);
    return fail(reduce_block_slabs(lay, heads, g, first ? g : nullptr, 1, count_prev ? g_prev : nullptr, count_prev, count,
                                   slab_stride, grad_stride, (hipStream_t)stream), "msst_block_bwd_reduce");
}

int msst_tokenize_bwd(const float* img, const float* pre_g, const float* pre_b, const float* w_emb,
                      const float* b_emb, const float* post_g, const float* post_b, const uint8_t* mask,
                      const float* dx0, float* slab, int nchunk, float* dpre_g, float* dpre_b,
                      float* dw_emb, float* db_emb, float* dpost_g, float* dpost_b, float* dpos_a,
                      float* dpos_b, int pos_split, float* dmask_token, int B, int S, int N, int P,
                      float emb_dropout_p, uint32_t seed, void* stream) {
    if (nchunk < 1) return fail(MSST_ERR_BADARG, "msst_tokenize_bwd");
    hipStream_t st = (hipStream_t)stream;
    TokBwdArgs a;
    a.drop = make_drop(emb_dropout_p, seed, 255);
    a.img = img; a.pre_g = pre_g; a.pre_b = pre_b; a.w_emb = w_emb; a.b_emb = b_emb; a.post_g = post_g;
    a.post_b = post_b; a.mask = mask; a.dx0 = dx0; a.slab = slab; a.B = B; a.S = S; a.N = N; a.T = S * N; a.P = P;
    int rc = launch_tokenize_bwd(a, nchunk, st);
    if (rc) return fail(rc, "msst_tokenize_bwd");
    const long ss = (long)N * 96 + 96 * P + 96 * 4 + 32;
    const long bs = (long)nchunk * ss;
    float* stage = slab + (long)S * nchunk * ss;  // [S][N][96] position-gradient staging
    float* dpos_dst = pos_split ? stage : dpos_a;
    const long v0 = (long)N * 96 + 96 * P;
    RSegBuilder rb;
    bool ok = true;
    for (int c = 0; c < S && ok; ++c) {   // per spectral block: position rows, embed weight, embed bias
        const float* sc = slab + c * bs;
        if (dpos_a) ok = rb.add(sc, ss, nchunk, dpos_dst + (long)c * N * 96, N * 96);   // null: position table applied by the caller
        ok = ok && rb.add(sc + N * 96, ss, nchunk, dw_emb + (long)c * 96 * P, 96 * P);
        ok = ok && rb.add(sc + v0, ss, nchunk, db_emb + (long)c * 96, 96);
        if (rb.r.nseg > MSST_MAX_RSEG - 8) {
            rc = launch_reduce_segs(rb.r, st);
            if (rc) return fail(rc, "msst_tokenize_bwd(reduce)");
            rb = RSegBuilder();
        }
    }
    // shared across blocks
    ok = ok && rb.add(slab + v0 + 96, ss, S * nchunk, dpost_g, 96);
    ok = ok && rb.add(slab + v0 + 192, ss, S * nchunk, dpost_b, 96);
    if (dmask_token) ok = ok && rb.add(slab + v0 + 288, ss, S * nchunk, dmask_token, 96);
    ok = ok && rb.add(slab + v0 + 384, ss, S * nchunk, dpre_g, P);
    ok = ok && rb.add(slab + v0 + 384 + 16, ss, S * nchunk, dpre_b, P);
    if (!ok) return fail(MSST_ERR_UNSUPPORTED, "msst_tokenize_bwd(reduce table)");
    rc = launch_reduce_segs(rb.r, st);
    if (!rc && pos_split && dpos_a) rc = launch_pos_split(stage, S, N, pos_split, dpos_a, dpos_b, st);
    return fail(rc, "msst_tokenize_bwd(reduce)");
}

int msst_cls_head_fwd(const float* y, const float* ln_g, const float* ln_b, const float* w, const float* b,
                      float* logits, int B, int S, int N, int n_classes, void* stream) {
    ClsArgs a;
    a.y = y; a.ln_g = ln_g; a.ln_b = ln_b; a.w = w; a.b = b; a.logits = logits;
    a.B = B; a.S = S; a.N = N; a.T = S * N; a.NC = n_classes;
    return fail(launch_cls_head_fwd(a, (hipStream_t)stream), "msst_cls_head_fwd");
}

int msst_cls_head_bwd(const float* y, const float* dlogits, const float* ln_g, const float* ln_b, const float* w,
                      float* dy, float* slab, float* dln_g, float* dln_b, float* dw, float* db, int B, int S,
                      int N, int n_classes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ClsBwdArgs a;
    a.y = y; a.dlogits = dlogits; a.ln_g = ln_g; a.ln_b = ln_b; a.w = w; a.dy = dy; a.slab = slab;
    a.B = B; a.S = S; a.N = N; a.T = S * N; a.NC = n_classes;
    int rc = launch_cls_head_bwd(a, st);
    if (rc) return fail(rc, "msst_cls_head_bwd");
    const long ss = (long)n_classes * 96 + n_classes + 192;
    RSegBuilder rb;
    bool ok = rb.add(slab, ss, B, dw, n_classes * 96);
    ok = ok && rb.add(slab + n_classes * 96, ss, B, db, n_classes);
    ok = ok && rb.add(slab + n_classes * 96 + n_classes, ss, B, dln_g, 96);
    ok = ok && rb.add(slab + n_classes * 96 + n_classes + 96, ss, B, dln_b, 96);
    if (!ok) return fail(MSST_ERR_UNSUPPORTED, "msst_cls_head_bwd");
    return fail(launch_reduce_segs(rb.r, st), "msst_cls_head_bwd(reduce)");
}

int msst_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows,
                       int D, float eps, void* stream) {
    return fail(launch_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, eps, (hipStream_t)stream), "msst_layernorm_fwd");
}

long msst_layernorm_bwd_slab(long rows, int D) { return rows < 1 || D < 1 ? 0 : (long)layernorm_bwd_grid(rows, D) * 2 * D; }

int msst_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, float* slab,
                       long rows, int D, float eps, void* stream) {
    if (!dgamma || !dbeta) return fail(MSST_ERR_BADARG, "msst_layernorm_bwd");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {   // "fully written" holds for an empty input too: the sums over no rows
        if (D < 1) return fail(MSST_ERR_BADARG, "msst_layernorm_bwd");
        if (hipMemsetAsync(dgamma, 0, sizeof(float) * D, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * D, st) != hipSuccess)
            return fail(MSST_ERR_BADARG, "msst_layernorm_bwd(memset)");
        return 0;
    }
    const int grid = layernorm_bwd_grid(rows, D);
    int rc = launch_layernorm_bwd(x, gamma, dy, dx, slab, grid, rows, D, eps, st);
    if (rc) return fail(rc, "msst_layernorm_bwd");
    RSegBuilder rb;
    bool ok = rb.add(slab, 2L * D, grid, dgamma, D);
    ok = ok && rb.add(slab + D, 2L * D, grid, dbeta, D);
    if (!ok) return fail(MSST_ERR_UNSUPPORTED, "msst_layernorm_bwd");
    return fail(launch_reduce_segs(rb.r, st), "msst_layernorm_bwd(reduce)");
}

int msst_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float clamp, float gscale, void* stream) {
    return fail(launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, clamp, gscale,
                             (hipStream_t)stream), "msst_adamw");
}

}  // extern "C"
