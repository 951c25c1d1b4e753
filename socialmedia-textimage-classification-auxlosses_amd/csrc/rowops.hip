// Bandwidth-bound row kernels: LayerNorm fwd/bwd, text-embedding gather + LN (+ its backward scatter), ViT patchify /
// token assembly, column sums (bias gradients), fp32 -> 16-bit casts (plain and transposed).
// One wave per row, 8-byte (4 x 16-bit) or 16-byte (4 x fp32) accesses per lane, fp32 arithmetic.
#include <cstdlib>
#include <type_traits>
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

static constexpr int MAXC = 4;   // 4-element chunks per lane: width <= 64*4*4 = 1024

// MMHIP_DETERMINISTIC=1 (read at every launch, so a test can flip it): fp32 atomics whose arrival order decides the last bits of a sum
// are replaced by single-writer sums in a fixed order -- the second stage of the column reductions runs one block per column group, the
// embedding backward stores its per-slot rows and a second kernel adds the rows of equal word / position id in slot order.  The
// reference's CPU path is deterministic; the default (atomic) path differs between identical runs by ~3e-8 relative in a gradient.
bool deterministic() {
    const char* e = getenv("MMHIP_DETERMINISTIC");
    return e && atoi(e) != 0;
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v) {
    typename Vec<T>::v4 x = *reinterpret_cast<const typename Vec<T>::v4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = to_f<T>(x[e]);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x[e];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v) {
    typename Vec<T>::v4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = from_f<T>(v[e]);
    *reinterpret_cast<typename Vec<T>::v4*>(p) = x;
}

// parity mode: 4 values into a plane pair (mmhip_kernels.h): hi = bf16(x) at p, lo = bf16(x - hi) `lo` elements behind
__device__ __forceinline__ void store4_pair(bf16_t* p, int lo, const float* v) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)v[e]; l[e] = (bf16_t)(v[e] - (float)h[e]); }
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + lo) = l;
}

// ------------------------------------------------------------------------------------------------ LayerNorm forward
// A wave takes LN_FWD_RPW rows and requests all of them before it uses the first: one row per wave left ~1 KB in flight per wave and the kernel
// latency-bound at a third of the HBM rate (round 4: 12.7 us for 12608 x 768 bf16 rows, 38 % of 8 TB/s; profiles/r05_rowop_bench.txt)
static constexpr int LN_FWD_RPW = 2;
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LNArgs a) {
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_FWD_RPW;
    if (row0 >= a.rows) return;
    const int nch = a.width >> 2;
    typedef typename Vec<T>::v4 v4;
    v4 raw[LN_FWD_RPW][MAXC];
#pragma unroll
    for (int i = 0; i < LN_FWD_RPW; ++i) {
        const int row = min(row0 + i, a.rows - 1);
#pragma unroll
        for (int t = 0; t < MAXC; ++t)
            if (lane + 64 * t < nch) raw[i][t] = *reinterpret_cast<const v4*>((const T*)a.x + (size_t)row * a.ldx + (lane + 64 * t) * 4);
    }
#pragma unroll
    for (int i = 0; i < LN_FWD_RPW; ++i) {
        const int row = row0 + i;
        if (row >= a.rows) break;
        float v[MAXC][4];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t)
            if (lane + 64 * t < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[t][e] = to_f<T>(raw[i][t][e]); s += v[t][e]; }
            }
        const float mean = wave_sum(s) / a.width;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t)
            if (lane + 64 * t < nch)
#pragma unroll
                for (int e = 0; e < 4; ++e) { float d = v[t][e] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / a.width + a.eps);
        T* y = (T*)a.y + (size_t)row * a.ldy;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nch) {
                float g[4], b[4], o[4];
                load4<float>(a.gamma + c * 4, g);
                load4<float>(a.beta + c * 4, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[t][e] - mean) * rstd * g[e] + b[e];
                if (a.y) store4<T>(y + c * 4, o);
                if (std::is_same<T, float>::value && a.y_pair) store4_pair((bf16_t*)a.y_pair + (size_t)row * a.ld_pair + c * 4, a.lo_pair, o);
            }
        }
        if (lane == 0 && a.mean) { a.mean[row] = mean; a.rstd[row] = rstd; }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres), g = dy * gamma; dgamma += dy * xhat, dbeta += dy.
// A block walks ROWS_PER_BLOCK rows (wave-strided), keeps the column partials in registers, reduces the 4 waves
// through LDS and issues one atomic per column per block.
static constexpr int LN_ROWS_PER_BLOCK = 16;
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LNBwdArgs a) {
    __shared__ float red[3][4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nch = a.width >> 2;
    const bool dropping = a.dx_drop != nullptr && a.drop.thresh16 != 0;
    float dg[MAXC][4], db[MAXC][4], dc[MAXC][4], gam[MAXC][4];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[t][e] = 0.f; db[t][e] = 0.f; dc[t][e] = 0.f; gam[t][e] = 0.f; }
        if (lane + 64 * t < nch) load4<float>(a.gamma + (lane + 64 * t) * 4, gam[t]);
    }
    const int r0 = blockIdx.x * LN_ROWS_PER_BLOCK;
    // all of this wave's rows are requested before the first one is used: 2 blocks per CU hold only 8 waves, so the
    // bytes in flight per wave, not the wave count, must cover the HBM latency (24 -> 96 KB in flight per CU)
    constexpr int RPW = LN_ROWS_PER_BLOCK / 4;
    typedef typename Vec<T>::v4 v4;
    v4 xraw[RPW][MAXC], dyraw[RPW][MAXC], rraw[RPW][MAXC];
    float mean_r[RPW], rstd_r[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = r0 + w + 4 * i;
        const bool live = row < a.rows;
        mean_r[i] = live ? a.mean[row] : 0.f;
        rstd_r[i] = live ? a.rstd[row] : 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (live && c < nch) {
                xraw[i][t] = *reinterpret_cast<const v4*>((const T*)a.x + (size_t)row * a.width + c * 4);
                dyraw[i][t] = *reinterpret_cast<const v4*>((const T*)a.dy + (size_t)row * a.width + c * 4);
                if (a.dres) rraw[i][t] = *reinterpret_cast<const v4*>((const T*)a.dres + (size_t)row * a.width + c * 4);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = r0 + w + 4 * i;
        if (row >= a.rows) break;
        const float mean = mean_r[i], rstd = rstd_r[i];
        float xh[MAXC][4], g[MAXC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xv = to_f<T>(xraw[i][t][e]), dv = to_f<T>(dyraw[i][t][e]);
                    xh[t][e] = (xv - mean) * rstd;
                    g[t][e] = dv * gam[t][e];
                    s1 += g[t][e];
                    s2 += g[t][e] * xh[t][e];
                    dg[t][e] += dv * xh[t][e];
                    db[t][e] += dv;
                }
            }
        }
        const float c1 = wave_sum(s1) / a.width, c2 = wave_sum(s2) / a.width;
        T* dx = (T*)a.dx + (size_t)row * a.width;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nch) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (g[t][e] - c1 - xh[t][e] * c2);
                if (a.dres) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += to_f<T>(rraw[i][t][e]);
                }
                store4<T>(dx + c * 4, o);
                if (dropping) {
                    const uint32_t e0 = (uint32_t)row * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.width + (uint32_t)c * 4u;
                    bool k0, k1, k2, k3;
                    mm_keep2(e0, a.drop, k0, k1);
                    mm_keep2(e0 + 2, a.drop, k2, k3);
                    o[0] = k0 ? o[0] * a.drop.keep_scale : 0.f;
                    o[1] = k1 ? o[1] * a.drop.keep_scale : 0.f;
                    o[2] = k2 ? o[2] * a.drop.keep_scale : 0.f;
                    o[3] = k3 ? o[3] * a.drop.keep_scale : 0.f;
                    if (!(std::is_same<T, float>::value && a.pair_out)) store4<T>((T*)a.dx_drop + (size_t)row * a.width + c * 4, o);
                }
                // parity mode: what the following matrix products read (the dropped gradient where dropout is on) as a plane pair
                if (std::is_same<T, float>::value && a.pair_out) {
                    bf16_t* pp = (bf16_t*)a.pair_out + (size_t)row * a.ld_pair + c * 4;
                    if (a.pair_hi_only) { bf16x4 h; h[0] = (bf16_t)o[0]; h[1] = (bf16_t)o[1]; h[2] = (bf16_t)o[2]; h[3] = (bf16_t)o[3]; *reinterpret_cast<bf16x4*>(pp) = h; }
                    else store4_pair(pp, a.lo_pair, o);
                }
                if (a.colsum_out) {      // sums of the values as the bias-gradient consumer sees them (16-bit rounded)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dc[t][e] += to_f<T>(from_f<T>(o[e]));
                }
            }
        }
    }
    const int nvec = a.colsum_out ? 3 : 2;
#pragma unroll
    for (int t = 0; t < MAXC; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * t) * 4 + e;
                red[0][w][c] = dg[t][e]; red[1][w][c] = db[t][e]; red[2][w][c] = dc[t][e];
            }
    __syncthreads();
    float* outs[3] = {a.dgamma, a.dbeta, a.colsum_out};
    for (int c = threadIdx.x; c < a.width; c += 256)
        for (int v = 0; v < nvec; ++v) {
            const float sres = red[v][0][c] + red[v][1][c] + red[v][2][c] + red[v][3][c];
            if (a.partial) a.partial[((size_t)v * gridDim.x + blockIdx.x) * a.width + c] = sres;   // [nvec][grid][width]
            else atomicAdd(outs[v] + c, sres * (a.alpha == 0.f ? 1.f : a.alpha));
        }
}
// out[c] += sum_i partial[i][c]   (i < n).  Block = 64 columns x 4 waves; grid.y splits the partial rows; each wave
// walks its rows with 4 independent accumulators; the 4 waves combine through LDS and one atomic per column per
// block finishes (grid.y-way contention only).
static constexpr int RP_SPLIT = 8;
struct ReduceOuts { float* out[3]; };
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial0, int n, int cols, ReduceOuts outs, float alpha, int single_writer) {
    __shared__ float red[4][64];
    const float* __restrict__ partial = partial0 + (size_t)blockIdx.z * n * cols;     // vector blockIdx.z of [nvec][n][cols]
    float* __restrict__ out = outs.out[blockIdx.z];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        const int stride = 4 * gridDim.y;
        int i = blockIdx.y * 4 + w;
        for (; i + 3 * stride < n; i += 4 * stride) {
            s0 += partial[(size_t)i * cols + c];
            s1 += partial[(size_t)(i + stride) * cols + c];
            s2 += partial[(size_t)(i + 2 * stride) * cols + c];
            s3 += partial[(size_t)(i + 3 * stride) * cols + c];
        }
        for (; i < n; i += stride) s0 += partial[(size_t)i * cols + c];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < cols) {
        const float v = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * alpha;
        if (single_writer) out[c] += v;          // gridDim.y == 1: this block is the column's only writer
        else atomicAdd(out + c, v);
    }
}
static inline void launch_reduce_partials(const float* partial, int n, int cols, float* out, hipStream_t s, float alpha = 1.0f,
                                          float* out1 = nullptr, float* out2 = nullptr) {
    const bool det = deterministic();
    const int split = (!det && n >= 4 * RP_SPLIT) ? RP_SPLIT : 1;
    const int nvec = out2 ? 3 : (out1 ? 2 : 1);
    ReduceOuts o{{out, out1, out2}};
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((cols + 63) / 64, split, nvec), dim3(256), 0, s, partial, n, cols, o, alpha, det ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------ text embeddings
// position ids: XLM-R  cumsum(ids != pad) * (ids != pad) + pad  (HF xlm_roberta :142-155); BERT arange(T).
__global__ __launch_bounds__(64) void pos_ids_kernel(const int64_t* ids, const int64_t* mask, int* pos_ids, float* maskbias, int T, int xlmr, int pad_id) {
    const int post = blockIdx.x, lane = threadIdx.x;
    int running = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < T;
        const bool nz = in && ids[(size_t)post * T + t] != pad_id;
        const unsigned long long bal = __ballot(nz);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull)) + (nz ? 1 : 0);
        if (in) {
            pos_ids[(size_t)post * T + t] = xlmr ? (nz ? running + pre + pad_id : pad_id) : t;
            maskbias[(size_t)post * T + t] = mask[(size_t)post * T + t] ? 0.f : -INFINITY;
        }
        running += __popcll(bal);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.posts * a.T) return;
    const int nch = a.H >> 2;
    const int64_t id = a.ids[row];
    const int pid = a.pos_ids[row];
    const float* wr = a.word + (size_t)id * a.H;
    const float* pr = a.pos + (size_t)pid * a.H;
    const float* tr = a.type + (a.type_ids ? (size_t)a.type_ids[row] * a.H : 0);
    float v[MAXC][4];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        const int c = lane + 64 * t;
        if (c < nch) {
            float x1[4], x2[4], x3[4];
            load4<float>(wr + c * 4, x1);
            load4<float>(pr + c * 4, x2);
            load4<float>(tr + c * 4, x3);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[t][e] = x1[e] + x3[e] + x2[e]; s += v[t][e]; }
        }
    }
    const float mean = wave_sum(s) / a.H;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < MAXC; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[t][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / a.H + a.eps);
    T* y = (T*)a.x + (size_t)row * a.H;
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        const int c = lane + 64 * t;
        if (c < nch) {
            float g[4], b[4], o[4], xh[4];
            load4<float>(a.gamma + c * 4, g);
            load4<float>(a.beta + c * 4, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xh[e] = (v[t][e] - mean) * rstd; o[e] = xh[e] * g[e] + b[e]; }
            if (a.xhat) store4<T>((T*)a.xhat + (size_t)row * a.H + c * 4, xh);
            if (a.drop.thresh16) {
                const uint32_t e0 = (uint32_t)row * (uint32_t)a.H + (uint32_t)c * 4u;
                bool k0, k1, k2, k3;
                mm_keep2(e0, a.drop, k0, k1);
                mm_keep2(e0 + 2, a.drop, k2, k3);
                o[0] = k0 ? o[0] * a.drop.keep_scale : 0.f;
                o[1] = k1 ? o[1] * a.drop.keep_scale : 0.f;
                o[2] = k2 ? o[2] * a.drop.keep_scale : 0.f;
                o[3] = k3 ? o[3] * a.drop.keep_scale : 0.f;
            }
            store4<T>(y + c * 4, o);
            if (std::is_same<T, float>::value && a.x_pair) store4_pair((bf16_t*)a.x_pair + (size_t)row * a.ld_pair + c * 4, a.lo_pair, o);
        }
    }
    if (lane == 0 && a.rstd) a.rstd[row] = rstd;
}

// backward: dropout -> LN backward (from saved xhat, rstd) -> scatter-add into word / position / type tables.
// Rows of nn.Embedding(padding_idx=...) get no gradient (word row pad_id; XLM-R position row pad_id).
// Block (t, g) = token slot t of posts 16g..16g+15, 4 posts per wave: every post of a slot normally shares one
// position id, so the position-row gradient is summed in registers, combined over the block's waves in LDS and
// leaves as ONE atomic row per block (same-address atomics from every post were the cost of the row-major
// version).  Word rows are scattered with lane-dense atomics (each wave instruction covers 256 contiguous bytes)
// staged through a wave-private LDS row.
static constexpr int EMB_POSTS_PER_BLOCK = 16;
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[3][4][1024];
    float (*rowbuf)[1024] = red[0];          // wave-private staging rows while the posts are walked; reductions afterwards
    __shared__ int wpid[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nch = a.H >> 2;
    const int tok = blockIdx.x, blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    const float al = a.alpha == 0.f ? 1.f : a.alpha;
    float dg[MAXC][4], db[MAXC][4], dt[MAXC][4], gam[MAXC][4], pacc[MAXC][4];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[t][e] = 0.f; db[t][e] = 0.f; dt[t][e] = 0.f; gam[t][e] = 0.f; pacc[t][e] = 0.f; }
        if (lane + 64 * t < nch) load4<float>(a.gamma + (lane + 64 * t) * 4, gam[t]);
    }
    int pid0 = -1;
    bool seen_bad = false;
    const int p0 = blockIdx.y * EMB_POSTS_PER_BLOCK + w * (EMB_POSTS_PER_BLOCK / 4);
    for (int pp = 0; pp < EMB_POSTS_PER_BLOCK / 4; ++pp) {
        const int post = p0 + pp;
        if (post >= a.posts) break;
        const int row = post * a.T + tok;
        const float rstd = a.rstd[row];
        float xh[MAXC][4], g[MAXC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nch) {
                float dv[4];
                load4<T>((const T*)a.dx + (size_t)row * a.H + c * 4, dv);
                load4<T>((const T*)a.xhat + (size_t)row * a.H + c * 4, xh[t]);
                if (a.drop.thresh16) {
                    const uint32_t e0 = (uint32_t)row * (uint32_t)a.H + (uint32_t)c * 4u;
                    bool k0, k1, k2, k3;
                    mm_keep2(e0, a.drop, k0, k1);
                    mm_keep2(e0 + 2, a.drop, k2, k3);
                    dv[0] = k0 ? dv[0] * a.drop.keep_scale : 0.f;
                    dv[1] = k1 ? dv[1] * a.drop.keep_scale : 0.f;
                    dv[2] = k2 ? dv[2] * a.drop.keep_scale : 0.f;
                    dv[3] = k3 ? dv[3] * a.drop.keep_scale : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[t][e] = dv[e] * gam[t][e];
                    s1 += g[t][e];
                    s2 += g[t][e] * xh[t][e];
                    dg[t][e] += dv[e] * xh[t][e];
                    db[t][e] += dv[e];
                }
            }
        }
        const float c1 = wave_sum(s1) / a.H, c2 = wave_sum(s2) / a.H;
        // the row sums are non-finite as soon as one element of the incoming gradient row is (inf, NaN; inf * 0 = NaN): the overflow guard
        if (!(fabsf(c1) <= 3.4028234e38f) || !(fabsf(c2) <= 3.4028234e38f)) seen_bad = true;
        const int64_t id = a.ids[row];
        const int pid = a.pos_ids[row];
        const bool type_on = !a.type_ids || a.type_ids[row] == 1;      // (wave-uniform: one row per wave at a time)
        float* wrow = (id != a.pad_id && !a.det_rows) ? a.dword + (size_t)id * a.H : nullptr;
        const bool pos_on = pid != a.pos_pad_id && !a.det_rows;      // deterministic mode: word and position rows are summed by embed_scatter_det_kernel
        if (pos_on && pid0 < 0) pid0 = pid;
        const bool pos_reg = pos_on && pid == pid0;
        float* prow = (pos_on && !pos_reg) ? a.dpos + (size_t)pid * a.H : nullptr;     // irregular slot: direct atomics
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nch) {
                f32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = rstd * (g[t][e] - c1 - xh[t][e] * c2);
                    if (type_on) dt[t][e] += o;
                    if (pos_reg) pacc[t][e] += o;
                    ov[e] = o * al;
                }
                *reinterpret_cast<f32x4*>(&rowbuf[w][c * 4]) = ov;
                if (a.det_rows) *reinterpret_cast<f32x4*>(a.det_rows + (size_t)row * a.H + c * 4) = ov;
            }
        }
        if (id != a.pad_id && a.row_state && lane == 0)        // 32-bit atomic OR: neighbouring rows' flags share the word
            atomicOr(reinterpret_cast<unsigned*>(a.row_state) + (id >> 2), 1u << (8 * (int)(id & 3)));
        if (wrow || prow) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < a.H; c += 64) {
                const float o = rowbuf[w][c];
                if (wrow) atomicAdd(wrow + c, o);
                if (prow) atomicAdd(prow + c, o);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int t = 0; t < MAXC; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * t) * 4 + e;
                red[0][w][c] = dg[t][e]; red[1][w][c] = db[t][e]; red[2][w][c] = dt[t][e];
            }
    if (lane == 0) wpid[w] = pid0;
    __syncthreads();
    for (int c = threadIdx.x; c < a.H; c += 256) {
        const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        const float st = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
        if (a.partial) {
            a.partial[(size_t)blk * a.H + c] = sg;
            a.partial[((size_t)nblk + blk) * a.H + c] = sb;
            a.partial[((size_t)2 * nblk + blk) * a.H + c] = st;
        } else {
            atomicAdd(a.dgamma + c, sg * al);
            atomicAdd(a.dbeta + c, sb * al);
            atomicAdd(a.dtype + (a.type_ids ? a.H : 0) + c, st * al);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAXC; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) rowbuf[w][(lane + 64 * t) * 4 + e] = pacc[t][e] * al;
    __syncthreads();
    for (int c = threadIdx.x; c < a.H; c += 256) {
        // position rows: waves that share an id leave as one atomic (the first such wave owns the sum)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pi = wpid[i];
            if (pi < 0) continue;
            bool owner = true;
            for (int j = 0; j < i; ++j) owner = owner && wpid[j] != pi;
            if (!owner) continue;
            float sp = rowbuf[i][c];
            for (int j = i + 1; j < 4; ++j) if (wpid[j] == pi) sp += rowbuf[j][c];
            atomicAdd(a.dpos + (size_t)pi * a.H + c, sp);
        }
    }
    if (seen_bad && a.status && lane == 0) {
        atomicAdd(a.status, 1u);
        atomicOr(a.status + 1, 1u);
    }
}

// deterministic second stage of the embedding backward (MMHIP_DETERMINISTIC=1): rows[slot] = the slot's gradient row (already scaled).
// Block b < n: slot b adds, in slot order, the rows of every slot with its word id -- if it is the FIRST slot with that id (a scan of the
// ids in front of it); block n + p: position id p adds the rows of its slots in slot order.  One writer per table row, plain += .
__global__ __launch_bounds__(64) void embed_scatter_det_kernel(const int64_t* __restrict__ ids, const int* __restrict__ pos_ids, const float* __restrict__ rows,
                                                               float* __restrict__ dword, float* __restrict__ dpos, int n, int H, int pad_id, int pos_pad_id) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bool any = false;
    float* dst;
    if (b < n) {
        const int64_t id = ids[b];
        if (id == pad_id) return;
        for (int j0 = 0; j0 < b; j0 += 64) {          // an earlier slot with this id owns the sum
            const int j = j0 + lane;
            if (__any(j < b && ids[j] == id)) return;
        }
        dst = dword + (size_t)id * H;
        for (int j = b; j < n; ++j) {
            if (ids[j] != id) continue;               // wave-uniform
            any = true;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < H) acc[i] += rows[(size_t)j * H + c]; }
        }
    } else {
        const int p = b - n;
        if (p == pos_pad_id) return;
        dst = dpos + (size_t)p * H;
        for (int j = 0; j < n; ++j) {
            if (pos_ids[j] != p) continue;
            any = true;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < H) acc[i] += rows[(size_t)j * H + c]; }
        }
    }
    if (!any) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < H) dst[c] += acc[i]; }
}

// ------------------------------------------------------------------------------------------------ ViT input
// pixels [B,3,img,img] fp32 NCHW -> patches [B*np*np, 3*ps*ps] 16-bit, column = c*ps*ps + i*ps + j
// (the flattening of the conv weight [H,3,ps,ps], HF vit :60,69).  8 consecutive j per thread.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, T* __restrict__ out, int B, int img, int ps) {
    const int np = img / ps, K = 3 * ps * ps, kc = K / 8;
    const size_t total = (size_t)B * np * np * kc;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ch = idx % kc;
        const size_t prow = idx / kc;
        const int col = ch * 8, c = col / (ps * ps), i = (col / ps) % ps, j = col % ps;
        const int b = prow / (np * np), py = (prow / np) % np, pxi = prow % np;
        const float* src = px + (((size_t)b * 3 + c) * img + (py * ps + i)) * img + pxi * ps + j;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        typename Vec<T>::v8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = from_f<T>(v0[e]); o[4 + e] = from_f<T>(v1[e]); }
        *reinterpret_cast<typename Vec<T>::v8*>(out + prow * K + col) = o;
    }
}
// any patch size (14 x 14 for CLIP-ViT-L/14: 588 columns), rows zero-padded to `ld` columns (the GEMM's k-step)
template <typename T>
__global__ __launch_bounds__(256) void patchify_any_kernel(const float* __restrict__ px, T* __restrict__ out, int B, int img, int ps, int ld) {
    const int np = img / ps, K = 3 * ps * ps;
    const size_t total = (size_t)B * np * np * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int col = idx % ld;
        const size_t prow = idx / ld;
        float v = 0.f;
        if (col < K) {
            const int c = col / (ps * ps), i = (col / ps) % ps, j = col % ps;
            const int b = prow / (np * np), py = (prow / np) % np, pxi = prow % np;
            v = px[(((size_t)b * 3 + c) * img + (py * ps + i)) * img + pxi * ps + j];
        }
        out[idx] = from_f<T>(v);
    }
}
// plane-pair forms (parity mode): out rows [hi(ld) | lo(ld)]
__global__ __launch_bounds__(256) void patchify_pair_kernel(const float* __restrict__ px, bf16_t* __restrict__ out, int B, int img, int ps, int ld) {
    const int np = img / ps, K = 3 * ps * ps;
    const size_t total = (size_t)B * np * np * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int col = idx % ld;
        const size_t prow = idx / ld;
        float v = 0.f;
        if (col < K) {
            const int c = col / (ps * ps), i = (col / ps) % ps, j = col % ps;
            const int b = prow / (np * np), py = (prow / np) % np, pxi = prow % np;
            v = px[(((size_t)b * 3 + c) * img + (py * ps + i)) * img + pxi * ps + j];
        }
        const bf16_t h = (bf16_t)v;
        out[prow * (2 * (size_t)ld) + col] = h;
        out[prow * (2 * (size_t)ld) + ld + col] = (bf16_t)(v - (float)h);
    }
}
__global__ __launch_bounds__(256) void cast_pad_pair_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols, int ld) {
    const size_t total = (size_t)rows * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = idx % ld;
        const size_t r = idx / ld;
        const float v = c < cols ? src[r * cols + c] : 0.f;
        const bf16_t h = (bf16_t)v;
        dst[r * (2 * (size_t)ld) + c] = h;
        dst[r * (2 * (size_t)ld) + ld + c] = (bf16_t)(v - (float)h);
    }
}
// dst[r][0..ld) = cast(src[r][0..cols)), zero beyond cols
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols, int ld) {
    const size_t total = (size_t)rows * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = idx % ld;
        const size_t r = idx / ld;
        dst[idx] = from_f<T>(c < cols ? src[r * cols + c] : 0.f);
    }
}
// x[b*P] = cls + pos[0]; x[b*P + 1 + p] = patches[b*(P-1) + p] + pos[1 + p]      (HF vit :146-157)
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const T* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ x, int B, int P, int H) {
    const int hc = H / 4;
    const size_t total = (size_t)B * P * hc;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (idx % hc) * 4;
        const size_t row = idx / hc;
        const int b = row / P, p = row % P;
        float v[4], q[4];
        if (p == 0) load4<float>(cls + c, v);
        else load4<T>(patches + ((size_t)b * (P - 1) + (p - 1)) * H + c, v);
        load4<float>(pos + (size_t)p * H + c, q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += q[e];
        store4<T>(x + row * H + c, v);
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] += sum_r x[r][c]   (bias gradients).  grid (ceil(cols/256), row chunks); lanes own 4 columns each.
static constexpr int COLSUM_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int rows, int cols, int ld, float* __restrict__ out, float* __restrict__ partial, float alpha) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        const int r1 = min(rows, (int)(blockIdx.y + 1) * COLSUM_ROWS);
        for (int r = blockIdx.y * COLSUM_ROWS + w; r < r1; r += 4) {
            float v[4];
            load4<T>(x + (size_t)r * ld + c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = s[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < cols) {
        const float sres = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (partial) partial[(size_t)blockIdx.y * cols + cc] = sres;
        else atomicAdd(out + cc, sres * alpha);
    }
}

// ------------------------------------------------------------------------------------------------ casts
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4<float>(src + i * 4, v);
        store4<T>(dst + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = from_f<T>(src[n4 * 4 + threadIdx.x]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * rows + r0 + r] = from_f<T>(tile[r][c]);
    }
}
// grouped weight refresh: for each matrix of the group read every fp32 64x64 tile once, write the 16-bit copy and
// (optionally) the transposed 16-bit copy -- one launch per layer instead of eight
// parity mode (round 4): the GEMM operand copies of the weights are PLANE PAIRS (mmhip_kernels.h), rows [hi(cols) | lo(cols)]: dst [rows, 2 cols],
// dstT [cols, 2 rows] -- written once per optimizer step instead of split before every matrix product
__global__ __launch_bounds__(256) void cast_dual_pair_kernel(CastGroup g) {
    __shared__ float tile[64][65];
    int id = blockIdx.x, pi = 0;
#pragma unroll
    for (int i = 1; i < CAST_MAX_GROUP; ++i)
        if (i < g.count && id >= g.m[i].tile_start) pi = i;
    const CastMat& M = g.m[pi];
    id -= M.tile_start;
    const int tc = (M.cols + 63) / 64;
    const int r0 = (id / tc) * 64, c0 = (id % tc) * 64;
    bf16_t* dst = (bf16_t*)M.dst;
    bf16_t* dstT = (bf16_t*)M.dstT;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < M.rows && c0 + c < M.cols) {
            load4<float>(M.src + (size_t)(r0 + r) * M.cols + c0 + c, v);
            store4_pair(dst + (size_t)(r0 + r) * (2 * M.cols) + c0 + c, M.cols, v);
        }
        tile[r][c] = v[0]; tile[r][c + 1] = v[1]; tile[r][c + 2] = v[2]; tile[r][c + 3] = v[3];
    }
    if (!dstT) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int c = i >> 4, r = (i & 15) * 4;
        if (c0 + c < M.cols && r0 + r < M.rows) {
            float v[4] = {tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]};
            store4_pair(dstT + (size_t)(c0 + c) * (2 * M.rows) + r0 + r, M.rows, v);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void cast_dual_kernel(CastGroup g) {
    __shared__ float tile[64][65];
    int id = blockIdx.x, pi = 0;
#pragma unroll
    for (int i = 1; i < CAST_MAX_GROUP; ++i)
        if (i < g.count && id >= g.m[i].tile_start) pi = i;
    const CastMat& M = g.m[pi];
    id -= M.tile_start;
    const int tc = (M.cols + 63) / 64;
    const int r0 = (id / tc) * 64, c0 = (id % tc) * 64;
    T* dst = (T*)M.dst;
    T* dstT = (T*)M.dstT;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {          // 16 float4 per row
        const int r = i >> 4, c = (i & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < M.rows && c0 + c < M.cols) {
            load4<float>(M.src + (size_t)(r0 + r) * M.cols + c0 + c, v);
            store4<T>(dst + (size_t)(r0 + r) * M.cols + c0 + c, v);
        }
        tile[r][c] = v[0]; tile[r][c + 1] = v[1]; tile[r][c + 2] = v[2]; tile[r][c + 3] = v[3];
    }
    if (!dstT) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int c = i >> 4, r = (i & 15) * 4;
        if (c0 + c < M.cols && r0 + r < M.rows) {
            float v[4] = {tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]};
            store4<T>(dstT + (size_t)(c0 + c) * M.rows + r0 + r, v);
        }
    }
}
// dst[r * dst_stride + c] (+)= src[r * H + c]: spread compact CLS rows into a full-row tensor
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows16_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, size_t dst_stride, int H, int add) {
    const int hc = H / 4;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * hc; idx += gridDim.x * 256) {
        const int rr = idx / hc, c = (idx % hc) * 4;
        float v[4];
        load4<T>(src + (size_t)rr * H + c, v);
        if (add) {
            float o[4];
            load4<T>(dst + (size_t)rr * dst_stride + c, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += o[e];
        }
        store4<T>(dst + (size_t)rr * dst_stride + c, v);
    }
}
// dx[post*T + t][:] = (t == 0) ? d[post][:] : 0        (gradient of the last hidden state: only CLS rows are read)
template <typename T>
__global__ __launch_bounds__(256) void scatter_cls_kernel(const float* __restrict__ d, T* __restrict__ dx, int posts, int Tn, int H, float scale) {
    const int hc = H / 4;
    const size_t total = (size_t)posts * Tn * hc;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (idx % hc) * 4;
        const size_t row = idx / hc;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row % Tn == 0) {
            load4<float>(d + (row / Tn) * H + c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= scale;
        }
        store4<T>(dx + row * H + c, v);
    }
}

// dst = dropout(src) on the linear element index (backward of a GEMM-epilogue dropout: same mask, same scale)
template <typename T>
__global__ __launch_bounds__(256) void dropout16_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t n, DropCfg d) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4<T>(src + i * 4, v);
        bool k0, k1, k2, k3;
        mm_keep2((uint32_t)(i * 4), d, k0, k1);
        mm_keep2((uint32_t)(i * 4 + 2), d, k2, k3);
        v[0] = k0 ? v[0] * d.keep_scale : 0.f;
        v[1] = k1 ? v[1] * d.keep_scale : 0.f;
        v[2] = k2 ? v[2] * d.keep_scale : 0.f;
        v[3] = k3 ? v[3] * d.keep_scale : 0.f;
        store4<T>(dst + i * 4, v);
    }
}
// out[r][0..H) (fp32, row stride ldo) = src[r * src_stride .. +H) (16-bit): CLS-row gather
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const T* __restrict__ src, size_t src_stride, float* __restrict__ out, int ldo, int rows, int H) {
    const int hc = H / 4;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * hc; idx += gridDim.x * 256) {
        const int r = idx / hc, c = (idx % hc) * 4;
        float v[4];
        load4<T>(src + (size_t)r * src_stride + c, v);
        *reinterpret_cast<f32x4*>(out + (size_t)r * ldo + c) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// ------------------------------------------------------------------------------------------------ launchers
#define DISPATCH_T(dtype, KERNEL, grid, block, stream, ...)                                        \
    do {                                                                                           \
        if ((dtype) == DT_BF16) hipLaunchKernelGGL((KERNEL<bf16_t>), grid, block, 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<f16_t>), grid, block, 0, stream, __VA_ARGS__);             \
    } while (0)

static inline int cap_grid(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

hipError_t launch_layernorm_fwd(const LNArgs& a, int dtype, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.width % 4 || a.width > 1024 || a.ldx % 4 || a.ldy % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, dim3((a.rows + 4 * LN_FWD_RPW - 1) / (4 * LN_FWD_RPW)), dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(ln_fwd_kernel<f16_t>, dim3((a.rows + 4 * LN_FWD_RPW - 1) / (4 * LN_FWD_RPW)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3((a.rows + 4 * LN_FWD_RPW - 1) / (4 * LN_FWD_RPW)), dim3(256), 0, s, a);
    return hipGetLastError();
}
size_t partial_floats_rows(int rows, int width, int nvec) {   // LN backward may use 3 vectors
    if (nvec < 3) nvec = 3;
    return (size_t)nvec * ((rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK) * width; }
size_t partial_floats_embed(int posts, int T, int width) { return (size_t)3 * T * ((posts + EMB_POSTS_PER_BLOCK - 1) / EMB_POSTS_PER_BLOCK) * width; }
size_t partial_floats_colsum(int rows, int cols) { return (size_t)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS) * cols; }

hipError_t launch_layernorm_bwd(const LNBwdArgs& a, int dtype, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.width % 4 || a.width > 1024) return hipErrorInvalidValue;
    const int grid = (a.rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    if (dtype == DT_BF16) hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(ln_bwd_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, a);
    if (a.partial && !a.defer_reduce) return launch_layernorm_bwd_reduce(a, s);
    return hipGetLastError();
}
// second stage of the column reductions (dgamma, dbeta[, colsum]); with LNBwdArgs::defer_reduce the caller issues it itself,
// on whichever stream it likes, once the first stage is ordered before it
hipError_t launch_layernorm_bwd_reduce(const LNBwdArgs& a, hipStream_t s) {
    if (a.rows <= 0 || !a.partial) return hipSuccess;
    const int grid = (a.rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    launch_reduce_partials(a.partial, grid, a.width, a.dgamma, s, a.alpha == 0.f ? 1.f : a.alpha, a.dbeta, a.colsum_out);
    return hipGetLastError();
}
hipError_t launch_embed_fwd(const EmbedArgs& a, int dtype, hipStream_t s) {
    if (a.posts <= 0) return hipSuccess;
    if (a.H % 4 || a.H > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pos_ids_kernel, dim3(a.posts), dim3(64), 0, s, a.ids, a.mask, a.pos_ids, a.maskbias, a.T, a.xlmr, a.pad_id);
    const int grid = (a.posts * a.T + 3) / 4;
    if (dtype == DT_BF16) hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(embed_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_embed_bwd(const EmbedBwdArgs& a, int dtype, hipStream_t s) {
    if (a.posts <= 0) return hipSuccess;
    if (a.H % 4 || a.H > 1024) return hipErrorInvalidValue;
    const dim3 grid(a.T, (a.posts + EMB_POSTS_PER_BLOCK - 1) / EMB_POSTS_PER_BLOCK);
    if (dtype == DT_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(embed_bwd_kernel<f16_t>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, s, a);
    if (a.partial) {
        launch_reduce_partials(a.partial, (int)(grid.x * grid.y), a.H, a.dgamma, s, a.alpha == 0.f ? 1.f : a.alpha, a.dbeta, a.dtype + (a.type_ids ? a.H : 0));
    }
    if (a.det_rows)
        hipLaunchKernelGGL(embed_scatter_det_kernel, dim3(a.posts * a.T + a.max_pos), dim3(64), 0, s, a.ids, a.pos_ids, a.det_rows, a.dword, a.dpos,
                           a.posts * a.T, a.H, a.pad_id, a.pos_pad_id);
    return hipGetLastError();
}
hipError_t launch_cast_pad(const float* src, void* dst, int rows, int cols, int ld, int dtype, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    if (ld < cols) return hipErrorInvalidValue;
    const int grid = cap_grid((size_t)rows * ld);
    if (dtype == DT_PAIR) { hipLaunchKernelGGL(cast_pad_pair_kernel, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, rows, cols, ld); return hipGetLastError(); }
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_pad_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, rows, cols, ld);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cast_pad_kernel<f16_t>, dim3(grid), dim3(256), 0, s, src, (f16_t*)dst, rows, cols, ld);
    else hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, rows, cols, ld);
    return hipGetLastError();
}
hipError_t launch_patchify(const float* pixels, void* out, int B, int img, int patch, int ld, int dtype, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (img % patch || ld < 3 * patch * patch) return hipErrorInvalidValue;
    if (dtype == DT_PAIR) {
        hipLaunchKernelGGL(patchify_pair_kernel, dim3(cap_grid((size_t)B * (img / patch) * (img / patch) * ld)), dim3(256), 0, s, pixels, (bf16_t*)out, B, img, patch, ld);
        return hipGetLastError();
    }
    if (patch % 8 || ld != 3 * patch * patch) {
        const int grid = cap_grid((size_t)B * (img / patch) * (img / patch) * ld);
        if (dtype == DT_BF16) hipLaunchKernelGGL(patchify_any_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, pixels, (bf16_t*)out, B, img, patch, ld);
        else if (dtype == DT_F16) hipLaunchKernelGGL(patchify_any_kernel<f16_t>, dim3(grid), dim3(256), 0, s, pixels, (f16_t*)out, B, img, patch, ld);
        else hipLaunchKernelGGL(patchify_any_kernel<float>, dim3(grid), dim3(256), 0, s, pixels, (float*)out, B, img, patch, ld);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * (img / patch) * (img / patch) * (3 * patch * patch / 8);
    if (dtype == DT_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(cap_grid(total)), dim3(256), 0, s, pixels, (bf16_t*)out, B, img, patch);
    else if (dtype == DT_F16) hipLaunchKernelGGL(patchify_kernel<f16_t>, dim3(cap_grid(total)), dim3(256), 0, s, pixels, (f16_t*)out, B, img, patch);
    else hipLaunchKernelGGL(patchify_kernel<float>, dim3(cap_grid(total)), dim3(256), 0, s, pixels, (float*)out, B, img, patch);
    return hipGetLastError();
}
hipError_t launch_vit_assemble(const void* patches, const float* cls, const float* pos, void* x, int B, int P, int H, int dtype, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t total = (size_t)B * P * (H / 4);
    if (dtype == DT_BF16) hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3(cap_grid(total)), dim3(256), 0, s, (const bf16_t*)patches, cls, pos, (bf16_t*)x, B, P, H);
    else if (dtype == DT_F16) hipLaunchKernelGGL(vit_assemble_kernel<f16_t>, dim3(cap_grid(total)), dim3(256), 0, s, (const f16_t*)patches, cls, pos, (f16_t*)x, B, P, H);
    else hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3(cap_grid(total)), dim3(256), 0, s, (const float*)patches, cls, pos, (float*)x, B, P, H);
    return hipGetLastError();
}
hipError_t launch_colsum(const void* x, int rows, int cols, int ld, float* out, int dtype, hipStream_t s, float* partial, float alpha) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    if (cols % 4 || ld % 4) return hipErrorInvalidValue;
    dim3 grid((cols + 255) / 256, (rows + COLSUM_ROWS - 1) / COLSUM_ROWS);
    if (dtype == DT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, rows, cols, ld, out, partial, alpha);
    else if (dtype == DT_F16) hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, rows, cols, ld, out, partial, alpha);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, rows, cols, ld, out, partial, alpha);
    if (partial) launch_reduce_partials(partial, (int)grid.y, cols, out, s, alpha);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void copy_ids_clamped_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out, size_t n, int64_t hi, unsigned* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int64_t v = in[i];
        const int64_t c = v < 0 ? 0 : (v >= hi ? hi - 1 : v);
        if (c != v && bad) atomicAdd(bad, 1u);
        out[i] = c;
    }
}
hipError_t launch_copy_ids_clamped(const int64_t* in, int64_t* out, size_t n, int64_t hi, unsigned* bad, hipStream_t s) {
    if (!n) return hipSuccess;
    if (hi < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_ids_clamped_kernel, dim3(cap_grid(n)), dim3(256), 0, s, in, out, n, hi, bad);
    return hipGetLastError();
}
hipError_t launch_cast(const float* src, void* dst, size_t n, int dtype, hipStream_t s) {
    if (!n) return hipSuccess;
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(cap_grid(n / 4 + 1)), dim3(256), 0, s, src, (bf16_t*)dst, n);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cast_kernel<f16_t>, dim3(cap_grid(n / 4 + 1)), dim3(256), 0, s, src, (f16_t*)dst, n);
    else hipLaunchKernelGGL(cast_kernel<float>, dim3(cap_grid(n / 4 + 1)), dim3(256), 0, s, src, (float*)dst, n);
    return hipGetLastError();
}
hipError_t launch_cast_transpose(const float* src, void* dst, int rows, int cols, int dtype, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, rows, cols);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cast_transpose_kernel<f16_t>, grid, dim3(256), 0, s, src, (f16_t*)dst, rows, cols);
    else hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, rows, cols);
    return hipGetLastError();
}
hipError_t launch_cast_group(const CastMat* mats, int count, int dtype, hipStream_t s) {
    CastGroup g;
    g.count = 0;
    int tiles = 0;
    for (int i = 0; i < count && i < CAST_MAX_GROUP; ++i) {
        CastMat m = mats[i];
        if (m.rows % 4 || m.cols % 4) return hipErrorInvalidValue;
        m.tile_start = tiles;
        tiles += ((m.rows + 63) / 64) * ((m.cols + 63) / 64);
        g.m[g.count++] = m;
    }
    if (!tiles) return hipSuccess;
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_dual_kernel<bf16_t>, dim3(tiles), dim3(256), 0, s, g);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cast_dual_kernel<f16_t>, dim3(tiles), dim3(256), 0, s, g);
    else if (dtype == DT_PAIR) hipLaunchKernelGGL(cast_dual_pair_kernel, dim3(tiles), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(cast_dual_kernel<float>, dim3(tiles), dim3(256), 0, s, g);
    return hipGetLastError();
}
hipError_t launch_scatter_rows16(const void* src, void* dst, int rows, size_t dst_stride, int H, int add, int dtype, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (H % 4 || dst_stride % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(scatter_rows16_kernel<bf16_t>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, rows, dst_stride, H, add);
    else if (dtype == DT_F16) hipLaunchKernelGGL(scatter_rows16_kernel<f16_t>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const f16_t*)src, (f16_t*)dst, rows, dst_stride, H, add);
    else hipLaunchKernelGGL(scatter_rows16_kernel<float>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const float*)src, (float*)dst, rows, dst_stride, H, add);
    return hipGetLastError();
}
hipError_t launch_dropout16(const void* src, void* dst, size_t n, const DropCfg& d, int dtype, hipStream_t s) {
    if (!n) return hipSuccess;
    if (n % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(dropout16_kernel<bf16_t>, dim3(cap_grid(n / 4)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n, d);
    else if (dtype == DT_F16) hipLaunchKernelGGL(dropout16_kernel<f16_t>, dim3(cap_grid(n / 4)), dim3(256), 0, s, (const f16_t*)src, (f16_t*)dst, n, d);
    else hipLaunchKernelGGL(dropout16_kernel<float>, dim3(cap_grid(n / 4)), dim3(256), 0, s, (const float*)src, (float*)dst, n, d);
    return hipGetLastError();
}
hipError_t launch_gather_rows_f32(const void* src, size_t src_stride, float* out, int ldo, int rows, int H, int dtype, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (H % 4 || ldo % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(gather_rows_f32_kernel<bf16_t>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const bf16_t*)src, src_stride, out, ldo, rows, H);
    else if (dtype == DT_F16) hipLaunchKernelGGL(gather_rows_f32_kernel<f16_t>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const f16_t*)src, src_stride, out, ldo, rows, H);
    else hipLaunchKernelGGL(gather_rows_f32_kernel<float>, dim3(cap_grid((size_t)rows * H / 4)), dim3(256), 0, s, (const float*)src, src_stride, out, ldo, rows, H);
    return hipGetLastError();
}
hipError_t launch_scatter_cls_rows(const float* d, void* dx, int posts, int T, int H, int dtype, hipStream_t s, float scale) {
    if (posts <= 0) return hipSuccess;
    const size_t total = (size_t)posts * T * (H / 4);
    if (dtype == DT_BF16) hipLaunchKernelGGL(scatter_cls_kernel<bf16_t>, dim3(cap_grid(total)), dim3(256), 0, s, d, (bf16_t*)dx, posts, T, H, scale);
    else if (dtype == DT_F16) hipLaunchKernelGGL(scatter_cls_kernel<f16_t>, dim3(cap_grid(total)), dim3(256), 0, s, d, (f16_t*)dx, posts, T, H, scale);
    else hipLaunchKernelGGL(scatter_cls_kernel<float>, dim3(cap_grid(total)), dim3(256), 0, s, d, (float*)dx, posts, T, H, scale);
    return hipGetLastError();
}

}  // namespace mmhip
