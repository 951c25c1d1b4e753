// mmhip early-fusion engine (BASELINE config 5): the LXMERT training step of the reference's models/mm_early.py:105-172 (class Lxmert:
// HF LxmertModel + linear_fusion / linear / linear_tim heads, max-pooled ITC embeddings) and :295-407 (the train loop's step body) as ONE
// native call path -- the counterpart of engine.hip for the late-fusion model.  Round 2/3 chained the block operators of capi_ops.hip by
// torch autograd (~40 nodes per step from Python, embeddings / dropout / max-pool / ITC / losses as ATen kernels); here the layer loop, the
// embeddings, the visual-feature encoder, the heads, the pooling, the ITC similarity, the fused loss, the backward, the weight gradients,
// AdamW and the operand refresh are enqueued from C++ on two internal HIP streams:
//   * language stream = the caller's stream (T tokens per post), vision stream = an internal one (36 boxes per post); they meet at the
//     cross-modality layers (events), exactly the dependency structure of HF LxmertEncoder.forward;
//   * every block is one of the composite operators of capi_ops.hip (self-attention block, cross-attention block, feed-forward block: the
//     same fused-epilogue GEMMs, attention and LayerNorm kernels the late-fusion engine runs);
//   * weight gradients leave per layer in grouped launches (gemm_tn_kernel, <= 8 problems per launch); the ONE cross-attention module of a
//     cross-modality layer is used in both directions (HF LxmertXLayer.cross_att), so its gradients are accumulated (atomics) from both;
//   * parameters live in ONE flat fp32 buffer ordered so that the gradient ranges of the backward stages are contiguous and finish in
//     address order: [pooler (never) | logit_scale (ITC) | linear_tim (ITM) | heads | x layers last -> first | language / relational
//     layers last -> first | visual-feature encoder | embeddings] -- the data-parallel caller exchanges a stage's range while the
//     stages below it still compute (mmhip_early_train_step's callback, the ABI of mmhip_train_step_dp).
#include <string>
#include <vector>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include "mmhip_common.h"
#include "mmhip_kernels.h"
#include "../../include/mmhip.h"

using namespace mmhip;

#define CHECK_HIP(expr)                       \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
#define CHECK_RC(expr)          \
    do {                        \
        int _r = (expr);        \
        if (_r) return _r;      \
    } while (0)

namespace {

struct AttOff { size_t qkv_w, qkv_b, o_w, o_b, ln_w, ln_b; };      // element offsets into the flat fp32 buffers
struct FfnOff { size_t w1, b1, w2, b2, ln_w, ln_b; };
struct Copy { size_t w, wT; };                                      // byte offsets of the operand copies in the workspace
struct AttW { Copy qkv, o; };
struct FfnW { Copy w1, w2; };
struct SelfAct { size_t qkv, att, lse, pre, mean, rstd, y, dpre, dd, datt, dqkv, dx; };
struct CrossAct { size_t qkv, att, lse, pre, mean, rstd, y, dpre, dd, datt, dqkv, dxq, dxc; };
struct FfnAct { size_t h, u, pre, mean, rstd, y, dpre, dd, du, dx; };
struct PlainLayer { AttOff att; FfnOff ffn; AttW aw; FfnW fw; SelfAct sa; FfnAct fa; size_t begin, end; };
struct XLayer {
    AttOff cross, lself, vself; FfnOff lffn, vffn;
    AttW cw, lw, vw; FfnW lfw, vfw;
    CrossAct cl, cv; SelfAct sl, sv; FfnAct fl, fv;
    size_t begin, end;
};

// ------------------------------------------------------------------------------------------------ small kernels of this path
// y = dropout((a + b) * alpha): the visual-feature encoder's average of its two LayerNorm outputs (b may be null: y = dropout(a * alpha), its backward)
template <typename T>
__global__ __launch_bounds__(256) void avg_drop_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n4, float alpha, DropCfg d) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        typename Vec<T>::v4 va = *reinterpret_cast<const typename Vec<T>::v4*>(a + i * 4);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = to_f<T>(va[e]);
        if (b) {
            typename Vec<T>::v4 vb = *reinterpret_cast<const typename Vec<T>::v4*>(b + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += to_f<T>(vb[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= alpha;
        if (d.thresh16) {
            bool k0, k1, k2, k3;
            mm_keep2((uint32_t)(i * 4), d, k0, k1);
            mm_keep2((uint32_t)(i * 4 + 2), d, k2, k3);
            v[0] = k0 ? v[0] * d.keep_scale : 0.f;
            v[1] = k1 ? v[1] * d.keep_scale : 0.f;
            v[2] = k2 ? v[2] * d.keep_scale : 0.f;
            v[3] = k3 ? v[3] * d.keep_scale : 0.f;
        }
        typename Vec<T>::v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(v[e]);
        *reinterpret_cast<typename Vec<T>::v4*>(y + i * 4) = o;
    }
}
hipError_t launch_avg_drop(const void* a, const void* b, void* y, size_t n, float alpha, const DropCfg& d, int dtype, hipStream_t s) {
    if (!n) return hipSuccess;
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL(avg_drop_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n4, alpha, d);
    else if (dtype == DT_F16) hipLaunchKernelGGL(avg_drop_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)a, (const f16_t*)b, (f16_t*)y, n4, alpha, d);
    else hipLaunchKernelGGL(avg_drop_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n4, alpha, d);
    return hipGetLastError();
}

// out[b][c] = max over the S rows of post b (rows whose mask is 0 count as -1e9: reference models/mm_early.py:139-143); arg[b][c] = first row
// that attains it (torch.max's tie rule on CUDA is unspecified; ties do not occur in floating-point activations).  Thread = 4 columns.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, const int64_t* __restrict__ mask, float* __restrict__ out, int* __restrict__ arg,
                                                      int posts, int S, int H) {
    const int hc = H / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= posts * hc) return;
    const int b = idx / hc, c = (idx % hc) * 4;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    for (int r = 0; r < S; ++r) {
        typename Vec<T>::v4 v = *reinterpret_cast<const typename Vec<T>::v4*>(x + ((size_t)b * S + r) * H + c);
        const bool live = !mask || mask[(size_t)b * S + r] != 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = live ? to_f<T>(v[e]) : -1e9f;
            if (f > best[e]) { best[e] = f; bi[e] = r; }
        }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)b * H + c) = f32x4{best[0], best[1], best[2], best[3]};
    if (arg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) arg[(size_t)b * H + c + e] = bi[e];
    }
}
// dx[b][r][c] = (r == arg[b][c] && b < live_posts) ? d[b][c] : 0 -- the whole [posts, S, H] tensor is written
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ d, const int* __restrict__ arg, T* __restrict__ dx, int posts, int live_posts, int S, int H) {
    const int hc = H / 4;
    const size_t total = (size_t)posts * S * hc;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % hc) * 4;
        const size_t row = idx / hc;
        const int b = (int)(row / S), r = (int)(row % S);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (d && b < live_posts) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (arg[(size_t)b * H + c + e] == r) v[e] = d[(size_t)b * H + c + e];
        }
        typename Vec<T>::v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(v[e]);
        *reinterpret_cast<typename Vec<T>::v4*>(dx + row * H + c) = o;
    }
}
// ITM posts: rows src[b] of the batch's ids / mask / token types into the second half of the engine's [2B, T] tensors
__global__ __launch_bounds__(256) void gather_rows_i64_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ src, int64_t* __restrict__ out, int B, int T) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx % T;
    const int64_t r = src[b];
    out[idx] = in[(size_t)(r < 0 ? 0 : (r >= B ? B - 1 : r)) * T + t];      // (a source row outside the batch would read outside the tensor)
}

// key biases at the packed cross-attention length S = max(T, boxes): live keys copy their bias, keys past the stream's own length get -inf
__global__ __launch_bounds__(256) void pad_bias_kernel(const float* __restrict__ in, float* __restrict__ out, int posts, int L, int S) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= posts * S) return;
    const int b = idx / S, k = idx % S;
    out[idx] = k < L ? (in ? in[(size_t)b * L + k] : 0.f) : -INFINITY;
}

// dW[n][c] += sum_m dY[m][n] X[m][c] for a handful of columns c (box_fc: 4 box coordinates) -- the generic fallback of gemm_tn walked all M rows in
// 4 threads per output row (410 us per step); here a block owns 64 n x 256 rows, thread = (n, c), coalesced over n, one atomic per output and block.
// colsum[n] += sum_m dY[m][n] from the threads of c == 0.  The targets are zero on entry (the step's AdamW clears the gradient buffer).
template <typename T>
__global__ __launch_bounds__(256) void tn_few_cols_kernel(const T* __restrict__ dy, int ldy, const T* __restrict__ x, int ldx, float* __restrict__ C, float* __restrict__ colsum,
                                                          int M, int Nn, int Nc) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), c = threadIdx.x >> 6;
    if (n >= Nn || c >= Nc) return;
    const int m0 = blockIdx.y * 256, m1 = min(M, m0 + 256);
    float acc = 0.f, cs = 0.f;
    for (int m = m0; m < m1; ++m) {
        const float d = to_f<T>(dy[(size_t)m * ldy + n]);
        acc += d * to_f<T>(x[(size_t)m * ldx + c]);
        cs += d;
    }
    atomicAdd(C + (size_t)n * Nc + c, acc);
    if (colsum && c == 0) atomicAdd(colsum + n, cs);
}
hipError_t launch_tn_few_cols(const void* dy, int ldy, const void* x, int ldx, float* C, float* colsum, int M, int Nn, int Nc, int dtype, hipStream_t s) {
    if (Nc > 4 || M <= 0) return hipErrorInvalidValue;
    const dim3 grid((Nn + 63) / 64, (M + 255) / 256);
    if (dtype == DT_BF16) hipLaunchKernelGGL(tn_few_cols_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, C, colsum, M, Nn, Nc);
    else if (dtype == DT_F16) hipLaunchKernelGGL(tn_few_cols_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)dy, ldy, (const f16_t*)x, ldx, C, colsum, M, Nn, Nc);
    else hipLaunchKernelGGL(tn_few_cols_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, ldy, (const float*)x, ldx, C, colsum, M, Nn, Nc);
    return hipGetLastError();
}

inline int cap(size_t work) { size_t g = (work + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

DropCfg drop_cfg(float p, uint64_t seed, uint32_t stream, bool on) {
    DropCfg d;
    d.seed = seed;
    d.stream = stream;
    uint32_t t = (on && p > 0.f) ? (uint32_t)lrintf(p * 65536.0f) : 0u;
    if (t > 65535u) t = 65535u;
    d.thresh16 = t;
    d.keep_scale = 1.0f / (1.0f - (float)t / 65536.0f);
    return d;
}

}  // namespace

struct mmhip_early {
    // mmhip_early_gemm_timing: events around every NT GEMM launch of the handle's steps while enabled
    std::vector<GemmTimingSink::Ev> timing_evs; GemmTimingSink timing{nullptr, 0, 0}; int timing_on = 0;
    unsigned* bad_index = nullptr;      // caller-owned device word: indices (token ids, token types, ITM source rows) that had to be clamped (mmhip_early_set_index_counter)
    mmhip_early_config cfg;
    std::vector<mmhip_param_info> params;
    size_t n_params = 0;
    // parameter offsets (elements)
    size_t pool_w, pool_b, logit_scale, tim_w, tim_b, fus_w, fus_b, lin_w, lin_b;
    size_t heads_begin, heads_end, vin_begin, vin_end, emb_begin, emb_end;
    size_t visn_fc_w, visn_fc_b, visn_ln_w, visn_ln_b, box_fc_w, box_fc_b, box_ln_w, box_ln_b;
    size_t word, pos, type, eln_w, eln_b;
    std::vector<PlainLayer> lang, rel;
    std::vector<XLayer> xl;
    // bound buffers
    float* P = nullptr; float* G = nullptr; char* ws = nullptr; size_t ws_need = 0;
    // workspace offsets (bytes)
    Copy c_visn_fc, c_box_fc;
    size_t ids_all, mask_all, tt_all, pos_ids, lbias, lbias_x, vbias, vbias_x;
    size_t x0, xhat, rstd_emb, dx0;
    size_t feats16, boxes16, vf_pre, vf_mean, vf_rstd, vf, bx_pre, bx_mean, bx_rstd, bx, v0, dv0, dvf, dbx, dvf_pre, dbx_pre;
    size_t h_z, h_fus, h_fusd, h_out, h_tim, h_embt, h_embv, h_argv, h_txt_n, h_img_n, h_txt_inv, h_img_inv, h_logits;
    size_t h_dout, h_dlogits, h_dtim, h_dembv, h_dembt, h_dfusd, h_dfus, h_dz, h_loss, g_dlang, g_dvisn, g_partial;
    // state of the last forward
    int B = 0, Bt = 0, T = 0, Nb = 0; bool itm = false, train = false, fwd_done = false, itc_done = false;
    uint64_t seed = 0;
    const float *bd_out = nullptr, *bd_embv = nullptr, *bd_tim = nullptr;
    hipStream_t side = nullptr;          // vision stream
    hipStream_t wside = nullptr;         // weight-gradient stream: a layer's grouped dW launches run beside the activation-gradient chains of the layers below
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_l = nullptr, ev_v = nullptr, ev_l2 = nullptr, ev_v2 = nullptr, ev_lw = nullptr, ev_vw = nullptr, ev_w = nullptr;
    int overlap = -1;

    template <typename U> U* wsp(size_t off) const { return reinterpret_cast<U*>(ws + off); }
    int dt() const { return cfg.dtype; }
    size_t esz() const { return cfg.dtype == MMHIP_BF16X3 ? 4 : 2; }
    int H() const { return cfg.hidden; }
    int S() const { return T > Nb ? T : Nb; }
};

namespace {

// ------------------------------------------------------------------------------------------------ layout
struct Builder {
    mmhip_early& e;
    size_t off = 0;
    size_t add(const std::string& name, int group, std::initializer_list<int64_t> dims) {
        mmhip_param_info p;
        memset(&p, 0, sizeof(p));
        strncpy(p.name, name.c_str(), sizeof(p.name) - 1);
        p.ndim = (int)dims.size();
        size_t n = 1;
        int i = 0;
        for (auto d : dims) { p.dims[i++] = d; n *= (size_t)d; }
        p.buffer = 1;
        p.group = group;
        p.offset = off;
        p.numel = n;
        e.params.push_back(p);
        off += (n + 3) & ~(size_t)3;
        return (size_t)p.offset;
    }
};
// HF LxmertAttention + LxmertAttentionOutput (transformers 4.25.1 names): <n>.<inner>.{query,key,value}, <n>.output.dense, <n>.output.LayerNorm.
// Q / K / V weights adjacent, then their biases: [Wq; Wk; Wv] is one [3H, H] matrix of the flat buffer.
void add_att(Builder& b, const std::string& n, const char* inner, int H, AttOff& o) {
    const int g = MMHIP_G_ALWAYS;
    const std::string q = n + "." + inner + ".";
    o.qkv_w = b.add(q + "query.weight", g, {H, H});
    b.add(q + "key.weight", g, {H, H});
    b.add(q + "value.weight", g, {H, H});
    o.qkv_b = b.add(q + "query.bias", g, {H});
    b.add(q + "key.bias", g, {H});
    b.add(q + "value.bias", g, {H});
    o.o_w = b.add(n + ".output.dense.weight", g, {H, H});
    o.o_b = b.add(n + ".output.dense.bias", g, {H});
    o.ln_w = b.add(n + ".output.LayerNorm.weight", g, {H});
    o.ln_b = b.add(n + ".output.LayerNorm.bias", g, {H});
}
void add_ffn(Builder& b, const std::string& inter, const std::string& out, int H, int I, FfnOff& o) {
    const int g = MMHIP_G_ALWAYS;
    o.w1 = b.add(inter + ".dense.weight", g, {I, H});
    o.b1 = b.add(inter + ".dense.bias", g, {I});
    o.w2 = b.add(out + ".dense.weight", g, {H, I});
    o.b2 = b.add(out + ".dense.bias", g, {H});
    o.ln_w = b.add(out + ".LayerNorm.weight", g, {H});
    o.ln_b = b.add(out + ".LayerNorm.bias", g, {H});
}

void build_layout(mmhip_early& e) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, I = c.inter, C = c.num_labels;
    Builder b{e};
    // never: the pooler is off the path (mm_early.py:132 takes the CLS row itself); ITC: logit_scale; ITM: linear_tim
    e.pool_w = b.add("model.pooler.dense.weight", MMHIP_G_NEVER, {H, H});
    e.pool_b = b.add("model.pooler.dense.bias", MMHIP_G_NEVER, {H});
    e.heads_begin = b.off;
    e.logit_scale = b.add("logit_scale", MMHIP_G_ITC, {});
    e.tim_w = b.add("linear_tim.weight", MMHIP_G_ITM, {2, H});
    e.tim_b = b.add("linear_tim.bias", MMHIP_G_ITM, {2});
    e.fus_w = b.add("linear_fusion.weight", MMHIP_G_ALWAYS, {H, H});
    e.fus_b = b.add("linear_fusion.bias", MMHIP_G_ALWAYS, {H});
    e.lin_w = b.add("linear.weight", MMHIP_G_ALWAYS, {C, H});
    e.lin_b = b.add("linear.bias", MMHIP_G_ALWAYS, {C});
    e.heads_end = b.off;
    e.xl.resize(c.x_layers);
    for (int i = c.x_layers - 1; i >= 0; --i) {
        XLayer& x = e.xl[i];
        const std::string p = "model.encoder.x_layers." + std::to_string(i) + ".";
        x.begin = b.off;
        add_ffn(b, p + "lang_inter", p + "lang_output", H, I, x.lffn);
        add_ffn(b, p + "visn_inter", p + "visn_output", H, I, x.vffn);
        add_att(b, p + "lang_self_att", "self", H, x.lself);
        add_att(b, p + "visn_self_att", "self", H, x.vself);
        add_att(b, p + "visual_attention", "att", H, x.cross);
        x.end = b.off;
    }
    e.lang.resize(c.l_layers);
    e.rel.resize(c.r_layers);
    const int depth = c.l_layers > c.r_layers ? c.l_layers : c.r_layers;
    for (int d = 0; d < depth; ++d) {          // depth d below the cross-modality layers: language layer L-1-d and relational layer R-1-d
        const int li = c.l_layers - 1 - d, ri = c.r_layers - 1 - d;
        if (li >= 0) {
            PlainLayer& l = e.lang[li];
            const std::string p = "model.encoder.layer." + std::to_string(li) + ".";
            l.begin = b.off;
            add_ffn(b, p + "intermediate", p + "output", H, I, l.ffn);
            add_att(b, p + "attention", "self", H, l.att);
            l.end = b.off;
        }
        if (ri >= 0) {
            PlainLayer& l = e.rel[ri];
            const std::string p = "model.encoder.r_layers." + std::to_string(ri) + ".";
            l.begin = b.off;
            add_ffn(b, p + "intermediate", p + "output", H, I, l.ffn);
            add_att(b, p + "attention", "self", H, l.att);
            l.end = b.off;
        }
    }
    const std::string v = "model.encoder.visn_fc.";
    e.vin_begin = b.off;
    e.visn_fc_w = b.add(v + "visn_fc.weight", MMHIP_G_ALWAYS, {H, c.feat_dim});
    e.visn_fc_b = b.add(v + "visn_fc.bias", MMHIP_G_ALWAYS, {H});
    e.visn_ln_w = b.add(v + "visn_layer_norm.weight", MMHIP_G_ALWAYS, {H});
    e.visn_ln_b = b.add(v + "visn_layer_norm.bias", MMHIP_G_ALWAYS, {H});
    e.box_fc_w = b.add(v + "box_fc.weight", MMHIP_G_ALWAYS, {H, c.pos_dim});
    e.box_fc_b = b.add(v + "box_fc.bias", MMHIP_G_ALWAYS, {H});
    e.box_ln_w = b.add(v + "box_layer_norm.weight", MMHIP_G_ALWAYS, {H});
    e.box_ln_b = b.add(v + "box_layer_norm.bias", MMHIP_G_ALWAYS, {H});
    e.vin_end = b.off;
    const std::string em = "model.embeddings.";
    e.emb_begin = b.off;
    e.eln_w = b.add(em + "LayerNorm.weight", MMHIP_G_ALWAYS, {H});
    e.eln_b = b.add(em + "LayerNorm.bias", MMHIP_G_ALWAYS, {H});
    e.type = b.add(em + "token_type_embeddings.weight", MMHIP_G_ALWAYS, {c.type_vocab, H});
    e.pos = b.add(em + "position_embeddings.weight", MMHIP_G_ALWAYS, {c.max_pos, H});
    e.word = b.add(em + "word_embeddings.weight", MMHIP_G_ALWAYS, {c.vocab, H});
    e.emb_end = b.off;
    e.n_params = b.off;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; }
};

void build_workspace(mmhip_early& e) {
    const mmhip_early_config& c = e.cfg;
    const size_t H = c.hidden, I = c.inter, C = c.num_labels, Z = e.esz();
    const size_t Pm = 2 * (size_t)c.max_posts, Tm = c.max_text_len, Nm = c.max_boxes, Sm = Tm > Nm ? Tm : Nm;
    const size_t ML = Pm * Tm, MV = Pm * Nm, MS = Pm * Sm, heads = c.heads;
    Carver w;
    auto cp = [&](Copy& k, size_t n, size_t kk, bool tr) { k.w = w.take(n * kk * Z); k.wT = tr ? w.take(n * kk * Z) : 0; };
    auto attw = [&](AttW& a) { cp(a.qkv, 3 * H, H, true); cp(a.o, H, H, true); };
    auto ffnw = [&](FfnW& f) { cp(f.w1, I, H, true); cp(f.w2, H, I, true); };
    auto self_act = [&](SelfAct& a, size_t M, size_t posts, size_t S) {
        a.qkv = w.take(M * 3 * H * Z); a.att = w.take(M * H * Z); a.lse = w.take(posts * heads * S * 4); a.pre = w.take(M * H * Z);
        a.mean = w.take(M * 4); a.rstd = w.take(M * 4); a.y = w.take(M * H * Z);
        a.dpre = w.take(M * H * Z); a.dd = w.take(M * H * Z); a.datt = w.take(M * H * Z); a.dqkv = w.take(M * 3 * H * Z); a.dx = w.take(M * H * Z);
    };
    auto ffn_act = [&](FfnAct& a, size_t M) {
        a.h = w.take(M * I * Z); a.u = w.take(M * I * Z); a.pre = w.take(M * H * Z); a.mean = w.take(M * 4); a.rstd = w.take(M * 4); a.y = w.take(M * H * Z);
        a.dpre = w.take(M * H * Z); a.dd = w.take(M * H * Z); a.du = w.take(M * I * Z); a.dx = w.take(M * H * Z);
    };
    auto cross_act = [&](CrossAct& a, size_t Mq, size_t Mc) {
        // compact tensors (round 5): qkv / dqkv hold Mq rows of Q columns and Mc rows of K | V columns, att / datt Mq rows; no padded blocks, no scratch copies
        a.qkv = w.take(MS * 3 * H * Z); a.att = w.take(Mq * H * Z); a.lse = w.take(Pm * heads * Sm * 4);
        a.pre = w.take(Mq * H * Z); a.mean = w.take(Mq * 4); a.rstd = w.take(Mq * 4); a.y = w.take(Mq * H * Z);
        a.dpre = w.take(Mq * H * Z); a.dd = w.take(Mq * H * Z); a.datt = w.take(Mq * H * Z); a.dqkv = w.take(MS * 3 * H * Z);
        a.dxq = w.take(Mq * H * Z); a.dxc = w.take(Mc * H * Z);
    };
    for (auto& l : e.lang) { attw(l.aw); ffnw(l.fw); self_act(l.sa, ML, Pm, Tm); ffn_act(l.fa, ML); }
    for (auto& l : e.rel) { attw(l.aw); ffnw(l.fw); self_act(l.sa, MV, Pm, Nm); ffn_act(l.fa, MV); }
    for (auto& x : e.xl) {
        attw(x.cw); attw(x.lw); attw(x.vw); ffnw(x.lfw); ffnw(x.vfw);
        cross_act(x.cl, ML, MV); cross_act(x.cv, MV, ML);
        self_act(x.sl, ML, Pm, Tm); self_act(x.sv, MV, Pm, Nm); ffn_act(x.fl, ML); ffn_act(x.fv, MV);
    }
    cp(e.c_visn_fc, H, c.feat_dim, false);
    cp(e.c_box_fc, H, c.pos_dim, false);
    e.ids_all = w.take(ML * 8); e.mask_all = w.take(ML * 8); e.tt_all = w.take(ML * 8); e.pos_ids = w.take(ML * 4);
    e.lbias = w.take(ML * 4); e.lbias_x = w.take(MS * 4); e.vbias = w.take(MV * 4); e.vbias_x = w.take(MS * 4);
    e.x0 = w.take(ML * H * Z); e.xhat = w.take(ML * H * Z); e.rstd_emb = w.take(ML * 4); e.dx0 = w.take(ML * H * Z);
    e.feats16 = w.take(MV * c.feat_dim * Z); e.boxes16 = w.take(MV * (size_t)c.pos_dim * Z + 256);
    e.vf_pre = w.take(MV * H * Z); e.vf_mean = w.take(MV * 4); e.vf_rstd = w.take(MV * 4); e.vf = w.take(MV * H * Z);
    e.bx_pre = w.take(MV * H * Z); e.bx_mean = w.take(MV * 4); e.bx_rstd = w.take(MV * 4); e.bx = w.take(MV * H * Z);
    e.v0 = w.take(MV * H * Z); e.dv0 = w.take(MV * H * Z); e.dvf = w.take(MV * H * Z); e.dbx = w.take(MV * H * Z);
    e.dvf_pre = w.take(MV * H * Z); e.dbx_pre = w.take(MV * H * Z);
    e.g_dlang = w.take(ML * H * Z); e.g_dvisn = w.take(MV * H * Z);
    {
        size_t pf = partial_floats_embed((int)Pm, (int)Tm, (int)H);
        const size_t pc = partial_floats_colsum((int)MV, (int)H);
        if (pc > pf) pf = pc;
        e.g_partial = w.take(pf * 4);
    }
    auto f = [&](size_t n) { return w.take(n * 4); };
    const size_t Bm = c.max_posts;
    e.h_z = f(Pm * H); e.h_fus = f(Bm * H); e.h_fusd = f(Bm * H); e.h_out = f(Bm * C); e.h_tim = f(Bm * 2); e.h_embt = f(Bm * H); e.h_embv = f(Bm * H);
    e.h_argv = f(Bm * H); e.h_txt_n = f(Bm * H); e.h_img_n = f(Bm * H); e.h_txt_inv = f(Bm); e.h_img_inv = f(Bm); e.h_logits = f(Bm * Bm);
    e.h_dout = f(Bm * C); e.h_dlogits = f(Bm * Bm); e.h_dtim = f(Bm * 2); e.h_dembv = f(Bm * H); e.h_dembt = f(Bm * H); e.h_dfusd = f(Bm * H); e.h_dfus = f(Bm * H);
    e.h_dz = f(Pm * H); e.h_loss = f(8);
    e.ws_need = w.off;
}

int side_init(mmhip_early& e) {
    if (e.overlap < 0) { const char* v = getenv("MMHIP_EARLY_STREAMS"); e.overlap = v ? atoi(v) : 1; }
    if (e.side) return 0;
    CHECK_HIP(pool_stream(POOL_VIT, &e.side));          // process-wide streams (mmhip_common.h: pool_stream): vision chain ...
    CHECK_HIP(pool_stream(POOL_SIDE, &e.wside));        // ... and weight gradients + AdamW, as in the late-fusion engine
    hipEvent_t* evs[9] = {&e.ev_fork, &e.ev_join, &e.ev_l, &e.ev_v, &e.ev_l2, &e.ev_v2, &e.ev_lw, &e.ev_vw, &e.ev_w};
    for (auto ev : evs) CHECK_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return 0;
}
// the vision stream of this call: the internal one, or the caller's when MMHIP_EARLY_STREAMS=0
inline hipStream_t vstream(const mmhip_early& e, hipStream_t s) { return e.overlap > 0 ? e.side : s; }
inline hipStream_t wstream(const mmhip_early& e, hipStream_t s) { return e.overlap > 0 ? e.wside : s; }
inline int order(const mmhip_early& e, hipEvent_t ev, hipStream_t from, hipStream_t to) {      // `to` continues after everything enqueued on `from` so far
    if (from == to) return 0;
    CHECK_HIP(hipEventRecord(ev, from));
    CHECK_HIP(hipStreamWaitEvent(to, ev, 0));
    return 0;
}

int refresh(mmhip_early& e, hipStream_t s) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, I = c.inter;
    std::vector<CastMat> mats;
    auto add = [&](size_t off, const Copy& k, int rows, int cols) { mats.push_back(CastMat{e.P + off, e.ws + k.w, k.wT ? e.ws + k.wT : nullptr, rows, cols, 0}); };
    auto att = [&](const AttOff& o, const AttW& a) { add(o.qkv_w, a.qkv, 3 * H, H); add(o.o_w, a.o, H, H); };
    auto ffn = [&](const FfnOff& o, const FfnW& f) { add(o.w1, f.w1, I, H); add(o.w2, f.w2, H, I); };
    for (auto& l : e.lang) { att(l.att, l.aw); ffn(l.ffn, l.fw); }
    for (auto& l : e.rel) { att(l.att, l.aw); ffn(l.ffn, l.fw); }
    for (auto& x : e.xl) { att(x.cross, x.cw); att(x.lself, x.lw); att(x.vself, x.vw); ffn(x.lffn, x.lfw); ffn(x.vffn, x.vfw); }
    add(e.visn_fc_w, e.c_visn_fc, H, c.feat_dim);
    add(e.box_fc_w, e.c_box_fc, H, c.pos_dim);
    for (size_t i = 0; i < mats.size(); i += CAST_MAX_GROUP) {
        const int n = (int)(mats.size() - i < (size_t)CAST_MAX_GROUP ? mats.size() - i : (size_t)CAST_MAX_GROUP);
        CHECK_HIP(launch_cast_group(mats.data() + i, n, e.dt(), s));
    }
    return 0;
}

// dropout seeds: one 64-bit seed per block and call, derived from the call's seed (the block operators use fixed stream ids inside)
inline uint64_t block_seed(const mmhip_early& e, int block) { return e.seed * 0x9E3779B97F4A7C15ull + (uint64_t)(block + 1) * 0xD1B54A32D192ED03ull; }

// ------------------------------------------------------------------------------------------------ blocks (thin wrappers over capi_ops.hip)
int self_fwd(mmhip_early& e, const AttOff& o, const AttW& w, SelfAct& a, const void* x, const float* bias, int posts, int S, int blk, hipStream_t s) {
    const float p_att = e.train ? e.cfg.p_attn : 0.f, p_hid = e.train ? e.cfg.p_hidden : 0.f;
    return mmhip_op_self_att_block_fwd(e.dt(), x, bias, e.ws + w.qkv.w, e.P + o.qkv_b, e.ws + w.o.w, e.P + o.o_b, e.P + o.ln_w, e.P + o.ln_b, e.cfg.ln_eps, posts, S,
                                       e.cfg.heads, p_att, p_hid, block_seed(e, blk), e.ws + a.qkv, e.ws + a.att, e.wsp<float>(a.lse), e.ws + a.pre,
                                       e.wsp<float>(a.mean), e.wsp<float>(a.rstd), e.ws + a.y, s);
}
int ffn_fwd(mmhip_early& e, const FfnOff& o, const FfnW& w, FfnAct& a, const void* x, int M, int blk, hipStream_t s) {
    const float p_hid = e.train ? e.cfg.p_hidden : 0.f;
    return mmhip_op_ffn_block_fwd(e.dt(), x, e.ws + w.w1.w, e.P + o.b1, e.ws + w.w2.w, e.P + o.b2, e.P + o.ln_w, e.P + o.ln_b, e.cfg.ln_eps, M, e.cfg.hidden, e.cfg.inter,
                                  p_hid, block_seed(e, blk), e.ws + a.h, e.ws + a.u, e.ws + a.pre, e.wsp<float>(a.mean), e.wsp<float>(a.rstd), e.ws + a.y, s);
}
int cross_fwd(mmhip_early& e, const AttOff& o, const AttW& w, CrossAct& a, const void* xq, const void* xc, const float* keybias, int posts, int Sq, int Sk, int blk,
              hipStream_t s) {
    const float p_att = e.train ? e.cfg.p_attn : 0.f, p_hid = e.train ? e.cfg.p_hidden : 0.f;
    return mmhip_op_cross_att_block_fwd(e.dt(), xq, xc, keybias, e.ws + w.qkv.w, e.P + o.qkv_b, e.ws + w.o.w, e.P + o.o_b, e.P + o.ln_w, e.P + o.ln_b, e.cfg.ln_eps, posts,
                                        Sq, Sk, e.cfg.heads, p_att, p_hid, block_seed(e, blk), e.ws + a.qkv, e.ws + a.att, e.wsp<float>(a.lse), nullptr, nullptr,
                                        nullptr, e.ws + a.pre, e.wsp<float>(a.mean), e.wsp<float>(a.rstd), e.ws + a.y, s);
}

// weight-gradient products of a group of blocks: queued while the blocks' backward is enqueued, launched grouped (<= 8 problems per launch)
struct TNQueue {
    std::vector<GemmTNProblem> q;
    void add(const void* dy, int lda, const void* x, int ldb, float* C, int M, int Nn, int Nc, float* colsum) {
        GemmTNProblem p;
        memset(&p, 0, sizeof(p));
        p.A = dy; p.B = x; p.C = C; p.M = M; p.Nn = Nn; p.Nc = Nc; p.lda = lda; p.ldb = ldb; p.ldc = Nc; p.colsum = colsum;
        q.push_back(p);
    }
    int flush(int dtype, int accumulate, hipStream_t s) {
        for (size_t i = 0; i < q.size(); i += GEMM_TN_MAX_GROUP) {
            const int n = (int)(q.size() - i < (size_t)GEMM_TN_MAX_GROUP ? q.size() - i : (size_t)GEMM_TN_MAX_GROUP);
            CHECK_HIP(launch_gemm_tn(q.data() + i, n, accumulate, dtype, 0, s, 1.0f));
        }
        q.clear();
        return 0;
    }
};

int self_bwd(mmhip_early& e, const AttOff& o, const AttW& w, SelfAct& a, const void* dy, const void* x, const float* bias, int posts, int S, int blk, TNQueue& tn,
             hipStream_t s) {
    const int H = e.cfg.hidden, M = posts * S;
    const float p_att = e.train ? e.cfg.p_attn : 0.f, p_hid = e.train ? e.cfg.p_hidden : 0.f;
    char* dd = p_hid > 0.f ? e.ws + a.dd : e.ws + a.dpre;
    CHECK_RC(mmhip_op_self_att_block_bwd(e.dt(), dy, bias, e.ws + w.qkv.wT, e.ws + w.o.wT, e.P + o.ln_w, posts, S, e.cfg.heads, p_att, p_hid, block_seed(e, blk), e.ws + a.qkv,
                                         e.ws + a.att, e.wsp<float>(a.lse), e.ws + a.pre, e.wsp<float>(a.mean), e.wsp<float>(a.rstd), e.G + o.ln_w, e.G + o.ln_b, e.ws + a.dpre,
                                         dd, e.ws + a.datt, e.ws + a.dqkv, e.ws + a.dx, s));
    tn.add(dd, H, e.ws + a.att, H, e.G + o.o_w, M, H, H, e.G + o.o_b);
    tn.add(e.ws + a.dqkv, 3 * H, x, H, e.G + o.qkv_w, M, 3 * H, H, e.G + o.qkv_b);      // [Wq; Wk; Wv] and their biases are contiguous
    return 0;
}
int ffn_bwd(mmhip_early& e, const FfnOff& o, const FfnW& w, FfnAct& a, const void* dy, const void* x, int M, int blk, TNQueue& tn, hipStream_t s) {
    const int H = e.cfg.hidden, I = e.cfg.inter;
    const float p_hid = e.train ? e.cfg.p_hidden : 0.f;
    char* dd = p_hid > 0.f ? e.ws + a.dd : e.ws + a.dpre;
    CHECK_RC(mmhip_op_ffn_block_bwd(e.dt(), dy, e.ws + w.w1.wT, e.ws + w.w2.wT, e.P + o.ln_w, M, H, I, p_hid, block_seed(e, blk), e.ws + a.u, e.ws + a.pre, e.wsp<float>(a.mean),
                                    e.wsp<float>(a.rstd), e.G + o.ln_w, e.G + o.ln_b, e.ws + a.dpre, dd, e.ws + a.du, e.ws + a.dx, s));
    tn.add(dd, H, e.ws + a.h, I, e.G + o.w2, M, H, I, e.G + o.b2);
    tn.add(e.ws + a.du, I, x, H, e.G + o.w1, M, I, H, e.G + o.b1);
    return 0;
}
int cross_bwd(mmhip_early& e, const AttOff& o, const AttW& w, CrossAct& a, const void* dy, const void* xq, const void* xc, const float* keybias, int posts, int Sq, int Sk,
              int blk, TNQueue& tn, hipStream_t s) {
    const int H = e.cfg.hidden, Mq = posts * Sq, Mc = posts * Sk;
    const size_t Z = e.esz();
    const float p_att = e.train ? e.cfg.p_attn : 0.f, p_hid = e.train ? e.cfg.p_hidden : 0.f;
    char* dd = p_hid > 0.f ? e.ws + a.dd : e.ws + a.dpre;
    CHECK_RC(mmhip_op_cross_att_block_bwd(e.dt(), dy, keybias, e.ws + w.qkv.wT, e.ws + w.o.wT, e.P + o.ln_w, posts, Sq, Sk, e.cfg.heads, p_att, p_hid, block_seed(e, blk),
                                          e.ws + a.qkv, e.ws + a.att, e.wsp<float>(a.lse), e.ws + a.pre, e.wsp<float>(a.mean), e.wsp<float>(a.rstd), e.G + o.ln_w, e.G + o.ln_b,
                                          e.ws + a.dpre, dd, nullptr, e.ws + a.datt, e.ws + a.dqkv, nullptr, nullptr, e.ws + a.dxq, e.ws + a.dxc, s));
    // weight-gradient operands as the block left them: att [Mq, H]; dQ = rows [0, Mq) of dqkv's columns [0, H), [dK | dV] = rows [0, Mc) of its columns [H, 3H)
    tn.add(dd, H, e.ws + a.att, H, e.G + o.o_w, Mq, H, H, e.G + o.o_b);
    tn.add(e.ws + a.dqkv, 3 * H, xq, H, e.G + o.qkv_w, Mq, H, H, e.G + o.qkv_b);
    tn.add(e.ws + a.dqkv + (size_t)H * Z, 3 * H, xc, H, e.G + o.qkv_w + (size_t)H * H, Mc, 2 * H, H, e.G + o.qkv_b + H);
    return 0;
}
// y += x over [rows, H] (the two gradient contributions a stream's tensor receives in a cross-modality layer)
inline int add_rows(mmhip_early& e, const void* x, void* y, int rows, hipStream_t s) {
    CHECK_HIP(launch_scatter_rows16(x, y, rows, (size_t)e.cfg.hidden, e.cfg.hidden, 1, e.dt(), s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ forward
int encoder_forward(mmhip_early& e, const float* feats, const float* boxes, hipStream_t s) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, Bt = e.Bt, B = e.B, T = e.T, Nb = e.Nb, ML = Bt * T, MV = Bt * Nb, S = e.S(), dt = e.dt();
    hipStream_t sv = vstream(e, s);
    CHECK_RC(order(e, e.ev_fork, s, sv));
    // ---- language stream: embeddings (word + position 0..T-1 + token type -> LayerNorm -> dropout), key bias from the attention mask
    {
        EmbedArgs ea;
        memset(&ea, 0, sizeof(ea));
        ea.ids = e.wsp<int64_t>(e.ids_all); ea.mask = e.wsp<int64_t>(e.mask_all); ea.type_ids = e.wsp<int64_t>(e.tt_all);
        ea.word = e.P + e.word; ea.pos = e.P + e.pos; ea.type = e.P + e.type; ea.gamma = e.P + e.eln_w; ea.beta = e.P + e.eln_b;
        ea.x = e.ws + e.x0; ea.xhat = e.ws + e.xhat; ea.rstd = e.wsp<float>(e.rstd_emb); ea.pos_ids = e.wsp<int>(e.pos_ids); ea.maskbias = e.wsp<float>(e.lbias);
        ea.posts = Bt; ea.T = T; ea.H = H; ea.xlmr = 0; ea.pad_id = 0; ea.eps = c.ln_eps;
        ea.drop = drop_cfg(c.p_hidden, e.seed, 1, e.train);
        CHECK_HIP(launch_embed_fwd(ea, dt, s));
        if (T < S) {          // key bias of the text at the packed cross-attention length (keys past T masked)
            hipLaunchKernelGGL(pad_bias_kernel, dim3((Bt * S + 255) / 256), dim3(256), 0, s, e.wsp<float>(e.lbias), e.wsp<float>(e.lbias_x), Bt, T, S);
            CHECK_HIP(hipGetLastError());
        }
    }
    if (Nb < S) {
        hipLaunchKernelGGL(pad_bias_kernel, dim3((Bt * S + 255) / 256), dim3(256), 0, sv, (const float*)nullptr, e.wsp<float>(e.vbias_x), Bt, Nb, S);
        CHECK_HIP(hipGetLastError());
    }
    // ---- vision stream: visual-feature encoder  visn = dropout((LN(visn_fc(feats)) + LN(box_fc(boxes))) / 2)   (HF LxmertVisualFeatureEncoder)
    {
        const size_t nf = (size_t)B * Nb * c.feat_dim, nb = (size_t)B * Nb * c.pos_dim, Z = e.esz();
        for (int half = 0; half < (e.itm ? 2 : 1); ++half) {          // the ITM posts see the same images as the posts they are paired with
            CHECK_HIP(launch_cast(feats, e.ws + e.feats16 + (size_t)half * nf * Z, nf, dt, sv));
            CHECK_HIP(launch_cast(boxes, e.ws + e.boxes16 + (size_t)half * nb * Z, nb, dt, sv));
        }
        GemmNTArgs a;
        memset(&a, 0, sizeof(a));
        a.A = e.ws + e.feats16; a.lda = c.feat_dim; a.B = e.ws + e.c_visn_fc.w; a.ldb = c.feat_dim; a.C = e.ws + e.vf_pre; a.ldc = H; a.M = MV; a.N = H; a.K = c.feat_dim;
        a.bias = e.P + e.visn_fc_b; a.flags = GEMM_BIAS;
        CHECK_HIP(launch_gemm_nt(a, dt, sv));
        LNArgs l1{e.ws + e.vf_pre, e.ws + e.vf, e.P + e.visn_ln_w, e.P + e.visn_ln_b, e.wsp<float>(e.vf_mean), e.wsp<float>(e.vf_rstd), MV, H, H, H, c.ln_eps};
        CHECK_HIP(launch_layernorm_fwd(l1, dt, sv));
        memset(&a, 0, sizeof(a));
        a.A = e.ws + e.boxes16; a.lda = c.pos_dim; a.B = e.ws + e.c_box_fc.w; a.ldb = c.pos_dim; a.C = e.ws + e.bx_pre; a.ldc = H; a.M = MV; a.N = H; a.K = c.pos_dim;
        a.bias = e.P + e.box_fc_b; a.flags = GEMM_BIAS;
        CHECK_HIP(launch_gemm_nt(a, dt, sv));
        LNArgs l2{e.ws + e.bx_pre, e.ws + e.bx, e.P + e.box_ln_w, e.P + e.box_ln_b, e.wsp<float>(e.bx_mean), e.wsp<float>(e.bx_rstd), MV, H, H, H, c.ln_eps};
        CHECK_HIP(launch_layernorm_fwd(l2, dt, sv));
        CHECK_HIP(launch_avg_drop(e.ws + e.bx, e.ws + e.vf, e.ws + e.v0, (size_t)MV * H, 0.5f, drop_cfg(c.p_hidden, e.seed, 2, e.train), dt, sv));
    }
    const char* lang = e.ws + e.x0;
    const char* visn = e.ws + e.v0;
    const float* lbias = e.wsp<float>(e.lbias);
    const float* vbias = e.wsp<float>(e.vbias);          // all boxes are live: zeros (mmhip_early_bind clears it once; never written)
    int blk = 16;
    for (size_t i = 0; i < e.rel.size(); ++i) {
        PlainLayer& l = e.rel[i];
        CHECK_RC(self_fwd(e, l.att, l.aw, l.sa, visn, vbias, Bt, Nb, blk++, sv));
        CHECK_RC(ffn_fwd(e, l.ffn, l.fw, l.fa, e.ws + l.sa.y, MV, blk++, sv));
        visn = e.ws + l.fa.y;
    }
    for (size_t i = 0; i < e.lang.size(); ++i) {
        PlainLayer& l = e.lang[i];
        CHECK_RC(self_fwd(e, l.att, l.aw, l.sa, lang, lbias, Bt, T, blk++, s));
        CHECK_RC(ffn_fwd(e, l.ffn, l.fw, l.fa, e.ws + l.sa.y, ML, blk++, s));
        lang = e.ws + l.fa.y;
    }
    // key biases at the packed cross-attention length S = max(T, Nb): keys past a stream's own length are masked
    const float* lbias_x = T == S ? lbias : e.wsp<float>(e.lbias_x);
    const float* vbias_x = Nb == S ? vbias : e.wsp<float>(e.vbias_x);
    for (size_t i = 0; i < e.xl.size(); ++i) {
        XLayer& x = e.xl[i];
        // each stream needs the other's output of the previous layer (HF LxmertXLayer.forward: both cross attentions read the layer's inputs)
        CHECK_RC(order(e, e.ev_l, s, sv));
        CHECK_RC(order(e, e.ev_v, sv, s));
        CHECK_RC(cross_fwd(e, x.cross, x.cw, x.cl, lang, visn, vbias_x, Bt, T, Nb, blk++, s));
        CHECK_RC(self_fwd(e, x.lself, x.lw, x.sl, e.ws + x.cl.y, lbias, Bt, T, blk++, s));
        CHECK_RC(ffn_fwd(e, x.lffn, x.lfw, x.fl, e.ws + x.sl.y, ML, blk++, s));
        CHECK_RC(cross_fwd(e, x.cross, x.cw, x.cv, visn, lang, lbias_x, Bt, Nb, T, blk++, sv));
        CHECK_RC(self_fwd(e, x.vself, x.vw, x.sv, e.ws + x.cv.y, vbias, Bt, Nb, blk++, sv));
        CHECK_RC(ffn_fwd(e, x.vffn, x.vfw, x.fv, e.ws + x.sv.y, MV, blk++, sv));
        lang = e.ws + x.fl.y;
        visn = e.ws + x.fv.y;
    }
    CHECK_RC(order(e, e.ev_join, sv, s));
    return 0;
}
const char* lang_final(const mmhip_early& e) { return !e.xl.empty() ? e.ws + e.xl.back().fl.y : (!e.lang.empty() ? e.ws + e.lang.back().fa.y : e.ws + e.x0); }
const char* visn_final(const mmhip_early& e) { return !e.xl.empty() ? e.ws + e.xl.back().fv.y : (!e.rel.empty() ? e.ws + e.rel.back().fa.y : e.ws + e.v0); }

SmallGemmArgs small(const void* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, int act = ACT_NONE, int acc = 0) {
    SmallGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.act = act; a.accumulate = acc;
    return a;
}

// reference models/mm_early.py:128-163: linear_output = linear(dropout(relu(linear_fusion(x_t[:, 0])))); the text / image embeddings of the ITC loss are
// max-pools over the tokens (masked, detached) and the boxes; out_tim = linear_tim(CLS row of the swapped-text pass)
int heads_forward(mmhip_early& e, float* out, float* emb_t, float* emb_v, float* out_tim, hipStream_t s) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, C = c.num_labels, B = e.B, Bt = e.Bt, T = e.T, Nb = e.Nb, dt = e.dt();
    float* z = e.wsp<float>(e.h_z);
    CHECK_HIP(launch_gather_rows_f32(lang_final(e), (size_t)T * H, z, H, Bt, H, dt, s));
    CHECK_HIP(launch_small_nt(small(z, H, e.P + e.fus_w, H, e.P + e.fus_b, e.wsp<float>(e.h_fus), H, B, H, H, ACT_RELU), DT_F32, s));
    CHECK_HIP(launch_elementwise(EW_DROPOUT, e.wsp<float>(e.h_fus), nullptr, e.wsp<float>(e.h_fusd), (size_t)B * H, 0.f, drop_cfg(c.p_head, e.seed, 3, e.train), s));
    CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_fusd), H, e.P + e.lin_w, H, e.P + e.lin_b, e.wsp<float>(e.h_out), C, B, C, H), DT_F32, s));
    if (e.itm) CHECK_HIP(launch_small_nt(small(z + (size_t)B * H, H, e.P + e.tim_w, H, e.P + e.tim_b, e.wsp<float>(e.h_tim), 2, B, 2, H), DT_F32, s));
    const int grid = (B * (H / 4) + 255) / 256;
    if (dt == DT_BF16) {
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)lang_final(e), e.wsp<int64_t>(e.mask_all), e.wsp<float>(e.h_embt), (int*)nullptr, B, T, H);
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)visn_final(e), (const int64_t*)nullptr, e.wsp<float>(e.h_embv), e.wsp<int>(e.h_argv), B, Nb, H);
    } else if (dt == DT_F16) {
        hipLaunchKernelGGL(maxpool_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)lang_final(e), e.wsp<int64_t>(e.mask_all), e.wsp<float>(e.h_embt), (int*)nullptr, B, T, H);
        hipLaunchKernelGGL(maxpool_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)visn_final(e), (const int64_t*)nullptr, e.wsp<float>(e.h_embv), e.wsp<int>(e.h_argv), B, Nb, H);
    } else {
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)lang_final(e), e.wsp<int64_t>(e.mask_all), e.wsp<float>(e.h_embt), (int*)nullptr, B, T, H);
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)visn_final(e), (const int64_t*)nullptr, e.wsp<float>(e.h_embv), e.wsp<int>(e.h_argv), B, Nb, H);
    }
    CHECK_HIP(hipGetLastError());
    if (out) CHECK_HIP(hipMemcpyAsync(out, e.ws + e.h_out, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s));
    if (emb_t) CHECK_HIP(hipMemcpyAsync(emb_t, e.ws + e.h_embt, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    if (emb_v) CHECK_HIP(hipMemcpyAsync(emb_v, e.ws + e.h_embv, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    if (out_tim && e.itm) CHECK_HIP(hipMemcpyAsync(out_tim, e.ws + e.h_tim, (size_t)B * 2 * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ backward
// heads: gradients of linear / linear_fusion / linear_tim; d lang_final (CLS rows only) into g_dlang, d visn_final (arg-max rows of the first B posts)
int heads_backward(mmhip_early& e, hipStream_t s) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, C = c.num_labels, B = e.B, Bt = e.Bt, T = e.T, Nb = e.Nb, dt = e.dt();
    const DropCfg nodrop = drop_cfg(0.f, 0, 0, false);
    float* z = e.wsp<float>(e.h_z);
    float* dz = e.wsp<float>(e.h_dz);
    float* dfusd = e.wsp<float>(e.h_dfusd);
    float* dfus = e.wsp<float>(e.h_dfus);
    const float* d_out = e.bd_out;
    CHECK_HIP(launch_small_nn(small(d_out, C, e.P + e.lin_w, H, nullptr, dfusd, H, B, H, C), s));
    CHECK_HIP(launch_elementwise(EW_DROPOUT, dfusd, nullptr, dfusd, (size_t)B * H, 0.f, drop_cfg(c.p_head, e.seed, 3, e.train), s));
    CHECK_HIP(launch_small_tn(small(d_out, C, e.wsp<float>(e.h_fusd), H, nullptr, e.G + e.lin_w, H, B, H, 0, 0, 1), DT_F32, C, s));
    CHECK_HIP(launch_bias_grad_f32(d_out, B, C, C, e.G + e.lin_b, 1, s));
    CHECK_HIP(launch_elementwise(EW_RELU_BWD, dfusd, e.wsp<float>(e.h_fus), dfus, (size_t)B * H, 0.f, nodrop, s));
    CHECK_HIP(launch_small_tn(small(dfus, H, z, H, nullptr, e.G + e.fus_w, H, B, H, 0, 0, 1), DT_F32, H, s));
    CHECK_HIP(launch_bias_grad_f32(dfus, B, H, H, e.G + e.fus_b, 1, s));
    CHECK_HIP(launch_small_nn(small(dfus, H, e.P + e.fus_w, H, nullptr, dz, H, B, H, H), s));
    if (e.itm) {
        if (e.bd_tim) {
            CHECK_HIP(launch_small_nn(small(e.bd_tim, 2, e.P + e.tim_w, H, nullptr, dz + (size_t)B * H, H, B, H, 2), s));
            CHECK_HIP(launch_small_tn(small(e.bd_tim, 2, z + (size_t)B * H, H, nullptr, e.G + e.tim_w, H, B, H, 0, 0, 1), DT_F32, 2, s));
            CHECK_HIP(launch_bias_grad_f32(e.bd_tim, B, 2, 2, e.G + e.tim_b, 1, s));
        } else {
            CHECK_HIP(hipMemsetAsync(dz + (size_t)B * H, 0, (size_t)B * H * 4, s));
        }
    }
    CHECK_HIP(launch_scatter_cls_rows(dz, e.ws + e.g_dlang, Bt, T, H, dt, s, 1.0f));
    const size_t total = (size_t)Bt * Nb * (H / 4);
    const float* dv = e.bd_embv;
    if (dt == DT_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(cap(total)), dim3(256), 0, s, dv, e.wsp<int>(e.h_argv), (bf16_t*)(e.ws + e.g_dvisn), Bt, B, Nb, H);
    else if (dt == DT_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, dim3(cap(total)), dim3(256), 0, s, dv, e.wsp<int>(e.h_argv), (f16_t*)(e.ws + e.g_dvisn), Bt, B, Nb, H);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(cap(total)), dim3(256), 0, s, dv, e.wsp<int>(e.h_argv), (float*)(e.ws + e.g_dvisn), Bt, B, Nb, H);
    CHECK_HIP(hipGetLastError());
    return 0;
}

// the layer inputs as the forward saw them
const char* lang_in_of_x(const mmhip_early& e, size_t i) { return i ? e.ws + e.xl[i - 1].fl.y : (!e.lang.empty() ? e.ws + e.lang.back().fa.y : e.ws + e.x0); }
const char* visn_in_of_x(const mmhip_early& e, size_t i) { return i ? e.ws + e.xl[i - 1].fv.y : (!e.rel.empty() ? e.ws + e.rel.back().fa.y : e.ws + e.v0); }

int num_stages(const mmhip_early& e) {
    const int depth = (int)(e.lang.size() > e.rel.size() ? e.lang.size() : e.rel.size());
    return 1 + (int)e.xl.size() + depth + 1;
}

// AdamW + operand refresh of one layer stage on the weight-gradient stream, right behind the stage's grouped dW launches: an HBM-bound update beside the
// MFMA-bound activation-gradient chains of the stages below instead of 1.4 + 0.45 ms at the end of the step (LXMERT: 213 M parameters, 6.8 GB of
// optimizer traffic).  Single rank only: under data parallelism the optimizer waits for the exchange.
struct StageOpt { AdamWArgs a; float* m; float* v; };
int stage_refresh(mmhip_early& e, int stage, hipStream_t s) {
    const int H = e.cfg.hidden, I = e.cfg.inter, X = (int)e.xl.size();
    std::vector<CastMat> mats;
    auto add = [&](size_t off, const Copy& k, int rows, int cols) { mats.push_back(CastMat{e.P + off, e.ws + k.w, k.wT ? e.ws + k.wT : nullptr, rows, cols, 0}); };
    auto att = [&](const AttOff& o, const AttW& a) { add(o.qkv_w, a.qkv, 3 * H, H); add(o.o_w, a.o, H, H); };
    auto ffn = [&](const FfnOff& o, const FfnW& f) { add(o.w1, f.w1, I, H); add(o.w2, f.w2, H, I); };
    if (stage <= X) {
        XLayer& x = e.xl[X - stage];
        att(x.cross, x.cw); att(x.lself, x.lw); att(x.vself, x.vw); ffn(x.lffn, x.lfw); ffn(x.vffn, x.vfw);
    } else {
        const int d = stage - X - 1, li = (int)e.lang.size() - 1 - d, ri = (int)e.rel.size() - 1 - d;
        if (li >= 0) { att(e.lang[li].att, e.lang[li].aw); ffn(e.lang[li].ffn, e.lang[li].fw); }
        if (ri >= 0) { att(e.rel[ri].att, e.rel[ri].aw); ffn(e.rel[ri].ffn, e.rel[ri].fw); }
    }
    for (size_t i = 0; i < mats.size(); i += CAST_MAX_GROUP) {
        const int n = (int)(mats.size() - i < (size_t)CAST_MAX_GROUP ? mats.size() - i : (size_t)CAST_MAX_GROUP);
        CHECK_HIP(launch_cast_group(mats.data() + i, n, e.dt(), s));
    }
    return 0;
}

// stage: 0 heads | 1 .. X cross-modality layers last -> first | X+1 .. X+D language / relational layers by depth below the cross layers | X+D+1 inputs
int backward_stage(mmhip_early& e, int stage, const char** dlang_io, const char** dvisn_io, hipStream_t s) {
    const mmhip_early_config& c = e.cfg;
    const int H = c.hidden, Bt = e.Bt, T = e.T, Nb = e.Nb, ML = Bt * T, MV = Bt * Nb, S = e.S(), dt = e.dt();
    const int X = (int)e.xl.size(), D = (int)(e.lang.size() > e.rel.size() ? e.lang.size() : e.rel.size());
    hipStream_t sv = vstream(e, s), sw = wstream(e, s);
    const float* lbias = e.wsp<float>(e.lbias);
    const float* vbias = e.wsp<float>(e.vbias);
    const float* lbias_x = T == S ? lbias : e.wsp<float>(e.lbias_x);
    const float* vbias_x = Nb == S ? vbias : e.wsp<float>(e.vbias_x);
    const int nl = (int)e.lang.size(), nr = (int)e.rel.size();
    if (stage == 0) {
        CHECK_RC(heads_backward(e, s));
        *dlang_io = e.ws + e.g_dlang;
        *dvisn_io = e.ws + e.g_dvisn;
        CHECK_RC(order(e, e.ev_fork, s, sv));          // the vision stream starts from d visn_final
        return 0;
    }
    if (stage <= X) {
        const size_t i = (size_t)(X - stage);
        XLayer& x = e.xl[i];
        // block indices as in the forward: 16 + 2 (nr + nl) + 6 i + {0 cross_l, 1 self_l, 2 ffn_l, 3 cross_v, 4 self_v, 5 ffn_v}
        const int b0 = 16 + 2 * (nr + nl) + 6 * (int)i;
        const char* lin = lang_in_of_x(e, i);
        const char* vin = visn_in_of_x(e, i);
        TNQueue tl, tv, tshared;
        CHECK_RC(ffn_bwd(e, x.lffn, x.lfw, x.fl, *dlang_io, e.ws + x.sl.y, ML, b0 + 2, tl, s));
        CHECK_RC(self_bwd(e, x.lself, x.lw, x.sl, e.ws + x.fl.dx, e.ws + x.cl.y, lbias, Bt, T, b0 + 1, tl, s));
        CHECK_RC(cross_bwd(e, x.cross, x.cw, x.cl, e.ws + x.sl.dx, lin, vin, vbias_x, Bt, T, Nb, b0 + 0, tshared, s));
        CHECK_RC(ffn_bwd(e, x.vffn, x.vfw, x.fv, *dvisn_io, e.ws + x.sv.y, MV, b0 + 5, tv, sv));
        CHECK_RC(self_bwd(e, x.vself, x.vw, x.sv, e.ws + x.fv.dx, e.ws + x.cv.y, vbias, Bt, Nb, b0 + 4, tv, sv));
        CHECK_RC(cross_bwd(e, x.cross, x.cw, x.cv, e.ws + x.sv.dx, vin, lin, lbias_x, Bt, Nb, T, b0 + 3, tshared, sv));
        // d lang_in = (queries of its own cross block) + (context of the vision stream's); likewise d visn_in
        CHECK_RC(order(e, e.ev_l, s, sv));
        CHECK_RC(order(e, e.ev_v, sv, s));
        CHECK_RC(add_rows(e, e.ws + x.cv.dxc, e.ws + x.cl.dxq, ML, s));
        CHECK_RC(add_rows(e, e.ws + x.cl.dxc, e.ws + x.cv.dxq, MV, sv));
        *dlang_io = e.ws + x.cl.dxq;
        *dvisn_io = e.ws + x.cv.dxq;
        // weight gradients: each stream's own blocks as plain stores; the ONE cross-attention module both directions used accumulates both
        // contributions, on the language stream (it has waited for the vision stream's cross block above)
        // ... all of it on the weight-gradient stream, behind both chains of this layer
        CHECK_RC(order(e, e.ev_lw, s, sw));
        CHECK_RC(order(e, e.ev_vw, sv, sw));
        CHECK_RC(tl.flush(dt, 0, sw));
        CHECK_RC(tv.flush(dt, 0, sw));
        CHECK_RC(tshared.flush(dt, 1, sw));
        return 0;
    }
    if (stage <= X + D) {
        const int d = stage - X - 1, li = nl - 1 - d, ri = nr - 1 - d;
        if (li >= 0) {
            PlainLayer& l = e.lang[li];
            const int b0 = 16 + 2 * nr + 2 * li;
            TNQueue tn;
            const char* xin = li ? e.ws + e.lang[li - 1].fa.y : e.ws + e.x0;
            CHECK_RC(ffn_bwd(e, l.ffn, l.fw, l.fa, *dlang_io, e.ws + l.sa.y, ML, b0 + 1, tn, s));
            CHECK_RC(self_bwd(e, l.att, l.aw, l.sa, e.ws + l.fa.dx, xin, lbias, Bt, T, b0, tn, s));
            CHECK_RC(order(e, e.ev_lw, s, sw));
            CHECK_RC(tn.flush(dt, 0, sw));
            *dlang_io = e.ws + l.sa.dx;
        }
        if (ri >= 0) {
            PlainLayer& l = e.rel[ri];
            const int b0 = 16 + 2 * ri;
            TNQueue tn;
            const char* xin = ri ? e.ws + e.rel[ri - 1].fa.y : e.ws + e.v0;
            CHECK_RC(ffn_bwd(e, l.ffn, l.fw, l.fa, *dvisn_io, e.ws + l.sa.y, MV, b0 + 1, tn, sv));
            CHECK_RC(self_bwd(e, l.att, l.aw, l.sa, e.ws + l.fa.dx, xin, vbias, Bt, Nb, b0, tn, sv));
            CHECK_RC(order(e, e.ev_vw, sv, sw));
            CHECK_RC(tn.flush(dt, 0, sw));
            *dvisn_io = e.ws + l.sa.dx;
        }
        return 0;
    }
    // ---- inputs: embeddings on the language stream, the visual-feature encoder on the vision stream
    {
        EmbedBwdArgs b;
        memset(&b, 0, sizeof(b));
        b.dx = *dlang_io; b.xhat = e.ws + e.xhat; b.rstd = e.wsp<float>(e.rstd_emb); b.gamma = e.P + e.eln_w;
        b.ids = e.wsp<int64_t>(e.ids_all); b.pos_ids = e.wsp<int>(e.pos_ids); b.type_ids = e.wsp<int64_t>(e.tt_all);
        b.dword = e.G + e.word; b.dpos = e.G + e.pos; b.dtype = e.G + e.type; b.dgamma = e.G + e.eln_w; b.dbeta = e.G + e.eln_b;
        b.posts = Bt; b.T = T; b.H = H; b.pad_id = 0; b.pos_pad_id = 0;          // HF LxmertEmbeddings: padding_idx = 0 on all three tables
        b.drop = drop_cfg(c.p_hidden, e.seed, 1, e.train);
        b.partial = e.wsp<float>(e.g_partial);
        b.alpha = 1.0f;
        CHECK_HIP(launch_embed_bwd(b, dt, s));
    }
    {
        // visn = dropout((f + bx) / 2): d f = d bx = dropout-backward(d visn) / 2
        CHECK_HIP(launch_avg_drop(*dvisn_io, nullptr, e.ws + e.dv0, (size_t)MV * H, 0.5f, drop_cfg(c.p_hidden, e.seed, 2, e.train), dt, sv));
        LNBwdArgs b1;
        memset(&b1, 0, sizeof(b1));
        b1.dy = e.ws + e.dv0; b1.x = e.ws + e.vf_pre; b1.gamma = e.P + e.visn_ln_w; b1.mean = e.wsp<float>(e.vf_mean); b1.rstd = e.wsp<float>(e.vf_rstd);
        b1.dx = e.ws + e.dvf_pre; b1.dgamma = e.G + e.visn_ln_w; b1.dbeta = e.G + e.visn_ln_b; b1.rows = MV; b1.width = H; b1.alpha = 1.0f;
        CHECK_HIP(launch_layernorm_bwd(b1, dt, sv));
        LNBwdArgs b2 = b1;
        b2.x = e.ws + e.bx_pre; b2.gamma = e.P + e.box_ln_w; b2.mean = e.wsp<float>(e.bx_mean); b2.rstd = e.wsp<float>(e.bx_rstd);
        b2.dx = e.ws + e.dbx_pre; b2.dgamma = e.G + e.box_ln_w; b2.dbeta = e.G + e.box_ln_b;
        CHECK_HIP(launch_layernorm_bwd(b2, dt, sv));
        TNQueue tn;
        tn.add(e.ws + e.dvf_pre, H, e.ws + e.feats16, c.feat_dim, e.G + e.visn_fc_w, MV, H, c.feat_dim, e.G + e.visn_fc_b);
        if (c.pos_dim <= 4) CHECK_HIP(launch_tn_few_cols(e.ws + e.dbx_pre, H, e.ws + e.boxes16, c.pos_dim, e.G + e.box_fc_w, e.G + e.box_fc_b, MV, H, c.pos_dim, dt, sv));
        else tn.add(e.ws + e.dbx_pre, H, e.ws + e.boxes16, c.pos_dim, e.G + e.box_fc_w, MV, H, c.pos_dim, e.G + e.box_fc_b);
        CHECK_RC(tn.flush(dt, 0, sv));
    }
    CHECK_RC(order(e, e.ev_join, sv, s));
    CHECK_RC(order(e, e.ev_w, sw, s));          // every weight gradient is final in the caller's stream order
    return 0;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int mmhip_early_create(const mmhip_early_config* cfg, mmhip_early_handle* out) {
    if (!cfg || !out) return MMHIP_E_INVALID;
    const mmhip_early_config& c = *cfg;
    if (c.hidden <= 0 || c.hidden % 128 || c.hidden > 1024 || c.heads * 64 != c.hidden || c.inter % 128) return MMHIP_E_INVALID;
    if (c.l_layers < 0 || c.r_layers < 0 || c.x_layers < 0 || c.vocab < 1 || c.type_vocab < 1 || c.type_vocab > 2) return MMHIP_E_INVALID;
    if (c.max_text_len < 1 || c.max_text_len > 128 || c.max_boxes < 1 || c.max_boxes > 128 || c.max_posts < 1 || c.max_posts > 1024) return MMHIP_E_INVALID;
    if (c.max_pos < c.max_text_len || c.feat_dim < 4 || c.feat_dim % 4 || c.pos_dim < 4 || c.pos_dim % 4 || c.num_labels < 1 || c.num_labels > 64) return MMHIP_E_INVALID;
    if (c.dtype != MMHIP_BF16 && c.dtype != MMHIP_F16 && c.dtype != MMHIP_BF16X3) return MMHIP_E_INVALID;
    mmhip_early* e = new (std::nothrow) mmhip_early();
    if (!e) return MMHIP_E_INVALID;
    e->cfg = c;
    build_layout(*e);
    build_workspace(*e);
    *out = e;
    return 0;
}
void mmhip_early_destroy(mmhip_early_handle h) {
    if (!h) return;
    for (auto& ev : h->timing_evs) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        for (hipEvent_t ev : {h->ev_fork, h->ev_join, h->ev_l, h->ev_v, h->ev_l2, h->ev_v2, h->ev_lw, h->ev_vw, h->ev_w})
            if (ev) (void)hipEventDestroy(ev);
        if (h->wside) (void)hipStreamSynchronize(h->wside);                       // pooled streams: drained, not destroyed
    }
    delete h;
}
int mmhip_early_param_count(mmhip_early_handle h) { return h ? (int)h->params.size() : MMHIP_E_INVALID; }
int mmhip_early_param_info_at(mmhip_early_handle h, int i, mmhip_param_info* out) {
    if (!h || !out || i < 0 || i >= (int)h->params.size()) return MMHIP_E_INVALID;
    *out = h->params[i];
    return 0;
}
uint64_t mmhip_early_numel(mmhip_early_handle h) { return h ? h->n_params : 0; }
uint64_t mmhip_early_workspace_bytes(mmhip_early_handle h) { return h ? h->ws_need : 0; }
int mmhip_early_bind(mmhip_early_handle h, float* params, float* grads, void* workspace, uint64_t workspace_bytes, void* stream) {
    if (!h || !params || !workspace) return MMHIP_E_INVALID;
    if (workspace_bytes < h->ws_need) return MMHIP_E_CAPACITY;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)workspace) & 255) return MMHIP_E_INVALID;
    h->P = params; h->G = grads; h->ws = (char*)workspace;
    h->fwd_done = false;
    const size_t Pm = 2 * (size_t)h->cfg.max_posts;
    CHECK_HIP(hipMemsetAsync(h->ws + h->vbias, 0, Pm * h->cfg.max_boxes * 4, (hipStream_t)stream));      // every box is a live key
    return 0;
}
int mmhip_early_set_index_counter(mmhip_early_handle h, uint32_t* device_word) {
    if (!h || ((uintptr_t)device_word & 3)) return MMHIP_E_INVALID;
    h->bad_index = device_word;
    return 0;
}
// Measurement only (the early-fusion counterpart of mmhip_gemm_timing): enable != 0 arms HIP events around every NT GEMM launch of the following
// mmhip_early_train_step calls (on the stream each launch goes to); ms / launches / flops (may be NULL) return the sums over the launches timed
// since the last reset -- synchronises.  Algorithmic FLOPs 2 M N K per launch.
int mmhip_early_gemm_timing(mmhip_early_handle h, int enable, int reset, double* ms, uint64_t* launches, double* flops) {
    if (!h) return MMHIP_E_INVALID;
    mmhip_early& e = *h;
    if (ms || launches || flops) {
        double tms = 0, tf = 0;
        for (size_t i = 0; i < e.timing.used; ++i) {
            CHECK_HIP(hipEventSynchronize(e.timing_evs[i].b));
            float t = 0;
            CHECK_HIP(hipEventElapsedTime(&t, e.timing_evs[i].a, e.timing_evs[i].b));
            tms += t; tf += e.timing_evs[i].flops;
        }
        if (ms) *ms = tms;
        if (launches) *launches = e.timing.used;
        if (flops) *flops = tf;
    }
    if (reset) e.timing.used = 0;
    if (enable && e.timing_evs.empty()) {
        e.timing_evs.resize(2048);
        for (auto& ev : e.timing_evs) { CHECK_HIP(hipEventCreate(&ev.a)); CHECK_HIP(hipEventCreate(&ev.b)); }
        e.timing.evs = e.timing_evs.data(); e.timing.capacity = e.timing_evs.size(); e.timing.used = 0;
    }
    e.timing_on = enable != 0;
    return 0;
}
int mmhip_early_refresh_weights(mmhip_early_handle h, void* stream) {
    if (!h || !h->ws) return MMHIP_E_STATE;
    return refresh(*h, (hipStream_t)stream);
}
int mmhip_early_num_stages(mmhip_early_handle h) { return h ? num_stages(*h) : MMHIP_E_INVALID; }
int mmhip_early_stage_grad_range(mmhip_early_handle h, int stage, uint64_t* begin, uint64_t* end) {
    if (!h || !begin || !end || stage < 0 || stage >= num_stages(*h)) return MMHIP_E_INVALID;
    const mmhip_early& e = *h;
    const int X = (int)e.xl.size(), D = (int)(e.lang.size() > e.rel.size() ? e.lang.size() : e.rel.size());
    if (stage == 0) { *begin = e.heads_begin; *end = e.heads_end; }
    else if (stage <= X) { *begin = e.xl[X - stage].begin; *end = e.xl[X - stage].end; }
    else if (stage <= X + D) {
        const int d = stage - X - 1, li = (int)e.lang.size() - 1 - d, ri = (int)e.rel.size() - 1 - d;
        *begin = li >= 0 ? e.lang[li].begin : e.rel[ri].begin;
        *end = ri >= 0 ? e.rel[ri].end : e.lang[li].end;
    } else { *begin = e.vin_begin; *end = e.emb_end; }
    return 0;
}

// tim_*: the swapped texts of the ITM pass (reference :146-161) as explicit tensors, OR itm_src [B] (device int64): row b of the ITM pass is
// row itm_src[b] of this batch (what the reference's sampling produces: MMEarly_Model.prepare_itm_inputs) -- gathered on the device
static int forward_impl(mmhip_early_handle h, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, const float* feats, const float* boxes,
                        const int64_t* tim_ids, const int64_t* tim_mask, const int64_t* tim_token_type_ids, const int64_t* itm_src, int B, int T, int Nb, int train,
                        uint64_t seed, float* out, float* emb_t, float* emb_v, float* out_tim, void* stream) {
    if (!h || !h->ws) return MMHIP_E_STATE;
    if (!ids || !mask || !feats || !boxes || B < 1 || T < 1 || Nb < 1) return MMHIP_E_INVALID;
    if ((tim_ids == nullptr) != (tim_mask == nullptr) || (tim_ids && itm_src)) return MMHIP_E_INVALID;
    mmhip_early& e = *h;
    if (B > e.cfg.max_posts || T > e.cfg.max_text_len || Nb > e.cfg.max_boxes) return MMHIP_E_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    e.B = B; e.T = T; e.Nb = Nb; e.itm = tim_ids != nullptr || itm_src != nullptr; e.Bt = e.itm ? 2 * B : B; e.train = train != 0; e.seed = seed;
    e.fwd_done = false; e.itc_done = false; e.bd_out = e.bd_embv = e.bd_tim = nullptr;
    CHECK_RC(side_init(e));
    const size_t nb = (size_t)B * T * 8;
    // indices are clamped into their tables on the way into the engine's copies (launch_copy_ids_clamped: why)
    const size_t nt = (size_t)B * T;
    CHECK_HIP(launch_copy_ids_clamped(ids, e.wsp<int64_t>(e.ids_all), nt, e.cfg.vocab, e.bad_index, s));
    CHECK_HIP(hipMemcpyAsync(e.ws + e.mask_all, mask, nb, hipMemcpyDeviceToDevice, s));
    if (token_type_ids) CHECK_HIP(launch_copy_ids_clamped(token_type_ids, e.wsp<int64_t>(e.tt_all), nt, e.cfg.type_vocab, e.bad_index, s));
    else CHECK_HIP(hipMemsetAsync(e.ws + e.tt_all, 0, nb, s));
    if (tim_ids) {
        CHECK_HIP(launch_copy_ids_clamped(tim_ids, e.wsp<int64_t>(e.ids_all) + nt, nt, e.cfg.vocab, e.bad_index, s));
        CHECK_HIP(hipMemcpyAsync(e.ws + e.mask_all + nb, tim_mask, nb, hipMemcpyDeviceToDevice, s));
        if (tim_token_type_ids) CHECK_HIP(launch_copy_ids_clamped(tim_token_type_ids, e.wsp<int64_t>(e.tt_all) + nt, nt, e.cfg.type_vocab, e.bad_index, s));
        else CHECK_HIP(hipMemsetAsync(e.ws + e.tt_all + nb, 0, nb, s));
    } else if (itm_src) {
        const int grid = (B * T + 255) / 256;
        for (size_t off : {e.ids_all, e.mask_all, e.tt_all})
            hipLaunchKernelGGL(gather_rows_i64_kernel, dim3(grid), dim3(256), 0, s, e.wsp<int64_t>(off), itm_src, e.wsp<int64_t>(off) + (size_t)B * T, B, T);
        CHECK_HIP(hipGetLastError());
    }
    CHECK_RC(encoder_forward(e, feats, boxes, s));
    CHECK_RC(heads_forward(e, out, emb_t, emb_v, out_tim, s));
    e.fwd_done = true;
    return 0;
}
int mmhip_early_forward(mmhip_early_handle h, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, const float* feats, const float* boxes,
                        const int64_t* tim_ids, const int64_t* tim_mask, const int64_t* tim_token_type_ids, int B, int T, int Nb, int train, uint64_t seed,
                        float* out, float* emb_t, float* emb_v, float* out_tim, void* stream) {
    return forward_impl(h, ids, mask, token_type_ids, feats, boxes, tim_ids, tim_mask, tim_token_type_ids, nullptr, B, T, Nb, train, seed, out, emb_t, emb_v, out_tim,
                        stream);
}

// loss mix of the reference's step (models/mm_early.py:366-379): (1 - b_itc - b_itm) CE(out, one-hot labels as probabilities, class weights)
// + b_itc clip_loss(logits_per_text) + b_itm CE(out_tim, lbl_tim), logits_per_text = normalise(emb_t) normalise(emb_v)^T exp(logit_scale) (:165-172).
// Leaves the output gradients in the handle for mmhip_early_backward(NULL ...) and ADDS d logit_scale to the gradient buffer.
int mmhip_early_loss(mmhip_early_handle h, const int64_t* onehot, const float* class_w, const int64_t* lbl_tim, float w_cls, float w_itc, float w_itm, float* loss,
                     float* logits_per_text, void* stream) {
    if (!h || !h->fwd_done) return MMHIP_E_STATE;
    if (!onehot) return MMHIP_E_INVALID;
    mmhip_early& e = *h;
    if (w_itm != 0.f && (!e.itm || !lbl_tim)) return MMHIP_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int B = e.B, H = e.cfg.hidden;
    if (w_itc != 0.f || logits_per_text) {
        ItcArgs it{e.wsp<float>(e.h_embt), e.wsp<float>(e.h_embv), e.P + e.logit_scale, e.wsp<float>(e.h_txt_n), e.wsp<float>(e.h_img_n), e.wsp<float>(e.h_txt_inv),
                   e.wsp<float>(e.h_img_inv), e.wsp<float>(e.h_logits), B, H};
        CHECK_HIP(launch_itc_fwd(it, s));
        e.itc_done = true;
        if (logits_per_text) CHECK_HIP(hipMemcpyAsync(logits_per_text, e.ws + e.h_logits, (size_t)B * B * 4, hipMemcpyDeviceToDevice, s));
    }
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.out_cls = e.wsp<float>(e.h_out); a.onehot = onehot; a.class_w = class_w;
    a.logits_per_text = w_itc != 0.f ? e.wsp<float>(e.h_logits) : nullptr;
    a.out_tim = w_itm != 0.f ? e.wsp<float>(e.h_tim) : nullptr; a.lbl_tim = lbl_tim;
    a.w_cls = w_cls; a.w_itc = w_itc; a.w_itm = w_itm;
    a.loss = e.wsp<float>(e.h_loss);
    a.d_out_cls = e.wsp<float>(e.h_dout);
    a.d_logits = w_itc != 0.f ? e.wsp<float>(e.h_dlogits) : nullptr;
    a.d_out_tim = w_itm != 0.f ? e.wsp<float>(e.h_dtim) : nullptr;
    a.B = B; a.C = e.cfg.num_labels;
    CHECK_HIP(launch_loss(a, s));
    if (loss) CHECK_HIP(hipMemcpyAsync(loss, a.loss, 16, hipMemcpyDeviceToDevice, s));
    e.bd_out = a.d_out_cls; e.bd_tim = a.d_out_tim; e.bd_embv = nullptr;
    if (w_itc != 0.f) {
        if (!e.G) return MMHIP_E_STATE;
        // the text embedding is detached (reference :139): only the image side and logit_scale receive the ITC gradient
        ItcBwdArgs ib{a.d_logits, e.wsp<float>(e.h_logits), e.wsp<float>(e.h_txt_n), e.wsp<float>(e.h_img_n), e.wsp<float>(e.h_txt_inv), e.wsp<float>(e.h_img_inv),
                      e.P + e.logit_scale, e.wsp<float>(e.h_dembt), e.wsp<float>(e.h_dembv), e.G + e.logit_scale, B, H};
        CHECK_HIP(launch_itc_bwd(ib, s));
        e.bd_embv = e.wsp<float>(e.h_dembv);
    }
    return 0;
}

// loss.backward() (reference :381): gradients of every parameter on the path, ADDED into / stored to the bound gradient buffer.  CONTRACT as
// mmhip_backward's: the gradient buffer is zero on entry over the ranges that receive gradients (LayerNorm weights, embedding rows, heads and
// the shared cross-attention module accumulate; the other weight gradients are plain stores).  NULL pointers: the gradients mmhip_early_loss
// left in the handle; else explicit fp32 output gradients d_out [B, C], d_emb_v [B, H] (may be NULL), d_out_tim [B, 2] (may be NULL).
static int backward_impl(mmhip_early& e, const float* d_out, const float* d_emb_v, const float* d_out_tim, hipStream_t s, mmhip_exchange_cb cb, void* user,
                         const StageOpt* opt = nullptr) {
    if (!e.fwd_done || !e.G) return MMHIP_E_STATE;
    if (d_out) { e.bd_out = d_out; e.bd_embv = d_emb_v; e.bd_tim = d_out_tim; }
    else if (!e.bd_out) return MMHIP_E_STATE;
    const char* dl = nullptr;
    const char* dv = nullptr;
    const int n = num_stages(e);
    hipStream_t sv = vstream(e, s);
    for (int st = 0; st < n; ++st) {
        CHECK_RC(backward_stage(e, st, &dl, &dv, s));
        if (opt && st >= 1 && st < n - 1) {
            uint64_t b = 0, en = 0;
            CHECK_RC(mmhip_early_stage_grad_range(&e, st, &b, &en));
            AdamWArgs a = opt->a;
            a.p = e.P + b; a.g = e.G + b; a.m = opt->m + b; a.v = opt->v + b; a.n = en - b;
            hipStream_t sw = wstream(e, s);
            CHECK_HIP(launch_adamw(a, sw));
            CHECK_RC(stage_refresh(e, st, sw));
        }
        if (cb && st >= 1) {
            // stage st-1's gradients are final once its work on the vision and the weight-gradient stream is ordered into the caller's stream
            CHECK_RC(order(e, e.ev_v2, sv, s));
            CHECK_RC(order(e, e.ev_l2, wstream(e, s), s));
            CHECK_RC(cb(user, st - 1));
        }
    }
    if (cb) { CHECK_RC(cb(user, n - 1)); CHECK_RC(cb(user, MMHIP_CB_WAIT_DENSE)); }
    return 0;
}
int mmhip_early_backward(mmhip_early_handle h, const float* d_out, const float* d_emb_v, const float* d_out_tim, void* stream) {
    if (!h) return MMHIP_E_STATE;
    return backward_impl(*h, d_out, d_emb_v, d_out_tim, (hipStream_t)stream, nullptr, nullptr);
}

// One training step of MMEarly_Model.train (reference models/mm_early.py:332-407: forward, loss mix, backward, optimizer.step) in ONE call:
// forward (train mode; the ITM pass batched with the main pass as 2B posts) + loss + backward + AdamW over the parameter ranges that receive a
// gradient for this flag set (torch skips `grad is None`: never the pooler; logit_scale only with ITC, linear_tim only with ITM) + operand refresh.
// on_stage (may be NULL): the data-parallel exchange hook of mmhip_train_step_dp -- stage st's gradient range (mmhip_early_stage_grad_range) is
// final in `stream` order when on_stage(user, st) is called; MMHIP_CB_WAIT_DENSE before the optimizer.
int mmhip_early_train_step(mmhip_early_handle h, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, const float* feats, const float* boxes,
                           const int64_t* itm_src, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T, int Nb, uint64_t seed,
                           int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int step, float grad_scale, float* loss, void* stream, mmhip_exchange_cb on_stage, void* user) {
    if (!h || !h->ws || !h->G) return MMHIP_E_STATE;
    if (!adam_m || !adam_v || !onehot || step < 1) return MMHIP_E_INVALID;
    if (use_itm && (!itm_src || !lbl_tim)) return MMHIP_E_INVALID;
    mmhip_early& e = *h;
    hipStream_t s = (hipStream_t)stream;
    struct SinkScope { bool on; explicit SinkScope(GemmTimingSink* k) : on(k != nullptr) { if (on) gemm_timing_sink(k); } ~SinkScope() { if (on) gemm_timing_sink(nullptr); } };
    SinkScope sink_scope(e.timing_on ? &e.timing : nullptr);
    CHECK_RC(forward_impl(h, ids, mask, token_type_ids, feats, boxes, nullptr, nullptr, nullptr, use_itm ? itm_src : nullptr, B, T, Nb, 1, seed, nullptr, nullptr, nullptr,
                          nullptr, stream));
    CHECK_RC(mmhip_early_loss(h, onehot, class_w, use_itm ? lbl_tim : nullptr, w_cls, use_itc ? w_itc : 0.f, use_itm ? w_itm : 0.f, loss, nullptr, stream));
    AdamWArgs a;
    memset(&a, 0, sizeof(a));
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    a.zero_grad = 1; a.grad_scale = grad_scale;
    const char* early_env = getenv("MMHIP_EARLY_ADAMW");
    const bool layer_opt = !on_stage && e.overlap != 0 && (early_env ? atoi(early_env) != 0 : true);
    StageOpt so{a, adam_m, adam_v};
    CHECK_RC(side_init(e));
    CHECK_RC(backward_impl(e, nullptr, nullptr, nullptr, s, on_stage, user, layer_opt && e.overlap > 0 ? &so : nullptr));
    const bool stepped = layer_opt && e.overlap > 0;
    uint64_t lb = 0, le = 0;          // the layer stages' ranges are contiguous: [first x layer's begin, vin_begin)
    if (stepped) { lb = e.heads_end; le = e.vin_begin; }
    // AdamW over the merged ranges of the active groups, in address order (the layer stages already stepped on the weight-gradient stream)
    bool act[6] = {false, use_itc != 0, use_itm != 0, false, true, false};
    uint64_t rb = 0, re = 0;
    bool open = false;
    auto flush = [&]() -> int {
        if (!open || re <= rb) return 0;
        a.p = e.P + rb; a.g = e.G + rb; a.m = adam_m + rb; a.v = adam_v + rb; a.n = re - rb;
        CHECK_HIP(launch_adamw(a, s));
        return 0;
    };
    for (const auto& p : e.params) {
        if (!act[p.group] || (stepped && p.offset >= lb && p.offset < le)) continue;
        const uint64_t b = p.offset, en = p.offset + ((p.numel + 3) & ~(uint64_t)3);
        if (open && b == re) { re = en; continue; }
        CHECK_RC(flush());
        rb = b; re = en; open = true;
    }
    CHECK_RC(flush());
    if (stepped) {          // only the visual-feature encoder's matrices are left to refresh
        CastMat m2[2] = {{e.P + e.visn_fc_w, e.ws + e.c_visn_fc.w, nullptr, e.cfg.hidden, e.cfg.feat_dim, 0}, {e.P + e.box_fc_w, e.ws + e.c_box_fc.w, nullptr, e.cfg.hidden, e.cfg.pos_dim, 0}};
        CHECK_HIP(launch_cast_group(m2, 2, e.dt(), s));
        return 0;
    }
    return refresh(e, s);
}

}  // extern "C"
