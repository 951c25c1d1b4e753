// Head kernels (fp32 arithmetic; B posts is 64..128, so these are latency-sized, not roofline-sized):
//   * cross-modal attention fusion for the CLS query only (reference models/mm_late.py:98-113, :195-210):
//     only row 0 of softmax(Q K^T) V is consumed, and K = x_v W_K^T + b_K, V = x_v W_V^T + b_V are affine in the
//     frozen image tokens, so   scores_j = (W_K^T q) . x_v[j] (+ const),  ctx = W_V (sum_j p_j x_v[j]) + b_V
//     -- exact algebra, 0.47 GF/post of fc_K/fc_V GEMMs become two [B,768]x[768,768] products.
//   * ITC similarity (HF vision_text_dual_encoder :261-274) and its backward
//   * fused loss forward+backward: weighted soft-target CE (run_mm_late.py:85), clip_loss (utils.py:225-231),
//     ITM CE (run_mm_late.py:97), mixed as models/mm_late.py:473-487
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// ------------------------------------------------------------------------------------------------ fusion attention
// One workgroup per text post.  Each wave takes image tokens j = w, w+4, ...: a lane owns 12 of the 768 columns
// (three 8-byte loads per token, independent across tokens, so the loads pipeline), dots reduce across the wave,
// weighted sums accumulate in registers and are combined across the 4 waves through LDS.
// 8 waves per post: the kernel is a latency chain over 197 x 768 image tokens read twice (300 KB per workgroup, 64-128 workgroups)
// on the critical path between the towers and the backward; eight waves keep twice the token loads in flight of four (40 KB of LDS partials)
static constexpr int FA_WAVES = 8, FA_THREADS = FA_WAVES * 64;
template <typename T>
__global__ __launch_bounds__(FA_THREADS) void fusion_attn_fwd_kernel(FusionAttnArgs a) {
    __shared__ __attribute__((aligned(16))) float q[1024];
    __shared__ float sc[FA_THREADS];
    __shared__ float red[FA_WAVES];
    __shared__ float part[FA_WAVES][1024];
    const int bt = blockIdx.x, b = bt % a.B, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < a.H; c += FA_THREADS) q[c] = a.qk[(size_t)bt * a.H + c];
    __syncthreads();
    const T* xv = (const T*)a.xv + (size_t)b * a.P * a.H;
    const int nch = a.H >> 2;                   // 4-element chunks; lane owns chunks lane, lane+64, ...
    // dots of 4 tokens at a time per wave: the four wave reductions interleave, hiding the cross-lane latency
    for (int j0 = w * 4; j0 < a.P; j0 += 4 * FA_WAVES) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = lane; ch < nch; ch += 64) {
            const f32x4 qq = *reinterpret_cast<const f32x4*>(q + ch * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, a.P - 1);
                typename Vec<T>::v4 x = *reinterpret_cast<const typename Vec<T>::v4*>(xv + (size_t)j * a.H + ch * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[u] += to_f<T>(x[e]) * qq[e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], o);
        if (lane < 4 && j0 + lane < a.P) sc[j0 + lane] = s[lane] * a.scale;
    }
    __syncthreads();
    const float v = threadIdx.x < a.P ? sc[threadIdx.x] : -INFINITY;
    const float mx = block_reduce(v, red, true);
    const float e = threadIdx.x < a.P ? __expf(v - mx) : 0.f;
    const float sum = block_reduce(e, red, false);
    if (threadIdx.x < a.P) {
        sc[threadIdx.x] = e / sum;
        a.prob[(size_t)bt * a.P + threadIdx.x] = e / sum;
    }
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) acc[t][ee] = 0.f;
    for (int j = w; j < a.P; j += FA_WAVES) {
        const float p = sc[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ch = lane + 64 * t;
            if (ch < nch) {
                typename Vec<T>::v4 x = *reinterpret_cast<const typename Vec<T>::v4*>(xv + (size_t)j * a.H + ch * 4);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) acc[t][ee] += p * to_f<T>(x[ee]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) part[w][(lane + 64 * t) * 4 + ee] = acc[t][ee];
    __syncthreads();
    for (int c = threadIdx.x; c < a.H; c += FA_THREADS) {
        float sres = 0.f;
#pragma unroll
        for (int ww = 0; ww < FA_WAVES; ++ww) sres += part[ww][c];
        a.xbar[(size_t)bt * a.H + c] = sres;
    }
}
template <typename T>
__global__ __launch_bounds__(FA_THREADS) void fusion_attn_bwd_kernel(FusionAttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dxb[1024];
    __shared__ float ds[FA_THREADS];
    __shared__ float red[FA_WAVES];
    __shared__ float part[FA_WAVES][1024];
    const int bt = blockIdx.x, b = bt % a.B, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < a.H; c += FA_THREADS) dxb[c] = a.dxbar[(size_t)bt * a.H + c];
    __syncthreads();
    const T* xv = (const T*)a.xv + (size_t)b * a.P * a.H;
    const int nch = a.H >> 2;
    for (int j0 = w * 4; j0 < a.P; j0 += 4 * FA_WAVES) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = lane; ch < nch; ch += 64) {
            const f32x4 qq = *reinterpret_cast<const f32x4*>(dxb + ch * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, a.P - 1);
                typename Vec<T>::v4 x = *reinterpret_cast<const typename Vec<T>::v4*>(xv + (size_t)j * a.H + ch * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[u] += to_f<T>(x[e]) * qq[e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], o);
        if (lane < 4 && j0 + lane < a.P) ds[j0 + lane] = s[lane];          // dp_j
    }
    __syncthreads();
    const float p = threadIdx.x < a.P ? a.prob[(size_t)bt * a.P + threadIdx.x] : 0.f;
    const float dp = threadIdx.x < a.P ? ds[threadIdx.x] : 0.f;
    const float tsum = block_reduce(p * dp, red, false);
    __syncthreads();
    if (threadIdx.x < a.P) ds[threadIdx.x] = p * (dp - tsum) * a.scale;
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) acc[t][ee] = 0.f;
    for (int j = w; j < a.P; j += FA_WAVES) {
        const float d = ds[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ch = lane + 64 * t;
            if (ch < nch) {
                typename Vec<T>::v4 x = *reinterpret_cast<const typename Vec<T>::v4*>(xv + (size_t)j * a.H + ch * 4);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) acc[t][ee] += d * to_f<T>(x[ee]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (lane + 64 * t < nch)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) part[w][(lane + 64 * t) * 4 + ee] = acc[t][ee];
    __syncthreads();
    for (int c = threadIdx.x; c < a.H; c += FA_THREADS) {
        float sres = 0.f;
#pragma unroll
        for (int ww = 0; ww < FA_WAVES; ++ww) sres += part[ww][c];
        a.dqk[(size_t)bt * a.H + c] = sres;
    }
}

// ------------------------------------------------------------------------------------------------ ITC
// block b < B: normalise txt_e[b]; block B + b: normalise img_e[b]
__global__ __launch_bounds__(256) void itc_norm_kernel(ItcArgs a) {
    __shared__ float red[4];
    const int i = blockIdx.x % a.B;
    const bool img = blockIdx.x >= a.B;
    const float* e = (img ? a.img_e : a.txt_e) + (size_t)i * a.E;
    float s = 0.f;
    for (int c = threadIdx.x; c < a.E; c += 256) s += e[c] * e[c];
    const float inv = 1.0f / sqrtf(block_reduce(s, red, false));
    float* n = (img ? a.img_n : a.txt_n) + (size_t)i * a.E;
    for (int c = threadIdx.x; c < a.E; c += 256) n[c] = e[c] * inv;
    if (threadIdx.x == 0) (img ? a.img_inv : a.txt_inv)[i] = inv;
}
// logits_per_text[i][j] = txt_n[i] . img_n[j] * exp(logit_scale); one wave per (i, j)
__global__ __launch_bounds__(256) void itc_logits_kernel(ItcArgs a) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= a.B * a.B) return;
    const int i = idx / a.B, j = idx % a.B;
    float s = 0.f;
    for (int c = lane; c < a.E; c += 64) s += a.txt_n[(size_t)i * a.E + c] * a.img_n[(size_t)j * a.E + c];
    s = wave_sum(s);
    if (lane == 0) a.logits[idx] = s * __expf(a.logit_scale[0]);
}
// block b < B: d txt_e[b]; block B + b: d img_e[b]; block 0 also adds d logit_scale = sum dlogits * logits
__global__ __launch_bounds__(256) void itc_bwd_kernel(ItcBwdArgs a) {
    __shared__ float dn[1024];
    __shared__ float red[4];
    const int i = blockIdx.x % a.B;
    const bool img = blockIdx.x >= a.B;
    const float es = __expf(a.logit_scale[0]);
    const float* other = img ? a.txt_n : a.img_n;
    const float* mine = (img ? a.img_n : a.txt_n) + (size_t)i * a.E;
    float dot = 0.f;
    for (int c = threadIdx.x; c < a.E; c += 256) {
        float s = 0.f;
        for (int j = 0; j < a.B; ++j) {
            const float dl = img ? a.dlogits[(size_t)j * a.B + i] : a.dlogits[(size_t)i * a.B + j];
            s += dl * other[(size_t)j * a.E + c];
        }
        s *= es;
        dn[c] = s;
        dot += s * mine[c];
    }
    dot = block_reduce(dot, red, false);
    const float inv = (img ? a.img_inv : a.txt_inv)[i];
    float* out = (img ? a.dimg_e : a.dtxt_e) + (size_t)i * a.E;
    for (int c = threadIdx.x; c < a.E; c += 256) out[c] = (dn[c] - mine[c] * dot) * inv;
    if (blockIdx.x == 0 && a.dlogit_scale) {
        float s = 0.f;
        for (int k = threadIdx.x; k < a.B * a.B; k += 256) s += a.dlogits[k] * a.logits[k];
        s = block_reduce(s, red, false);
        if (threadIdx.x == 0) a.dlogit_scale[0] += s;
    }
}

// ------------------------------------------------------------------------------------------------ losses (one block)
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
    __shared__ float red[4];
    __shared__ float rowlse[1024], collse[1024];
    const int B = a.B, C = a.C, t = threadIdx.x;
    // ---- classification: -(1/B) sum_b sum_c w_c y_bc log_softmax(out)_bc   (soft-target CE, normalised by B)
    float lc = 0.f;
    int correct = 0;
    for (int b = t; b < B; b += 256) {
        const float* z = a.out_cls + (size_t)b * C;
        float mx = -INFINITY;
        int am = 0, ay = 0;
        long ymax = -1;
        for (int c = 0; c < C; ++c) {
            if (z[c] > mx) { mx = z[c]; am = c; }
            const long y = a.onehot[(size_t)b * C + c];
            if (y > ymax) { ymax = y; ay = c; }
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(z[c] - mx);
        const float lse = mx + __logf(se);
        float wy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float w = a.class_w ? a.class_w[c] : 1.f;
            const float y = (float)a.onehot[(size_t)b * C + c];
            lc -= w * y * (z[c] - lse);
            wy += w * y;
        }
        if (a.d_out_cls)
            for (int c = 0; c < C; ++c) {
                const float w = a.class_w ? a.class_w[c] : 1.f;
                const float y = (float)a.onehot[(size_t)b * C + c];
                a.d_out_cls[(size_t)b * C + c] = a.w_cls * (__expf(z[c] - lse) * wy - w * y) / B;
            }
        correct += (am == ay);
    }
    lc = block_reduce(lc, red, false) / B;
    const float fc = block_reduce((float)correct, red, false);
    // ---- ITC: (CE(S, arange) + CE(S^T, arange)) / 2
    float li = 0.f;
    if (a.logits_per_text) {
        const float* S = a.logits_per_text;
        for (int i = t; i < B; i += 256) {
            float mr = -INFINITY, mc = -INFINITY;
            for (int j = 0; j < B; ++j) { mr = fmaxf(mr, S[(size_t)i * B + j]); mc = fmaxf(mc, S[(size_t)j * B + i]); }
            float sr = 0.f, scn = 0.f;
            for (int j = 0; j < B; ++j) { sr += __expf(S[(size_t)i * B + j] - mr); scn += __expf(S[(size_t)j * B + i] - mc); }
            rowlse[i] = mr + __logf(sr);
            collse[i] = mc + __logf(scn);
            li += (rowlse[i] - S[(size_t)i * B + i]) + (collse[i] - S[(size_t)i * B + i]);
        }
        li = block_reduce(li, red, false) / (2.f * B);
        __syncthreads();
        if (a.d_logits)
            for (int k = t; k < B * B; k += 256) {
                const int i = k / B, j = k % B;
                const float s = S[k];
                a.d_logits[k] = a.w_itc * (__expf(s - rowlse[i]) + __expf(s - collse[j]) - (i == j ? 2.f : 0.f)) / (2.f * B);
            }
    }
    // ---- ITM: index-target CE, mean
    float lm = 0.f;
    if (a.out_tim) {
        for (int b = t; b < B; b += 256) {
            const float z0 = a.out_tim[b * 2], z1 = a.out_tim[b * 2 + 1];
            const float mx = fmaxf(z0, z1), lse = mx + __logf(__expf(z0 - mx) + __expf(z1 - mx));
            const int y = (int)a.lbl_tim[b];
            lm += lse - (y ? z1 : z0);
            if (a.d_out_tim) {
                a.d_out_tim[b * 2] = a.w_itm * (__expf(z0 - lse) - (y == 0 ? 1.f : 0.f)) / B;
                a.d_out_tim[b * 2 + 1] = a.w_itm * (__expf(z1 - lse) - (y == 1 ? 1.f : 0.f)) / B;
            }
        }
        lm = block_reduce(lm, red, false) / B;
    }
    if (t == 0) {
        a.loss[0] = a.w_cls * lc + (a.logits_per_text ? a.w_itc * li : 0.f) + (a.out_tim ? a.w_itm * lm : 0.f);
        a.loss[1] = lc; a.loss[2] = li; a.loss[3] = lm;
        if (a.n_correct) a.n_correct[0] = (int)(fc + 0.5f);
    }
}

// ------------------------------------------------------------------------------------------------ small elementwise
__global__ __launch_bounds__(256) void elementwise_kernel(int op, const float* a, const float* b, float* out, size_t n, float alpha, DropCfg drop) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        switch (op) {
            case EW_TANH_BWD: v = a[i] * (1.f - b[i] * b[i]); break;             // a = dy, b = y
            case EW_RELU_BWD: v = b[i] > 0.f ? a[i] : 0.f; break;                // a = dy, b = y
            case EW_DROPOUT: v = (drop.thresh16 && !mm_keep((uint32_t)i, drop)) ? 0.f : a[i] * (drop.thresh16 ? drop.keep_scale : 1.f); break;
            case EW_ADD: v = a[i] + alpha * b[i]; break;
            default: v = a[i]; break;
        }
        out[i] = v;
    }
}
// out[c] (+)= sum_r d[r][c], fp32, small row counts
__global__ __launch_bounds__(256) void bias_grad_f32_kernel(const float* d, int rows, int cols, int ld, float* out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += d[(size_t)r * ld + c];
    out[c] = accumulate ? out[c] + s : s;
}

// ------------------------------------------------------------------------------------------------ fused AdamW
// torch.optim.AdamW semantics (decoupled decay first, then moments; models/mm_late.py:420-422) over one flat fp32 buffer.
// a gradient element that is inf / NaN (f16 mode: an overflow in the 16-bit gradient chain) must not reach m / v / p: it is read as
// 0 and counted; the trainer lowers the loss scale / raises when the counter moves (MMLate_Model.train)
__device__ __forceinline__ float guard_finite(float g, bool& bad) {
    const bool ok = fabsf(g) <= 3.4028234e38f;       // false for NaN
    bad = bad || !ok;
    return ok ? g : 0.f;
}
__global__ __launch_bounds__(256) void adamw_kernel(AdamWArgs a) {
    const size_t n4 = a.n / 4;
    if (a.skip && *a.skip) {          // the step is void: leave p / m / v alone, clear the gradient for the next step
        if (a.zero_grad) {
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<f32x4*>(a.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) a.g[n4 * 4 + threadIdx.x] = 0.f;
        }
        return;
    }
    bool bad = false;
    const float one_m_b1 = 1.f - a.beta1, one_m_b2 = 1.f - a.beta2;
    const float decay = 1.f - a.lr * a.wd, step = a.lr / a.bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], g = reinterpret_cast<f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = guard_finite(g[e] * a.grad_scale, bad);
            p[e] *= decay;
            m[e] = m[e] + one_m_b1 * (ge - m[e]);
            v[e] = v[e] * a.beta2 + one_m_b2 * ge * ge;
            const float denom = sqrtf(v[e]) / a.bc2_sqrt + a.eps;
            p[e] -= step * (m[e] / denom);
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.zero_grad) reinterpret_cast<f32x4*>(a.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (bad && a.nonfinite) atomicAdd(a.nonfinite, 1u);
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        bool bad1 = false;
        const float ge = guard_finite(a.g[i] * a.grad_scale, bad1);
        float p = a.p[i] * decay;
        const float m = a.m[i] + one_m_b1 * (ge - a.m[i]);
        const float v = a.v[i] * a.beta2 + one_m_b2 * ge * ge;
        p -= step * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (a.zero_grad) a.g[i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_fusion_attn_fwd(const FusionAttnArgs& a, int dtype, hipStream_t s) {
    if (a.Bt <= 0) return hipSuccess;
    if (a.P > FA_THREADS || a.H > 1024 || a.H % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(fusion_attn_fwd_kernel<bf16_t>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(fusion_attn_fwd_kernel<f16_t>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    else hipLaunchKernelGGL(fusion_attn_fwd_kernel<float>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fusion_attn_bwd(const FusionAttnBwdArgs& a, int dtype, hipStream_t s) {
    if (a.Bt <= 0) return hipSuccess;
    if (a.P > FA_THREADS || a.H > 1024 || a.H % 4) return hipErrorInvalidValue;
    if (dtype == DT_BF16) hipLaunchKernelGGL(fusion_attn_bwd_kernel<bf16_t>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(fusion_attn_bwd_kernel<f16_t>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    else hipLaunchKernelGGL(fusion_attn_bwd_kernel<float>, dim3(a.Bt), dim3(FA_THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_itc_fwd(const ItcArgs& a, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(itc_norm_kernel, dim3(2 * a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(itc_logits_kernel, dim3((a.B * a.B + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_itc_bwd(const ItcBwdArgs& a, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (a.E > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(itc_bwd_kernel, dim3(2 * a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_loss(const LossArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.B > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_elementwise(int op, const float* a, const float* b, float* out, size_t n, float alpha, const DropCfg& drop, hipStream_t s) {
    if (!n) return hipSuccess;
    size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(elementwise_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, s, op, a, b, out, n, alpha, drop);
    return hipGetLastError();
}
hipError_t launch_bias_grad_f32(const float* d, int rows, int cols, int ld, float* out, int accumulate, hipStream_t s) {
    if (cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(bias_grad_f32_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, d, rows, cols, ld, out, accumulate);
    return hipGetLastError();
}
// Row-lazy variant for embedding tables (the word table is 69 % of Bernice's parameters and <= B*T of its 250 002 rows see
// a gradient per step).  One wave per row per iteration; the state byte is wave-uniform.
//   state 0                : g = m = v = 0  ->  p *= (1 - lr*wd)   (exactly what the dense kernel computes; 8 B/param)
//   ROW_HAS_GRAD           : full update, gradient cleared, state -> ROW_HAS_MOMENTS
//   ROW_HAS_MOMENTS only   : full update with g = 0 (gradient neither read nor cleared)
__global__ __launch_bounds__(256) void adamw_rows_kernel(AdamWArgs a, int rows, int width, uint8_t* __restrict__ state) {
    const int lane = threadIdx.x & 63;
    const int nch = width >> 2;
    const float one_m_b1 = 1.f - a.beta1, one_m_b2 = 1.f - a.beta2;
    const float decay = 1.f - a.lr * a.wd, step = a.lr / a.bc1;
    bool bad = false;
    if (a.skip && *a.skip) {          // void step: rows that received a gradient lose it (and the flag), nothing else moves
        for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
            const int st = state[row];
            if (!(st & ROW_HAS_GRAD) || !a.zero_grad) continue;
            f32x4* __restrict__ g4 = reinterpret_cast<f32x4*>(a.g + (size_t)row * width);
            for (int c = lane; c < nch; c += 64) g4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane == 0) state[row] = (uint8_t)(st & ~ROW_HAS_GRAD);
        }
        return;
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int st = state[row];
        f32x4* __restrict__ p4 = reinterpret_cast<f32x4*>(a.p + (size_t)row * width);
        // 1 - lr*wd rounds to exactly 1.0f whenever lr*wd < 2^-24 -- true for the reference's defaults (1e-5 x 2.5e-4 = 2.5e-9;
        // torch's own param.mul_(1 - lr*wd) multiplies fp32 parameters by exactly 1.0f there): the decay-only rows are then
        // bit-identical without being read or written at all (1.5 GB of traffic per step for Bernice's word table)
        if (st == 0 && decay == 1.0f) continue;
        if (st == 0) {
            for (int c = lane; c < nch; c += 64) {
                f32x4 p = p4[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] *= decay;
                p4[c] = p;
            }
            continue;
        }
        f32x4* __restrict__ g4 = reinterpret_cast<f32x4*>(a.g + (size_t)row * width);
        f32x4* __restrict__ m4 = reinterpret_cast<f32x4*>(a.m + (size_t)row * width);
        f32x4* __restrict__ v4 = reinterpret_cast<f32x4*>(a.v + (size_t)row * width);
        const bool has_g = st & ROW_HAS_GRAD;
        for (int c = lane; c < nch; c += 64) {
            f32x4 p = p4[c], m = m4[c], v = v4[c];
            f32x4 g = has_g ? g4[c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = guard_finite(g[e] * a.grad_scale, bad);
                p[e] *= decay;
                m[e] = m[e] + one_m_b1 * (ge - m[e]);
                v[e] = v[e] * a.beta2 + one_m_b2 * ge * ge;
                const float denom = sqrtf(v[e]) / a.bc2_sqrt + a.eps;
                p[e] -= step * (m[e] / denom);
            }
            p4[c] = p; m4[c] = m; v4[c] = v;
            if (has_g && a.zero_grad) g4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int nst = ROW_HAS_MOMENTS | ((has_g && !a.zero_grad) ? ROW_HAS_GRAD : 0);
        if (lane == 0 && nst != st) state[row] = (uint8_t)nst;
    }
    if (bad && a.nonfinite) atomicAdd(a.nonfinite, 1u);
}
hipError_t launch_adamw_rows(const AdamWArgs& a, int rows, int width, uint8_t* row_state, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (width % 4 || !row_state) return hipErrorInvalidValue;
    const int g = (rows + 3) / 4;
    hipLaunchKernelGGL(adamw_rows_kernel, dim3(g > 16384 ? 16384 : g), dim3(256), 0, s, a, rows, width, row_state);
    return hipGetLastError();
}
hipError_t launch_adamw(const AdamWArgs& a, hipStream_t s) {
    if (!a.n) return hipSuccess;
    size_t g = (a.n / 4 + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)(g < 1 ? 1 : (g > 16384 ? 16384 : g))), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mmhip
