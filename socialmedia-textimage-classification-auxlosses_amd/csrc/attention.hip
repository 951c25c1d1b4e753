// Fused self-attention forward / backward for one (post, head) per workgroup; head dim 64, S <= 224 keys.
// The whole K / V of a head stays in LDS (S=128: 16 KB each; S=197: 28 KB each), scores never touch HBM.
//
// MFMA 32x32x16.  Forward computes S^T = K.Q^T so a lane owns one query column: the soft-max is a reduction over
// the lane's registers (+ one cross-half shuffle), and the fp32 accumulator tile P^T is, after conversion, directly
// the B operand of O^T = V^T.P^T (MI355X guide "accumulator tile as the next MFMA's operand").  V^T fragments come
// from the row-major V image with the transposing LDS read ds_read_b64_tr_b16.
//
// Backward (text tower only, S <= 128): wave w owns key tile w; S = Q.K^T and dP = dO.V^T are computed with the key on
// the lane, so P and dS are the B operands of dV^T = dO^T.P and dK^T = Q^T.dS; only dS crosses LDS, once, for dQ.
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

static constexpr int HD = 64;          // head dim
static constexpr float LOG2E = 1.4426950408889634f;
// v_exp_f32 itself: exp2f() wraps it in a denormal-range rescue (compare, select, add, ldexp: five more VALU instructions per element);
// a soft-max weight below 2^-126 of the row maximum is zero either way
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// byte offset of 16-B chunk `ch` (0..7) of row `row` in a [rows][64] 16-bit image read by rows (ds_read_b128)
__device__ __forceinline__ int rowimg_off(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }
// ... in an image read with the transposing read (4 rows x 16 cols blocks): flip the 64-B half on rows 2,3 mod 4
__device__ __forceinline__ int trimg_off(int row, int ch) { return row * 128 + ((ch ^ (((row >> 1) & 1) << 2)) << 4); }

// A-operand fragment X^T (32 "rows" = feature columns c0..c0+31 of the image, k = 16 image rows) for k-order
// element j of lane half h = image row r0 + 8*(j>>2) + 4*h + (j&3)   (the accumulator-as-operand order)
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 tr_frag_acc_order(const char* img, int r0, int c0, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int col = c0 + 16 * g1 + 4 * p;
    const int ra = r0 + 4 * h + q, rb = ra + 8;
    return join_tr<T>(lds_read_tr4(img, trimg_off(ra, col >> 3) + (col & 7) * 2), lds_read_tr4(img, trimg_off(rb, col >> 3) + (col & 7) * 2));
}
// same, natural k order: element j of lane half h = image row r0 + 8*h + j
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 tr_frag_natural(const char* img, int r0, int c0, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int col = c0 + 16 * g1 + 4 * p;
    const int ra = r0 + 8 * h + q, rb = ra + 4;
    return join_tr<T>(lds_read_tr4(img, trimg_off(ra, col >> 3) + (col & 7) * 2), lds_read_tr4(img, trimg_off(rb, col >> 3) + (col & 7) * 2));
}

template <typename T>
__device__ __forceinline__ void stage_image(char* img, const T* src, int ld, int rows_valid, int rows_pad, bool tr_layout, int tid, int nthreads) {
    typedef typename Vec<T>::v8 v8;
    for (int idx = tid; idx < rows_pad * 8; idx += nthreads) {
        const int row = idx >> 3, ch = idx & 7;
        const int gr = min(row, rows_valid - 1);
        v8 v = *reinterpret_cast<const v8*>(src + (size_t)gr * ld + ch * 8);
        *reinterpret_cast<v8*>(img + (tr_layout ? trimg_off(row, ch) : rowimg_off(row, ch))) = v;
    }
}

// ------------------------------------------------------------------------------------------------ forward
// NW waves per (post, head); wave w takes query tiles w, w+NW, ...  Online soft-max over the key tiles keeps one 32x32
// score tile live at a time (~100 VGPRs -> 4 waves per SIMD) instead of all of them (490 VGPRs, 1 wave per SIMD).
// DROP (compile time, so that neither instance branches per element): 0 = no dropout, 1 = one hash per element, 2 = one hash per PAIR of
// consecutive keys (S even: the element index of an even key is even, mm_keep2)
template <typename T, int NKT, int NW, int DROP>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int SP = NKT * 32;
    char* Kimg = smem;                       // [SP][64] row image
    char* Vimg = smem + SP * 128;            // [SP][64] tr image
    float* mb = reinterpret_cast<float*>(smem + 2 * SP * 128);   // [SP] additive key bias * log2e
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = blockIdx.x, post = blockIdx.y;
    const int S = a.S;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // the post's queries and keys (cross attention: Sq != Sk)
    const int qr = a.q_rps > 0 ? a.q_rps : S, kr = a.kv_rps > 0 ? a.kv_rps : S, cr = a.ctx_rps > 0 ? a.ctx_rps : S;      // rows per post of Q | K, V | ctx
    const T* base = (const T*)a.qkv + (size_t)post * qr * a.ld_qkv + head * HD;                  // the post's query rows
    const T* kvb = (const T*)a.qkv + (size_t)post * kr * a.ld_qkv + a.hidden + head * HD;        // ... its key rows (values a.hidden columns on)
    const int nkt = min(NKT, (Sk + 31) / 32);          // key tiles that hold a key: the others are neither staged nor multiplied
    stage_image<T>(Kimg, kvb, a.ld_qkv, Sk, nkt * 32, false, tid, NW * 64);
    stage_image<T>(Vimg, kvb + a.hidden, a.ld_qkv, Sk, nkt * 32, true, tid, NW * 64);
    for (int k = tid; k < nkt * 32; k += NW * 64) {
        float b = (k < Sk) ? (a.maskbias ? a.maskbias[(size_t)post * S + k] : 0.f) : -INFINITY;
        mb[k] = b * LOG2E;
    }
    __syncthreads();
    const int r = lane & 31, h2 = lane >> 5;
    const float sc = a.scale * LOG2E;
    const int nqt = a.q_tiles > 0 ? min((Sq + 31) / 32, a.q_tiles) : (Sq + 31) / 32;
    for (int qt = w; qt < nqt; qt += NW) {
        const int q = qt * 32 + r;
        const int qrow = min(q, Sq - 1);
        const T* qp = base + (size_t)qrow * a.ld_qkv + 8 * h2;
        v8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const v8*>(qp + 16 * s);
        const uint32_t ebase = (uint32_t)(((size_t)post * a.heads + head) * S + (uint32_t)qrow) * (uint32_t)S;
        float m_run = -INFINITY, l_run = 0.f;
        f32x16 oacc[2] = {f32x16{}, f32x16{}};
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            // S^T tile: acc[reg] = score(key = kt*32 + (reg&3) + 8*(reg>>2) + 4*h2, query q)
            f32x16 acc = f32x16{};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v8 kf = lds_read8<T>(Kimg, rowimg_off(kt * 32 + r, 2 * s + h2));
                acc = mfma32(kf, qf[s], acc);
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b = *reinterpret_cast<const f32x4*>(mb + kt * 32 + 8 * g + 4 * h2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[4 * g + e] * sc + b[e];
                    acc[4 * g + e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;        // a fully masked tile contributes nothing
            const float alpha = fast_exp2(m_run - m_safe);                  // first tile: exp2(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(acc[e] - m_safe);
                acc[e] = p;
                psum += p;
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                v8 pf;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int reg = 8 * s2 + j;
                    float p0 = acc[reg], p1 = acc[reg + 1];
                    if (DROP) {
                        const int key = kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;      // reg even: keys key, key + 1
                        bool k0, k1;
                        if (DROP == 2) mm_keep2(ebase + (uint32_t)key, a.drop, k0, k1);
                        else { k0 = mm_keep(ebase + (uint32_t)key, a.drop); k1 = mm_keep(ebase + (uint32_t)key + 1u, a.drop); }
                        p0 = k0 ? p0 * a.drop.keep_scale : 0.f;
                        p1 = k1 ? p1 * a.drop.keep_scale : 0.f;
                    }
                    pf[j] = from_f<T>(p0);
                    pf[j + 1] = from_f<T>(p1);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    v8 vf = tr_frag_acc_order<T>(Vimg, kt * 32 + 16 * s2, dt * 32, lane);
                    oacc[dt] = mfma32(vf, pf, oacc[dt]);
                }
            }
        }
        if (a.lse && h2 == 0 && q < Sq) a.lse[((size_t)post * a.heads + head) * S + q] = (m_run + log2f(l_run)) * (1.0f / LOG2E);
        const float inv = 1.0f / l_run;
        // oacc[dt][reg] = O(query q, d = dt*32 + (reg&3) + 8*(reg>>2) + 4*h2), still to be divided by the soft-max sum
        if (q < Sq) {
            T* op = (T*)a.ctx + ((size_t)post * cr + q) * a.ld_ctx + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = from_f<T>(oacc[dt][4 * g + e] * inv);
                    *reinterpret_cast<v4*>(op + dt * 32 + 8 * g + 4 * h2) = o;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// LDS: Qtr, dOtr, Ktr images [SP][64] (tr layout), dS [64][SP] 16-bit row image, lse / D / maskbias floats.
__device__ __forceinline__ int ds_off(int row, int ch, int sp_chunks) {   // [64 rows][SP keys] 16-bit, 16-B chunks swizzled
    return row * (sp_chunks * 16) + ((ch ^ (row & (sp_chunks - 1) & 15)) << 4);
}

template <typename T, int NKT, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int SP = NKT * 32;
    constexpr int SPC = SP / 8;                 // 16-B chunks per dS row (4, 8, 16)
    char* Qtr = smem;
    char* dOtr = smem + SP * 128;
    char* Ktr = smem + 2 * SP * 128;
    char* dSimg = smem + 3 * SP * 128;          // 64 * SP * 2 bytes
    float* lse2 = reinterpret_cast<float*>(dSimg + 64 * SP * 2);   // lse * log2e
    float* Dv = lse2 + SP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = blockIdx.x, post = blockIdx.y;
    const int S = a.S;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // the post's queries and keys (cross attention: Sq != Sk)
    const size_t qrow0 = (size_t)post * (a.q_rps > 0 ? a.q_rps : S), krow0 = (size_t)post * (a.kv_rps > 0 ? a.kv_rps : S), crow0 = (size_t)post * (a.ctx_rps > 0 ? a.ctx_rps : S);
    const int nkt = min(NKT, (Sk + 31) / 32);          // key tiles that hold a key
    const T* qb = (const T*)a.qkv + qrow0 * a.ld_qkv + head * HD;
    const T* kb = (const T*)a.qkv + krow0 * a.ld_qkv + a.hidden + head * HD;
    const T* vb = kb + a.hidden;
    const T* dob = (const T*)a.dctx + crow0 * a.ld_ctx + head * HD;
    const T* ob = (const T*)a.ctx + crow0 * a.ld_ctx + head * HD;
    stage_image<T>(Qtr, qb, a.ld_qkv, Sq, SP, true, tid, 256);
    stage_image<T>(dOtr, dob, a.ld_ctx, Sq, SP, true, tid, 256);
    stage_image<T>(Ktr, kb, a.ld_qkv, Sk, SP, true, tid, 256);
    for (int q = tid; q < SP; q += 256) {
        float d = 0.f, l = 0.f;
        if (q < Sq) {
            const T* o = ob + (size_t)q * a.ld_ctx;
            const T* g = dob + (size_t)q * a.ld_ctx;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v8 ov = *reinterpret_cast<const v8*>(o + c * 8), gv = *reinterpret_cast<const v8*>(g + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += to_f<T>(ov[e]) * to_f<T>(gv[e]);
            }
            l = a.lse[((size_t)post * a.heads + head) * S + q] * LOG2E;
        }
        Dv[q] = d;
        lse2[q] = l;
    }
    const int r = lane & 31, h2 = lane >> 5;
    const bool has_keys = w < nkt;              // wave w owns keys 32w .. 32w+31 (a wave without a live key only takes part in the dQ products)
    const int key = w * 32 + r;
    const int krow = min(key, Sk - 1);
    v8 kf[4], vf[4];
    float mbk = 0.f;
    if (has_keys) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const v8*>(kb + (size_t)krow * a.ld_qkv + 16 * s + 8 * h2);
            vf[s] = *reinterpret_cast<const v8*>(vb + (size_t)krow * a.ld_qkv + 16 * s + 8 * h2);
        }
        mbk = (key < Sk) ? (a.maskbias ? a.maskbias[(size_t)post * S + key] : 0.f) : -INFINITY;
        mbk *= LOG2E;
    }
    __syncthreads();
    const float sc = a.scale * LOG2E;
    // dropout element index of (query q, this lane's key) = e_lane + q * S in 32-bit wrap-around arithmetic -- the forward's
    // (uint32)(((post * heads + head) * S + q)) * S + key.  No clamps: a query or key past S has p = 0 and its mask is never used.
    const uint32_t e_lane = (uint32_t)(((size_t)post * a.heads + head) * S) * (uint32_t)S + (uint32_t)key + (uint32_t)(4 * h2) * (uint32_t)S;
    f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};
    constexpr int NQT = NKT;
    int qlim = a.q_tiles > 0 ? min(NQT, a.q_tiles) : NQT;            // later tiles carry a zero d ctx: nothing to do
    qlim = min(qlim, (Sq + 31) / 32);                                // ... or hold no live query at all
    for (int pair = 0; pair < (qlim + 1) / 2; ++pair) {
        if (has_keys) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int qt = pair * 2 + t2;
                if (qt >= qlim) break;
                const int q0 = qt * 32;
                const uint32_t e_tile = e_lane + (uint32_t)q0 * (uint32_t)S;      // + (8 g + e) * S per element: wave-uniform addends
                f32x16 sacc = f32x16{}, pacc = f32x16{};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // row reads of the already-staged Q / dO images (one image serves the row reads here and the transposed reads of
                    // dV / dK below): the fragments used to come from global memory again, four waves re-reading every query row with
                    // the load latency exposed in front of each tile's first MFMA.  Image rows past S hold row S-1 (stage_image).
                    v8 qf = lds_read8<T>(Qtr, trimg_off(q0 + r, 2 * s + h2));
                    v8 gf = lds_read8<T>(dOtr, trimg_off(q0 + r, 2 * s + h2));
                    sacc = mfma32(qf, kf[s], sacc);      // S[q][key]
                    pacc = mfma32(gf, vf[s], pacc);      // dP[q][key]
                }
                // element reg <-> query q0 + (reg&3) + 8*(reg>>2) + 4*h2, key = this lane's key
                v8 pfrag[2], dsfrag[2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + q0 + 8 * g + 4 * h2);
                    f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + q0 + 8 * g + 4 * h2);
                    uint32_t e_el = e_tile + (uint32_t)g * (8u * (uint32_t)S);
#pragma unroll
                    for (int e = 0; e < 4; ++e, e_el += (uint32_t)S) {
                        const int reg = 4 * g + e;
                        const int q = q0 + 8 * g + 4 * h2 + e;
                        float p = fast_exp2(sacc[reg] * sc + mbk - l4[e]);
                        if (q >= Sq) p = 0.f;
                        float pd = p, dpd = pacc[reg];
                        if (DROP) {
                            const bool kp = mm_keep(e_el, a.drop);
                            pd = kp ? p * a.drop.keep_scale : 0.f;
                            dpd = kp ? dpd * a.drop.keep_scale : 0.f;
                        }
                        const float ds = p * (dpd - d4[e]) * a.scale;
                        pfrag[reg >> 3][reg & 7] = from_f<T>(pd);
                        dsfrag[reg >> 3][reg & 7] = from_f<T>(ds);
                        // dS row image for dQ: row = t2*32 + (q - q0), column = key
                        const int drow = t2 * 32 + 8 * g + 4 * h2 + e;
                        *reinterpret_cast<T*>(dSimg + ds_off(drow, key >> 3, SPC) + (key & 7) * 2) = from_f<T>(ds);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        v8 gt = tr_frag_acc_order<T>(dOtr, q0 + 16 * s2, dt * 32, lane);   // dO^T[d][q]
                        dv[dt] = mfma32(gt, pfrag[s2], dv[dt]);
                        v8 qtf = tr_frag_acc_order<T>(Qtr, q0 + 16 * s2, dt * 32, lane);   // Q^T[d][q]
                        dk[dt] = mfma32(qtf, dsfrag[s2], dk[dt]);
                    }
            }
        }
        __syncthreads();
        {   // dQ tile (query tile pair*2 + (w>>1), d tile w&1) = dS[q][:] . K[:, d]
            const int qi = w >> 1, dt = w & 1;
            const int qt = pair * 2 + qi;
            if (qt < qlim) {
                f32x16 dq = f32x16{};
#pragma unroll
                for (int ks = 0; ks < SP / 16; ++ks) {
                    if (ks >= nkt * 2) break;          // columns of dS past the last live key tile were never written
                    v8 dsf = lds_read8<T>(dSimg, ds_off(qi * 32 + r, 2 * ks + h2, SPC));
                    v8 ktf = tr_frag_natural<T>(Ktr, ks * 16, dt * 32, lane);
                    dq = mfma32(ktf, dsf, dq);          // dQ^T = K^T . dS^T: the same two fragments with the roles swapped, so the lane owns a query row
                }
                // dq[reg] = dQ(query qt*32 + r, d = dt*32 + (reg&3) + 8*(reg>>2) + 4*h2): four consecutive d per register quad -> 8-byte stores
                const int q = qt * 32 + r;
                if (q < Sq) {
                    T* dqp = (T*)a.dqkv + (qrow0 + q) * a.ld_qkv + head * HD + dt * 32 + 4 * h2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(dq[4 * g + e]);
                        *reinterpret_cast<v4*>(dqp + 8 * g) = o;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (has_keys && key < Sk) {
        // dk[dt][reg] = dK(key, d = dt*32 + (reg&3) + 8*(reg>>2) + 4*h2)
        T* dkp = (T*)a.dqkv + (krow0 + key) * a.ld_qkv + a.hidden + head * HD;
        T* dvp = dkp + a.hidden;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o1[e] = from_f<T>(dk[dt][4 * g + e]); o2[e] = from_f<T>(dv[dt][4 * g + e]); }
                *reinterpret_cast<v4*>(dkp + dt * 32 + 8 * g + 4 * h2) = o1;
                *reinterpret_cast<v4*>(dvp + dt * 32 + 8 * g + 4 * h2) = o2;
            }
    }
}


// ------------------------------------------------------------------------------------------------ parity mode (bf16x3)
// The same two kernels on fp32 tensors: every MFMA operand is the pair hi = bf16(x), lo = bf16(x - hi) and every product the three MFMAs
// lo.hi + hi.lo + hi.hi (mma3_32), fp32 everywhere else.  The K / V (backward: Q / dO / K) images are split while they are staged, so LDS
// holds two 16-bit images per tensor; P and dS are split in registers where the 16-bit kernels convert them.  (Round 2 ran this mode's
// attention on the vector ALUs, one query per thread, one wave per SIMD: 26 ms of a 71 ms step.)
__device__ __forceinline__ void stage_image_x3(char* hi, char* lo, const float* src, int ld, int rows_valid, int rows_pad, bool tr_layout, int tid, int nthreads) {
    for (int idx = tid; idx < rows_pad * 8; idx += nthreads) {
        const int row = idx >> 3, ch = idx & 7;
        const int gr = min(row, rows_valid - 1);
        float v[8];
        load8(src + (size_t)gr * ld + ch * 8, v);
        const Frag3 f = split8(v);
        const int off = tr_layout ? trimg_off(row, ch) : rowimg_off(row, ch);
        *reinterpret_cast<bf16x8*>(hi + off) = f.hi;
        *reinterpret_cast<bf16x8*>(lo + off) = f.lo;
    }
}
// the same images from a plane pair (mmhip_kernels.h): the planes are copied as they are
__device__ __forceinline__ void stage_image_pp(char* hi, char* lo, const bf16_t* src, int ld, int lo_off, int rows_valid, int rows_pad, bool tr_layout, int tid, int nthreads) {
    for (int idx = tid; idx < rows_pad * 8; idx += nthreads) {
        const int row = idx >> 3, ch = idx & 7;
        const int gr = min(row, rows_valid - 1);
        const bf16_t* p = src + (size_t)gr * ld + ch * 8;
        const int off = tr_layout ? trimg_off(row, ch) : rowimg_off(row, ch);
        *reinterpret_cast<bf16x8*>(hi + off) = *reinterpret_cast<const bf16x8*>(p);
        *reinterpret_cast<bf16x8*>(lo + off) = *reinterpret_cast<const bf16x8*>(p + lo_off);
    }
}
// X = fp32 tensor or plane pair: image staging, 8-element fragments, 4-element values and stores through one interface
template <bool PAIR> struct X3IO;
template <> struct X3IO<false> {
    typedef float elem;
    static __device__ __forceinline__ void stage(char* hi, char* lo, const elem* src, int ld, int, int rv, int rp, bool tr, int tid, int nt) { stage_image_x3(hi, lo, src, ld, rv, rp, tr, tid, nt); }
    static __device__ __forceinline__ Frag3 frag(const elem* p, int) { float v[8]; load8(p, v); return split8(v); }
    static __device__ __forceinline__ f32x4 load4(const elem* p, int) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store4(elem* p, int, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ void store1(elem* p, int, float v) { *p = v; }
};
template <> struct X3IO<true> {
    typedef bf16_t elem;
    static __device__ __forceinline__ void stage(char* hi, char* lo, const elem* src, int ld, int lo_off, int rv, int rp, bool tr, int tid, int nt) { stage_image_pp(hi, lo, src, ld, lo_off, rv, rp, tr, tid, nt); }
    static __device__ __forceinline__ Frag3 frag(const elem* p, int lo_off) {
        Frag3 f;
        f.hi = *reinterpret_cast<const bf16x8*>(p);
        f.lo = *reinterpret_cast<const bf16x8*>(p + lo_off);
        return f;
    }
    static __device__ __forceinline__ f32x4 load4(const elem* p, int lo_off) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(p), l = *reinterpret_cast<const bf16x4*>(p + lo_off);
        return f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
    }
    static __device__ __forceinline__ void store4(elem* p, int lo_off, f32x4 v) {
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)v[e]; l[e] = (bf16_t)(v[e] - (float)h[e]); }
        *reinterpret_cast<bf16x4*>(p) = h;
        *reinterpret_cast<bf16x4*>(p + lo_off) = l;
    }
    static __device__ __forceinline__ void store1(elem* p, int lo_off, float v) { const bf16_t h = (bf16_t)v; p[0] = h; p[lo_off] = (bf16_t)(v - (float)h); }
};
__device__ __forceinline__ Frag3 lds_pair(const char* hi, const char* lo, int off) {
    Frag3 f;
    f.hi = lds_read8<bf16_t>(hi, off);
    f.lo = lds_read8<bf16_t>(lo, off);
    return f;
}
__device__ __forceinline__ Frag3 global_pair(const float* p) {
    float v[8];
    load8(p, v);
    return split8(v);
}

template <int NKT, int NW, bool PAIR>
__global__ __launch_bounds__(NW * 64) void attn_fwd_x3_kernel(AttnArgs a) {
    typedef X3IO<PAIR> IO;
    typedef typename IO::elem E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 32, IMG = SP * 128;
    char* Kh = smem;                         // [SP][64] row images of K (hi, lo), tr images of V (hi, lo)
    char* Kl = smem + IMG;
    char* Vh = smem + 2 * IMG;
    char* Vl = smem + 3 * IMG;
    float* mb = reinterpret_cast<float*>(smem + 4 * IMG);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = blockIdx.x, post = blockIdx.y;
    const int S = a.S;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // as in attn_fwd_kernel: queries / keys of a post, rows per post, live key tiles
    const int qr = a.q_rps > 0 ? a.q_rps : S, kr = a.kv_rps > 0 ? a.kv_rps : S, cr = a.ctx_rps > 0 ? a.ctx_rps : S;
    const E* base = (const E*)a.qkv + (size_t)post * qr * a.ld_qkv + head * HD;
    const E* kvb = (const E*)a.qkv + (size_t)post * kr * a.ld_qkv + a.hidden + head * HD;
    const int nkt = min(NKT, (Sk + 31) / 32);
    IO::stage(Kh, Kl, kvb, a.ld_qkv, a.lo_qkv, Sk, nkt * 32, false, tid, NW * 64);
    IO::stage(Vh, Vl, kvb + a.hidden, a.ld_qkv, a.lo_qkv, Sk, nkt * 32, true, tid, NW * 64);
    for (int k = tid; k < nkt * 32; k += NW * 64) {
        float b = (k < Sk) ? (a.maskbias ? a.maskbias[(size_t)post * S + k] : 0.f) : -INFINITY;
        mb[k] = b * LOG2E;
    }
    __syncthreads();
    const int r = lane & 31, h2 = lane >> 5;
    const float sc = a.scale * LOG2E;
    const int nqt = a.q_tiles > 0 ? min((Sq + 31) / 32, a.q_tiles) : (Sq + 31) / 32;
    const bool dropping = a.drop.thresh16 != 0;
    for (int qt = w; qt < nqt; qt += NW) {
        const int q = qt * 32 + r;
        const int qrow = min(q, Sq - 1);
        const E* qp = base + (size_t)qrow * a.ld_qkv + 8 * h2;
        Frag3 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = IO::frag(qp + 16 * s, a.lo_qkv);
        const uint32_t ebase = (uint32_t)(((size_t)post * a.heads + head) * S + (uint32_t)qrow) * (uint32_t)S;
        float m_run = -INFINITY, l_run = 0.f;
        f32x16 oacc[2] = {f32x16{}, f32x16{}};
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 acc = f32x16{};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mma3_32(lds_pair(Kh, Kl, rowimg_off(kt * 32 + r, 2 * s + h2)), qf[s], acc);
            float tmax = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b = *reinterpret_cast<const f32x4*>(mb + kt * 32 + 8 * g + 4 * h2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[4 * g + e] * sc + b[e];
                    acc[4 * g + e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_safe);
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(acc[e] - m_safe);
                acc[e] = p;
                psum += p;
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int reg = 8 * s2 + j;
                    float p = acc[reg];
                    if (dropping) {
                        const int key = kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                        p = mm_keep(ebase + (uint32_t)key, a.drop) ? p * a.drop.keep_scale : 0.f;
                    }
                    pv[j] = p;
                }
                const Frag3 pf = split8(pv);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    Frag3 vf;
                    vf.hi = tr_frag_acc_order<bf16_t>(Vh, kt * 32 + 16 * s2, dt * 32, lane);
                    vf.lo = tr_frag_acc_order<bf16_t>(Vl, kt * 32 + 16 * s2, dt * 32, lane);
                    oacc[dt] = mma3_32(vf, pf, oacc[dt]);
                }
            }
        }
        if (a.lse && h2 == 0 && q < Sq) a.lse[((size_t)post * a.heads + head) * S + q] = (m_run + log2f(l_run)) * (1.0f / LOG2E);
        const float inv = 1.0f / l_run;
        if (q < Sq) {
            E* op = (E*)a.ctx + ((size_t)post * cr + q) * a.ld_ctx + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = oacc[dt][4 * g + e] * inv;
                    IO::store4(op + dt * 32 + 8 * g + 4 * h2, a.lo_ctx, o);
                }
        }
    }
}

// Parity-mode forward for sequences whose split K / V images do not fit the CU's LDS at once (S > 288: the 577 image tokens of
// CLIP-ViT-L/14 @336 need 4 x 608 x 128 B = 311 KB).  The keys are walked in CHUNKS of CKT x 32 (288 keys = 147 KB of images): a chunk is
// staged once and every wave runs all its query tiles over it before the next chunk replaces it; the online soft-max state of a wave's (at
// most QPW = 3) query tiles -- running maximum, running sum, 32 x 64 output accumulator -- stays in registers across the chunks, so the
// result is the same online soft-max as attn_fwd_x3_kernel's, merely with two more workgroup barriers per chunk.  S <= QPW * 8 * 32 = 768.
template <bool PAIR>
__global__ __launch_bounds__(512) void attn_fwd_x3_long_kernel(AttnArgs a) {
    typedef X3IO<PAIR> IO;
    typedef typename IO::elem E;
    constexpr int NW = 8, CKT = 9, QPW = 3, IMG = CKT * 32 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kh = smem;
    char* Kl = smem + IMG;
    char* Vh = smem + 2 * IMG;
    char* Vl = smem + 3 * IMG;
    float* mb = reinterpret_cast<float*>(smem + 4 * IMG);          // [QPW * NW * 32] additive key bias * log2e of ALL keys
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = blockIdx.x, post = blockIdx.y;
    const int S = a.S, nkt = (S + 31) / 32;
    const E* base = (const E*)a.qkv + (size_t)post * S * a.ld_qkv + head * HD;
    for (int k = tid; k < nkt * 32; k += NW * 64) {
        float b = (k < S) ? (a.maskbias ? a.maskbias[(size_t)post * S + k] : 0.f) : -INFINITY;
        mb[k] = b * LOG2E;
    }
    const int r = lane & 31, h2 = lane >> 5;
    const float sc = a.scale * LOG2E;
    const int nqt = a.q_tiles > 0 ? min(nkt, a.q_tiles) : nkt;
    const bool dropping = a.drop.thresh16 != 0;
    float m_run[QPW], l_run[QPW];
    f32x16 oacc[QPW][2];
#pragma unroll
    for (int j = 0; j < QPW; ++j) { m_run[j] = -INFINITY; l_run[j] = 0.f; oacc[j][0] = f32x16{}; oacc[j][1] = f32x16{}; }
    for (int c0 = 0; c0 < nkt; c0 += CKT) {
        const int ckt = min(CKT, nkt - c0), krow0 = c0 * 32;
        __syncthreads();                                   // every wave is done with the previous chunk's images
        IO::stage(Kh, Kl, base + a.hidden + (size_t)krow0 * a.ld_qkv, a.ld_qkv, a.lo_qkv, S - krow0, ckt * 32, false, tid, NW * 64);
        IO::stage(Vh, Vl, base + 2 * a.hidden + (size_t)krow0 * a.ld_qkv, a.ld_qkv, a.lo_qkv, S - krow0, ckt * 32, true, tid, NW * 64);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            const int qt = w + j * NW;
            if (qt >= nqt) continue;
            const int qrow = min(qt * 32 + r, S - 1);
            const E* qp = base + (size_t)qrow * a.ld_qkv + 8 * h2;
            Frag3 qf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = IO::frag(qp + 16 * s, a.lo_qkv);
            const uint32_t ebase = (uint32_t)(((size_t)post * a.heads + head) * S + (uint32_t)qrow) * (uint32_t)S;
#pragma unroll 1
            for (int kt = 0; kt < ckt; ++kt) {
                f32x16 acc = f32x16{};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mma3_32(lds_pair(Kh, Kl, rowimg_off(kt * 32 + r, 2 * s + h2)), qf[s], acc);
                float tmax = -INFINITY;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(mb + krow0 + kt * 32 + 8 * g + 4 * h2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[4 * g + e] * sc + b[e];
                        acc[4 * g + e] = v;
                        tmax = fmaxf(tmax, v);
                    }
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(m_run[j], tmax);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2(m_run[j] - m_safe);
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = fast_exp2(acc[e] - m_safe);
                    acc[e] = p;
                    psum += p;
                }
                psum += __shfl_xor(psum, 32);
                l_run[j] = l_run[j] * alpha + psum;
                m_run[j] = m_new;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[j][dt][e] *= alpha;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float pv[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int reg = 8 * s2 + jj;
                        float p = acc[reg];
                        if (dropping) {
                            const int key = krow0 + kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                            p = mm_keep(ebase + (uint32_t)key, a.drop) ? p * a.drop.keep_scale : 0.f;
                        }
                        pv[jj] = p;
                    }
                    const Frag3 pf = split8(pv);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        Frag3 vf;
                        vf.hi = tr_frag_acc_order<bf16_t>(Vh, kt * 32 + 16 * s2, dt * 32, lane);
                        vf.lo = tr_frag_acc_order<bf16_t>(Vl, kt * 32 + 16 * s2, dt * 32, lane);
                        oacc[j][dt] = mma3_32(vf, pf, oacc[j][dt]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int qt = w + j * NW, q = qt * 32 + r;
        if (qt >= nqt || q >= S) continue;
        if (a.lse && h2 == 0) a.lse[((size_t)post * a.heads + head) * S + q] = (m_run[j] + log2f(l_run[j])) * (1.0f / LOG2E);
        const float inv = 1.0f / l_run[j];
        E* op = (E*)a.ctx + ((size_t)post * S + q) * a.ld_ctx + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = oacc[j][dt][4 * g + e] * inv;
                IO::store4(op + dt * 32 + 8 * g + 4 * h2, a.lo_ctx, o);
            }
    }
}

// NP1 (plane pairs only; AttnBwdArgs::nprod = 1, the engine's one-product backward): the scores S = Q.K^T keep their three products -- the
// probabilities are RE-computed against the forward's log-sum-exp, and scores off by one bf16 rounding (2e-3) would leave every row of P summing to
// something else than 1 (measured: the all-one-product kernel took the worst gradient tensor of a 12-layer model from 6e-3 to 1.4e-2) -- while
// dP = dO.V^T, dV = P^T.dO, dK = dS^T.Q and dQ = dS.K take ONE product of the hi planes, like the backward's GEMMs; d ctx is read from its hi plane only
// (its producer does not write the other under that policy).
template <int NKT, bool PAIR, bool NP1 = false>
__global__ __launch_bounds__(256) void attn_bwd_x3_kernel(AttnBwdArgs a) {
    static_assert(PAIR || !NP1, "the one-product form reads plane pairs");
    typedef X3IO<PAIR> IO;
    typedef typename IO::elem E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 32, IMG = SP * 128, DSI = 64 * SP * 2;
    constexpr int SPC = SP / 8;
    char* Qh = smem;
    char* Ql = smem + IMG;
    char* dOh = smem + 2 * IMG;
    char* dOl = smem + 3 * IMG;
    char* Kh = smem + 4 * IMG;
    char* Kl = smem + 5 * IMG;
    char* dSh = smem + 6 * IMG;
    char* dSl = dSh + DSI;
    float* lse2 = reinterpret_cast<float*>(dSl + DSI);
    float* Dv = lse2 + SP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = blockIdx.x, post = blockIdx.y;
    const int S = a.S;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // as in attn_bwd_kernel
    const size_t qrow0 = (size_t)post * (a.q_rps > 0 ? a.q_rps : S), krow0 = (size_t)post * (a.kv_rps > 0 ? a.kv_rps : S), crow0 = (size_t)post * (a.ctx_rps > 0 ? a.ctx_rps : S);
    const int nkt = min(NKT, (Sk + 31) / 32);
    const E* qb = (const E*)a.qkv + qrow0 * a.ld_qkv + head * HD;
    const E* kb = (const E*)a.qkv + krow0 * a.ld_qkv + a.hidden + head * HD;
    const E* vb = kb + a.hidden;
    const E* dob = (const E*)a.dctx + crow0 * a.ld_ctx + head * HD;
    const E* ob = (const E*)a.ctx + crow0 * a.ld_ctx + head * HD;
    IO::stage(Qh, Ql, qb, a.ld_qkv, a.lo_qkv, Sq, SP, true, tid, 256);
    if constexpr (NP1) stage_image<bf16_t>(dOh, (const bf16_t*)dob, a.ld_ctx, Sq, SP, true, tid, 256);
    else IO::stage(dOh, dOl, dob, a.ld_ctx, a.lo_ctx, Sq, SP, true, tid, 256);
    IO::stage(Kh, Kl, kb, a.ld_qkv, a.lo_qkv, Sk, SP, true, tid, 256);
    for (int q = tid; q < SP; q += 256) {
        float d = 0.f, l = 0.f;
        if (q < Sq) {
            const E* o = ob + (size_t)q * a.ld_ctx;
            const E* g = dob + (size_t)q * a.ld_ctx;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 ov = IO::load4(o + c * 4, a.lo_ctx);
                f32x4 gv;
                if constexpr (NP1) { const bf16x4 h = *reinterpret_cast<const bf16x4*>((const bf16_t*)g + c * 4); gv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
                else gv = IO::load4(g + c * 4, a.lo_ctx);
#pragma unroll
                for (int e = 0; e < 4; ++e) d += ov[e] * gv[e];
            }
            l = a.lse[((size_t)post * a.heads + head) * S + q] * LOG2E;
        }
        Dv[q] = d;
        lse2[q] = l;
    }
    const int r = lane & 31, h2 = lane >> 5;
    const bool has_keys = w < nkt;
    const int key = w * 32 + r;
    const int krow = min(key, Sk - 1);
    Frag3 kf[4], vf[4];
    float mbk = 0.f;
    if (has_keys) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = IO::frag(kb + (size_t)krow * a.ld_qkv + 16 * s + 8 * h2, a.lo_qkv);
            vf[s] = IO::frag(vb + (size_t)krow * a.ld_qkv + 16 * s + 8 * h2, a.lo_qkv);
        }
        mbk = (key < Sk) ? (a.maskbias ? a.maskbias[(size_t)post * S + key] : 0.f) : -INFINITY;
        mbk *= LOG2E;
    }
    __syncthreads();
    const float sc = a.scale * LOG2E;
    const bool dropping = a.drop.thresh16 != 0;
    f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};
    constexpr int NQT = NKT;
    const int qlim = min(a.q_tiles > 0 ? min(NQT, a.q_tiles) : NQT, (Sq + 31) / 32);
    for (int pair = 0; pair < (qlim + 1) / 2; ++pair) {
        if (has_keys) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int qt = pair * 2 + t2;
                if (qt >= qlim) break;
                const int q0 = qt * 32;
                f32x16 sacc = f32x16{}, pacc = f32x16{};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int off = trimg_off(q0 + r, 2 * s + h2);
                    sacc = mma3_32(lds_pair(Qh, Ql, off), kf[s], sacc);       // S[q][key]
                    if constexpr (NP1) pacc = mfma32(lds_read8<bf16_t>(dOh, off), vf[s].hi, pacc);
                    else pacc = mma3_32(lds_pair(dOh, dOl, off), vf[s], pacc);     // dP[q][key]
                }
                float pdv[16], dsv[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + q0 + 8 * g + 4 * h2);
                    f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + q0 + 8 * g + 4 * h2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * g + e;
                        const int q = q0 + 8 * g + 4 * h2 + e;
                        float p = fast_exp2(sacc[reg] * sc + mbk - l4[e]);
                        if (q >= Sq) p = 0.f;
                        float pd = p, dpd = pacc[reg];
                        if (dropping) {
                            const uint32_t eidx = (uint32_t)(((size_t)post * a.heads + head) * S + (uint32_t)min(q, S - 1)) * (uint32_t)S + (uint32_t)min(key, S - 1);
                            const bool kp = mm_keep(eidx, a.drop);
                            pd = kp ? p * a.drop.keep_scale : 0.f;
                            dpd = kp ? dpd * a.drop.keep_scale : 0.f;
                        }
                        const float ds = p * (dpd - d4[e]) * a.scale;
                        pdv[reg] = pd;
                        dsv[reg] = ds;
                        const int drow = t2 * 32 + 8 * g + 4 * h2 + e;
                        const int doff = ds_off(drow, key >> 3, SPC) + (key & 7) * 2;
                        const bf16_t dh = (bf16_t)ds;
                        *reinterpret_cast<bf16_t*>(dSh + doff) = dh;
                        if constexpr (!NP1) *reinterpret_cast<bf16_t*>(dSl + doff) = (bf16_t)(ds - (float)dh);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const Frag3 pfrag = split8(pdv + 8 * s2), dsfrag = split8(dsv + 8 * s2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        Frag3 gt, qtf;
                        gt.hi = tr_frag_acc_order<bf16_t>(dOh, q0 + 16 * s2, dt * 32, lane);      // dO^T[d][q]
                        qtf.hi = tr_frag_acc_order<bf16_t>(Qh, q0 + 16 * s2, dt * 32, lane);      // Q^T[d][q]
                        if constexpr (NP1) {
                            dv[dt] = mfma32(gt.hi, pfrag.hi, dv[dt]);
                            dk[dt] = mfma32(qtf.hi, dsfrag.hi, dk[dt]);
                        } else {
                            gt.lo = tr_frag_acc_order<bf16_t>(dOl, q0 + 16 * s2, dt * 32, lane);
                            dv[dt] = mma3_32(gt, pfrag, dv[dt]);
                            qtf.lo = tr_frag_acc_order<bf16_t>(Ql, q0 + 16 * s2, dt * 32, lane);
                            dk[dt] = mma3_32(qtf, dsfrag, dk[dt]);
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            const int qi = w >> 1, dt = w & 1;
            const int qt = pair * 2 + qi;
            if (qt < qlim) {
                f32x16 dq = f32x16{};
#pragma unroll
                for (int ks = 0; ks < SP / 16; ++ks) {
                    if (ks >= nkt * 2) break;          // columns of dS past the last live key tile were never written
                    Frag3 ktf;
                    ktf.hi = tr_frag_natural<bf16_t>(Kh, ks * 16, dt * 32, lane);
                    if constexpr (NP1) {
                        dq = mfma32(lds_read8<bf16_t>(dSh, ds_off(qi * 32 + r, 2 * ks + h2, SPC)), ktf.hi, dq);
                    } else {
                        ktf.lo = tr_frag_natural<bf16_t>(Kl, ks * 16, dt * 32, lane);
                        dq = mma3_32(lds_pair(dSh, dSl, ds_off(qi * 32 + r, 2 * ks + h2, SPC)), ktf, dq);
                    }
                }
                E* dqp = (E*)a.dqkv + qrow0 * a.ld_qkv + head * HD + dt * 32 + r;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int q = qt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                    if (q < Sq) IO::store1(dqp + (size_t)q * a.ld_qkv, a.lo_qkv, dq[reg]);
                }
            }
        }
        __syncthreads();
    }
    if (has_keys && key < Sk) {
        E* dkp = (E*)a.dqkv + (krow0 + key) * a.ld_qkv + a.hidden + head * HD;
        E* dvp = dkp + a.hidden;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o1[e] = dk[dt][4 * g + e]; o2[e] = dv[dt][4 * g + e]; }
                IO::store4(dkp + dt * 32 + 8 * g + 4 * h2, a.lo_qkv, o1);
                IO::store4(dvp + dt * 32 + 8 * g + 4 * h2, a.lo_qkv, o2);
            }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// rows per post must hold the post's live rows
static bool attn_rows_ok(int S, int Sq_live, int Sk_live, int q_rps, int kv_rps, int ctx_rps) {
    const int Sq = Sq_live > 0 ? Sq_live : S, Sk = Sk_live > 0 ? Sk_live : S;
    return q_rps >= 0 && kv_rps >= 0 && ctx_rps >= 0 && (q_rps == 0 || q_rps >= Sq) && (kv_rps == 0 || kv_rps >= Sk) && (ctx_rps == 0 || ctx_rps >= Sq);
}
template <typename T, int NKT, int NW, int DROP>
static void launch_fwd_td(const AttnArgs& a, hipStream_t s) {
    const int lds = 2 * NKT * 32 * 128 + NKT * 32 * 4;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NKT, NW, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((attn_fwd_kernel<T, NKT, NW, DROP>), dim3(a.heads, a.posts), dim3(NW * 64), lds, s, a);
}
template <typename T, int NKT, int NW>
static void launch_fwd_t(const AttnArgs& a, hipStream_t s) {
    if (!a.drop.thresh16) launch_fwd_td<T, NKT, NW, 0>(a, s);
    else if (a.S & 1) launch_fwd_td<T, NKT, NW, 1>(a, s);
    else launch_fwd_td<T, NKT, NW, 2>(a, s);
}
template <typename T>
static hipError_t launch_fwd_d(const AttnArgs& a, hipStream_t s) {
    if (a.S <= 32) launch_fwd_t<T, 1, 1>(a, s);
    else if (a.S <= 64) launch_fwd_t<T, 2, 2>(a, s);
    else if (a.S <= 128) launch_fwd_t<T, 4, 4>(a, s);
    else if (a.S <= 224) launch_fwd_t<T, 7, 8>(a, s);
    else if (a.S <= 288) launch_fwd_t<T, 9, 8>(a, s);       // CLIP-ViT-L/14 @224: 257 tokens
    else if (a.S <= 608) launch_fwd_t<T, 19, 8>(a, s);      // @336: 577 tokens; K + V of a head = 152 KB of the CU's 160 KB LDS
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template <int NKT, int NW, bool PAIR>
static void launch_fwd_x3_tp(const AttnArgs& a, hipStream_t s) {
    const int lds = 4 * NKT * 32 * 128 + NKT * 32 * 4;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn_fwd_x3_kernel<NKT, NW, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((attn_fwd_x3_kernel<NKT, NW, PAIR>), dim3(a.heads, a.posts), dim3(NW * 64), lds, s, a);
}
template <int NKT, int NW>
static void launch_fwd_x3_t(const AttnArgs& a, hipStream_t s) {
    if (a.pair) launch_fwd_x3_tp<NKT, NW, true>(a, s);
    else launch_fwd_x3_tp<NKT, NW, false>(a, s);
}
static bool x3_mfma_attention() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MMHIP_X3_FAST"); on = e ? atoi(e) : 1; }
    return on != 0;
}
hipError_t launch_attn_fwd(const AttnArgs& a, int dtype, hipStream_t s) {
    if (a.hidden != a.heads * HD || a.ld_qkv % 8 || a.ld_ctx % 8 || a.S < 1 || a.Sq_live > a.S || a.Sk_live > a.S) return hipErrorInvalidValue;
    if (!attn_rows_ok(a.S, a.Sq_live, a.Sk_live, a.q_rps, a.kv_rps, a.ctx_rps)) return hipErrorInvalidValue;
    const bool cross = a.Sq_live > 0 || a.Sk_live > 0 || a.q_rps > 0 || a.kv_rps > 0 || a.ctx_rps > 0;      // live counts / compact rows: the MFMA kernels only
    if (dtype == DT_F32) {
        // parity mode: split operands on the matrix cores; two 16-bit images per tensor fit the CU's LDS up to S = 288 (148 KB), longer
        // sequences (up to 768) walk the keys in LDS-sized chunks (attn_fwd_x3_long_kernel); MMHIP_X3_FAST=0: fp32 on the vector ALUs
        if (!x3_mfma_attention() || ((uintptr_t)a.qkv & 15) || ((uintptr_t)a.ctx & 15) || a.S > 768) {
            if (a.pair) return hipErrorInvalidValue;          // the fp32 ALU kernels read fp32 tensors only
            return launch_attn_fwd_f32(a, s);
        }
        if (a.S > 288) {
            if (cross) return hipErrorInvalidValue;          // (the chunked kernel of the 577-token image tower: self-attention only)
            const int lds = 4 * 9 * 32 * 128 + 768 * 4;
            static bool done[2] = {false, false};
            if (a.pair) {
                if (!done[1]) { (void)hipFuncSetAttribute((const void*)attn_fwd_x3_long_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done[1] = true; }
                hipLaunchKernelGGL((attn_fwd_x3_long_kernel<true>), dim3(a.heads, a.posts), dim3(512), lds, s, a);
            } else {
                if (!done[0]) { (void)hipFuncSetAttribute((const void*)attn_fwd_x3_long_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done[0] = true; }
                hipLaunchKernelGGL((attn_fwd_x3_long_kernel<false>), dim3(a.heads, a.posts), dim3(512), lds, s, a);
            }
            return hipGetLastError();
        }
        if (a.S <= 32) launch_fwd_x3_t<1, 1>(a, s);
        else if (a.S <= 64) launch_fwd_x3_t<2, 2>(a, s);
        else if (a.S <= 128) launch_fwd_x3_t<4, 4>(a, s);
        else if (a.S <= 224) launch_fwd_x3_t<7, 8>(a, s);
        else launch_fwd_x3_t<9, 8>(a, s);
        return hipGetLastError();
    }
    return dtype == DT_BF16 ? launch_fwd_d<bf16_t>(a, s) : launch_fwd_d<f16_t>(a, s);
}
template <typename T, int NKT, bool DROP>
static void launch_bwd_td(const AttnBwdArgs& a, hipStream_t s) {
    const int lds = 3 * NKT * 32 * 128 + 64 * NKT * 32 * 2 + 2 * NKT * 32 * 4;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, NKT, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((attn_bwd_kernel<T, NKT, DROP>), dim3(a.heads, a.posts), dim3(256), lds, s, a);
}
template <typename T, int NKT>
static void launch_bwd_t(const AttnBwdArgs& a, hipStream_t s) {
    if (a.drop.thresh16) launch_bwd_td<T, NKT, true>(a, s);
    else launch_bwd_td<T, NKT, false>(a, s);
}
template <typename T>
static hipError_t launch_bwd_d(const AttnBwdArgs& a, hipStream_t s) {
    if (a.S <= 32) launch_bwd_t<T, 1>(a, s);
    else if (a.S <= 64) launch_bwd_t<T, 2>(a, s);
    else if (a.S <= 128) launch_bwd_t<T, 4>(a, s);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template <int NKT, bool PAIR, bool NP1 = false>
static void launch_bwd_x3_tp(const AttnBwdArgs& a, hipStream_t s) {
    const int lds = 6 * NKT * 32 * 128 + 2 * 64 * NKT * 32 * 2 + 2 * NKT * 32 * 4;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn_bwd_x3_kernel<NKT, PAIR, NP1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((attn_bwd_x3_kernel<NKT, PAIR, NP1>), dim3(a.heads, a.posts), dim3(256), lds, s, a);
}
template <int NKT>
static void launch_bwd_x3_t(const AttnBwdArgs& a, hipStream_t s) {
    if (a.pair && a.nprod == 1) launch_bwd_x3_tp<NKT, true, true>(a, s);
    else if (a.pair) launch_bwd_x3_tp<NKT, true>(a, s);
    else launch_bwd_x3_tp<NKT, false>(a, s);
}
hipError_t launch_attn_bwd(const AttnBwdArgs& a, int dtype, hipStream_t s) {
    if (a.hidden != a.heads * HD || a.ld_qkv % 8 || a.ld_ctx % 8 || a.S < 1 || a.Sq_live > a.S || a.Sk_live > a.S) return hipErrorInvalidValue;
    if (!attn_rows_ok(a.S, a.Sq_live, a.Sk_live, a.q_rps, a.kv_rps, a.ctx_rps)) return hipErrorInvalidValue;
    if (dtype == DT_F32) {
        auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        if (!x3_mfma_attention() || a.S > 128 || !al(a.qkv) || !al(a.ctx) || !al(a.dctx) || !al(a.dqkv)) {
            if (a.pair) return hipErrorInvalidValue;
            return launch_attn_bwd_f32(a, s);
        }
        if (a.S <= 32) launch_bwd_x3_t<1>(a, s);
        else if (a.S <= 64) launch_bwd_x3_t<2>(a, s);
        else launch_bwd_x3_t<4>(a, s);
        return hipGetLastError();
    }
    return dtype == DT_BF16 ? launch_bwd_d<bf16_t>(a, s) : launch_bwd_d<f16_t>(a, s);
}

}  // namespace mmhip
