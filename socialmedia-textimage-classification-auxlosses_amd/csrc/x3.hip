// Parity mode (dtype MMHIP_BF16X3): activations stay fp32 in HBM and every Linear runs on the bf16 matrix cores as THREE
// products of split operands,  x = hi + lo (hi = bf16(x), lo = bf16(x - hi)):   x.w ~= hi.hi + lo.hi + hi.lo
// (the lo.lo term is 2^-16 relative).  SURVEY.md 7.3 measured this policy at 1.8e-5 on the per-post logits against the
// fp32 reference, where single-pass bf16 operands give 7e-3 and f16 9e-4 -- it is the mode in which north_star's 1e-3
// logit tolerance is asserted (tests/test_gpu_model.py).  Throughput is secondary here: fragments come straight from
// global memory (an MFMA 16x16x32 operand fragment is 16 rows x 8 consecutive k per lane group = whole 128-byte lines of an
// fp32 row), no LDS, no barriers; the split is done in registers.  Attention (1.5 % of the FLOPs) is plain fp32 on the
// vector ALUs with K / V (and Q / dO in the backward) resident in LDS.
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

// D += A.B with both operands split; the two small products first
__device__ __forceinline__ f32x4 mma3(const Frag3& a, const Frag3& b, f32x4 c) {
    c = mfma16(a.lo, b.hi, c);
    c = mfma16(a.hi, b.lo, c);
    return mfma16(a.hi, b.hi, c);
}

// operand fragment of 8 consecutive k of one row: an fp32 row split in registers, or the two planes of a plane pair as they are
// (mmhip_kernels.h: hi at base[row * ld + k], lo `lo` elements behind; ld, lo in 16-bit elements)
// hi_only (GemmNTArgs::nprod = 1): the lo plane is not read -- under the one-product backward its producers do not write it
__device__ __forceinline__ Frag3 frag_any(const void* base, size_t row, int ld, int k, int pair, int lo, int hi_only = 0) {
    if (pair) {
        const bf16_t* p = (const bf16_t*)base + row * (size_t)ld + k;
        Frag3 f;
        f.hi = *reinterpret_cast<const bf16x8*>(p);
        if (hi_only) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f.lo[e] = (bf16_t)0.0f;
        } else {
            f.lo = *reinterpret_cast<const bf16x8*>(p + lo);
        }
        return f;
    }
    float v[8];
    load8((const float*)base + row * (size_t)ld + k, v);
    return split8(v);
}
__device__ __forceinline__ float value_any(const void* base, size_t row, int ld, int k, int pair, int lo, int hi_only = 0) {
    if (pair) { const bf16_t* p = (const bf16_t*)base + row * (size_t)ld + k; return hi_only ? (float)p[0] : (float)p[0] + (float)p[lo]; }
    return ((const float*)base)[row * (size_t)ld + k];
}
// C row store of 4 consecutive columns: fp32, or a plane pair (GEMM_OUT_PAIR)
__device__ __forceinline__ void store4_any(const GemmNTArgs& a, int m, int n, const float* v) {
    if (a.flags & GEMM_OUT_PAIR) {
        bf16_t* c = (bf16_t*)a.C + (size_t)m * a.ldc + n;
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)v[e]; l[e] = (bf16_t)(v[e] - (float)h[e]); }
        *reinterpret_cast<bf16x4*>(c) = h;
        if (!(a.flags & GEMM_OUT_PAIR_HI)) *reinterpret_cast<bf16x4*>(c + a.c_lo) = l;
    } else {
        *reinterpret_cast<f32x4*>((float*)a.C + (size_t)m * a.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// ------------------------------------------------------------------------------------------------ NT
// epilogue of four consecutive columns n .. n+3 of row m (bias already added): the flags of gemm.hip's fused epilogue on fp32 / plane-pair tensors
__device__ __forceinline__ void x3_epilogue4(const GemmNTArgs& a, int fl, int m, int n, float (&v)[4]) {
    if (fl & GEMM_AUX_PRE) *reinterpret_cast<f32x4*>((float*)a.aux + (size_t)m * a.ldaux + n) = f32x4{v[0], v[1], v[2], v[3]};
    if (fl & GEMM_GELU) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) mm_gelu2(v[e], v[e + 1]);
    }
    if (fl & GEMM_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    }
    if (fl & GEMM_QGELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mm_qgelu(v[e]);
    }
    if (fl & GEMM_MUL_GELU_GRAD) {
        const f32x4 u = *reinterpret_cast<const f32x4*>((const float*)a.mul_in + (size_t)m * a.ldmul + n);
#pragma unroll
        for (int e = 0; e < 4; e += 2) { const f32x2_t gg = mm_gelu_grad2(u[e], u[e + 1]); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
    }
    if ((fl & GEMM_DROPOUT) && a.drop.thresh16) {
        const uint32_t e0 = (uint32_t)m * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            bool k0, k1;
            mm_keep2(e0 + e, a.drop, k0, k1);
            v[e] = k0 ? v[e] * a.drop.keep_scale : 0.f;
            v[e + 1] = k1 ? v[e + 1] * a.drop.keep_scale : 0.f;
        }
    }
    if (fl & GEMM_RESIDUAL) {
        const f32x4 r = *reinterpret_cast<const f32x4*>((const float*)a.residual + (size_t)m * a.ldres + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    store4_any(a, m, n, v);
}

// C[M,N] = epilogue(A[M,K] . B[N,K]^T), all fp32 in memory (or plane pairs, see frag_any).  Block = 4 waves (2 x 2), 128 x 128 tile, wave 64 x 64.
// Swapped MFMA operands (D = B_frag . A_frag^T): D row = n offset 4*(lane>>4) + reg, D column = m offset lane&15, so a
// lane holds 4 consecutive columns of one output row -> 16-byte stores and the same fused epilogue as gemm.hip.
__global__ __launch_bounds__(256) void gemm_nt_x3_kernel(GemmNTArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
    const int l15 = lane & 15, kc = lane >> 4;
    const int m0 = blockIdx.y * 128 + wm * 64, n0 = blockIdx.x * 128 + wn * 64;
    if (m0 >= a.M || n0 >= a.N) return;
    size_t arow[4], brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { arow[i] = (size_t)min(m0 + i * 16 + l15, a.M - 1); brow[i] = (size_t)min(n0 + i * 16 + l15, a.N - 1); }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag3 na[4], nb[4];
    const int h1 = a.nprod == 1;          // one product: hi planes only (the small terms multiply zeros)
#pragma unroll
    for (int i = 0; i < 4; ++i) { na[i] = frag_any(a.A, arow[i], a.lda, kc * 8, a.a_pair, a.a_lo, h1); nb[i] = frag_any(a.B, brow[i], a.ldb, kc * 8, a.b_pair, a.b_lo, h1); }
#pragma unroll 1
    for (int k0 = 0; k0 < a.K; k0 += 32) {
        Frag3 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i] = na[i]; bf[i] = nb[i]; }
        if (k0 + 32 < a.K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                na[i] = frag_any(a.A, arow[i], a.lda, k0 + 32 + kc * 8, a.a_pair, a.a_lo, h1);
                nb[i] = frag_any(a.B, brow[i], a.ldb, k0 + 32 + kc * 8, a.b_pair, a.b_lo, h1);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma3(bf[j], af[i], acc[i][j]);
    }
    const int fl = a.flags;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + 4 * kc;
        if (n >= a.N) continue;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (fl & GEMM_BIAS) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * 16 + l15;
            if (m >= a.M) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[e];
            x3_epilogue4(a, fl, m, n, v);
        }
    }
}

// Few rows (M <= 128: the CLS-row GEMMs of the last text layer, 64 x 768 x 3072 and the like).  The 128 x 128 kernel above puts such a problem on
// N / 128 = 6 workgroups with half their waves idle and walks all of K behind one prefetched step: 138 us for 64 x 768 x 3072 (round 5,
// profiles/r05_nt_shapes_bf16x3_bwd1.txt).  Here a workgroup takes a 64-row x 16-column tile (N / 16 = 48 ... 192 workgroups) and its four waves a
// QUARTER OF K each (same fragments, same products); the four partial tiles meet in 16 KB of LDS -- little enough to sit on a CU beside a 128 KB
// workgroup of the other tower's GEMM (a first version with 64 x 64 tiles and 64 KB waited for whole CUs) -- and wave w finishes row fragment w through
// the same epilogue.  Deterministic (fixed order of the four partial sums).  K % 128 == 0, N % 16 == 0.
__global__ __launch_bounds__(256) void gemm_nt_x3_small_kernel(GemmNTArgs a) {
    __shared__ f32x4 red[4][4][64];          // [wave][row fragment][lane]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, kc = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 16;
    size_t arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = (size_t)min(m0 + i * 16 + l15, a.M - 1);
    const size_t brow = (size_t)min(n0 + l15, a.N - 1);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int h1 = a.nprod == 1;
    const int kq = a.K >> 2, kb = w * kq, ke = kb + kq;
    Frag3 na[4], nb;
#pragma unroll
    for (int i = 0; i < 4; ++i) na[i] = frag_any(a.A, arow[i], a.lda, kb + kc * 8, a.a_pair, a.a_lo, h1);
    nb = frag_any(a.B, brow, a.ldb, kb + kc * 8, a.b_pair, a.b_lo, h1);
#pragma unroll 1
    for (int k0 = kb; k0 < ke; k0 += 32) {
        Frag3 af[4], bf = nb;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = na[i];
        if (k0 + 32 < ke) {
#pragma unroll
            for (int i = 0; i < 4; ++i) na[i] = frag_any(a.A, arow[i], a.lda, k0 + 32 + kc * 8, a.a_pair, a.a_lo, h1);
            nb = frag_any(a.B, brow, a.ldb, k0 + 32 + kc * 8, a.b_pair, a.b_lo, h1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mma3(bf, af[i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[w][i][lane] = acc[i];
    __syncthreads();
    const int fl = a.flags, i = w, n = n0 + 4 * kc, m = m0 + i * 16 + l15;
    f32x4 t = red[0][i][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) { const f32x4 p = red[q][i][lane]; t[0] += p[0]; t[1] += p[1]; t[2] += p[2]; t[3] += p[3]; }
    if (n >= a.N || m >= a.M) return;
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (fl & GEMM_BIAS) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e] + b4[e];
    x3_epilogue4(a, fl, m, n, v);
}

// generic fp32 NT for shapes the tiled kernel does not take (K % 32, N % 4, unaligned): plain fp32 FMA, same epilogue
__global__ __launch_bounds__(256) void slow_nt_f32_kernel(GemmNTArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= a.N) return;
    float v = 0.f;
    for (int k = 0; k < a.K; ++k) v = fmaf(value_any(a.A, m, a.lda, k, a.a_pair, a.a_lo, a.nprod == 1), value_any(a.B, n, a.ldb, k, a.b_pair, a.b_lo, a.nprod == 1), v);
    const int fl = a.flags;
    if (fl & GEMM_BIAS) v += a.bias[n];
    if (fl & GEMM_AUX_PRE) ((float*)a.aux)[(size_t)m * a.ldaux + n] = v;
    if (fl & GEMM_GELU) v = mm_gelu(v);
    if (fl & GEMM_TANH) v = tanhf(v);
    if (fl & GEMM_QGELU) v = mm_qgelu(v);
    if (fl & GEMM_MUL_GELU_GRAD) v *= mm_gelu_grad(((const float*)a.mul_in)[(size_t)m * a.ldmul + n]);
    if ((fl & GEMM_DROPOUT) && a.drop.thresh16) v = mm_keep((uint32_t)m * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.N + (uint32_t)n, a.drop) ? v * a.drop.keep_scale : 0.f;
    if (fl & GEMM_RESIDUAL) v += ((const float*)a.residual)[(size_t)m * a.ldres + n];
    if (fl & GEMM_OUT_PAIR) {
        bf16_t* c = (bf16_t*)a.C + (size_t)m * a.ldc + n;
        const bf16_t h = (bf16_t)v;
        c[0] = h;
        c[a.c_lo] = (bf16_t)(v - (float)h);
    } else {
        ((float*)a.C)[(size_t)m * a.ldc + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------ split planes
// The fast form of the parity mode: an fp32 operand is written once as bf16 planes hi = bf16(x), lo = bf16(x - hi), three copies laid out so
// that ONE ordinary bf16 GEMM over a three times longer reduction index computes  hi.hi + lo.hi + hi.lo :
//   NT (reduction along the row):     A' = [hi | lo | hi]  (M x 3K),   B' = [hi | hi | lo]  (N x 3K)   -> gemm_nt8_kernel, fp32 epilogue
//   TN (reduction over the rows):     A' = [hi ; lo ; hi]  (3M x Nn),  B' = [hi ; hi ; lo]  (3M x Nc)  -> gemm_tn_kernel
// so the deep-pipelined kernels of gemm8.hip / gemm.hip run unchanged (fp32 accumulation of the three products in the MFMA accumulators).
// Cost of the copies: 4 B read + 6 B written per operand element, a few per cent of the three-pass product.
// Thread = 8 consecutive elements of a row; the three copies of the row go to dst + o[k] + row * dld.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int ld, int rows, int cols, bf16_t* __restrict__ dst, int dld,
                                                     size_t o_hi0, size_t o_lo, size_t o_hi1) {
    const int per_row = cols >> 3;
    const size_t total = (size_t)rows * per_row;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / per_row), c = (int)(idx - (size_t)r * per_row) << 3;
        float v[8];
        load8(src + (size_t)r * ld + c, v);
        const Frag3 f = split8(v);
        bf16_t* d = dst + (size_t)r * dld + c;
        *reinterpret_cast<bf16x8*>(d + o_hi0) = f.hi;
        *reinterpret_cast<bf16x8*>(d + o_lo) = f.lo;
        *reinterpret_cast<bf16x8*>(d + o_hi1) = f.hi;
    }
}
static void launch_split3(const float* src, int ld, int rows, int cols, void* dst, int dld, size_t o_hi0, size_t o_lo, size_t o_hi1, hipStream_t s) {
    const size_t threads = (size_t)rows * (cols >> 3);
    const int grid = (int)((threads + 255) / 256 > 16384 ? 16384 : (threads + 255) / 256);
    hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, s, src, ld, rows, cols, (bf16_t*)dst, dld, o_hi0, o_lo, o_hi1);
}
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
size_t x3_nt_scratch_bytes(int M, int N, int K) { return up256((size_t)M * 3 * K * 2) + up256((size_t)N * 3 * K * 2); }
size_t x3_tn_scratch_bytes(int M, int Nn, int Nc) { return up256((size_t)3 * M * Nn * 2) + up256((size_t)3 * M * Nc * 2); }

// the split-plane form through the deep-pipelined kernel; false = rules not met
static bool nt_x3_fast(const GemmNTArgs& a, hipStream_t s) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MMHIP_X3_FAST"); on = e ? atoi(e) : 1; }
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool pairs = a.a_pair && a.b_pair;          // plane pairs written by the producers: no scratch, no copies
    if (a.a_pair != a.b_pair) return false;
    if (!on || (!pairs && !a.x3_ws) || a.force_slow || a.M <= 128 || a.N % 128 || a.K % 64 || a.lda % 4 || a.ldb % 4 || !al(a.A) || !al(a.B)) return false;
    if (!pairs && x3_nt_scratch_bytes(a.M, a.N, a.K) > a.x3_ws_bytes) return false;
    int bn = 0;
    double best = 0;
    const long tm = (a.M + 255) / 256;
    for (int cand : {256, 192, 128}) {
        if (a.N % cand) continue;
        const long t = tm * (a.N / cand);
        const double u = (double)t / (double)(((t + 255) / 256) * 256) + (cand == 256 ? 0.08 : (cand == 192 ? 0.04 : 0.0));      // near ties -> wider
        if (u > best) { best = u; bn = cand; }
    }
    if (pairs) {
        GemmNTArgs b = a;
        b.x3_ws = nullptr; b.splitk_ws = nullptr;
        if (a.nprod == 1) { b.a_pair = b.b_pair = 0; }      // hi . hi only: the hi planes are ordinary bf16 matrices of leading dimension lda / ldb
        return launch_gemm_nt8(b, DT_F32, bn, 1, s);
    }
    GemmNTArgs b = a;
    char* pa = (char*)a.x3_ws;
    char* pb = pa + up256((size_t)a.M * 3 * a.K * 2);
    b.A = pa; b.lda = 3 * a.K; b.B = pb; b.ldb = 3 * a.K; b.K = 3 * a.K; b.x3_ws = nullptr; b.splitk_ws = nullptr;
    launch_split3((const float*)a.A, a.lda, a.M, a.K, pa, 3 * a.K, 0, (size_t)a.K, (size_t)2 * a.K, s);
    launch_split3((const float*)a.B, a.ldb, a.N, a.K, pb, 3 * a.K, 0, (size_t)2 * a.K, (size_t)a.K, s);
    if (launch_gemm_nt8(b, DT_F32, bn, 1, s)) return true;
    return false;          // (the two copies above are harmless: the caller falls back to the direct kernel)
}

hipError_t launch_gemm_nt_x3(const GemmNTArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (nt_x3_fast(a, s)) return hipGetLastError();
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool fast = !a.force_slow && a.K % 32 == 0 && a.N % 4 == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.ldc % 4 == 0 && al(a.A) && al(a.B) && al(a.C) &&
                      (!a.a_pair || (a.a_lo % 8 == 0 && a.lda % 8 == 0)) && (!a.b_pair || (a.b_lo % 8 == 0 && a.ldb % 8 == 0)) && (!(a.flags & GEMM_OUT_PAIR) || a.c_lo % 4 == 0) &&
                      (!(a.flags & GEMM_RESIDUAL) || (a.ldres % 4 == 0 && al(a.residual))) && (!(a.flags & GEMM_AUX_PRE) || (a.ldaux % 4 == 0 && al(a.aux))) &&
                      (!(a.flags & GEMM_MUL_GELU_GRAD) || (a.ldmul % 4 == 0 && al(a.mul_in))) && (!(a.flags & GEMM_BIAS) || al(a.bias));
    static int small_on = -1;
    if (small_on < 0) { const char* e = getenv("MMHIP_X3_SMALL"); small_on = e ? atoi(e) : 1; }
    if (fast && small_on && a.M <= 128 && a.K % 128 == 0 && a.N % 16 == 0) {          // few rows: K split over the waves of a 64 x 16 tile (MMHIP_X3_SMALL=0: the 128 x 128 kernel)
        hipLaunchKernelGGL(gemm_nt_x3_small_kernel, dim3(a.N / 16, (a.M + 63) / 64), dim3(256), 0, s, a);
    } else if (fast) hipLaunchKernelGGL(gemm_nt_x3_kernel, dim3((a.N + 127) / 128, (a.M + 127) / 128), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(slow_nt_f32_kernel, dim3((a.N + 255) / 256, a.M), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ TN
// C[n][c] (+)= alpha * sum_m A[m][n] * B[m][c]  (weight gradients dW = dY^T . X; A, B fp32 [M, .], C fp32), and optionally
// colsum[n] (+)= alpha * sum_m A[m][n].  The reduction index is the row index of both operands, so a fragment is 8 strided
// dword loads per lane (16 consecutive columns per lane group: 64-byte segments).  D = X_frag^T . dY_frag: D row = column c
// offset 4*(lane>>4) + reg, D column = n offset lane&15 -> a lane stores 4 consecutive c of one output row n.
__global__ __launch_bounds__(256) void gemm_tn_x3_kernel(GemmTNProblem P, int accumulate, float alpha) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wn_ = w >> 1, wc_ = w & 1;
    const int l15 = lane & 15, kc = lane >> 4;
    const int n0 = blockIdx.y * 128 + wn_ * 64, c0 = blockIdx.x * 128 + wc_ * 64;
    if (n0 >= P.Nn || c0 >= P.Nc) return;
    int ncol[4], ccol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ncol[i] = min(n0 + i * 16 + l15, P.Nn - 1); ccol[i] = min(c0 + i * 16 + l15, P.Nc - 1); }
    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool with_colsum = P.colsum != nullptr && blockIdx.x == 0 && wc_ == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll 1
    for (int mb = 0; mb < P.M; mb += 32) {
        Frag3 yf[4], xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float ry[8], rx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = mb + 8 * kc + e;
                const bool in = m < P.M;
                const int mr = in ? m : P.M - 1;
                const float y = value_any(P.A, mr, P.lda, ncol[i], P.pair, P.a_lo, P.nprod == 1), x = value_any(P.B, mr, P.ldb, ccol[i], P.pair, P.b_lo, P.nprod == 1);
                ry[e] = in ? y : 0.f;
                rx[e] = in ? x : 0.f;
            }
            yf[i] = split8(ry);
            xf[i] = split8(rx);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma3(xf[j], yf[i], acc[i][j]);
        if (with_colsum) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { accb[i] = mfma16(ones, yf[i].lo, accb[i]); accb[i] = mfma16(ones, yf[i].hi, accb[i]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + i * 16 + l15;
        if (n >= P.Nn) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * 16 + 4 * kc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c + e >= P.Nc) continue;
                float* dst = P.C + (size_t)n * P.ldc + c + e;
                if (accumulate == 1) atomicAdd(dst, acc[i][j][e] * alpha);
                else *dst = acc[i][j][e] * alpha;
            }
        }
        if (with_colsum && kc == 0) {       // every D row holds the same column sums: take row 0
            float* dst = P.colsum + n;
            if (accumulate == 1) atomicAdd(dst, accb[i][0] * alpha);
            else *dst = accb[i][0] * alpha;
        }
    }
}

hipError_t launch_gemm_tn_x3(const GemmTNProblem* probs, int count, int accumulate, hipStream_t s, float alpha, void* ws, size_t ws_bytes) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MMHIP_X3_FAST"); on = e ? atoi(e) : 1; }
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    // split-plane form: the problems that fit the grouped bf16 kernel's shape rules (gemm.hip) go through it with three times the rows
    GemmTNProblem fastp[GEMM_TN_MAX_GROUP];
    int nfast = 0;
    size_t used = 0;
    bool taken[64] = {false};
    for (int i = 0; i < count && i < 64 && on; ++i) {
        const GemmTNProblem& P = probs[i];
        if (P.pair) {          // plane pairs: the grouped kernel walks the planes itself (gemm.hip), no scratch
            if (P.M <= 0 || P.M % 64 || P.Nn % 256 || P.Nc % 128 || P.lda % 8 || P.ldb % 8 || P.a_lo % 8 || P.b_lo % 8 || !al(P.A) || !al(P.B) || nfast == GEMM_TN_MAX_GROUP) continue;
            fastp[nfast] = P;
            if (P.nprod == 1) fastp[nfast].pair = 0;      // hi . hi only: the hi planes as ordinary bf16 matrices (the column sums then cover dY's hi plane)
            ++nfast;
            taken[i] = true;
            continue;
        }
        if (!ws) continue;
        if (P.M <= 0 || P.M % 64 || P.Nn % 256 || P.Nc % 128 || P.lda % 4 || P.ldb % 4 || !al(P.A) || !al(P.B) || nfast == GEMM_TN_MAX_GROUP) continue;
        const size_t need = x3_tn_scratch_bytes(P.M, P.Nn, P.Nc);
        if (used + need > ws_bytes) continue;
        char* pa = (char*)ws + used;
        char* pb = pa + up256((size_t)3 * P.M * P.Nn * 2);
        used += need;
        const size_t pla = (size_t)P.M * P.Nn, plb = (size_t)P.M * P.Nc;
        launch_split3((const float*)P.A, P.lda, P.M, P.Nn, pa, P.Nn, 0, pla, 2 * pla, s);
        launch_split3((const float*)P.B, P.ldb, P.M, P.Nc, pb, P.Nc, 0, 2 * plb, plb, s);
        GemmTNProblem Q = P;
        Q.A = pa; Q.lda = P.Nn; Q.B = pb; Q.ldb = P.Nc; Q.M = 3 * P.M;
        Q.colsum_rows = 2 * P.M;          // bias gradient = column sums of dY = of its hi and lo planes: the first two of the three stacked blocks
        fastp[nfast++] = Q;
        taken[i] = true;
    }
    if (nfast) { hipError_t e = launch_gemm_tn(fastp, nfast, accumulate, DT_BF16, 0, s, alpha); if (e != hipSuccess) return e; }
    for (int i = 0; i < count; ++i) {
        const GemmTNProblem& P = probs[i];
        if (i < 64 && taken[i]) continue;
        if (P.M <= 0 || P.Nn <= 0 || P.Nc <= 0) continue;
        hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((P.Nc + 127) / 128, (P.Nn + 127) / 128), dim3(256), 0, s, P, accumulate, alpha);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ attention, fp32
// One workgroup per (post, head); K and V of the head in LDS (fp32), one query per thread.  Same conventions as
// attention.hip: additive key bias, dropout on the normalised probabilities with element index ((post*heads+head)*S+q)*S+key,
// LSE in natural-log units of the scaled, biased scores, q_tiles = number of 32-row query tiles to compute.
static constexpr int HD = 64;
__device__ __forceinline__ float dot64(const float* q, const float* __restrict__ k) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(k + d);
        s0 = fmaf(q[d], kv[0], s0); s1 = fmaf(q[d + 1], kv[1], s1); s2 = fmaf(q[d + 2], kv[2], s2); s3 = fmaf(q[d + 3], kv[3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld, int rows, int tid, int nthr) {
    for (int idx = tid; idx < rows * 16; idx += nthr) {
        const int row = idx >> 4, c = (idx & 15) * 4;
        *reinterpret_cast<f32x4*>(dst + row * HD + c) = *reinterpret_cast<const f32x4*>(src + (size_t)row * ld + c);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = a.S, head = blockIdx.x, post = blockIdx.y, tid = threadIdx.x;
    float* Ks = sm;
    float* Vs = sm + (size_t)S * HD;
    float* mb = sm + (size_t)2 * S * HD;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // queries / keys of a post and rows per post, as in attn_fwd_kernel
    const int qr = a.q_rps > 0 ? a.q_rps : S, kr = a.kv_rps > 0 ? a.kv_rps : S, cr = a.ctx_rps > 0 ? a.ctx_rps : S;
    const float* base = (const float*)a.qkv + (size_t)post * qr * a.ld_qkv + head * HD;
    const float* kvb = (const float*)a.qkv + (size_t)post * kr * a.ld_qkv + a.hidden + head * HD;
    stage_rows(Ks, kvb, a.ld_qkv, Sk, tid, 256);
    stage_rows(Vs, kvb + a.hidden, a.ld_qkv, Sk, tid, 256);
    for (int k = tid; k < Sk; k += 256) mb[k] = a.maskbias ? a.maskbias[(size_t)post * S + k] : 0.f;
    __syncthreads();
    const int qlim = a.q_tiles > 0 ? min(Sq, a.q_tiles * 32) : Sq;
    const bool dropping = a.drop.thresh16 != 0;
    for (int q = tid; q < qlim; q += 256) {
        float qv[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(base + (size_t)q * a.ld_qkv + d);
            qv[d] = x[0]; qv[d + 1] = x[1]; qv[d + 2] = x[2]; qv[d + 3] = x[3];
        }
        float mx = -INFINITY;
        for (int key = 0; key < Sk; ++key) mx = fmaxf(mx, dot64(qv, Ks + key * HD) * a.scale + mb[key]);
        const float msafe = (mx == -INFINITY) ? 0.f : mx;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
        float l = 0.f;
        const uint32_t ebase = (uint32_t)(((size_t)post * a.heads + head) * S + (uint32_t)q) * (uint32_t)S;
        for (int key = 0; key < Sk; ++key) {
            const float p = __expf(dot64(qv, Ks + key * HD) * a.scale + mb[key] - msafe);
            l += p;
            float pd = p;
            if (dropping) pd = mm_keep(ebase + (uint32_t)key, a.drop) ? p * a.drop.keep_scale : 0.f;
            const float* vr = Vs + key * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + d);
                o[d] = fmaf(pd, vv[0], o[d]); o[d + 1] = fmaf(pd, vv[1], o[d + 1]); o[d + 2] = fmaf(pd, vv[2], o[d + 2]); o[d + 3] = fmaf(pd, vv[3], o[d + 3]);
            }
        }
        if (a.lse) a.lse[((size_t)post * a.heads + head) * S + q] = msafe + __logf(l);
        const float inv = 1.0f / l;
        float* op = (float*)a.ctx + ((size_t)post * cr + q) * a.ld_ctx + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<f32x4*>(op + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    }
}

// backward: LDS holds Q, dO, K, V of the head (fp32), lse, D = rowsum(dO . O) and the key bias.  Waves 0-1 (thread = key)
// accumulate dK, waves 2-3 (thread = key) dV, then waves 0-1 (thread = query) dQ: every output element is written by
// exactly one thread -- no atomics, bitwise reproducible.  S <= 128.
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = a.S, head = blockIdx.x, post = blockIdx.y, tid = threadIdx.x;
    float* Qs = sm;
    float* Gs = Qs + (size_t)S * HD;       // dO
    float* Ks = Gs + (size_t)S * HD;
    float* Vs = Ks + (size_t)S * HD;
    float* lse = Vs + (size_t)S * HD;
    float* Dv = lse + S;
    float* mb = Dv + S;
    const int Sq = a.Sq_live > 0 ? a.Sq_live : S, Sk = a.Sk_live > 0 ? a.Sk_live : S;      // as in attn_bwd_kernel
    const size_t qrow0 = (size_t)post * (a.q_rps > 0 ? a.q_rps : S), krow0 = (size_t)post * (a.kv_rps > 0 ? a.kv_rps : S), crow0 = (size_t)post * (a.ctx_rps > 0 ? a.ctx_rps : S);
    const float* qb = (const float*)a.qkv + qrow0 * a.ld_qkv + head * HD;
    const float* kb = (const float*)a.qkv + krow0 * a.ld_qkv + a.hidden + head * HD;
    const float* dob = (const float*)a.dctx + crow0 * a.ld_ctx + head * HD;
    const float* ob = (const float*)a.ctx + crow0 * a.ld_ctx + head * HD;
    stage_rows(Qs, qb, a.ld_qkv, Sq, tid, 256);
    stage_rows(Ks, kb, a.ld_qkv, Sk, tid, 256);
    stage_rows(Vs, kb + a.hidden, a.ld_qkv, Sk, tid, 256);
    stage_rows(Gs, dob, a.ld_ctx, Sq, tid, 256);
    for (int k = tid; k < Sk; k += 256) mb[k] = a.maskbias ? a.maskbias[(size_t)post * S + k] : 0.f;
    for (int q = tid; q < Sq; q += 256) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(ob + (size_t)q * a.ld_ctx + c), g = *reinterpret_cast<const f32x4*>(dob + (size_t)q * a.ld_ctx + c);
            d += o[0] * g[0] + o[1] * g[1] + o[2] * g[2] + o[3] * g[3];
        }
        Dv[q] = d;
        lse[q] = a.lse[((size_t)post * a.heads + head) * S + q];
    }
    __syncthreads();
    const int qlim = a.q_tiles > 0 ? min(Sq, a.q_tiles * 32) : Sq;      // later queries carry a zero d ctx
    const bool dropping = a.drop.thresh16 != 0;
    const uint32_t hbase = (uint32_t)(((size_t)post * a.heads + head) * S);
    const int role = tid >> 7, key = tid & 127;        // role 0: dK, role 1: dV
    if (key < Sk) {
        float kr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { kr[d] = Ks[key * HD + d]; acc[d] = 0.f; }
        const float mbk = mb[key];
        float* outp = (float*)a.dqkv + (krow0 + key) * a.ld_qkv + (role == 0 ? 1 : 2) * a.hidden + head * HD;
        if (role == 0) {
            float vr[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) vr[d] = Vs[key * HD + d];
            for (int q = 0; q < qlim; ++q) {
                const float p = __expf(dot64(kr, Qs + q * HD) * a.scale + mbk - lse[q]);
                float dpd = dot64(vr, Gs + q * HD);
                if (dropping) dpd = mm_keep((hbase + (uint32_t)q) * (uint32_t)S + (uint32_t)key, a.drop) ? dpd * a.drop.keep_scale : 0.f;
                const float ds = p * (dpd - Dv[q]) * a.scale;
                const float* qr = Qs + q * HD;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(qr + d);
                    acc[d] = fmaf(ds, x[0], acc[d]); acc[d + 1] = fmaf(ds, x[1], acc[d + 1]); acc[d + 2] = fmaf(ds, x[2], acc[d + 2]); acc[d + 3] = fmaf(ds, x[3], acc[d + 3]);
                }
            }
        } else {
            for (int q = 0; q < qlim; ++q) {
                float pd = __expf(dot64(kr, Qs + q * HD) * a.scale + mbk - lse[q]);
                if (dropping) pd = mm_keep((hbase + (uint32_t)q) * (uint32_t)S + (uint32_t)key, a.drop) ? pd * a.drop.keep_scale : 0.f;
                const float* gr = Gs + q * HD;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(gr + d);
                    acc[d] = fmaf(pd, x[0], acc[d]); acc[d + 1] = fmaf(pd, x[1], acc[d + 1]); acc[d + 2] = fmaf(pd, x[2], acc[d + 2]); acc[d + 3] = fmaf(pd, x[3], acc[d + 3]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<f32x4*>(outp + d) = f32x4{acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
    }
    if (tid < qlim) {       // dQ: thread = query
        const int q = tid;
        float qr[HD], gr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { qr[d] = Qs[q * HD + d]; gr[d] = Gs[q * HD + d]; acc[d] = 0.f; }
        const float lq = lse[q], dq = Dv[q];
        for (int k = 0; k < Sk; ++k) {
            const float p = __expf(dot64(qr, Ks + k * HD) * a.scale + mb[k] - lq);
            float dpd = dot64(gr, Vs + k * HD);
            if (dropping) dpd = mm_keep((hbase + (uint32_t)q) * (uint32_t)S + (uint32_t)k, a.drop) ? dpd * a.drop.keep_scale : 0.f;
            const float ds = p * (dpd - dq) * a.scale;
            const float* kr = Ks + k * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(kr + d);
                acc[d] = fmaf(ds, x[0], acc[d]); acc[d + 1] = fmaf(ds, x[1], acc[d + 1]); acc[d + 2] = fmaf(ds, x[2], acc[d + 2]); acc[d + 3] = fmaf(ds, x[3], acc[d + 3]);
            }
        }
        float* outp = (float*)a.dqkv + (qrow0 + q) * a.ld_qkv + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<f32x4*>(outp + d) = f32x4{acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
    }
}

hipError_t launch_attn_fwd_f32(const AttnArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)2 * a.S * HD + a.S) * 4;
    if (a.ld_qkv % 4 || a.ld_ctx % 4 || lds > 160 * 1024) return hipErrorInvalidValue;
    static size_t set = 0;
    if (lds > set) { (void)hipFuncSetAttribute((const void*)attn_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = 160 * 1024; }
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(a.heads, a.posts), dim3(256), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_attn_bwd_f32(const AttnBwdArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)4 * a.S * HD + 3 * a.S) * 4;
    if (a.ld_qkv % 4 || a.ld_ctx % 4 || a.S > 128) return hipErrorInvalidValue;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)attn_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(a.heads, a.posts), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace mmhip
