// Shared device helpers for the mmhip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

namespace mmhip {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MM_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define MM_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

// ---------------------------------------------------------------- 16-bit activation types
template <typename T> struct Vec;
template <> struct Vec<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Vec<f16_t> { typedef f16x8 v8; typedef f16x4 v4; };
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <> struct Vec<float> { typedef f32x8 v8; typedef f32x4 v4; };      // parity mode (bf16x3): activations stay fp32

template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

// D(16x16) += A(16x32) * B(32x16); lane l: A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15]; D: col=l&15,row=4(l>>4)+reg
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// D(32x32) += A(32x16) * B(16x32); lane l: A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31]; D: col=l&31,row=(reg&3)+8(reg>>2)+4(l>>5)
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------- parity mode (bf16x3): an fp32 value as two bf16, x ~= hi + lo
struct Frag3 { bf16x8 hi, lo; };
__device__ __forceinline__ Frag3 split8(const float* v) {
    Frag3 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)v[e];
        f.hi[e] = h;
        f.lo[e] = (bf16_t)(v[e] - (float)h);
    }
    return f;
}
__device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
}
__device__ __forceinline__ f32x16 mma3_32(const Frag3& a, const Frag3& b, f32x16 c) {      // the two small products first
    c = mfma32(a.lo, b.hi, c);
    c = mfma32(a.hi, b.lo, c);
    return mfma32(a.hi, b.hi, c);
}

// 16-byte LDS read of 8 elements
template <typename T> __device__ __forceinline__ typename Vec<T>::v8 lds_read8(const char* lds, int byte_off) {
    return *reinterpret_cast<const typename Vec<T>::v8*>(lds + byte_off);
}
// transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a 4-row x 16-column block; lane 4q+p gives the
// address of row q, columns 4p..4p+3; lane i receives column i, rows 0..3 in elements 0..3.
__device__ __forceinline__ s16x4 lds_read_tr4(const char* lds, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + byte_off));
}
template <typename T> __device__ __forceinline__ typename Vec<T>::v8 join_tr(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Vec<T>::v8, r);
}

// ---------------------------------------------------------------- counter-based dropout RNG
// Bit-identical to oracle/mm_oracle.py:rng_u32 (the oracle replays the very same masks).
__device__ __host__ __forceinline__ uint32_t mm_rng_u32(uint32_t idx, uint32_t stream, uint64_t seed) {
    uint32_t x = idx ^ (uint32_t)seed;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x += stream * 0x85EBCA77u + (uint32_t)(seed >> 32);
    x ^= x >> 13;
    x *= 0xC2B2AE3Du;
    x ^= x >> 16;
    return x;
}
struct DropCfg {
    uint64_t seed;
    uint32_t stream;
    uint32_t thresh16;   // drop when the element's 16 random bits < thresh16 (0 => dropout off)
    float keep_scale;    // 1 / (1 - thresh16/65536)
};
// element e uses 16 bits of the hash of (e >> 1): low half for even e, high half for odd e
__device__ __forceinline__ bool mm_keep(uint32_t e, const DropCfg& d) {
    uint32_t h = mm_rng_u32(e >> 1, d.stream, d.seed);
    uint32_t r = (e & 1u) ? (h >> 16) : (h & 0xFFFFu);
    return r >= d.thresh16;
}
// keep flags for two consecutive elements e (even) and e+1 with one hash
__device__ __forceinline__ void mm_keep2(uint32_t e_even, const DropCfg& d, bool& k0, bool& k1) {
    uint32_t h = mm_rng_u32(e_even >> 1, d.stream, d.seed);
    k0 = (h & 0xFFFFu) >= d.thresh16;
    k1 = (h >> 16) >= d.thresh16;
}

// ---------------------------------------------------------------- math
// erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, i.e. fp32 round-off level)
__device__ __forceinline__ float mm_erf(float x) {
    float ax = fabsf(x);
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);      // v_rcp_f32 (1 ulp), not the IEEE division sequence
    float y = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float r = 1.0f - y * __expf(-ax * ax);
    return copysignf(r, x);
}
// exact (erf) GELU without the division: GELU(x) = max(x, 0) - |x| Phi(-|x|), and the normal tail is smooth in the log domain --
// Phi(-a) = 2^-(1 + P(a)) with P of degree 8 on [0, 5.5] (beyond it |x| Phi(-|x|) < 1.1e-7).  |error| <= 4.8e-7 absolute over all x
// (tools/gelu_fit.py: fit and check against erfc in double), the fp32 floor of the form; one v_exp_f32 and ten plain VALU operations per value
// instead of v_rcp_f32 + v_exp_f32 + fifteen (the Abramowitz-Stegun erf above): in the FC1 epilogue of the encoder GEMMs the activation was
// a fifth of the launch (profiles/r05_gemm_epilogue_cost.txt).
__device__ __forceinline__ float mm_gelu(float x) {
    const float a = __builtin_amdgcn_fmed3f(fabsf(x), 0.f, 5.5f);          // one v_med3_f32 (fminf / fmaxf also quiet their operands: two instructions each)
    float p = 9.896420750e-08f;
    p = fmaf(p, a, -2.559779944e-07f);
    p = fmaf(p, a, -4.382965926e-05f);
    p = fmaf(p, a, 8.528624312e-04f);
    p = fmaf(p, a, -8.322801441e-03f);
    p = fmaf(p, a, 5.372542515e-02f);
    p = fmaf(p, a, 4.586065114e-01f);
    p = fmaf(p, a, 1.151219487e+00f);
    p = fmaf(p, a, 9.999963641e-01f);
    return fmaf(x, 0.f, __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f) - a * __builtin_amdgcn_exp2f(-p));          // + 0 x: v_med3_f32 drops a NaN operand, GELU(NaN) must stay NaN (the overflow guard)
}
// two values at a time: the polynomial as v_pk_fma_f32 (hipcc does not pack the scalar form -- its constants are 32-bit literals of v_fmaak_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mm_gelu2(float& x0, float& x1) {
    const f32x2_t a = {__builtin_amdgcn_fmed3f(fabsf(x0), 0.f, 5.5f), __builtin_amdgcn_fmed3f(fabsf(x1), 0.f, 5.5f)};
    f32x2_t p = {9.896420750e-08f, 9.896420750e-08f};
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-2.559779944e-07f, -2.559779944e-07f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-4.382965926e-05f, -4.382965926e-05f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){8.528624312e-04f, 8.528624312e-04f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-8.322801441e-03f, -8.322801441e-03f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){5.372542515e-02f, 5.372542515e-02f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){4.586065114e-01f, 4.586065114e-01f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){1.151219487e+00f, 1.151219487e+00f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){9.999963641e-01f, 9.999963641e-01f});
    const f32x2_t t = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const f32x2_t r = {__builtin_amdgcn_fmed3f(x0, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x1, 0.f, 3.0e38f)};
    const f32x2_t x = {x0, x1};
    const f32x2_t o = __builtin_elementwise_fma(x, (f32x2_t){0.f, 0.f}, __builtin_elementwise_fma(-a, t, r));      // + 0 x: v_med3_f32 drops a NaN operand; GELU(NaN) stays NaN
    x0 = o[0];
    x1 = o[1];
}
// gelu'(x) = Phi(x) + x phi(x) from the same tail polynomial: r = Phi(-a) - a phi(a), gelu' = 1 - r for x >= 0, r below; |error| <= 1.3e-6
__device__ __forceinline__ f32x2_t mm_gelu_grad2(float x0, float x1) {
    const f32x2_t a = {__builtin_amdgcn_fmed3f(fabsf(x0), 0.f, 5.5f), __builtin_amdgcn_fmed3f(fabsf(x1), 0.f, 5.5f)};
    f32x2_t p = {9.896420750e-08f, 9.896420750e-08f};
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-2.559779944e-07f, -2.559779944e-07f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-4.382965926e-05f, -4.382965926e-05f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){8.528624312e-04f, 8.528624312e-04f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){-8.322801441e-03f, -8.322801441e-03f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){5.372542515e-02f, 5.372542515e-02f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){4.586065114e-01f, 4.586065114e-01f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){1.151219487e+00f, 1.151219487e+00f});
    p = __builtin_elementwise_fma(p, a, (f32x2_t){9.999963641e-01f, 9.999963641e-01f});
    const f32x2_t e = __builtin_elementwise_fma(a * a, (f32x2_t){-0.72134752044f, -0.72134752044f}, (f32x2_t){-1.32574806473f, -1.32574806473f});   // log2 phi(a)
    const f32x2_t T = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const f32x2_t ph = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    const f32x2_t h = (f32x2_t){0.5f, 0.5f} - __builtin_elementwise_fma(-a, ph, T);
    return (f32x2_t){fmaf(x0, 0.f, 0.5f + copysignf(h[0], x0)), fmaf(x1, 0.f, 0.5f + copysignf(h[1], x1))};          // (+ 0 x: NaN in, NaN out)
}
__device__ __forceinline__ float mm_qgelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float mm_gelu_grad(float x) {
    // cdf(x) + x pdf(x).  erf(x / sqrt2) = 1 - poly(t) exp(-x^2 / 2) and pdf(x) = exp(-x^2 / 2) / sqrt(2 pi) share ONE exponential
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float y = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __expf(-ax * ax);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - y * e, x));
    return cdf + x * (0.3989422804014327f * e);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin dispatch),
// so give each XCD a contiguous chunk of logical tile ids (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}


// ---- process-wide side streams, one set per device.  ROCm maps HIP streams onto a small pool of hardware queues (4 unless
// GPU_MAX_HW_QUEUES says otherwise) as they are created.  An engine created after other engines have come and gone -- a training model
// after the thirteen parity models of bench.py -- got streams that shared a hardware queue with the caller's stream or with one another,
// and silently lost the overlap the streams exist for (measured, same box: strict-dtype step 29.4 ms instead of 24.5; 25.0 with
// GPU_MAX_HW_QUEUES=8).  So the engines borrow streams that are created once per process and device, in a fixed order, and never
// destroyed: every engine of the process sees the mapping the first one saw.  Two engines driven at the same time share them, which
// orders their side work but keeps every dependency (events) intact.
enum { POOL_SIDE = 0, POOL_VIT = 1, POOL_VIT_HI = 2, POOL_STREAMS = 3 };
inline hipError_t pool_stream(int which, hipStream_t* out) {
    static std::mutex mu;
    static hipStream_t pool[16][POOL_STREAMS] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    hipError_t r = hipGetDevice(&dev);
    if (r != hipSuccess) return r;
    if (dev < 0 || dev >= 16 || which < 0 || which >= POOL_STREAMS) return hipErrorInvalidValue;
    if (!pool[dev][0]) {
        // MMHIP_SIDE_PRIO: HIP priority of the backward's side stream (lower = higher; default: equal to the caller's)
        const char* v = getenv("MMHIP_SIDE_PRIO");
        int least = 0, greatest = 0;
        if ((r = hipDeviceGetStreamPriorityRange(&least, &greatest)) != hipSuccess) return r;
        hipStream_t s[POOL_STREAMS] = {};
        r = v ? hipStreamCreateWithPriority(&s[POOL_SIDE], hipStreamNonBlocking, atoi(v)) : hipStreamCreateWithFlags(&s[POOL_SIDE], hipStreamNonBlocking);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&s[POOL_VIT], hipStreamNonBlocking);
        if (r == hipSuccess) r = hipStreamCreateWithPriority(&s[POOL_VIT_HI], hipStreamNonBlocking, greatest);
        if (r != hipSuccess) {
            for (auto q : s) if (q) (void)hipStreamDestroy(q);
            return r;
        }
        for (int i = 0; i < POOL_STREAMS; ++i) pool[dev][i] = s[i];
    }
    *out = pool[dev][which];
    return hipSuccess;
}

}  // namespace mmhip
