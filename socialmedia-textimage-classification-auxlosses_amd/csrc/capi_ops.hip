// Operator-level C entry points (parity tests call the very launchers the engine uses) and the hardware-layout probe.
#include <cstring>
#include <cmath>
#include "mmhip_common.h"
#include <vector>
#include "mmhip_kernels.h"
#include "../../include/mmhip.h"

using namespace mmhip;

#define CHECK_HIP(expr)                       \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

static DropCfg drop_of(float p, uint64_t seed, uint32_t stream) {
    DropCfg d;
    d.seed = seed;
    d.stream = stream;
    uint32_t t = p > 0.f ? (uint32_t)lrintf(p * 65536.0f) : 0u;
    if (t > 65535u) t = 65535u;
    d.thresh16 = t;
    d.keep_scale = 1.0f / (1.0f - (float)t / 65536.0f);
    return d;
}

// ------------------------------------------------------------------------------------------------ probe
// Index-coded operands make every lane's view of the MFMA / transposing-read layouts observable:
//   out[0    .. 1023]  D of mfma_f32_16x16x32_bf16 with A[i][k] = (k==0) * (i+1),  B[k][j] = (k==0) * (j+1)*32
//                      -> D[i][j] = (i+1)*(j+1)*32, stored as out[lane*4 + reg]
//   out[1024 .. 2047]  D of mfma_f32_32x32x16_bf16, same construction, out[1024 + lane*16 + reg]
//   out[2048 .. 2303]  ds_read_b64_tr_b16 of a [16 rows][64 cols] bf16 image holding row*64+col; lane's address =
//                      row (lane>>4)*4 + ((lane>>2)&3), col 4*(lane&3); out[2048 + lane*4 + e]
//   out[2304 .. 2815]  k-order probe of mfma_f32_16x16x32: A[i][k] = 1 for all, B[k][j] = (j==0) ? 2^k' ... (see test)
__global__ __launch_bounds__(64) void probe_kernel(int32_t* out) {
    __shared__ __attribute__((aligned(16))) bf16_t img[16 * 64];
    const int lane = threadIdx.x;
    {   // 16x16x32: lane l holds A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15]
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * (lane >> 4) + j;
            a[j] = (bf16_t)(k == 0 ? (float)((lane & 15) + 1) : 0.f);
            b[j] = (bf16_t)(k == 0 ? (float)(((lane & 15) + 1) * 32) : 0.f);
        }
        f32x4 d = mfma16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
        for (int r = 0; r < 4; ++r) out[lane * 4 + r] = (int)d[r];
    }
    {   // 32x32x16: lane l holds A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31]
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * (lane >> 5) + j;
            a[j] = (bf16_t)(k == 0 ? (float)((lane & 31) + 1) : 0.f);
            b[j] = (bf16_t)(k == 0 ? (float)(((lane & 31) + 1) * 32) : 0.f);
        }
        f32x16 d = mfma32(a, b, f32x16{});
        for (int r = 0; r < 16; ++r) out[1024 + lane * 16 + r] = (int)d[r];
    }
    for (int i = lane; i < 16 * 64; i += 64) img[i] = (bf16_t)(float)(i % 256);   // value = (row*64+col) mod 256 (exact in bf16)
    __syncthreads();
    {
        const int row = (lane >> 4) * 4 + ((lane >> 2) & 3), col = 4 * (lane & 3);
        s16x4 t = lds_read_tr4(reinterpret_cast<const char*>(img), (row * 64 + col) * 2);
        bf16x4 tb = __builtin_bit_cast(bf16x4, t);
        for (int e = 0; e < 4; ++e) out[2048 + lane * 4 + e] = (int)(float)tb[e];
    }
    {   // k-order: A[i][k] = k+1 (all rows), B[k][j] = (k == j) for j < 16 (k < 16) -> D[i][j] = j+1;  second half k>=16: B[k][j] = (k-16==j)*64
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * (lane >> 4) + j;
            a[j] = (bf16_t)(float)(k + 1);
            b[j] = (bf16_t)((k & 15) == (lane & 15) ? (k < 16 ? 1.f : 64.f) : 0.f);
        }
        f32x4 d = mfma16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
        for (int r = 0; r < 4; ++r) out[2304 + lane * 4 + r] = (int)d[r];
    }
}

extern "C" {

int mmhip_op_probe_layouts(int32_t* out, void* stream) {
    if (!out) return MMHIP_E_INVALID;
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    CHECK_HIP(hipGetLastError());
    return 0;
}

int mmhip_op_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const float* bias, int act, void* aux_pre, int ldaux, const void* mul_gelu_grad_of, int ldmul,
                     float p_drop, uint64_t seed, uint32_t stream_id, const void* residual, int ldres, int out_f32,
                     int force_slow, void* stream) {
    if (!A || !B || !C || M < 0 || N < 1 || K < 1 || (dtype != MMHIP_BF16 && dtype != MMHIP_F16 && dtype != MMHIP_F32)) return MMHIP_E_INVALID;
    GemmNTArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    if (bias) { a.bias = bias; a.flags |= GEMM_BIAS; }
    if (act == 1) a.flags |= GEMM_GELU;
    if (act == 2) a.flags |= GEMM_TANH;
    if (act == 3) a.flags |= GEMM_QGELU;
    if (aux_pre && (force_slow & 2)) { a.aux = aux_pre; a.flags |= GEMM_DEBUG_TS | ((force_slow & 4) ? GEMM_DEBUG_CYC : 0); }    // profiling hooks of gemm8.hip
    else if (aux_pre) { a.aux = aux_pre; a.ldaux = ldaux; a.flags |= GEMM_AUX_PRE; }
    if (mul_gelu_grad_of) { a.mul_in = mul_gelu_grad_of; a.ldmul = ldmul; a.flags |= GEMM_MUL_GELU_GRAD; }
    if (p_drop > 0.f) { a.drop = drop_of(p_drop, seed, stream_id); a.flags |= GEMM_DROPOUT; }
    if (residual) { a.residual = residual; a.ldres = ldres; a.flags |= GEMM_RESIDUAL; }
    if (out_f32) a.flags |= GEMM_OUT_F32;
    a.force_slow = force_slow & 1;
    a.tile = force_slow >> 4;      // bits 4.. select the tile variant (test / tuning hook)
    CHECK_HIP(launch_gemm_nt(a, dtype, (hipStream_t)stream));
    return 0;
}

int mmhip_op_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int Nn, int Nc,
                     int accumulate, int force_slow, float* colsum, void* stream) {
    if (!A || !B || !C || M < 1 || Nn < 1 || Nc < 1 || (dtype != MMHIP_BF16 && dtype != MMHIP_F16 && dtype != MMHIP_F32)) return MMHIP_E_INVALID;
    GemmTNProblem p{A, B, C, M, Nn, Nc, lda, ldb, ldc, 0, colsum};
    CHECK_HIP(launch_gemm_tn(&p, 1, accumulate, dtype, force_slow, (hipStream_t)stream));
    return 0;
}

int mmhip_op_gemm_tn_group(int dtype, const mmhip_tn_problem* problems, int count, int accumulate, void* stream) {
    if (!problems || count < 0 || (dtype != MMHIP_BF16 && dtype != MMHIP_F16 && dtype != MMHIP_F32)) return MMHIP_E_INVALID;
    std::vector<GemmTNProblem> ps((size_t)count);
    for (int i = 0; i < count; ++i) {
        const mmhip_tn_problem& q = problems[i];
        if (!q.A || !q.B || !q.C || q.M < 1 || q.Nn < 1 || q.Nc < 1) return MMHIP_E_INVALID;
        ps[i] = GemmTNProblem{q.A, q.B, q.C, q.M, q.Nn, q.Nc, q.lda, q.ldb, q.ldc, 0, q.colsum};
    }
    if (count) CHECK_HIP(launch_gemm_tn(ps.data(), count, accumulate, dtype, 0, (hipStream_t)stream));
    return 0;
}

int mmhip_op_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                           int rows, int width, float eps, void* stream) {
    if (!x || !y || !gamma || !beta) return MMHIP_E_INVALID;
    LNArgs a{x, y, gamma, beta, mean, rstd, rows, width, width, width, eps};
    CHECK_HIP(launch_layernorm_fwd(a, dtype, (hipStream_t)stream));
    return 0;
}
int mmhip_op_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                           void* dx, const void* dres, float* dgamma, float* dbeta, int rows, int width, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) return MMHIP_E_INVALID;
    LNBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.x = x; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.dx = dx; a.dres = dres; a.dgamma = dgamma; a.dbeta = dbeta;
    a.rows = rows; a.width = width; a.alpha = 1.0f;
    CHECK_HIP(launch_layernorm_bwd(a, dtype, (hipStream_t)stream));
    return 0;
}
int mmhip_op_attn_fwd(int dtype, const void* qkv, const float* maskbias, void* ctx, float* lse, int posts, int S, int heads,
                      float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    if (!qkv || !ctx || posts < 1 || S < 1 || heads < 1) return MMHIP_E_INVALID;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.maskbias = maskbias; a.ctx = ctx; a.lse = lse; a.posts = posts; a.S = S; a.heads = heads;
    a.hidden = heads * 64; a.ld_qkv = 3 * a.hidden; a.ld_ctx = a.hidden; a.scale = 0.125f;
    a.drop = drop_of(p_drop, seed, stream_id);
    CHECK_HIP(launch_attn_fwd(a, dtype, (hipStream_t)stream));
    return 0;
}
int mmhip_op_attn_bwd(int dtype, const void* qkv, const float* maskbias, const void* ctx, const void* dctx, const float* lse,
                      void* dqkv, int posts, int S, int heads, float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    if (!qkv || !ctx || !dctx || !lse || !dqkv || posts < 1 || S < 1 || heads < 1) return MMHIP_E_INVALID;
    AttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.maskbias = maskbias; a.ctx = ctx; a.dctx = dctx; a.lse = lse; a.dqkv = dqkv; a.posts = posts; a.S = S; a.heads = heads;
    a.hidden = heads * 64; a.ld_qkv = 3 * a.hidden; a.ld_ctx = a.hidden; a.scale = 0.125f;
    a.drop = drop_of(p_drop, seed, stream_id);
    CHECK_HIP(launch_attn_bwd(a, dtype, (hipStream_t)stream));
    return 0;
}
int mmhip_op_colsum(int dtype, const void* x, int rows, int cols, int ld, float* out, void* stream) {
    if (!x || !out) return MMHIP_E_INVALID;
    CHECK_HIP(launch_colsum(x, rows, cols, ld, out, dtype, (hipStream_t)stream));
    return 0;
}
int mmhip_op_cast(int dtype, const float* src, void* dst, uint64_t n, int transpose_rows, int transpose_cols, void* stream) {
    if (!src || !dst) return MMHIP_E_INVALID;
    if (transpose_rows > 0) CHECK_HIP(launch_cast_transpose(src, dst, transpose_rows, transpose_cols, dtype, (hipStream_t)stream));
    else CHECK_HIP(launch_cast(src, dst, n, dtype, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
