// bf16/fp16 MFMA GEMMs for gfx950.
//
//   gemm_nt   C[M,N]  = epilogue(A[M,K] . B[N,K]^T)        (forward Linear: x . W^T; dX with a W^T copy)
//   gemm_tn   C[Nn,Nc] = A[M,Nn]^T . B[M,Nc]   (fp32 out)  (weight gradients dW = dY^T . X), grouped, no split-K
//   simple_*  fp32-accumulate fallbacks for shapes the tiled kernels do not take (tiny heads, ragged tests)
//
// Tiling: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered.
// LDS images are XOR-swizzled on the *source* address (the LDS-DMA destination is lane-linear) and on the read.
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

// ------------------------------------------------------------------------------------------------ NT
static constexpr int BM = 128, BN = 128, BK = 64;
static constexpr int NT_LDS_BYTES = 4 * 64 * 68 * 4;   // epilogue staging (69632) >= 2 x (16K + 16K) operand buffers

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int tilesN = a.N / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (id / tilesN) * BM, n0 = (id % tilesN) * BN;
    const T* __restrict__ A = (const T*)a.A;
    const T* __restrict__ B = (const T*)a.B;

    // ---- LDS-DMA staging: one wave instruction = 8 tile rows x 128 B; lane -> (row, 16-B slot)
    const int lrow = lane >> 3, slot = lane & 7;
    const T* asrc[4];
    const T* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (w * 4 + i) * 8 + lrow;
        int chunk = slot ^ (row & 7);                       // source-side swizzle
        int gm = min(m0 + row, a.M - 1);                    // rows past M read a valid row, never stored
        asrc[i] = A + (size_t)gm * a.lda + chunk * 8;
        bsrc[i] = B + (size_t)(n0 + row) * a.ldb + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* base = smem + buf * 32768 + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(MM_GLB(asrc[i] + k0), MM_LDS(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(MM_GLB(bsrc[i] + k0), MM_LDS(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base16 + (lane&15); 16-B chunk = kk*4 + (lane>>4), XOR (row&7) == (lane&7)
    const int frag_row = (lane & 15) * 128;
    const int sw = lane & 7, kc = lane >> 4;
    const int nk = a.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(cur ^ 1, (t + 1) * BK);
        const char* As = smem + cur * 32768 + wm * (64 * 128) + frag_row;
        const char* Bs = smem + cur * 32768 + 16384 + wn * (64 * 128) + frag_row;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + kc) ^ sw) << 4;
            v8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds_read8<T>(As, i * (16 * 128) + coff);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = lds_read8<T>(Bs, j * (16 * 128) + coff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> wave-private fp32 LDS tile -> row-contiguous 16-B global stores
    float* ep = reinterpret_cast<float*>(smem + w * (64 * 68 * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(i * 16 + (lane >> 4) * 4 + r) * 68 + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    const int fl = a.flags;
    const int c8 = (lane & 7) * 8;
    const int n = n0 + wn * 64 + c8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (fl & GEMM_BIAS) {
        f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + (lane >> 3);
        const int m = m0 + wm * 64 + row;
        if (m >= a.M) continue;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + row * 68 + c8), v1 = *reinterpret_cast<const f32x4*>(ep + row * 68 + c8 + 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bias8[e]; v[4 + e] = v1[e] + bias8[4 + e]; }
        if (fl & GEMM_AUX_PRE) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f<T>(v[e]);
            *reinterpret_cast<v8*>((T*)a.aux + (size_t)m * a.ldaux + n) = o;
        }
        if (fl & GEMM_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = mm_gelu(v[e]);
        }
        if (fl & GEMM_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
        }
        if (fl & GEMM_MUL_GELU_GRAD) {
            v8 u = *reinterpret_cast<const v8*>((const T*)a.mul_in + (size_t)m * a.ldmul + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= mm_gelu_grad(to_f<T>(u[e]));
        }
        if ((fl & GEMM_DROPOUT) && a.drop.thresh16) {
            const uint32_t e0 = (uint32_t)m * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                bool k0, k1;
                mm_keep2(e0 + e, a.drop, k0, k1);
                v[e] = k0 ? v[e] * a.drop.keep_scale : 0.f;
                v[e + 1] = k1 ? v[e + 1] * a.drop.keep_scale : 0.f;
            }
        }
        if (fl & GEMM_RESIDUAL) {
            v8 r = *reinterpret_cast<const v8*>((const T*)a.residual + (size_t)m * a.ldres + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += to_f<T>(r[e]);
        }
        if (fl & GEMM_OUT_F32) {
            float* c = (float*)a.C + (size_t)m * a.ldc + n;
            *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(c + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f<T>(v[e]);
            *reinterpret_cast<v8*>((T*)a.C + (size_t)m * a.ldc + n) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ TN (grouped)
// C[n][c] = sum_m A[m][n] * B[m][c].  Both operands have the reduction index as their row index, so the MFMA
// fragments are fetched with the hardware transposing read ds_read_b64_tr_b16 from [m][n] / [m][c] LDS images
// (256-B rows, 16-B chunk index XOR-ed with ((row&3)<<2 | (row>>2)&3): conflict-free for the tr reads).
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTNGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn_ = w >> 1, wc_ = w & 1;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_TN_MAX_GROUP; ++i)
        if (i < g.count && id >= g.p[i].tile_start) pi = i;
    const GemmTNProblem& P = g.p[pi];
    id -= P.tile_start;
    const int tilesC = P.Nc / 128;
    const int n0 = (id / tilesC) * 128, c0 = (id % tilesC) * 128;
    const T* __restrict__ A = (const T*)P.A;
    const T* __restrict__ B = (const T*)P.B;

    // staging: one wave instruction = 4 reduction rows x 256 B; lane -> (row, 16-B slot)
    const int lrow = lane >> 4, slot = lane & 15;
    const T* asrc[4];
    const T* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (w * 4 + i) * 4 + lrow;                              // 0..63 within the m-step
        int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
        asrc[i] = A + (size_t)row * P.lda + n0 + chunk * 8;
        bsrc[i] = B + (size_t)row * P.ldb + c0 + chunk * 8;
    }
    auto stage = [&](int buf, int mstep) {
        char* base = smem + buf * 32768 + w * 4096;
        const size_t ao = (size_t)mstep * 64 * P.lda, bo = (size_t)mstep * 64 * P.ldb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(MM_GLB(asrc[i] + ao), MM_LDS(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(MM_GLB(bsrc[i] + bo), MM_LDS(base + 16384 + i * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: lane = 16 g + 4 q + p; rows 8g+q (and +4), columns col16 + 4p..4p+3
    const int gq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int nsteps = P.M / 64;
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        const int cur = t & 1;
        if (t + 1 < nsteps) stage(cur ^ 1, t + 1);
        const char* As = smem + cur * 32768;
        const char* Bs = As + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r0 = kk * 32 + gq * 8 + q4, r1 = r0 + 4;
            const int f0 = ((r0 & 3) << 2) | ((r0 >> 2) & 3), f1 = ((r1 & 3) << 2) | ((r1 >> 2) & 3);
            v8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = (wn_ * 64 + i * 16 + p4 * 4) >> 3, in = (p4 & 1) * 8;
                af[i] = join_tr<T>(lds_read_tr4(As, r0 * 256 + ((ch ^ f0) << 4) + in), lds_read_tr4(As, r1 * 256 + ((ch ^ f1) << 4) + in));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = (wc_ * 64 + j * 16 + p4 * 4) >> 3, in = (p4 & 1) * 8;
                bf[j] = join_tr<T>(lds_read_tr4(Bs, r0 * 256 + ((ch ^ f0) << 4) + in), lds_read_tr4(Bs, r1 * 256 + ((ch ^ f1) << 4) + in));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    // D[row = n][col = c]: col = lane&15, row = 4*(lane>>4) + reg
    float* C = P.C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn_ * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int c = c0 + wc_ * 64 + j * 16 + (lane & 15);
                float* dst = C + (size_t)n * P.ldc + c;
                if (g.accumulate) atomicAdd(dst, acc[i][j][r]);
                else *dst = acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------------ simple fallbacks
// Small / ragged problems (heads with 2-4 outputs, B posts as the row count): plain fp32-accumulate kernels.
// NT: out[m][n] = act(sum_k A[m][k] W[n][k] + b[n]); one wave per output column n, lanes stride over k.
template <typename TA>
__global__ __launch_bounds__(256) void small_nt_kernel(SmallGemmArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const float* W = a.W + (size_t)n * a.ldw;
    const float bn = a.bias ? a.bias[n] : 0.f;
    for (int m = 0; m < a.M; ++m) {
        const TA* x = (const TA*)a.A + (size_t)m * a.lda;
        float s = 0.f;
        for (int k = lane; k < a.K; k += 64) s += (float)x[k] * W[k];
        s = wave_sum(s);
        if (lane == 0) {
            s += bn;
            if (a.act == ACT_TANH) s = tanhf(s);
            else if (a.act == ACT_RELU) s = fmaxf(s, 0.f);
            float* o = a.out + (size_t)m * a.ldo + n;
            *o = a.accumulate ? *o + s : s;
        }
    }
}
// NN: out[m][i] = sum_o A[m][o] W[o][i]   (thread per (m, i); W rows read coalesced)
__global__ __launch_bounds__(256) void small_nn_kernel(SmallGemmArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (i >= a.N) return;
    const float* x = (const float*)a.A + (size_t)m * a.lda;
    float s = 0.f;
    for (int o = 0; o < a.K; ++o) s += x[o] * a.W[(size_t)o * a.ldw + i];
    float* dst = a.out + (size_t)m * a.ldo + i;
    *dst = a.accumulate ? *dst + s : s;
}
// TN: out[n][c] = sum_m A[m][n] B[m][c]    (thread per (n, c); B rows read coalesced; TB = float or 16-bit)
template <typename TB>
__global__ __launch_bounds__(256) void small_tn_kernel(SmallGemmArgs a) {
    const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (c >= a.N) return;
    const float* A = (const float*)a.A;
    const TB* B = (const TB*)a.W;
    float s = 0.f;
    for (int m = 0; m < a.M; ++m) s += A[(size_t)m * a.lda + n] * (float)B[(size_t)m * a.ldw + c];
    float* dst = a.out + (size_t)n * a.ldo + c;
    *dst = a.accumulate ? *dst + s : s;
}
// generic slow NT/TN for 16-bit operands of any shape (used when the MFMA kernels' shape rules do not hold)
template <typename T>
__global__ __launch_bounds__(256) void slow_nt_kernel(GemmNTArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= a.N) return;
    const T* x = (const T*)a.A + (size_t)m * a.lda;
    const T* wv = (const T*)a.B + (size_t)n * a.ldb;
    float v = 0.f;
    for (int k = 0; k < a.K; ++k) v += to_f<T>(x[k]) * to_f<T>(wv[k]);
    const int fl = a.flags;
    if (fl & GEMM_BIAS) v += a.bias[n];
    if (fl & GEMM_AUX_PRE) ((T*)a.aux)[(size_t)m * a.ldaux + n] = from_f<T>(v);
    if (fl & GEMM_GELU) v = mm_gelu(v);
    if (fl & GEMM_TANH) v = tanhf(v);
    if (fl & GEMM_MUL_GELU_GRAD) v *= mm_gelu_grad(to_f<T>(((const T*)a.mul_in)[(size_t)m * a.ldmul + n]));
    if ((fl & GEMM_DROPOUT) && a.drop.thresh16) v = mm_keep((uint32_t)m * (uint32_t)a.N + (uint32_t)n, a.drop) ? v * a.drop.keep_scale : 0.f;
    if (fl & GEMM_RESIDUAL) v += to_f<T>(((const T*)a.residual)[(size_t)m * a.ldres + n]);
    if (fl & GEMM_OUT_F32) ((float*)a.C)[(size_t)m * a.ldc + n] = v;
    else ((T*)a.C)[(size_t)m * a.ldc + n] = from_f<T>(v);
}
template <typename T>
__global__ __launch_bounds__(256) void slow_tn_kernel(GemmTNProblem P, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (c >= P.Nc) return;
    const T* A = (const T*)P.A;
    const T* B = (const T*)P.B;
    float s = 0.f;
    for (int m = 0; m < P.M; ++m) s += to_f<T>(A[(size_t)m * P.lda + n]) * to_f<T>(B[(size_t)m * P.ldb + c]);
    float* dst = P.C + (size_t)n * P.ldc + c;
    *dst = accumulate ? *dst + s : s;
}

// ------------------------------------------------------------------------------------------------ launchers
static bool nt_fast_ok(const GemmNTArgs& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return a.N % BN == 0 && a.K % BK == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && al(a.A) && al(a.B) && al(a.C) &&
           (!(a.flags & GEMM_RESIDUAL) || (a.ldres % 8 == 0 && al(a.residual))) &&
           (!(a.flags & GEMM_AUX_PRE) || (a.ldaux % 8 == 0 && al(a.aux))) &&
           (!(a.flags & GEMM_MUL_GELU_GRAD) || (a.ldmul % 8 == 0 && al(a.mul_in))) &&
           (!(a.flags & GEMM_BIAS) || al(a.bias)) && a.M > 0;
}

hipError_t launch_gemm_nt(const GemmNTArgs& a, int dtype, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (nt_fast_ok(a) && !a.force_slow) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
            attr_done = true;
        }
        const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
        if (dtype == DT_BF16) hipLaunchKernelGGL(gemm_nt_kernel<bf16_t>, dim3(grid), dim3(256), NT_LDS_BYTES, s, a);
        else hipLaunchKernelGGL(gemm_nt_kernel<f16_t>, dim3(grid), dim3(256), NT_LDS_BYTES, s, a);
    } else {
        dim3 grid((a.N + 255) / 256, a.M);
        if (dtype == DT_BF16) hipLaunchKernelGGL(slow_nt_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(slow_nt_kernel<f16_t>, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_tn(const GemmTNProblem* probs, int count, int accumulate, int dtype, int force_slow, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_done = true;
    }
    GemmTNGroup g;
    g.count = 0;
    g.accumulate = accumulate;
    int tiles = 0;
    auto flush = [&]() {
        if (!g.count) return;
        if (dtype == DT_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(tiles), dim3(256), 65536, s, g);
        else hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(tiles), dim3(256), 65536, s, g);
        g.count = 0;
        tiles = 0;
    };
    for (int i = 0; i < count; ++i) {
        GemmTNProblem P = probs[i];
        auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        bool fast = !force_slow && P.M > 0 && P.M % 64 == 0 && P.Nn % 128 == 0 && P.Nc % 128 == 0 && P.lda % 8 == 0 && P.ldb % 8 == 0 && al(P.A) && al(P.B);
        if (fast) {
            P.tile_start = tiles;
            tiles += (P.Nn / 128) * (P.Nc / 128);
            g.p[g.count++] = P;
            if (g.count == GEMM_TN_MAX_GROUP) flush();
        } else if (P.M > 0) {
            dim3 grid((P.Nc + 255) / 256, P.Nn);
            if (dtype == DT_BF16) hipLaunchKernelGGL(slow_tn_kernel<bf16_t>, grid, dim3(256), 0, s, P, accumulate);
            else hipLaunchKernelGGL(slow_tn_kernel<f16_t>, grid, dim3(256), 0, s, P, accumulate);
        }
    }
    flush();
    return hipGetLastError();
}

hipError_t launch_small_nt(const SmallGemmArgs& a, int a_dtype, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    dim3 grid((a.N + 3) / 4);
    if (a_dtype == DT_F32) hipLaunchKernelGGL(small_nt_kernel<float>, grid, dim3(256), 0, s, a);
    else if (a_dtype == DT_BF16) hipLaunchKernelGGL(small_nt_kernel<bf16_t>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(small_nt_kernel<f16_t>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_small_nn(const SmallGemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    hipLaunchKernelGGL(small_nn_kernel, dim3((a.N + 255) / 256, a.M), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_small_tn(const SmallGemmArgs& a, int b_dtype, int n_rows, hipStream_t s) {
    if (n_rows <= 0 || a.N <= 0) return hipSuccess;
    dim3 grid((a.N + 255) / 256, n_rows);
    if (b_dtype == DT_F32) hipLaunchKernelGGL(small_tn_kernel<float>, grid, dim3(256), 0, s, a);
    else if (b_dtype == DT_BF16) hipLaunchKernelGGL(small_tn_kernel<bf16_t>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(small_tn_kernel<f16_t>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mmhip
