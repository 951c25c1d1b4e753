// bf16/fp16 MFMA GEMMs for gfx950.
//
//   gemm_nt   C[M,N]  = epilogue(A[M,K] . B[N,K]^T)        (forward Linear: x . W^T; dX with a W^T copy)
//   gemm_tn   C[Nn,Nc] = A[M,Nn]^T . B[M,Nc]   (fp32 out)  (weight gradients dW = dY^T . X), grouped, no split-K
//   simple_*  fp32-accumulate fallbacks for shapes the tiled kernels do not take (tiny heads, ragged tests)
//
// Tiling: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered.
// LDS images are XOR-swizzled on the *source* address (the LDS-DMA destination is lane-linear) and on the read.
#include <cstdio>
#include <cstdlib>
#include "mmhip_common.h"
#include <cstdlib>
#include <type_traits>
#include "mmhip_kernels.h"

namespace mmhip {

// ------------------------------------------------------------------------------------------------ NT
// Tile variants (BM x BN, waves WM x WN, each wave (BM/WM) x (BN/WN)):
//   128x128, 2x2 (256 thr, 64 KB operand LDS, 2 blocks/CU)   64 FLOP per operand byte staged
//   256x128, 4x2 (512 thr, 96 KB, 1 block/CU)                85
//   256x256, 2x4 (512 thr, 128 KB, 1 block/CU)               128
// At the full MFMA rate a CU consumes 2 * 9.8 TFLOP/s; the L1/TA path delivers ~64 B/clk, so 128x128 is fill-bound.
static constexpr int BK = 64;

// counted wait: at most N of this wave's vector-memory operations (LDS-DMA loads here) may still be in flight
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NL > 0: role-specialised variant -- WM*WN consumer waves (ds_read + MFMA only) and NL loader waves that stream the
// operand tiles into an NS-deep LDS ring with LDS-DMA (counted vmcnt); one raw s_barrier per k-step couples them.
template <int BM, int BN, int WM, int WN, int NS, int NL = 0>
struct NTCfg {
    static constexpr int NW = WM * WN, NLOAD = NL ? NL : NW, NTHR = (NW + NL) * 64;
    static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int AI = BM / 8 / NLOAD, BI = BN / 8 / NLOAD;  // LDS-DMA wave-instructions per loading wave per stage
    static constexpr int EP_LD = TN + 4;                            // fp32 staging row stride (floats)
    static constexpr int EP_WAVE = 32 * EP_LD * 4;                  // 32-row chunk per wave
    static constexpr int LDS = (NS * STAGE > NW * EP_WAVE) ? NS * STAGE : NW * EP_WAVE;
    static constexpr int BLOCKS_PER_CU = (LDS <= 80 * 1024) ? 2 : 1;
    static constexpr int LPS = AI + BI;                             // LDS-DMA instructions per wave per stage
};

template <typename T, int BM, int BN, int WM, int WN, int NS, int NL = 0, bool SK = false>
__global__ __launch_bounds__((WM * WN + NL) * 64, (NTCfg<BM, BN, WM, WN, NS, NL>::BLOCKS_PER_CU * (WM * WN + NL)) / 4)
void gemm_nt_kernel(GemmNTArgs a) {
    using C = NTCfg<BM, BN, WM, WN, NS, NL>;
    if constexpr (SK) {      // split-K: slice blockIdx.y of K, plain fp32 partial products [slice][M][N] (launch_nt_splitk)
        const int z = blockIdx.y, Ks = a.K / (int)gridDim.y;
        a.A = (const T*)a.A + (size_t)z * Ks;
        a.B = (const T*)a.B + (size_t)z * Ks;
        a.K = Ks;
        a.C = a.splitk_ws + (size_t)z * a.M * a.N;
        a.ldc = a.N;
        a.flags = GEMM_OUT_F32;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    // tile raster: column groups of GW tiles, all row bands inside a group before the next group, and each XCD gets a
    // contiguous run of that order -- the blocks co-resident on one XCD then share <= GW weight panels and a few row
    // bands, which fit its 4 MB L2 (PMC: 74 % -> L2 hit rate with the plain row-major order on N = 3072).
    const int tilesN = a.N / BN, tilesM = (a.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GW = 8;
    const int per_group = tilesM * GW;
    const int grp = id / per_group, rem = id - grp * per_group;
    const int gw = min(GW, tilesN - grp * GW);
    const int m0 = (rem / gw) * BM, n0 = (grp * GW + rem % gw) * BN;
    const T* __restrict__ A = (const T*)a.A;
    const T* __restrict__ B = (const T*)a.B;

    // ---- LDS-DMA staging: one wave instruction = 8 tile rows x 128 B; lane -> (row, 16-B slot)
    const int lrow = lane >> 3, slot = lane & 7;
    const int lw = NL ? (w - C::NW) & (C::NLOAD - 1) : w;          // index among the loading waves
    const T* asrc[C::AI];
    const T* bsrc[C::BI];
#pragma unroll
    for (int i = 0; i < C::AI; ++i) {
        const int row = (lw * C::AI + i) * 8 + lrow;
        const int chunk = slot ^ (row & 7);                 // source-side swizzle
        const int gm = min(m0 + row, a.M - 1);              // rows past M read a valid row, never stored
        asrc[i] = A + (size_t)gm * a.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < C::BI; ++i) {
        const int row = (lw * C::BI + i) * 8 + lrow;
        const int chunk = slot ^ (row & 7);
        bsrc[i] = B + (size_t)(n0 + row) * a.ldb + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* base = smem + buf * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::AI; ++i)
            __builtin_amdgcn_global_load_lds(MM_GLB(asrc[i] + k0), MM_LDS(base + (lw * C::AI + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::BI; ++i)
            __builtin_amdgcn_global_load_lds(MM_GLB(bsrc[i] + k0), MM_LDS(base + C::A_BYTES + (lw * C::BI + i) * 1024), 16, 0, 0);
    };

    f32x4 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base16 + (lane&15); 16-B chunk = kk*4 + (lane>>4), XOR (row&7) == (lane&7)
    const int frag_row = (lane & 15) * 128;
    const int sw = lane & 7, kc = lane >> 4;
    const int nk = a.K / BK;
    auto compute = [&](int buf) {
        const char* As = smem + buf * C::STAGE + wm * (C::TM * 128) + frag_row;
        const char* Bs = smem + buf * C::STAGE + C::A_BYTES + wn * (C::TN * 128) + frag_row;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + kc) ^ sw) << 4;
            v8 af[C::FM], bf[C::FN];
#pragma unroll
            for (int j = 0; j < C::FN; ++j) bf[j] = lds_read8<T>(Bs, j * (16 * 128) + coff);
#pragma unroll
            for (int i = 0; i < C::FM; ++i) af[i] = lds_read8<T>(As, i * (16 * 128) + coff);
#pragma unroll
            for (int i = 0; i < C::FM; ++i)
#pragma unroll
                for (int j = 0; j < C::FN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
    };
    if constexpr (NL > 0) {
        // ---- role-specialised ring.  Step t: loaders wait until their share of tile t has landed (at most NS-2 younger
        // tiles still in flight), everybody meets at one s_barrier (tile t complete; buffer (t-1)%NS free), loaders restage
        // that buffer with tile t+NS-1, consumers multiply tile t.
        if (w >= C::NW) {
#pragma unroll
            for (int p = 0; p < NS - 1; ++p)
                if (p < nk) stage(p, p * BK);
            int sbuf = NS - 1;
            for (int t = 0; t < nk; ++t) {
                const int ahead = nk - 1 - t;
                if (ahead >= NS - 2) wait_vmcnt<(NS - 2) * C::LPS>();
                else if (NS > 3 && ahead == 1) wait_vmcnt<C::LPS>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (t + NS - 1 < nk) stage(sbuf, (t + NS - 1) * BK);
                sbuf = (sbuf + 1 == NS) ? 0 : sbuf + 1;
            }
            __builtin_amdgcn_s_barrier();          // matches the consumers' "operand buffers free" barrier below
            return;                                // loaders take no part in the epilogue
        }
        int buf = 0;
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");         // keep the tile's ds_reads below the barrier
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
        }
        __builtin_amdgcn_s_barrier();
    } else if constexpr (NS == 2) {
        // two buffers, one tile of prefetch; __syncthreads drains the LDS-DMA queue (vmcnt(0)) at every step
        stage(0, 0);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) stage(cur ^ 1, (t + 1) * BK);
            compute(cur);
            __syncthreads();
        }
    } else {
        // NS-deep ring, NS-1 tiles of LDS-DMA in flight across the (raw) barriers: per step one counted wait for the
        // oldest tile, one s_barrier (everyone's share of that tile landed, everyone finished reading the buffer that
        // is restaged next), restage, compute.
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nk) stage(p, p * BK);
        int buf = 0, sbuf = NS - 1;
        for (int t = 0; t < nk; ++t) {
            const int ahead = nk - 1 - t;                          // tiles issued after tile t
            if (ahead >= NS - 2) wait_vmcnt<(NS - 2) * C::LPS>();
            else if (NS > 3 && ahead == 1) wait_vmcnt<C::LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nk) stage(sbuf, (t + NS - 1) * BK);
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            sbuf = (sbuf + 1 == NS) ? 0 : sbuf + 1;
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> wave-private fp32 LDS chunk (32 rows) -> row-contiguous 16-B global stores
    float* ep = reinterpret_cast<float*>(smem + w * C::EP_WAVE);
    const int fl = a.flags;
    constexpr int LPR = C::TN / 8;                 // lanes per row (8 outputs = 16 B each)
    constexpr int RPP = 64 / LPR;                  // rows per pass; lanes >= RPP*LPR idle when 64 % LPR != 0
    constexpr int PASSES = (32 + RPP - 1) / RPP;
    const bool ep_lane = lane < RPP * LPR;
    const int c8 = (lane % LPR) * 8;
    const int n = n0 + wn * C::TN + c8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if ((fl & GEMM_BIAS) && ep_lane) {
        f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    // Every global LOAD of the epilogue is issued before its first STORE: vmcnt retires in issue order, so a load issued
    // behind stores can only be consumed once those stores have completed -- one residual load per pass used to serialise
    // the eight passes of a wave on HBM write latency.  `pre` = the residual rows (or, without a residual, the mul_in rows).
    constexpr int NCH = (C::FM + 1) / 2;
    const bool pre_res = (fl & GEMM_RESIDUAL) != 0, pre_mul = !pre_res && (fl & GEMM_MUL_GELU_GRAD);
    v8 pre[NCH][PASSES];
    if (pre_res || pre_mul) {
        const T* pb = pre_res ? (const T*)a.residual : (const T*)a.mul_in;
        const int pld = pre_res ? a.ldres : a.ldmul;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = p * RPP + lane / LPR;
                const int m = m0 + wm * C::TM + ch * 32 + row;
                if (!ep_lane || row >= 32 || ch * 32 + row >= C::TM || m >= a.M) continue;
                pre[ch][p] = *reinterpret_cast<const v8*>(pb + (size_t)m * pld + n);
            }
    }
    // the staging region is wave-private and LDS operations of one wave execute in order: no barrier inside the loop
#pragma unroll
    for (int ch = 0; ch < (C::FM + 1) / 2; ++ch) {     // FM odd (160-row tiles): the last chunk holds one 16-row fragment
        // compiler-level ordering only (other lanes of this wave read what this lane writes): keep chunk ch's reads
        // above chunk ch+1's writes and the writes above the reads; no instruction is emitted
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            if (ch * 2 + ii >= C::FM) break;
#pragma unroll
            for (int j = 0; j < C::FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(ii * 16 + (lane >> 4) * 4 + r) * C::EP_LD + j * 16 + (lane & 15)] = acc[ch * 2 + ii][j][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = p * RPP + lane / LPR;
            const int m = m0 + wm * C::TM + ch * 32 + row;
            if (!ep_lane || row >= 32 || ch * 32 + row >= C::TM || m >= a.M) continue;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + row * C::EP_LD + c8), v1 = *reinterpret_cast<const f32x4*>(ep + row * C::EP_LD + c8 + 4);
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
                for (int e = 0; e < 8; e += 2) mm_gelu2(v[e], v[e + 1]);
            }
            if (fl & GEMM_TANH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
            }
            if (fl & GEMM_QGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = mm_qgelu(v[e]);
            }
            if (fl & GEMM_MUL_GELU_GRAD) {
                v8 u = pre[ch][p];
                if (!pre_mul) u = *reinterpret_cast<const v8*>((const T*)a.mul_in + (size_t)m * a.ldmul + n);
#pragma unroll
                for (int e = 0; e < 8; e += 2) { const f32x2_t gg = mm_gelu_grad2(to_f<T>(u[e]), to_f<T>(u[e + 1])); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
            }
            if ((fl & GEMM_DROPOUT) && a.drop.thresh16) {
                const uint32_t e0 = (uint32_t)m * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    bool k0, k1;
                    mm_keep2(e0 + e, a.drop, k0, k1);
                    v[e] = k0 ? v[e] * a.drop.keep_scale : 0.f;
                    v[e + 1] = k1 ? v[e + 1] * a.drop.keep_scale : 0.f;
                }
            }
            if (fl & GEMM_RESIDUAL) {
                const v8 r = pre[ch][p];
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
}

// ------------------------------------------------------------------------------------------------ TN (grouped)
// C[n][c] = sum_m A[m][n] * B[m][c].  Both operands have the reduction index as their row index, so the MFMA
// fragments are fetched with the hardware transposing read ds_read_b64_tr_b16 from [m][n] / [m][c] LDS images
// (rows of TILE*2 bytes; the low 4 bits of the 16-B chunk index are XOR-ed with ((row&3)<<2 | (row>>2)&3):
// conflict-free for the transposing reads, and a permutation inside each 256-B group so LDS-DMA stays line-friendly).
template <int BNN, int BNC, int WN, int WC, int NS, int NL = 0>
struct TNCfg {
    static constexpr int NW = WN * WC, NLOAD = NL ? NL : NW, NTHR = (NW + NL) * 64;
    static constexpr int A_ROW = BNN * 2, B_ROW = BNC * 2;                 // bytes per reduction row
    static constexpr int A_BYTES = 64 * A_ROW, B_BYTES = 64 * B_ROW, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_RPI = 1024 / A_ROW, B_RPI = 1024 / B_ROW;       // reduction rows per LDS-DMA wave instruction
    static constexpr int AI = 64 / A_RPI / NLOAD, BI = 64 / B_RPI / NLOAD;
    static constexpr int LPS = AI + BI;
    static constexpr int LDS = NS * STAGE;
    static constexpr int BLOCKS_PER_CU = (LDS <= 80 * 1024) ? 2 : 1;
};

template <typename T, int BNN, int BNC, int WN, int WC, int NS, int NL = 0>
__global__ __launch_bounds__((WN * WC + NL) * 64, (TNCfg<BNN, BNC, WN, WC, NS, NL>::BLOCKS_PER_CU * (WN * WC + NL)) / 4)
void gemm_tn_kernel(GemmTNGroup g) {
    using C = TNCfg<BNN, BNC, WN, WC, NS, NL>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn_ = w / WC, wc_ = w % WC;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_TN_MAX_GROUP; ++i)
        if (i < g.count && id >= g.p[i].tile_start) pi = i;
    const GemmTNProblem& P = g.p[pi];
    id -= P.tile_start;
    // Tile walk: consecutive ids (one XCD's chunk, running side by side in lockstep along m) share the band of the LARGER operand, so that band
    // crosses the fabric once instead of once per band of the other operand -- dW2 = df^T . h: B = h is four times A (round 5: its 24 column
    // bands were each fetched by three XCDs; the other three problems of a layer have the larger operand in A and keep the row-major walk)
    const int tilesC = P.Nc / BNC, tilesN = P.Nn / BNN;
    const bool c_major = P.Nc > P.Nn;
    const int n0 = (c_major ? id % tilesN : id / tilesC) * BNN, c0 = (c_major ? id / tilesN : id % tilesC) * BNC;
    const T* __restrict__ A = (const T*)P.A;
    const T* __restrict__ B = (const T*)P.B;

    auto fsw = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
    // staging: lane -> (row within the instruction, 16-B slot within the row); source chunk = slot ^ f(row)
    const int lw = NL ? (w - C::NW) & (C::NLOAD - 1) : w;                    // index among the loading waves
    const T* asrc[C::AI];
    const T* bsrc[C::BI];
#pragma unroll
    for (int i = 0; i < C::AI; ++i) {
        constexpr int SPR = C::A_ROW / 16;                                  // slots per row
        const int row = (lw * C::AI + i) * C::A_RPI + lane / SPR, slot = lane % SPR;
        asrc[i] = A + (size_t)row * P.lda + n0 + (slot ^ fsw(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < C::BI; ++i) {
        constexpr int SPR = C::B_ROW / 16;
        const int row = (lw * C::BI + i) * C::B_RPI + lane / SPR, slot = lane % SPR;
        bsrc[i] = B + (size_t)row * P.ldb + c0 + (slot ^ fsw(row)) * 8;
    }
    // parity mode, plane pairs (mmhip_kernels.h): the M rows are walked three times -- (A hi, B hi), (A lo, B hi), (A hi, B lo)
    const int msteps = P.M / 64;
    const int npr = P.pair ? (P.nprod == 2 ? 2 : 3) : 1;          // products per 64-row slice (GemmTNProblem::nprod; 1 comes in as plain hi planes)
    auto stage = [&](int buf, int mstep) {
        char* base = smem + buf * C::STAGE;
        size_t ao, bo;
        if (P.pair) {
            // the three products of one 64-row slice follow each other (second uses of a slice served from L2), g.pair_serial = the round-3 order
            const int seg = g.pair_serial ? mstep / msteps : mstep % npr, r = g.pair_serial ? mstep - seg * msteps : mstep / npr;
            ao = (size_t)r * 64 * P.lda + (seg == 1 ? (size_t)P.a_lo : 0);
            bo = (size_t)r * 64 * P.ldb + (seg == 2 ? (size_t)P.b_lo : 0);
        } else {
            ao = g.accumulate == 2 ? 0 : (size_t)mstep * 64 * P.lda;
            bo = g.accumulate == 2 ? 0 : (size_t)mstep * 64 * P.ldb;
        }
#pragma unroll
        for (int i = 0; i < C::AI; ++i)
            __builtin_amdgcn_global_load_lds(MM_GLB(asrc[i] + ao), MM_LDS(base + (lw * C::AI + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::BI; ++i)
            __builtin_amdgcn_global_load_lds(MM_GLB(bsrc[i] + bo), MM_LDS(base + C::A_BYTES + (lw * C::BI + i) * 1024), 16, 0, 0);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of A (bias gradient): the waves of the first column tile multiply their A fragments by a B fragment of
    // ones as well -- every column of that product is sum_m A[m][n]
    const bool with_colsum = P.colsum != nullptr && c0 == 0 && wc_ == 0;
    f32x4 accb[4];
    v8 ones;
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = from_f<T>(1.0f);

    // transposed-read addressing: lane = 16 g + 4 q + p; rows 8g+q (and +4), columns col16 + 4p..4p+3
    const int gq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    bool cs_now = with_colsum;
    auto compute = [&](int buf) {
        const char* As = smem + buf * C::STAGE;
        const char* Bs = As + C::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r0 = kk * 32 + gq * 8 + q4, r1 = r0 + 4;
            const int f0 = fsw(r0), f1 = fsw(r1);
            v8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = (wn_ * 64 + i * 16 + p4 * 4) >> 3, in = (p4 & 1) * 8;
                af[i] = join_tr<T>(lds_read_tr4(As, r0 * C::A_ROW + ((ch ^ f0) << 4) + in), lds_read_tr4(As, r1 * C::A_ROW + ((ch ^ f1) << 4) + in));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = (wc_ * 64 + j * 16 + p4 * 4) >> 3, in = (p4 & 1) * 8;
                bf[j] = join_tr<T>(lds_read_tr4(Bs, r0 * C::B_ROW + ((ch ^ f0) << 4) + in), lds_read_tr4(Bs, r1 * C::B_ROW + ((ch ^ f1) << 4) + in));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
            if (cs_now) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = mfma16(af[i], ones, accb[i]);
            }
        }
    };
    const int nsteps = npr * msteps;
    // the column sums may cover the first rows only (parity mode: the hi and lo planes of A = the first two of the three passes)
    const int cs_steps = P.pair ? nsteps : (P.colsum_rows > 0 ? P.colsum_rows / 64 : nsteps);
    const bool pair_ilv = P.pair && !g.pair_serial;          // (pairs: the column sums take A hi and A lo = the first two of every three steps)
    if constexpr (NL > 0) {
        // role-specialised ring (see gemm_nt_kernel): loader waves stream, consumer waves multiply, one barrier per step
        if (w >= C::NW) {
#pragma unroll
            for (int p = 0; p < NS - 1; ++p)
                if (p < nsteps) stage(p, p);
            int sbuf = NS - 1;
            for (int t = 0; t < nsteps; ++t) {
                const int ahead = nsteps - 1 - t;
                if (ahead >= NS - 2) wait_vmcnt<(NS - 2) * C::LPS>();
                else if (NS > 3 && ahead == 1) wait_vmcnt<C::LPS>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (t + NS - 1 < nsteps) stage(sbuf, t + NS - 1);
                sbuf = (sbuf + 1 == NS) ? 0 : sbuf + 1;
            }
            return;
        }
        int buf = 0;
        for (int t = 0; t < nsteps; ++t) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cs_now = with_colsum && t < cs_steps && (!P.pair || (pair_ilv ? t % npr != 2 : t < 2 * msteps));
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
        }
    } else if constexpr (NS == 2) {
        stage(0, 0);
        __syncthreads();
        for (int t = 0; t < nsteps; ++t) {
            const int cur = t & 1;
            if (t + 1 < nsteps) stage(cur ^ 1, t + 1);
            cs_now = with_colsum && t < cs_steps && (!P.pair || (pair_ilv ? t % npr != 2 : t < 2 * msteps));
            compute(cur);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nsteps) stage(p, p);
        int buf = 0, sbuf = NS - 1;
        for (int t = 0; t < nsteps; ++t) {
            const int ahead = nsteps - 1 - t;
            if (ahead >= NS - 2) wait_vmcnt<(NS - 2) * C::LPS>();
            else if (NS > 3 && ahead == 1) wait_vmcnt<C::LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nsteps) stage(sbuf, t + NS - 1);
            cs_now = with_colsum && t < cs_steps && (!P.pair || (pair_ilv ? t % npr != 2 : t < 2 * msteps));
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            sbuf = (sbuf + 1 == NS) ? 0 : sbuf + 1;
        }
    }
    // D[row = n][col = c]: col = lane&15, row = 4*(lane>>4) + reg
    float* Cp = P.C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn_ * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int c = c0 + wc_ * 64 + j * 16 + (lane & 15);
                float* dst = Cp + (size_t)n * P.ldc + c;
                if (g.accumulate == 1) atomicAdd(dst, acc[i][j][r] * g.alpha);
                else *dst = acc[i][j][r] * g.alpha;
            }
    if (with_colsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = P.colsum + n0 + wn_ * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (g.accumulate == 1) atomicAdd(dst, accb[i][r] * g.alpha);
                else *dst = accb[i][r] * g.alpha;
            }
    }
}

// ------------------------------------------------------------------------------------------------ small fp32 GEMM
// Heads (B = 64..128 posts as the row count; 2-4 labels): exact fp32 on the matrix cores.
//   out[m][n] (+)= act( sum_k A(m,k) * B(k,n) + bias[n] ),   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn]
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit-exact fp32): lane l holds A[l&31][k = l>>5], B[k = l>>5][l&31].
// A workgroup owns one 32x32 output tile; its 8 waves split K in 8-wide slices (wave w takes slices w, w+8, ...),
// each lane fetching 4 consecutive k per operand (one 16-byte load when that operand is k-contiguous), and the four
// partial tiles are summed through LDS.  K order inside a slice is permuted identically for A and B.
static constexpr int SG_WAVES = 8;       // the K loop is latency-bound (two dependent global loads per step): split it wide
template <typename TA>
__global__ __launch_bounds__(SG_WAVES * 64) void small_gemm_kernel(SmallGemmArgs a) {
    __shared__ float red[SG_WAVES][32][32];      // lanes index the last dimension: conflict-free without padding
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int m = min(m0 + r, a.M - 1), n = min(n0 + r, a.N - 1);       // out-of-range rows/cols compute garbage, never stored
    const TA* Ap = (const TA*)a.A + (size_t)m * a.sam;
    const float* Bp = a.W + (size_t)n * a.sbn;
    f32x16 acc = f32x16{};
    for (int k0 = w * 8 + h * 4; k0 < a.K; k0 += 8 * SG_WAVES) {
        float av[4], bv[4];
        if (a.sak == 1 && k0 + 3 < a.K && sizeof(TA) == 4 && ((((uintptr_t)(Ap + k0)) & 15) == 0)) {
            f32x4 t = *reinterpret_cast<const f32x4*>(Ap + k0);
            av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = (k0 + e < a.K) ? (float)Ap[(size_t)(k0 + e) * a.sak] : 0.f;
        }
        if (a.sbk == 1 && k0 + 3 < a.K && ((((uintptr_t)(Bp + k0)) & 15) == 0)) {
            f32x4 t = *reinterpret_cast<const f32x4*>(Bp + k0);
            bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (k0 + e < a.K) ? Bp[(size_t)(k0 + e) * a.sbk] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
    // D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[w][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] = acc[reg];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += SG_WAVES * 64) {
        const int row = i >> 5, col = i & 31;
        const int gm = m0 + row, gn = n0 + col;
        if (gm < a.M && gn < a.N) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < SG_WAVES; ++ww) v += red[ww][row][col];
            if (a.bias) v += a.bias[gn];
            if (a.act == ACT_TANH) v = tanhf(v);
            else if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
            float* o = a.out + (size_t)gm * a.ldo + gn;
            *o = a.accumulate ? *o + v : v;
        }
    }
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
    if (fl & GEMM_QGELU) v = mm_qgelu(v);
    if (fl & GEMM_MUL_GELU_GRAD) v *= mm_gelu_grad(to_f<T>(((const T*)a.mul_in)[(size_t)m * a.ldmul + n]));
    if ((fl & GEMM_DROPOUT) && a.drop.thresh16) v = mm_keep((uint32_t)m * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.N + (uint32_t)n, a.drop) ? v * a.drop.keep_scale : 0.f;
    if (fl & GEMM_RESIDUAL) v += to_f<T>(((const T*)a.residual)[(size_t)m * a.ldres + n]);
    if (fl & GEMM_OUT_F32) ((float*)a.C)[(size_t)m * a.ldc + n] = v;
    else ((T*)a.C)[(size_t)m * a.ldc + n] = from_f<T>(v);
}
template <typename T>
__global__ __launch_bounds__(256) void slow_tn_kernel(GemmTNProblem P, int accumulate, float alpha) {
    const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (c >= P.Nc) return;
    const T* A = (const T*)P.A;
    const T* B = (const T*)P.B;
    float s = 0.f;
    for (int m = 0; m < P.M; ++m) s += to_f<T>(A[(size_t)m * P.lda + n]) * to_f<T>(B[(size_t)m * P.ldb + c]);
    float* dst = P.C + (size_t)n * P.ldc + c;
    *dst = accumulate ? *dst + s * alpha : s * alpha;
}

// ------------------------------------------------------------------------------------------------ launchers
static bool nt_fast_ok(const GemmNTArgs& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return (a.N % 128 == 0 || a.N % 192 == 0 || a.N % 96 == 0) && a.K % BK == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && al(a.A) && al(a.B) && al(a.C) &&
           (!(a.flags & GEMM_RESIDUAL) || (a.ldres % 8 == 0 && al(a.residual))) &&
           (!(a.flags & GEMM_AUX_PRE) || (a.ldaux % 8 == 0 && al(a.aux))) &&
           (!(a.flags & GEMM_MUL_GELU_GRAD) || (a.ldmul % 8 == 0 && al(a.mul_in))) &&
           (!(a.flags & GEMM_BIAS) || al(a.bias)) && a.M > 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int NL = 0>
static void launch_nt_t(const GemmNTArgs& a, hipStream_t s) {
    using C = NTCfg<BM, BN, WM, WN, NS, NL>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BM, BN, WM, WN, NS, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); done = true; }
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WM, WN, NS, NL>), dim3(grid), dim3(C::NTHR), C::LDS, s, a);
}

// few rows, long K: K / 384 slices side by side, then the epilogue over their sum (two launches instead of one 128-row block per
// 128 columns walking all of K: 64 x 768 x 3072 took 54 us)
static int splitk_slices(const GemmNTArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MMHIP_SPLITK"); on = e ? atoi(e) : 1; }
    if (!on || !a.splitk_ws || a.tile || a.M > 128 || a.K < 1536 || a.K % 384 || a.N % 128 || (a.flags & (GEMM_TANH | GEMM_QGELU))) return 0;
    return a.K / 384;
}
template <typename T>
static void launch_nt_splitk(const GemmNTArgs& a, int slices, hipStream_t s) {
    using C = NTCfg<128, 128, 2, 2, 2, 0>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, 128, 128, 2, 2, 2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); done = true; }
    const int grid = ((a.M + 127) / 128) * (a.N / 128);
    hipLaunchKernelGGL((gemm_nt_kernel<T, 128, 128, 2, 2, 2, 0, true>), dim3(grid, slices), dim3(C::NTHR), C::LDS, s, a);
}

// MMHIP_TILE_MAP="MxNxK:tile,MxNxK/FLAGS:tile,..." -- per-shape tile override for same-box A/B runs of the whole step (an entry with
// /FLAGS matches that epilogue flag set only)
struct TileMapEntry { int M, N, K, flags, tile; };
static int tile_map_lookup(const GemmNTArgs& a) {
    static TileMapEntry tab[32];
    static int n = -1;
    if (n < 0) {
        n = 0;
        const char* e = getenv("MMHIP_TILE_MAP");
        while (e && *e && n < 32) {
            TileMapEntry t{0, 0, 0, -1, 0};
            int used = 0;
            if (sscanf(e, "%dx%dx%d/%d:%d%n", &t.M, &t.N, &t.K, &t.flags, &t.tile, &used) == 5 && used > 0) { tab[n++] = t; }
            else { t.flags = -1; if (sscanf(e, "%dx%dx%d:%d%n", &t.M, &t.N, &t.K, &t.tile, &used) == 4 && used > 0) tab[n++] = t; else break; }
            e += used;
            if (*e == ',') ++e;
        }
    }
    for (int i = 0; i < n; ++i)
        if (tab[i].M == a.M && tab[i].N == a.N && tab[i].K == a.K && (tab[i].flags < 0 || tab[i].flags == a.flags)) return tab[i].tile;
    return 0;
}
// tile choice: explicit (a.tile / MMHIP_NT_TILE / MMHIP_TILE_MAP) or measured rules.  Tiles: 1 = 128x128 (two blocks per CU), 6 = 128x192,
// 9 = role-specialised 256x128, 10 = 128x96, 12 = role-specialised 256x96, 20 / 21 = 128x128 on a 4- / 3-deep ring (gemm.hip); 13-18 = the deep-pipelined kernel of gemm8.hip
// (13 / 15 = 256x256 one-shot / persistent, 14 / 16 = 256x128, 17 / 18 = 256x192)
static int choose_nt_tile(const GemmNTArgs& a) {
    { const int m = tile_map_lookup(a); if (m) return m; }
    static int env = -1;
    if (env < 0) { const char* e = getenv("MMHIP_NT_TILE"); env = e ? atoi(e) : 0; }
    int t = a.tile ? a.tile : env;
    if ((t == 13 || t == 15) && a.N % 256) t = 0;
    if ((t == 17 || t == 18 || t == 6) && a.N % 192) t = 0;
    if ((t == 14 || t == 16 || t == 1 || t == 9) && a.N % 128) t = 0;
    if ((t == 10 || t == 12) && a.N % 96) t = 0;
    if ((t == 20 || t == 21) && a.N % 128) t = 0;
    if (t == 1 || t == 6 || t == 9 || t == 10 || t == 12 || (t >= 13 && t <= 18) || t == 20 || t == 21) return t;
    // Rules measured inside the training step (same-box A/B of bench.py, profiles/r02_step_ab*.txt, r03_*): the image-tower-sized GEMMs
    // (M >= 12000 rows) take the deep-pipelined persistent tile that fills the rounds of 256 workgroups best; the 8192-row text GEMMs
    // outside the forward's CU partition (i.e. the backward's) keep 128 x 128 at two blocks per CU, and the long-K 768-wide ones the
    // role-specialised tiles.  MMHIP_NT8=0 turns the deep-pipelined kernel off here; MMHIP_NT8_MINM moves the row threshold.
    static int nt8 = -1, nt8_minm = -1;
    if (nt8 < 0) { const char* e = getenv("MMHIP_NT8"); nt8 = e ? atoi(e) : 1; }
    if (nt8_minm < 0) { const char* e = getenv("MMHIP_NT8_MINM"); nt8_minm = e ? atoi(e) : 12000; }
    if (nt8 && a.M >= nt8_minm && a.N % 128 == 0 && a.K % 64 == 0) {
        const long tm = (a.M + 255) / 256;
        const long t256 = a.N % 256 == 0 ? tm * (a.N / 256) : 0, t128 = tm * (a.N / 128);
        const double u256 = t256 ? (double)t256 / (double)(((t256 + 255) / 256) * 256) : 0.0;
        const double u128 = (double)t128 / (double)(((t128 + 255) / 256) * 256);
        const bool narrow_long = a.N <= 768 && a.K >= 2048 && a.M <= 8192;
        if (!narrow_long) {
            // 256 x 192 where it fills clearly more of the chip than 256 x 256 (M = 12608, N = 768: 200 tiles instead of 150)
            if (a.N % 192 == 0) {
                const long t192 = tm * (a.N / 192);
                const double u192 = (double)t192 / (double)(((t192 + 255) / 256) * 256);
                if (u192 >= u256 + 0.1 && u192 >= 0.30) return t192 > 256 ? 18 : 17;
            }
            if (u256 >= 0.30) return 15;
            if (u128 >= 0.70) return 16;
        }
    }
    // long K on a grid that leaves one block per CU at most (config 5's 4096- and 1152-row GEMMs): nothing else on the CU hides the
    // 2-stage variant's load latency; the 3-deep ring does (microbench profiles/r04_early_gemm.txt: 4096x768x3072 40.1 -> 35.0 us,
    // 1152x768x3072 40.2 -> 33.4; K = 768 shapes are unchanged -- their 13-15 us are launch + prologue + epilogue).  MMHIP_RING3=0: off
    static int ring3 = -1;
    if (ring3 < 0) { const char* e = getenv("MMHIP_RING3"); ring3 = e ? atoi(e) : 1; }
    if (ring3 && a.N % 128 == 0 && a.K >= 2048 && (long)((a.M + 127) / 128) * (a.N / 128) <= 256) return 21;
    if (a.N % 128 == 0 && a.N <= 768 && a.K >= 2048 && a.M >= 4096) return 9;
    if (a.N % 128 == 0) return 1;
    if (a.N % 192 == 0) return 6;
    return 10;
}

template <typename T>
static void launch_nt_d(const GemmNTArgs& a, hipStream_t s) {
    const int tile = choose_nt_tile(a);
    if (tile >= 13 && tile <= 18) {      // deep-pipelined persistent-capable tiles (gemm8.hip): 13 / 15 = 256x256, 14 / 16 = 256x128, 17 / 18 = 256x192; even >= 16 and 15: persistent
        const int dt = sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? DT_BF16 : DT_F16;
        const int bn = tile >= 17 ? 192 : ((tile == 13 || tile == 15) ? 256 : 128);
        if (launch_gemm_nt8(a, dt, bn, tile == 15 || tile == 16 || tile == 18, s)) return;
    }
    switch ((tile >= 13 && tile <= 18) ? 1 : tile) {
        case 20: launch_nt_t<T, 128, 128, 2, 2, 4>(a, s); break;      // 128 x 128 on a 4-deep LDS ring (one block per CU): grids of <= 256 tiles, where
                                                                      // nothing else on the CU hides the load latency of the 2-stage variant
        case 21: launch_nt_t<T, 128, 128, 2, 2, 3>(a, s); break;      // ... 3-deep
        case 12: launch_nt_t<T, 256, 96, 4, 2, 3, 4>(a, s); break;    // role-specialised (8 MFMA + 4 LDS-DMA loader waves), 256 tiles for 8192 x 768: one tile per CU
        case 10: launch_nt_t<T, 128, 96, 2, 2, 2>(a, s); break;       // N that only 96 divides
        case 9: launch_nt_t<T, 256, 128, 4, 2, 3, 4>(a, s); break;    // role-specialised, 3-stage ring
        case 6: launch_nt_t<T, 128, 192, 2, 2, 2>(a, s); break;       // N that only 192 divides
        default: launch_nt_t<T, 128, 128, 2, 2, 2>(a, s); break;      // 128 x 128, two blocks per CU
    }
}

static bool debug_force_slow() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MMHIP_FORCE_SLOW"); v = e ? atoi(e) : 0; }
    return v != 0;
}

static thread_local GemmTimingSink* g_timing_sink = nullptr;
void gemm_timing_sink(GemmTimingSink* sink) { g_timing_sink = sink; }
static hipError_t launch_gemm_nt_untimed(const GemmNTArgs& a, int dtype, hipStream_t s);
hipError_t launch_gemm_nt(const GemmNTArgs& a, int dtype, hipStream_t s) {
    GemmTimingSink* k = g_timing_sink;
    if (!k || k->used >= k->capacity || a.M <= 0 || a.N <= 0) return launch_gemm_nt_untimed(a, dtype, s);
    GemmTimingSink::Ev& ev = k->evs[k->used];
    ev.flops = 2.0 * a.M * (double)a.N * a.K;
    hipError_t r = hipEventRecord(ev.a, s);
    if (r != hipSuccess) return r;
    r = launch_gemm_nt_untimed(a, dtype, s);
    if (r != hipSuccess) return r;
    r = hipEventRecord(ev.b, s);
    if (r == hipSuccess) k->used++;
    return r;
}
static hipError_t launch_gemm_nt_untimed(const GemmNTArgs& a, int dtype, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (dtype == DT_F32) return launch_gemm_nt_x3(a, s);
    if (nt_fast_ok(a) && !a.force_slow && !debug_force_slow()) {
        if (const int slices = splitk_slices(a)) {
            if (dtype == DT_BF16) launch_nt_splitk<bf16_t>(a, slices, s);
            else launch_nt_splitk<f16_t>(a, slices, s);
            return launch_splitk_finish(a, dtype, slices, s);
        }
        if (dtype == DT_BF16) launch_nt_d<bf16_t>(a, s);
        else launch_nt_d<f16_t>(a, s);
    } else {
        dim3 grid((a.N + 255) / 256, a.M);
        if (dtype == DT_BF16) hipLaunchKernelGGL(slow_nt_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(slow_nt_kernel<f16_t>, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

template <typename T, int BNN, int BNC, int WN, int WC, int NS, int NL = 0>
static void launch_tn_t(const GemmTNGroup& g, int tiles, hipStream_t s) {
    using C = TNCfg<BNN, BNC, WN, WC, NS, NL>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, BNN, BNC, WN, WC, NS, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); done = true; }
    hipLaunchKernelGGL((gemm_tn_kernel<T, BNN, BNC, WN, WC, NS, NL>), dim3(tiles), dim3(C::NTHR), C::LDS, s, g);
}

// variant: 1 = 128x128 2-stage, 4 (default, fastest inside the step: 13.5 vs 14.2 ms) = role-specialised 256x128,
// 2 = 128x128 4-stage ring, 3 = 256x128 3-stage ring, 4 = role-specialised 256x128 (8 MFMA + 4 loader waves, 3-stage
// ring), 5 = role-specialised 128x128 (4 + 4, 4-stage ring); env MMHIP_TN_TILE overrides
hipError_t launch_gemm_tn(const GemmTNProblem* probs, int count, int accumulate, int dtype, int force_slow, hipStream_t s, float alpha, void* x3_ws, size_t x3_ws_bytes) {
    if (dtype == DT_F32) return launch_gemm_tn_x3(probs, count, accumulate, s, alpha, force_slow ? nullptr : x3_ws, x3_ws_bytes);
    static int env = -1;
    if (env < 0) { const char* e = getenv("MMHIP_TN_TILE"); env = e ? atoi(e) : 0; }
    int variant = (force_slow >> 4) ? (force_slow >> 4) : (env ? env : 4);
    if (variant == 3 || variant == 4)          // the 256-row tiles need every problem's Nn to be a multiple of 256
        for (int i = 0; i < count; ++i)
            if (probs[i].Nn % 256) { variant = 1; break; }
    force_slow = (force_slow & 1) | (debug_force_slow() ? 1 : 0);
    const int bnn = (variant == 3 || variant == 4) ? 256 : 128;
    GemmTNGroup g;
    g.count = 0;
    g.accumulate = accumulate;
    {
        static int ilv = -1;
        if (ilv < 0) { const char* e = getenv("MMHIP_X3_INTERLEAVE"); ilv = e ? atoi(e) : 1; }
        g.pair_serial = !ilv;
    }
    g.alpha = alpha;
    int tiles = 0;
    auto flush = [&]() {
        if (!g.count) return;
        if (dtype == DT_BF16) {
            if (variant == 4) launch_tn_t<bf16_t, 256, 128, 4, 2, 3, 4>(g, tiles, s);
            else if (variant == 5) launch_tn_t<bf16_t, 128, 128, 2, 2, 4, 4>(g, tiles, s);
            else if (variant == 3) launch_tn_t<bf16_t, 256, 128, 4, 2, 3>(g, tiles, s);
            else if (variant == 2) launch_tn_t<bf16_t, 128, 128, 2, 2, 4>(g, tiles, s);
            else launch_tn_t<bf16_t, 128, 128, 2, 2, 2>(g, tiles, s);
        } else {
            if (variant == 4) launch_tn_t<f16_t, 256, 128, 4, 2, 3, 4>(g, tiles, s);
            else if (variant == 5) launch_tn_t<f16_t, 128, 128, 2, 2, 4, 4>(g, tiles, s);
            else if (variant == 3) launch_tn_t<f16_t, 256, 128, 4, 2, 3>(g, tiles, s);
            else if (variant == 2) launch_tn_t<f16_t, 128, 128, 2, 2, 4>(g, tiles, s);
            else launch_tn_t<f16_t, 128, 128, 2, 2, 2>(g, tiles, s);
        }
        g.count = 0;
        tiles = 0;
    };
    for (int i = 0; i < count; ++i) {
        GemmTNProblem P = probs[i];
        auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        bool fast = !force_slow && P.M > 0 && P.M % 64 == 0 && P.Nn % bnn == 0 && P.Nc % 128 == 0 && P.lda % 8 == 0 && P.ldb % 8 == 0 && al(P.A) && al(P.B);
        if (fast) {
            P.tile_start = tiles;
            tiles += (P.Nn / bnn) * (P.Nc / 128);
            g.p[g.count++] = P;
            if (g.count == GEMM_TN_MAX_GROUP) flush();
        } else if (P.M > 0) {
            dim3 grid((P.Nc + 255) / 256, P.Nn);
            if (dtype == DT_BF16) hipLaunchKernelGGL(slow_tn_kernel<bf16_t>, grid, dim3(256), 0, s, P, accumulate, alpha);
            else hipLaunchKernelGGL(slow_tn_kernel<f16_t>, grid, dim3(256), 0, s, P, accumulate, alpha);
            if (P.colsum) {       // generic shapes: the column sums take their own pass
                if (accumulate != 1) { hipError_t e = hipMemsetAsync(P.colsum, 0, (size_t)P.Nn * 4, s); if (e != hipSuccess) return e; }
                // (colsum_rows: the parity mode's stacked [hi; lo; hi] planes count hi + lo once -- x3.hip)
                hipError_t e = launch_colsum(P.A, P.colsum_rows > 0 ? P.colsum_rows : P.M, P.Nn, P.lda, P.colsum, dtype, s, nullptr, alpha);
                if (e != hipSuccess) return e;
            }
        }
    }
    flush();
    return hipGetLastError();
}

hipError_t launch_small_gemm(const SmallGemmArgs& a, int a_dtype, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    dim3 grid((a.N + 31) / 32, (a.M + 31) / 32);
    if (a_dtype == DT_F32) hipLaunchKernelGGL(small_gemm_kernel<float>, grid, dim3(SG_WAVES * 64), 0, s, a);
    else if (a_dtype == DT_BF16) hipLaunchKernelGGL(small_gemm_kernel<bf16_t>, grid, dim3(SG_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(small_gemm_kernel<f16_t>, grid, dim3(SG_WAVES * 64), 0, s, a);
    return hipGetLastError();
}
// out[M,N] = act(A[M,K] W[N,K]^T + b)
hipError_t launch_small_nt(const SmallGemmArgs& a0, int a_dtype, hipStream_t s) {
    SmallGemmArgs a = a0;
    a.sam = a.lda; a.sak = 1; a.sbk = 1; a.sbn = a.ldw;
    return launch_small_gemm(a, a_dtype, s);
}
// out[M,N] = A[M,K] W[K,N]
hipError_t launch_small_nn(const SmallGemmArgs& a0, hipStream_t s) {
    SmallGemmArgs a = a0;
    a.sam = a.lda; a.sak = 1; a.sbk = a.ldw; a.sbn = 1;
    return launch_small_gemm(a, DT_F32, s);
}
// out[n_rows,N] = A[M,n_rows]^T B[M,N]   (a.M = reduction length, a.W = B fp32)
hipError_t launch_small_tn(const SmallGemmArgs& a0, int b_dtype, int n_rows, hipStream_t s) {
    if (b_dtype != DT_F32) return hipErrorInvalidValue;
    SmallGemmArgs a = a0;
    a.K = a0.M; a.M = n_rows;
    a.sam = 1; a.sak = a0.lda; a.sbk = a0.ldw; a.sbn = 1;
    return launch_small_gemm(a, DT_F32, s);
}

}  // namespace mmhip
