// Deep-pipelined NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T), 16-bit operands, fp32 accumulate.
//
// One 512-thread workgroup per CU (8 waves = 4 along M x 2 along N), a 256 x BN output tile (BN = 256 or 128), BK = 64.
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template", re-derived for this tile family):
//   * operand K-tiles live in an LDS ring (2 buffers of 64 KB for BN = 256, 3 of 48 KB for BN = 128), filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KB pieces = 8 tile rows x 128 B) that stays in flight ACROSS barriers: every wait is a
//     counted s_waitcnt vmcnt(N), every barrier a raw s_barrier; no vmcnt(0) in the main loop;
//   * a K-tile is consumed in NPH = BN/64 phases of 16 MFMA 16x16x32 per wave; a phase is two barrier intervals, L (this
//     phase's ds_read_b128 fragment reads + GPP LDS-DMA pieces of a later K-tile) and C (the MFMAs).  Waves 4-7 run one
//     interval behind waves 0-3, so on every SIMD one wave multiplies while its partner reads: the matrix pipe never
//     waits for LDS;
//   * every LDS slot is read in exactly one phase and restaged two phases later (WAR distance the guide asks for); a slot
//     is read one phase after the counted wait + barrier that retired its DMA (RAW);
//   * swapped MFMA operands (D = B_frag . A_frag^T) with the B rows of a fragment pair permuted so that a lane ends up
//     with 8 CONSECUTIVE output columns of one row: the fused epilogue (bias / GELU / gelu' / dropout / residual / aux)
//     works on registers and leaves as 16-byte row stores -- no LDS round trip, no barrier, no LDS space;
//   * persistent: a workgroup walks its list of tiles and the DMA stream runs on across tile boundaries, so the next
//     tile's first K-tiles land while the epilogue of the current one runs (K = 768 is only 12 K-tiles).
#include <type_traits>
#include "mmhip_common.h"
#include "mmhip_kernels.h"

namespace mmhip {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int BN>
struct P8 {
    static constexpr int BM = 256, BK = 64, TN = BN / 2, FN = TN / 16, NPH = BN / 64;
    static constexpr int NBUF = BN == 128 ? 3 : 2;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, KT = A_BYTES + B_BYTES;
    static constexpr int LDS = NBUF * KT;
    static constexpr int GPP = BN == 128 ? 3 : 2;          // LDS-DMA instructions per wave and phase
    static constexpr int NSRC = NPH * GPP;
};

// staging plan.  (phase p, slot jj) -> operand, piece index (a function of the wave), target K-tile offset
//   BN = 256: p=2: A rows 0-127 of K-tile kt+2 | p=3: A rows 128-255 of kt+2 | p=0: B columns of phases 0,1 of kt+1 | p=1: phases 2,3 of kt+1
//   BN = 128: p=0: A rows 0-127 + B columns of phase 0 of kt+2 | p=1: A rows 128-255 + B columns of phase 1 of kt+2
template <int BN> __device__ __forceinline__ constexpr bool st_is_b(int p, int jj) { return BN == 256 ? (p < 2) : (jj == 2); }
template <int BN> __device__ __forceinline__ constexpr int st_ahead(int p) { return BN == 256 ? (p < 2 ? 1 : 2) : 2; }
template <int BN> __device__ __forceinline__ int st_piece(int p, int jj, int w) {
    if (BN == 256) {
        const int q = w * 2 + jj;
        if (p == 2) return q;
        if (p == 3) return 16 + q;
        if (p == 0) return q + (q & 8);
        return q + 8 + (q & 8);
    }
    if (jj < 2) return p * 16 + w * 2 + jj;
    return p * 4 + w + (w & 4);
}

// epilogue classes (compile-time, so that each instance carries only its own code): the flag sets the engine uses
//   EP_PLAIN   [bias] [dropout] [residual]          (QKV / AO / FC2 forward, dX GEMMs)
//   EP_GELU    bias [aux = pre-activation] GELU     (FC1 forward)
//   EP_MULG    * gelu'(mul_in) [residual]           (dFC2)
//   EP_ANY     every flag at run time               (tests, tanh, fp32 output)
enum { EP_PLAIN = 0, EP_GELU = 1, EP_MULG = 2, EP_ANY = 3 };

template <typename T, int EPI>
__device__ __forceinline__ void epilogue8(const GemmNTArgs& a, float* v, int m, int n, const typename Vec<T>::v8& pre) {
    // `pre`: this lane's 8 values of the residual (EP_PLAIN / EP_ANY) or of mul_in (EP_MULG), loaded by the caller BEFORE the
    // first store of the tile: vmcnt retires in issue order, so a load issued behind stores can only be consumed once those
    // stores have completed -- one such load per output fragment serialised the whole epilogue on HBM write latency
    typedef typename Vec<T>::v8 v8;
    const int fl = a.flags;
    if ((EPI == EP_GELU || EPI == EP_ANY) && (fl & GEMM_AUX_PRE)) {
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f<T>(v[e]);
        *reinterpret_cast<v8*>((T*)a.aux + (size_t)m * a.ldaux + n) = o;
    }
    if (EPI == EP_GELU || (EPI == EP_ANY && (fl & GEMM_GELU))) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mm_gelu(v[e]);
    }
    if (EPI == EP_ANY && (fl & GEMM_TANH)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    }
    if (EPI == EP_ANY && (fl & GEMM_QGELU)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mm_qgelu(v[e]);
    }
    if (EPI == EP_MULG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mm_gelu_grad(to_f<T>(pre[e]));
    }
    if (EPI == EP_ANY && (fl & GEMM_MUL_GELU_GRAD)) {
        v8 u = *reinterpret_cast<const v8*>((const T*)a.mul_in + (size_t)m * a.ldmul + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mm_gelu_grad(to_f<T>(u[e]));
    }
    if ((EPI == EP_PLAIN || EPI == EP_ANY) && (fl & GEMM_DROPOUT) && a.drop.thresh16) {
        const uint32_t e0 = (uint32_t)m * (uint32_t)(a.drop_row_mul ? a.drop_row_mul : 1) * (uint32_t)a.N + (uint32_t)n;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            bool k0, k1;
            mm_keep2(e0 + e, a.drop, k0, k1);
            v[e] = k0 ? v[e] * a.drop.keep_scale : 0.f;
            v[e + 1] = k1 ? v[e + 1] * a.drop.keep_scale : 0.f;
        }
    }
    if ((EPI == EP_PLAIN || EPI == EP_ANY) && (fl & GEMM_RESIDUAL)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += to_f<T>(pre[e]);
    }
    if (EPI == EP_MULG && (fl & GEMM_RESIDUAL)) {
        v8 r = *reinterpret_cast<const v8*>((const T*)a.residual + (size_t)m * a.ldres + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += to_f<T>(r[e]);
    }
    if (EPI == EP_ANY && (fl & GEMM_OUT_F32)) {
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

template <typename T, int BN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(GemmNTArgs a, int persistent) {
    using C = P8<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1, grp = w >> 2;
    const int tilesN = a.N / BN, tilesM = (a.M + C::BM - 1) / C::BM, ntiles = tilesN * tilesM;
    const int nk = a.K / C::BK;
    const char* __restrict__ Ab = (const char*)a.A;
    const char* __restrict__ Bb = (const char*)a.B;

    // ---- work list: tiles first, first + stride, ... in the column-group raster (see gemm.hip), XCD-contiguous
    const int nwg = gridDim.x;
    int first, stride, count;
    if (persistent) {
        // workgroups b, b+8, ... share an XCD: XCD x owns a contiguous run of the raster and its workgroups walk it round
        // by round, so the tiles resident on an XCD at any time are neighbours
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = (nwg + 7 - x) / 8;     // workgroups on this XCD
        const int q = ntiles >> 3, r = ntiles & 7;
        const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, n_x = q + (x < r ? 1 : 0);
        first = lo + j; stride = wpx; count = j < n_x ? (n_x - j + wpx - 1) / wpx : 0;
    } else {
        first = xcd_remap(blockIdx.x, nwg); stride = 0; count = 1;
    }
    if (count <= 0) return;
    constexpr int GW = 8;
    auto tile_origin = [&](int id, int& m0, int& n0) {
        const int per_group = tilesM * GW;
        const int g = id / per_group, rem = id - g * per_group;
        const int gw = min(GW, tilesN - g * GW);
        m0 = (rem / gw) * C::BM;
        n0 = (g * GW + rem % gw) * BN;
    };

    // ---- the DMA stream.  K-tiles are numbered globally over the work list (tile t, K-tile k -> t*nk + k); the staging
    // group of phase p issues its n-th stage for global K-tile n, so each group carries its own running state (no
    // division in the loop): per-lane source pointers (advanced 128 bytes per stage, re-based when the group enters the
    // next tile), K-tiles left in its tile, its tile index, its LDS buffer.
    const int lrow = lane >> 3, lslot = lane & 7;
    const int total_kt = count * nk;
    const char* src[C::NPH][C::GPP];
    int g_rem[C::NPH], g_tile[C::NPH], g_buf[C::NPH], g_done[C::NPH];
    auto rebase = [&](auto pc_, int id) {
        constexpr int p = decltype(pc_)::value;
        int m0, n0;
        tile_origin(id, m0, n0);
#pragma unroll
        for (int jj = 0; jj < C::GPP; ++jj) {
            const int pc = st_piece<BN>(p, jj, w), row = pc * 8 + lrow;
            if (st_is_b<BN>(p, jj)) {
                const int sw = (lrow & 3) | ((pc & 1) << 2);
                src[p][jj] = Bb + ((size_t)(n0 + row) * a.ldb + (size_t)((lslot ^ sw) * 8)) * 2;
            } else {
                const int gm = min(m0 + row, a.M - 1);          // rows past M read a valid row, never stored
                src[p][jj] = Ab + ((size_t)gm * a.lda + (size_t)((lslot ^ lrow) * 8)) * 2;
            }
        }
    };
    auto stage = [&](auto pc_) {
        constexpr int p = decltype(pc_)::value;
        if (g_done[p] >= total_kt) return;
        if (g_rem[p] == 0) {
            g_tile[p] += 1;
            g_rem[p] = nk;
            rebase(pc_, first + g_tile[p] * stride);
        }
        char* base = smem + g_buf[p] * C::KT;
#pragma unroll
        for (int jj = 0; jj < C::GPP; ++jj) {
            const int pc = st_piece<BN>(p, jj, w);
            char* dst = base + (st_is_b<BN>(p, jj) ? C::A_BYTES : 0) + pc * 1024;
            __builtin_amdgcn_global_load_lds(MM_GLB(src[p][jj]), MM_LDS(dst), 16, 0, 0);
            src[p][jj] += 128;
        }
        g_rem[p] -= 1;
        g_done[p] += 1;
        g_buf[p] = (g_buf[p] + 1 == C::NBUF) ? 0 : g_buf[p] + 1;
    };
    auto init_group = [&](auto pc_) {
        constexpr int p = decltype(pc_)::value;
        g_rem[p] = nk; g_tile[p] = 0; g_buf[p] = 0; g_done[p] = 0;
        rebase(pc_, first);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    f32x4 acc[4][C::FN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // fragment read offsets.  A tile: row = wm*64 + 16 i + (lane&15); B tile (permuted): row = wn*TN + 32 j' + 8((lane&15)>>2) + 4 h + (lane&3);
    // 16-byte chunk = (kk*4 + (lane>>4)) ^ (lane&7) in both
    const int l15 = lane & 15, kc = lane >> 4, sw7 = lane & 7;
    const int a_row_off = (wm * 64 + l15) * 128;
    const int b_row_off = C::A_BYTES + (wn * C::TN + 8 * (l15 >> 2) + (lane & 3)) * 128;
    const int ch0 = ((0 * 4 + kc) ^ sw7) << 4, ch1 = ((1 * 4 + kc) ^ sw7) << 4;
    v8 af[4][2], bf[2][2];

    // ---- prologue: the groups the steady state would have issued before K-tile 0
    init_group(I0{});
    init_group(I1{});
    if constexpr (BN == 256) {
        init_group(I2{});
        init_group(I3{});
        stage(I2{}); stage(I3{}); stage(I0{}); stage(I1{});
        if (total_kt > 1) { stage(I2{}); stage(I3{}); wait_vm<6>(); }
        else wait_vm<2>();
    } else {
        stage(I0{}); stage(I1{});
        if (total_kt > 1) { stage(I0{}); stage(I1{}); wait_vm<6>(); }
        else wait_vm<0>();
    }
    raw_barrier();
    if (grp == 1) raw_barrier();          // waves 4-7 run one interval behind

    int gkt = 0, cbuf = 0;
    for (int t = 0; t < count; ++t) {
#pragma unroll 1
        for (int k = 0; k < nk; ++k, ++gkt) {
            const char* Ks = smem + cbuf * C::KT;
            cbuf = (cbuf + 1 == C::NBUF) ? 0 : cbuf + 1;
            auto phase = [&](auto pc_) {
                constexpr int p = decltype(pc_)::value;
                // ---- L interval: fragment reads of this phase, DMA of a later K-tile, counted wait
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf[h][0] = lds_read8<T>(Ks, b_row_off + (32 * p + 4 * h) * 128 + ch0);
                    bf[h][1] = lds_read8<T>(Ks, b_row_off + (32 * p + 4 * h) * 128 + ch1);
                }
                if (p == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        af[i][0] = lds_read8<T>(Ks, a_row_off + i * (16 * 128) + ch0);
                        af[i][1] = lds_read8<T>(Ks, a_row_off + i * (16 * 128) + ch1);
                    }
                }
                stage(pc_);
                if (BN == 256) {
                    if (p == 1) { if (gkt + 1 < total_kt) wait_vm<8>(); else wait_vm<0>(); }
                    if (p == 3) { if (gkt + 2 < total_kt) wait_vm<6>(); else if (gkt + 1 < total_kt) wait_vm<2>(); }
                } else {
                    if (p == 1) { if (gkt + 2 < total_kt) wait_vm<6>(); else if (gkt + 1 < total_kt) wait_vm<0>(); }
                }
                raw_barrier();
                // ---- C interval
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) acc[i][2 * p + h] = mfma16(bf[h][kk], af[i][kk], acc[i][2 * p + h]);
                __builtin_amdgcn_s_setprio(0);
                raw_barrier();
            };
            phase(I0{});
            phase(I1{});
            if constexpr (C::NPH == 4) {
                phase(I2{});
                phase(I3{});
            }
        }
        // ---- epilogue of tile t, from registers: lane holds C[m][n .. n+7], m = m0 + wm*64 + 16 i + (lane&15),
        // n = n0 + wn*TN + 32 j' + 8 (lane>>4); values 0-3 from fragment 2j', 4-7 from fragment 2j'+1
        int m0, n0;
        tile_origin(first + t * stride, m0, n0);
        // every load of the epilogue first (bias; residual or mul_in of all the lane's fragments), then compute, then stores.
        // Rows past M: loads read (clamped) row M-1, stores are masked -- C may alias the residual (x += f(x) in the image tower),
        // so a duplicate of row M-1 must never be stored: another wave could read it back as residual.
        const int nb = n0 + wn * C::TN + 8 * kc;
        float bias8[C::FN / 2][8];
#pragma unroll
        for (int jp = 0; jp < C::FN / 2; ++jp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[jp][e] = 0.f;
            if (EPI == EP_GELU || (a.flags & GEMM_BIAS)) {
                f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + nb + 32 * jp), b1 = *reinterpret_cast<const f32x4*>(a.bias + nb + 32 * jp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias8[jp][e] = b0[e]; bias8[jp][4 + e] = b1[e]; }
            }
        }
        // (BN = 256: two chunks of two fragment-pair columns -- 64 more registers for all sixteen residual fragments would spill)
#pragma unroll
        for (int jc = 0; jc < C::FN / 2; jc += 2) {
            v8 pre[4][2];
            const T* pbase = (EPI == EP_MULG) ? (const T*)a.mul_in : (const T*)a.residual;
            const int pld = (EPI == EP_MULG) ? a.ldmul : a.ldres;
            const bool want = (EPI == EP_MULG) || ((EPI == EP_PLAIN || EPI == EP_ANY) && (a.flags & GEMM_RESIDUAL));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    if (EPI != EP_GELU && want) {
                        const int m = min(m0 + wm * 64 + i * 16 + l15, a.M - 1);
                        pre[i][j2] = *reinterpret_cast<const v8*>(pbase + (size_t)m * pld + nb + 32 * (jc + j2));
                    }
                }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int jp = jc + j2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wm * 64 + i * 16 + l15;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jp][e] + bias8[jp][e]; v[4 + e] = acc[i][2 * jp + 1][e] + bias8[jp][4 + e]; }
                    if (m < a.M) epilogue8<T, EPI>(a, v, m, nb + 32 * jp, pre[i][j2]);
                }
            }
        }
        zero_acc();
    }
    if (grp == 0) raw_barrier();          // matches the extra interval of waves 4-7
}

static bool nt8_ok(const GemmNTArgs& a, int bn) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return a.N % bn == 0 && a.K % 64 == 0 && a.K >= 64 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && al(a.A) && al(a.B) && al(a.C) &&
           (!(a.flags & GEMM_RESIDUAL) || (a.ldres % 8 == 0 && al(a.residual))) &&
           (!(a.flags & GEMM_AUX_PRE) || (a.ldaux % 8 == 0 && al(a.aux))) &&
           (!(a.flags & GEMM_MUL_GELU_GRAD) || (a.ldmul % 8 == 0 && al(a.mul_in))) &&
           (!(a.flags & GEMM_BIAS) || al(a.bias)) && a.M > 0;
}

template <typename T, int BN, int EPI>
static void launch_nt8_e(const GemmNTArgs& a, int persistent, hipStream_t s) {
    using C = P8<BN>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<T, BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); done = true; }
    const int ntiles = ((a.M + C::BM - 1) / C::BM) * (a.N / BN);
    const int grid = persistent ? (ntiles < 256 ? ntiles : 256) : ntiles;
    hipLaunchKernelGGL((gemm_nt8_kernel<T, BN, EPI>), dim3(grid), dim3(512), C::LDS, s, a, persistent && ntiles > 256 ? 1 : 0);
}
template <typename T, int BN>
static void launch_nt8_t(const GemmNTArgs& a, int persistent, hipStream_t s) {
    const int f = a.flags;
    if (f == (GEMM_BIAS | GEMM_GELU) || f == (GEMM_BIAS | GEMM_GELU | GEMM_AUX_PRE)) launch_nt8_e<T, BN, EP_GELU>(a, persistent, s);
    else if (f == GEMM_MUL_GELU_GRAD || f == (GEMM_MUL_GELU_GRAD | GEMM_RESIDUAL)) launch_nt8_e<T, BN, EP_MULG>(a, persistent, s);
    else if (!(f & ~(GEMM_BIAS | GEMM_DROPOUT | GEMM_RESIDUAL))) launch_nt8_e<T, BN, EP_PLAIN>(a, persistent, s);
    else launch_nt8_e<T, BN, EP_ANY>(a, persistent, s);
}

// bn: 256 or 128.  Returns false when the shape rules of the kernel do not hold (caller falls back).
bool launch_gemm_nt8(const GemmNTArgs& a, int dtype, int bn, int persistent, hipStream_t s) {
    if (!nt8_ok(a, bn)) return false;
    if (dtype == DT_BF16) { if (bn == 256) launch_nt8_t<bf16_t, 256>(a, persistent, s); else launch_nt8_t<bf16_t, 128>(a, persistent, s); }
    else { if (bn == 256) launch_nt8_t<f16_t, 256>(a, persistent, s); else launch_nt8_t<f16_t, 128>(a, persistent, s); }
    return true;
}

}  // namespace mmhip
