// Deep-pipelined NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T), 16-bit operands, fp32 accumulate.
//
// One 512-thread workgroup per CU (8 waves = 4 along M x 2 along N), a 256 x BN output tile (BN = 256, 192 or 128), BK = 64.
// Common to the kernel below (cdna_hip_programming.md, "The 256^2 8-phase template", re-derived for this tile family):
//   * operand K-tiles live in an LDS ring (2 buffers of 64 / 56 KB for BN = 256 / 192, 3 of 48 KB for BN = 128), filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KB pieces = 8 tile rows x 128 B) that stays in flight ACROSS barriers: every wait is a
//     counted s_waitcnt vmcnt(N), every barrier a raw s_barrier; no vmcnt(0) in the main loop;
//   * a K-tile is consumed in NPH = BN/64 phases of 16 MFMA 16x16x32 per wave;
//   * swapped MFMA operands (D = B_frag . A_frag^T) with the B rows of a fragment pair permuted so that a lane ends up
//     with 8 CONSECUTIVE output columns of one row: the fused epilogue (bias / GELU / gelu' / dropout / residual / aux)
//     works on registers and leaves as 16-byte row stores -- no LDS round trip, no barrier, no LDS space;
//   * persistent: a workgroup walks its list of tiles and the DMA stream runs on across tile boundaries, so the next
//     tile's first K-tiles land while the epilogue of the current one runs (K = 768 is only 12 K-tiles).
// Round 2 ran a phase as two barrier intervals (fragment reads + DMA issue | MFMAs) with waves 4-7 one interval behind waves 0-3;
// round 3 replaced that schedule by the interleaved one documented at gemm_nt8_kernel (same-box: +20-25 % per tile,
// profiles/r03_gemm8_interleaved.txt); the two-interval kernel, its BK = 32 deep-ring and four-wave variants are gone.
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
// Diagnostic builds only (tools/gemm8_diag.sh: WRONG results, timing experiments that say where a K-tile's cycles go): each macro removes one
// ingredient of the K loop.  None is defined in the product build.
#ifdef MMHIP_DIAG_NOBARRIER
#define KLOOP_BARRIER() asm volatile("" ::: "memory")
#else
#define KLOOP_BARRIER() raw_barrier()
#endif
#ifdef MMHIP_DIAG_NOWAITVM
#define KLOOP_WAIT_VM(N) asm volatile("" ::: "memory")
#else
#define KLOOP_WAIT_VM(N) wait_vm<N>()
#endif

template <int BN>
struct P8 {
    static constexpr int BM = 256, BK = 64, TN = BN / 2, FN = TN / 16, NPH = BN / 64;
    static constexpr int NBUF = BN == 128 ? 3 : 2;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, KT = A_BYTES + B_BYTES;
    static constexpr int LDS = NBUF * KT;
};

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
#ifdef MMHIP_DIAG_NOEPI
    if (a.force_slow != 77) {          // (never 77: neither the epilogue's arithmetic nor its stores)
        float sum = 0.f;
        for (int e = 0; e < 8; ++e) sum += v[e];
        if (sum == 1.2345e-30f) *((float*)a.C) = sum;
        return;
    }
#endif
#ifdef MMHIP_DIAG_NOSTORE
#define EPI_STORE_GUARD if (a.force_slow == 77)          // (never 77: the epilogue's arithmetic stays, its stores do not happen)
#else
#define EPI_STORE_GUARD
#endif
    if ((EPI == EP_GELU || EPI == EP_ANY) && (fl & GEMM_AUX_PRE)) {
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f<T>(v[e]);
        EPI_STORE_GUARD *reinterpret_cast<v8*>((T*)a.aux + (size_t)m * a.ldaux + n) = o;
    }
    if (EPI == EP_GELU && (fl & GEMM_QGELU)) {          // the CLIP tower's quick-GELU shares the class of the exact GELU (bias [+ stash] + activation)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mm_qgelu(v[e]);
    } else if (EPI == EP_GELU || (EPI == EP_ANY && (fl & GEMM_GELU))) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) mm_gelu2(v[e], v[e + 1]);
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
        for (int e = 0; e < 8; e += 2) { const f32x2_t gg = mm_gelu_grad2(to_f<T>(pre[e]), to_f<T>(pre[e + 1])); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
    }
    if (EPI == EP_ANY && (fl & GEMM_MUL_GELU_GRAD)) {
        v8 u = *reinterpret_cast<const v8*>((const T*)a.mul_in + (size_t)m * a.ldmul + n);
#pragma unroll
        for (int e = 0; e < 8; e += 2) { const f32x2_t gg = mm_gelu_grad2(to_f<T>(u[e]), to_f<T>(u[e + 1])); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
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
    } else if (std::is_same<T, float>::value && (fl & GEMM_OUT_PAIR)) {
        // parity mode: C as a plane pair (mmhip_kernels.h) -- the following matrix products read hi and lo as they are: two 16-byte stores
        // instead of the two of the fp32 row, no split pass later
        const Frag3 f = split8(v);
        bf16_t* c = (bf16_t*)a.C + (size_t)m * a.ldc + n;
        *reinterpret_cast<bf16x8*>(c) = f.hi;
        if (!(fl & GEMM_OUT_PAIR_HI)) *reinterpret_cast<bf16x8*>(c + a.c_lo) = f.lo;
    } else {
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f<T>(v[e]);
        EPI_STORE_GUARD *reinterpret_cast<v8*>((T*)a.C + (size_t)m * a.ldc + n) = o;
    }
}

// epilogue of one 256 x (2 TN) tile from the accumulators: lane holds C[m][n .. n+7], m = m0 + wm*64 + 16 i + (lane&15),
// n = n0 + wn*TN + 32 j' + 8 (lane>>4); values 0-3 from fragment 2j', 4-7 from fragment 2j'+1
// lds_bias: the problem's whole bias vector in LDS (gemm_nt8_kernel copies it there once per workgroup) or null.  From LDS the eight
// values of a fragment pair are read where they are used -- an LDS read does not queue behind the tile's global stores the way a
// global load does (vmcnt retires in issue order), so nothing has to be fetched up front and held in 32 registers.
template <typename T, int FN, int TN, int EPI, int NI = 4, bool LB = false>
__device__ __forceinline__ void tile_epilogue8(const GemmNTArgs& a, f32x4 (&acc)[NI][FN], int m0, int n0, int wm, int wn, int l15, int kc,
                                               const float* lds_bias = nullptr) {
    typedef typename Vec<T>::v8 v8;
    constexpr int WROWS = 16 * NI;              // rows of the tile owned by one wave
        // every load of the epilogue first (bias; residual or mul_in of all the lane's fragments), then compute, then stores.
        // Rows past M: loads read (clamped) row M-1, stores are masked -- C may alias the residual (x += f(x) in the image tower),
        // so a duplicate of row M-1 must never be stored: another wave could read it back as residual.
        const int nb = n0 + wn * TN + 8 * kc;
        float bias8[FN / 2][8];
#pragma unroll
        for (int jp = 0; jp < FN / 2; ++jp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[jp][e] = 0.f;
            if (LB) continue;
            if (EPI == EP_GELU || (a.flags & GEMM_BIAS)) {
                f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + nb + 32 * jp), b1 = *reinterpret_cast<const f32x4*>(a.bias + nb + 32 * jp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias8[jp][e] = b0[e]; bias8[jp][4 + e] = b1[e]; }
            }
        }
        // (BN = 256: two chunks of two fragment-pair columns -- 64 more registers for all sixteen residual fragments would spill; the
        // interleaved kernel (LB) takes one column per chunk: its K loop keeps a second set of B fragments)
        constexpr int CW = LB ? 1 : 2;
#pragma unroll
        for (int jc = 0; jc < FN / 2; jc += CW) {
            v8 pre[NI][CW];
            const T* pbase = (EPI == EP_MULG) ? (const T*)a.mul_in : (const T*)a.residual;
            const int pld = (EPI == EP_MULG) ? a.ldmul : a.ldres;
            const bool want = (EPI == EP_MULG) || ((EPI == EP_PLAIN || EPI == EP_ANY) && (a.flags & GEMM_RESIDUAL));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j2 = 0; j2 < CW; ++j2) {
                    if (EPI != EP_GELU && want && jc + j2 < FN / 2) {
                        const int m = min(m0 + wm * WROWS + i * 16 + l15, a.M - 1);
                        pre[i][j2] = *reinterpret_cast<const v8*>(pbase + (size_t)m * pld + nb + 32 * (jc + j2));
                    }
                }
#pragma unroll
            for (int j2 = 0; j2 < CW; ++j2) {
                const int jp = jc + j2;
                if (jp >= FN / 2) continue;
                if (LB) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds_bias + nb + 32 * jp), b1 = *reinterpret_cast<const f32x4*>(lds_bias + nb + 32 * jp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bias8[jp][e] = b0[e]; bias8[jp][4 + e] = b1[e]; }
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int m = m0 + wm * WROWS + i * 16 + l15;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jp][e] + bias8[jp][e]; v[4 + e] = acc[i][2 * jp + 1][e] + bias8[jp][4 + e]; }
                    if (m < a.M) epilogue8<T, EPI>(a, v, m, nb + 32 * jp, pre[i][j2]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The K-loop schedule ("interleaved", round 3).  One or two problems of equal N and K per launch (GemmNTPair): the tiles of problem 0
// come first in the work list, then those of problem 1.
// In-kernel stamps of the round-2 two-interval schedule (profiles/r02_gemm8_stamps.txt) put a phase at 770-960 cycles for 256 cycles of MFMA per
// wave: the wave group in its L interval (fragment reads, two LDS-DMA issues at 100-185 cycles each, counted wait) needs longer
// than the 290 cycles its partner group multiplies, and the partner then stands at the barrier.  Here no wave ever sits in a
// read-only interval:
//   * every wave runs the same stream, no skew between wave groups, ONE barrier per phase (was two);
//   * a phase's 16 MFMAs multiply fragments that were requested during the PREVIOUS phase: the ds_read_b128 of the next phase's
//     B fragments (4), in the K-tile's last phase also the next K-tile's A fragments (8), and the phase's two LDS-DMA pieces are
//     spread between the MFMAs of the wave itself (sched_group_barrier), so their issue and latency hide under its own and its
//     SIMD partner's MFMAs; B fragments alternate between two register sets by phase parity, A fragments by K-tile parity;
//   * an LDS slot is free as soon as the barrier that ends the phase in which it was READ INTO REGISTERS has passed, i.e. a whole
//     phase before its MFMAs run: the A image of K-tile kt is free during all of kt.  Phase p of K-tile kt therefore restages,
//     for K-tile kt+2: p=0 A rows 0-127 | p=1 A rows 128-255 | p=2 B columns of phases 0,1 | p=3 B columns of phases 2,3 --
//     into slots last read in phases (kt-1,3) | (kt-1,3) | (kt-1,3),(kt,0) | (kt,1),(kt,2);
//   * RAW: one counted wait per phase, vmcnt(DMAs issued in the last four phases) -- everything issued five or more phases ago
//     has landed in every wave before the barrier that ends the phase; the earliest read of a piece comes five phases after its
//     issue (B columns of phase 0: issued (kt,2), read in (kt+1,3)).  Eight 1 KB pieces per wave = 64 KB per CU stay in flight
//     across barriers; at the end of the work list the count follows the pieces actually issued;
//   * at a tile boundary the last phase requests nothing; after the epilogue the wave reads the next tile's first fragments and
//     one extra barrier keeps a fast wave's next LDS-DMA off slots a slower wave has not read yet.
// Staging plan of the interleaved schedule, per tile width (a K-tile = NPH phases of 16 MFMAs per wave; phase p of K-tile kt issues its
// pieces for K-tile kt + NBUF into the buffer of K-tile kt, whose slots were read into registers one phase before their MFMAs):
//   BN = 256 (NBUF 2)  p0: A rows 0-127 (2 per wave) | p1: A rows 128-255 (2) | p2: B columns of phases 0,1 (2) | p3: B phases 2,3 (2);   wait 8 | 8 | 8 | 8
//   BN = 192 (NBUF 2)  p0: A rows 0-127 (2), A rows 128-255 (2) | p1: B phase 0 (1), B phase 1 (1) | p2: B phase 2 (1);                  wait 8 | 8 | 10
//   BN = 128 (NBUF 3)  p0: A rows 0-127 (2) + B phase 0 (1) | p1: A rows 128-255 (2) + B phase 1 (1);                                     wait 9 | 12
// `wait` = the vmcnt of the counted wait that ends phase p: how many of the wave's youngest pieces may stay in flight so that every
// slot requested in the NEXT phase has landed (derivation per width next to I8::wait).
template <int BN>
struct I8 {
    using C = P8<BN>;
    static constexpr int NPH = C::NPH, NBUF = C::NBUF, TN = C::TN;
    static constexpr int PIECES = (C::A_BYTES + C::B_BYTES) / 1024 / 8;            // per wave and K-tile: 8 | 7 | 6
    __host__ __device__ static constexpr int cnt(int p) { return BN == 256 ? 2 : (BN == 192 ? (p == 0 ? 4 : 1 + (p == 1)) : 3); }
    // BN = 192: slots free at the start of (kt,0): all of A(kt), B(kt,0) [read in (kt-1,2)]; B(kt,1) is read in (kt,0), B(kt,2) in (kt,1).
    //   p0 issues A (4); p1 issues B phase 0 and B phase 1 (2: both free by then); p2 issues B phase 2 (1).
    //   reads in (kt+1,2) [A(kt+2), B(kt+2,0)]: issued (kt,0), (kt,1)#1 -> behind them (kt,1)#2, (kt,2), (kt+1,0), (kt+1,1) = 1+1+4+2 = 8 at the end of (kt+1,1)
    //   reads in (kt+2,0) [B(kt+2,1)]: issued (kt,1)#2 -> behind it (kt,2), (kt+1,0), (kt+1,1), (kt+1,2) = 1+4+2+1 = 8 at the end of (kt+1,2)... that wait is p2's
    //   reads in (kt+2,1) [B(kt+2,2)]: issued (kt,2)   -> behind it (kt+1,0..2), (kt+2,0) = 4+2+1+4 = 11 at the end of (kt+2,0) -> 10 (even, conservative)
    // BN = 128 (three buffers, K-tile kt+3): reads in (kt+2,1) [A(kt+3), B(kt+3,0)]: issued (kt,0), (kt,1) -> behind them (kt+1,0), (kt+1,1), (kt+2,0) = 9
    //   at the end of (kt+2,0); reads in (kt+3,0) [B(kt+3,1)]: issued (kt,1) -> (kt+1,0) .. (kt+2,1) = 12 at the end of (kt+2,1)
    __host__ __device__ static constexpr int wait(int p) { return BN == 256 ? 8 : (BN == 192 ? (p == 0 ? 10 : 8) : (p == 0 ? 9 : 12)); }
    // is piece jj of phase p a B piece; its index among the operand's 1 KB pieces (8 rows x 128 B) for wave w
    __host__ __device__ static constexpr int issued_through(int p) { return p == 0 ? cnt(0) : cnt(p) + issued_through(p - 1); }   // pieces of phases 0 .. p
    __host__ __device__ static constexpr bool is_b(int p, int jj) { return BN == 256 ? p >= 2 : (BN == 192 ? p >= 1 : jj == 2); }
    __device__ static int piece(int p, int jj, int w) {
        const int bh = (w >> 2) * (TN / 8) + (w & 3);                  // + 4 * phase: this wave's piece of a phase's B columns (both halves of the tile)
        if (BN == 256) { const int q = w * 2 + jj; return p == 0 ? q : (p == 1 ? 16 + q : (p == 2 ? q + (q & 8) : q + 8 + (q & 8))); }
        if (BN == 192) return p == 0 ? w * 4 + jj : (p == 1 ? bh + 4 * jj : bh + 8);
        return jj < 2 ? p * 16 + w * 2 + jj : bh + 4 * p;
    }
};

// ET = element type of everything the epilogue touches (C, aux, residual, mul_in): T, or float in the parity mode (bf16x3), whose 16-bit
// operands are the split planes of fp32 tensors (x3.hip) -- the K loop is the same
// SEG (parity mode, round 4): A and B are plane pairs written by their producers (mmhip_kernels.h); the reduction runs over THREE segments of
// K / 64 K-tiles each -- (A hi, B hi), (A lo, B hi), (A hi, B lo) -- by moving the operands' base pointers between the planes at the segment
// ends: one launch, no scratch copies of the operands (round 3 split every fp32 operand into [hi | lo | hi] copies before each call).
// SEG = 2: the same three products interleaved per 64-deep k slice (below); 1: three serial segments; 0: plain operands.
template <typename T, int BN, int EPI, typename ET = T, int SEG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(GemmNTPair g, int persistent) {
    using C = P8<BN>;
    using P = I8<BN>;
    constexpr int NPH = P::NPH, NBUF = P::NBUF, GPP = BN == 256 ? 2 : (BN == 192 ? 4 : 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tilesN = g.p[0].N / BN;
    const int tilesM0 = (g.p[0].M + C::BM - 1) / C::BM, tiles0 = tilesN * tilesM0;
    const int tilesM1 = g.count > 1 ? (g.p[1].M + C::BM - 1) / C::BM : 0, ntiles = tiles0 + tilesN * tilesM1;
    const int kseg = g.p[0].K / C::BK;               // K-tiles of one plane
    // SEG = 2 may run only the first two products of every slice -- (A hi, B hi), (A lo, B hi): GemmNTArgs::nprod = 2, B rounded to its hi plane
    const int np2 = SEG == 2 && g.p[0].nprod == 2 ? 1 : 0;
    const int nk = SEG ? (np2 ? 2 : 3) * kseg : kseg;
    // byte steps of the operand pointers at the two segment ends (on top of the ordinary 128 bytes): A hi -> A lo -> A hi, B hi -> B hi -> B lo
    const int segA1 = SEG == 1 ? (g.p[0].a_lo - g.p[0].K) * 2 : 0, segA2 = SEG == 1 ? -(g.p[0].a_lo + g.p[0].K) * 2 : 0;
    const int segB1 = SEG == 1 ? -g.p[0].K * 2 : 0, segB2 = SEG == 1 ? (g.p[0].b_lo - g.p[0].K) * 2 : 0;
    // interleaved order (g.seg_interleave): the three products of ONE 64-deep k slice follow each other -- (A hi, B hi), (A lo, B hi), (A hi, B lo), then
    // the next slice -- so the second use of a B hi / A hi slice comes one / two K-tiles after the first and is served from the XCD's L2
    // instead of, K / 64 tiles later, from beyond it (the K loop runs at the rate its operands arrive: DESIGN.md 7c)
    const int aLo2 = SEG == 2 ? g.p[0].a_lo * 2 : 0, bLo2 = SEG == 2 ? g.p[0].b_lo * 2 : 0;          // byte steps after the products 0, 1, 2 of a slice: A +lo, -lo, +128; B 0, +lo, 128 - lo
    const int nwg = gridDim.x;
    int first, stride, count;
    if (persistent) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = (nwg + 7 - x) / 8;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, n_x = q + (x < r ? 1 : 0);
        first = lo + j; stride = wpx; count = j < n_x ? (n_x - j + wpx - 1) / wpx : 0;
    } else {
        first = xcd_remap(blockIdx.x, nwg); stride = 0; count = 1;
    }
    if (count <= 0) return;
#ifdef MMHIP_DIAG_STAGGER
    if ((blockIdx.x >> 3) & 1)          // every other workgroup of an XCD starts MMHIP_DIAG_STAGGER x 3.4 us late: do the epilogues of a round then overlap the others' K loops?
        for (int i = 0; i < MMHIP_DIAG_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const int GW = g.gw > 0 ? g.gw : 8;
    auto tile_origin = [&](int id, int& m0, int& n0) -> int {
        const int which = id >= tiles0 ? 1 : 0;
        if (which) id -= tiles0;
        const int per_group = (which ? tilesM1 : tilesM0) * GW;
        const int cg = id / per_group, rem = id - cg * per_group;
        const int gw = min(GW, tilesN - cg * GW);
        m0 = (rem / gw) * C::BM;
        n0 = (cg * GW + rem % gw) * BN;
        return which;
    };

    // ---- the DMA stream: staging group p = the phase that issues it; its n-th issue is for global K-tile n (K-tiles numbered over the
    // work list).  issue() is branch-free (it sits between the MFMAs of a phase); advance() moves the group to its next K-tile afterwards.
    // Past the end of the work list a group keeps issuing (into slots nobody reads any more, from the first tile's rows): every phase
    // issues its fixed number of pieces, so the counted waits are constants and the K loop has no tail case.
    // A piece's source = the operand's base pointer + K offset (wave-uniform: scalar registers) + a per-lane 32-bit byte offset.
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* gA[NPH];
    const char* gB[NPH];
    unsigned voff[NPH][GPP];
    int g_rem[NPH], g_tile[NPH], g_inc[NPH], g_ph[NPH];
    auto rebase = [&](auto pc_, int id) {
        constexpr int p = decltype(pc_)::value;
        int m0, n0;
        const GemmNTArgs& a = g.p[tile_origin(id, m0, n0)];
        gA[p] = (const char*)a.A;
        gB[p] = (const char*)a.B;
#pragma unroll
        for (int jj = 0; jj < P::cnt(p); ++jj) {
            const int pc = P::piece(p, jj, w), row = pc * 8 + lrow;
            if (P::is_b(p, jj)) {
                const int sw = (lrow & 3) | ((pc & 1) << 2);
                voff[p][jj] = ((unsigned)(n0 + row) * (unsigned)a.ldb + (unsigned)((lslot ^ sw) * 8)) * 2u;
            } else {
                const int gm = min(m0 + row, a.M - 1);          // rows past M read a valid row, never stored
                voff[p][jj] = ((unsigned)gm * (unsigned)a.lda + (unsigned)((lslot ^ lrow) * 8)) * 2u;
            }
        }
    };
    auto issue = [&](auto pc_, const char* buf) {          // the phase's pieces into K-tile buffer `buf`
        constexpr int p = decltype(pc_)::value;
#ifdef MMHIP_DIAG_NODMA
        return;
#endif
#pragma unroll
        for (int jj = 0; jj < P::cnt(p); ++jj) {
            const char* src = (P::is_b(p, jj) ? gB[p] : gA[p]) + voff[p][jj];
            __builtin_amdgcn_global_load_lds(MM_GLB(src), MM_LDS(buf + (P::is_b(p, jj) ? C::A_BYTES : 0) + P::piece(p, jj, w) * 1024), 16, 0, 0);
        }
    };
    auto advance = [&](auto pc_) {
        constexpr int p = decltype(pc_)::value;
        if (--g_rem[p] == 0) {
            g_rem[p] = nk;
            g_ph[p] = 0;
            g_tile[p] += 1;
            const bool live = g_tile[p] < count;
            g_inc[p] = live ? 128 : 0;
            rebase(pc_, first + (live ? g_tile[p] : 0) * stride);
        } else {
            if constexpr (SEG == 2) {
                // g_ph = the product (0, 1, 2) just issued within its k slice (never moving while the group runs on past the end of the work
                // list: g_inc = 0 there)
                const int lv = g_inc[p] >> 7, ph = g_ph[p];
                if (np2) {          // two products per slice: A +lo, then 128 - lo; B 0, then +128
                    gA[p] += lv * (ph == 0 ? aLo2 : 128 - aLo2);
                    gB[p] += lv * (ph == 0 ? 0 : 128);
                    g_ph[p] = ph ^ 1;
                } else {
                    gA[p] += lv * (ph == 0 ? aLo2 : (ph == 1 ? -aLo2 : 128));
                    gB[p] += lv * (ph == 0 ? 0 : (ph == 1 ? bLo2 : 128 - bLo2));
                    g_ph[p] = ph == 2 ? 0 : ph + 1;
                }
            } else if constexpr (SEG == 1) {
                // g_rem K-tiles of the group's current tile are still to be issued: the next one opens segment 1 / segment 2
                const int j1 = g_rem[p] == 2 * kseg ? (g_inc[p] >> 7) : 0, j2 = g_rem[p] == kseg ? (g_inc[p] >> 7) : 0;
                gA[p] += g_inc[p] + j1 * segA1 + j2 * segA2;
                gB[p] += g_inc[p] + j1 * segB1 + j2 * segB2;
            } else {
                gA[p] += g_inc[p];
                gB[p] += g_inc[p];
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    auto for_phases = [&](auto&& f) {
        f(I0{});
        f(I1{});
        if constexpr (NPH >= 3) f(I2{});
        if constexpr (NPH == 4) f(I3{});
    };

    f32x4 acc[4][C::FN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // fragment read offsets (layout of gemm_nt8_kernel): A row = wm*64 + 16 i + (lane&15); B (permuted) row = wn*TN + 32 p + 8((lane&15)>>2) + 4 h + (lane&3);
    // 16-byte chunk = (kk*4 + (lane>>4)) ^ (lane&7)
    const int l15 = lane & 15, kc = lane >> 4, sw7 = lane & 7;
    const int a_row_off = (wm * 64 + l15) * 128;
    const int b_row_off = C::A_BYTES + (wn * C::TN + 8 * (l15 >> 2) + (lane & 3)) * 128;
    const int ch[2] = {((0 * 4 + kc) ^ sw7) << 4, ((1 * 4 + kc) ^ sw7) << 4};
    v8 af[4][2], bf[2][2][2];          // A fragments [i][kk] (reloaded in place in a K-tile's last phase); B [set][h][kk], sets alternate by phase
    auto read_b = [&](auto set_, const char* Ks, int p) {
        constexpr int set = decltype(set_)::value;
#ifdef MMHIP_DIAG_NOLDS
        if (g.gw != 12345) return;          // (never true: the fragments keep what the prologue read; the compiler cannot drop the reads' code)
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bf[set][h][kk] = lds_read8<T>(Ks, b_row_off + (32 * p + 4 * h) * 128 + ch[kk]);
    };
    // the builtin, not inline asm: hipcc's own wait insertion then knows that nothing is outstanding on the LGKM counter and puts no
    // `lgkmcnt(0)` of its own in front of the next phase's first MFMA (behind that phase's first fragment requests)
    auto lgkm0 = [&]() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };

    // ---- the bias vectors of the launch's problems into LDS behind the K-tile buffers
    float* lds_bias_all = reinterpret_cast<float*>(smem + C::LDS);
    for (int pi = 0; pi < g.count; ++pi) {
        const GemmNTArgs& a = g.p[pi];
        if (!(a.flags & GEMM_BIAS)) {
            for (int c = tid * 4; c < a.N; c += 2048) *reinterpret_cast<f32x4*>(lds_bias_all + pi * a.N + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int c = tid * 4; c < a.N; c += 2048) *reinterpret_cast<f32x4*>(lds_bias_all + pi * a.N + c) = *reinterpret_cast<const f32x4*>(a.bias + c);
        }
    }
    // ---- prologue: K-tiles 0 .. NBUF-1, i.e. what the phases of K-tiles -NBUF .. -1 would have issued
    for_phases([&](auto pc_) { constexpr int p = decltype(pc_)::value; g_rem[p] = nk; g_tile[p] = 0; g_inc[p] = 128; g_ph[p] = 0; rebase(pc_, first); });
#pragma unroll
    for (int b = 0; b < NBUF; ++b) for_phases([&](auto pc_) { issue(pc_, smem + b * C::KT); advance(pc_); });
    wait_vm<(NBUF - 1) * P::PIECES>();          // K-tile 0 has landed
    raw_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) af[i][kk] = lds_read8<T>(smem, a_row_off + i * (16 * 128) + ch[kk]);
    read_b(I0{}, smem, 0);
    lgkm0();
    raw_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int cb = 0;                                  // buffer of the current K-tile
    constexpr int NST1 = 4 * (C::FN / 2);        // global stores of one output tensor per wave and tile
    int grace0 = 0;
    for (int t = 0; t < count; ++t) {
        // one K-tile; S0 = the B register set its phase 0 multiplies (sets alternate by phase: with three phases per K-tile also by K-tile)
        auto ktile = [&](auto s0_, int grace) {
            constexpr int S0 = decltype(s0_)::value;
            const int nb = cb + 1 == NBUF ? 0 : cb + 1;
            const char* Ks = smem + cb * C::KT;          // this K-tile; also the buffer K-tile + NBUF is staged into
            const char* Kn = smem + nb * C::KT;
            cb = nb;
            for_phases([&](auto pc_) {
                constexpr int p = decltype(pc_)::value;
                constexpr int bs = (S0 + p) & 1;
                constexpr int nd = P::cnt(p);
                if constexpr (p < NPH - 1) {
                    // request the B fragments of the next phase, issue the phase's pieces, 16 MFMAs
                    read_b(std::integral_constant<int, bs ^ 1>{}, Ks, p + 1);
                    issue(pc_, Ks);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int h = 0; h < 2; ++h) acc[i][2 * p + h] = mfma16(bf[bs][h][kk], af[i][kk], acc[i][2 * p + h]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                    for (int d = 0; d < nd; ++d) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, nd <= 2 ? 4 : 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 12 - (nd <= 2 ? 4 : 2) * nd, 0);
                } else {
                    // last phase: the next K-tile's B fragments of phase 0 (other register set) and its A fragments, each A fragment into
                    // the registers of the one whose last two MFMAs have just been issued
                    read_b(std::integral_constant<int, bs ^ 1>{}, Kn, 0);
                    issue(pc_, Ks);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) acc[i][2 * p + h] = mfma16(bf[bs][h][kk], af[i][kk], acc[i][2 * p + h]);
                            af[i][kk] = lds_read8<T>(Kn, a_row_off + i * (16 * 128) + ch[kk]);
                        }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (r < nd * 2 && (r & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                advance(pc_);
                // every slot the next phase requests has landed (this wave's pieces).  In the first K-tile behind an epilogue that only
                // stored (no residual / gelu' loads), those `grace` stores sit inside the window of pieces that may stay in flight as long
                // as the window still reaches back to pieces issued before the epilogue: the K loop does not wait for HBM writes
                if (grace && P::wait(p) > P::issued_through(p)) {
                    if (grace > NST1) KLOOP_WAIT_VM(P::wait(p) + 2 * NST1); else KLOOP_WAIT_VM(P::wait(p) + NST1);
                } else {
                    KLOOP_WAIT_VM(P::wait(p));
                }
                lgkm0();                           // this wave's fragment requests are back: the slots they read may be restaged after the barrier
                KLOOP_BARRIER();
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if constexpr (NPH & 1) {
            int k = 0;
#pragma unroll 1
            for (; k + 2 <= nk; k += 2) { ktile(I0{}, k == 0 ? grace0 : 0); ktile(I1{}, 0); }
            if (k < nk) ktile(I0{}, k == 0 ? grace0 : 0);
        } else {
#pragma unroll 1
            for (int k = 0; k < nk; ++k) ktile(I0{}, k == 0 ? grace0 : 0);
        }
        // ---- epilogue of tile t from the accumulators (register epilogue of gemm_nt8_kernel, bias from LDS)
        int m0, n0;
        const int which = tile_origin(first + t * stride, m0, n0);
        const GemmNTArgs& a = g.p[which];
        tile_epilogue8<ET, C::FN, C::TN, EPI, 4, true>(a, acc, m0, n0, wm, wn, l15, kc, lds_bias_all + which * a.N);
        zero_acc();
        // stores this wave has just issued and nothing else (a full tile: every row store executes): see `grace` in the K loop
        grace0 = 0;
        if (std::is_same<T, ET>::value && m0 + C::BM <= a.M) {          // (fp32 stores are two instructions each: no grace in the parity mode)
            if (EPI == EP_GELU) grace0 = (a.flags & GEMM_AUX_PRE) ? 2 * NST1 : NST1;
            else if (EPI == EP_PLAIN && !(a.flags & GEMM_RESIDUAL)) grace0 = NST1;
        }
        if (t + 1 < count) {
            // the fragments requested in the tile's last phase are not kept across the epilogue (it needs the registers): request the
            // next tile's first fragments again; the barrier keeps a faster wave's next LDS-DMA off slots this wave has not read yet
            const char* Ks = smem + cb * C::KT;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) af[i][kk] = lds_read8<T>(Ks, a_row_off + i * (16 * 128) + ch[kk]);
            read_b(I0{}, Ks, 0);
            lgkm0();
            raw_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vm<0>();          // no LDS-DMA may outlive the workgroup's LDS allocation
}

// split-K finish (launch_nt_splitk in gemm.hip): thread = 8 consecutive columns of one row; sums the slices' fp32 partial products,
// adds the bias and runs the run-time epilogue of the fused kernels (aux, GELU, gelu', dropout, residual, output type)
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(GemmNTArgs a, int slices) {
    typedef typename Vec<T>::v8 v8;
    const int per_row = a.N >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.M * per_row) return;
    const int m = idx / per_row, n = (idx - m * per_row) << 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int z = 0; z < slices; ++z) {
        const float* p = a.splitk_ws + ((size_t)z * a.M + m) * a.N + n;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += x0[e]; v[4 + e] += x1[e]; }
    }
    if (a.flags & GEMM_BIAS) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    v8 pre;
#pragma unroll
    for (int e = 0; e < 8; ++e) pre[e] = from_f<T>(0.f);
    if (a.flags & GEMM_RESIDUAL) pre = *reinterpret_cast<const v8*>((const T*)a.residual + (size_t)m * a.ldres + n);
    epilogue8<T, EP_ANY>(a, v, m, n, pre);
}
hipError_t launch_splitk_finish(const GemmNTArgs& a, int dtype, int slices, hipStream_t s) {
    const int threads = a.M * (a.N >> 3);
    if (dtype == DT_BF16) hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3((threads + 255) / 256), dim3(256), 0, s, a, slices);
    else hipLaunchKernelGGL(splitk_finish_kernel<f16_t>, dim3((threads + 255) / 256), dim3(256), 0, s, a, slices);
    return hipGetLastError();
}

static bool nt8_ok(const GemmNTArgs& a, int bn) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.a_pair != a.b_pair || (a.a_pair && (a.a_lo % 8 || a.b_lo % 8))) return false;      // plane pairs: both operands, 16-byte aligned planes
    if ((a.flags & GEMM_OUT_PAIR) && a.c_lo % 8) return false;
    return a.N % bn == 0 && a.K % 64 == 0 && a.K >= 64 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && al(a.A) && al(a.B) && al(a.C) &&
           (!(a.flags & GEMM_RESIDUAL) || (a.ldres % 8 == 0 && al(a.residual))) &&
           (!(a.flags & GEMM_AUX_PRE) || (a.ldaux % 8 == 0 && al(a.aux))) &&
           (!(a.flags & GEMM_MUL_GELU_GRAD) || (a.ldmul % 8 == 0 && al(a.mul_in))) &&
           (!(a.flags & GEMM_BIAS) || al(a.bias)) && a.M > 0;
}

static int nt8_grid(int ntiles) { return ntiles <= 256 ? ntiles : 256; }      // workgroups of a persistent launch: one per CU
template <typename T, int BN, int EPI, typename ET = T, int SEG = 0>
static void launch_nt8_e(const GemmNTPair& g, int persistent, hipStream_t s) {
    using C = P8<BN>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<T, BN, EPI, ET, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    int ntiles = 0;
    for (int i = 0; i < g.count; ++i) ntiles += ((g.p[i].M + C::BM - 1) / C::BM) * (g.p[i].N / BN);
    const int cap = g.p[0].grid > 0 ? (g.p[0].grid < ntiles ? g.p[0].grid : ntiles) : 0;
    const int grid = persistent ? (cap ? cap : nt8_grid(ntiles)) : ntiles;
    // the bias vectors live in LDS behind the K-tile buffers (nt8_fits: they fit the CU's 160 KB)
    size_t lds = C::LDS;
    for (int i = 0; i < g.count; ++i) lds += (size_t)g.p[i].N * 4;
    GemmNTPair g2 = g;
    {   // column groups of equal width (N = 2304 in 256-wide tiles: 5 + 4 instead of 8 + 1 -- the XCD that walks a one-tile-wide group reads every
        // A tile for a single use: 12608 x 2304 x 768 59.0 -> 56.8 us, L2 hit rate 0.70 -> 0.74, same box)
        const int tn = g.p[0].N / BN, ng = (tn + 7) / 8;
        g2.gw = (tn + ng - 1) / ng;
    }
    hipLaunchKernelGGL((gemm_nt8_kernel<T, BN, EPI, ET, SEG>), dim3(grid), dim3(512), lds, s, g2, persistent && ntiles > grid ? 1 : 0);
}
// epilogue class that covers a flag set (a pair uses the class that covers both)
static int nt8_class(int f) {
    f &= ~(GEMM_OUT_PAIR | GEMM_OUT_PAIR_HI);          // where the result goes does not change the class
    if (f == (GEMM_BIAS | GEMM_GELU) || f == (GEMM_BIAS | GEMM_GELU | GEMM_AUX_PRE) || f == (GEMM_BIAS | GEMM_QGELU)) return EP_GELU;
    if (f == GEMM_MUL_GELU_GRAD || f == (GEMM_MUL_GELU_GRAD | GEMM_RESIDUAL)) return EP_MULG;
    if (!(f & ~(GEMM_BIAS | GEMM_DROPOUT | GEMM_RESIDUAL))) return EP_PLAIN;
    return EP_ANY;
}
// (Round 5 built the variant the round-4 review asked for -- four-wave workgroups on 256 x 128 tiles, two per CU, so that one tile's epilogue overlaps
// the other's K loop -- and measured it 25-45 % behind this kernel: profiles/r05_two_wg_gemm.txt.  The overlap worked; the feed did not: 1.5 x the operand
// bytes per MFMA through the 64 B/clk L1 path of the CU, and no room in 160 KB for two BK = 64 rings.  Removed; the history has gemm_nt4_kernel.)
template <typename T, int BN, typename ET = T, int SEG = 0>
static void launch_nt8_t(const GemmNTPair& g, int persistent, hipStream_t s) {
    int c = nt8_class(g.p[0].flags);
    if (g.count > 1 && nt8_class(g.p[1].flags) != c) c = EP_ANY;
    if (c == EP_GELU) launch_nt8_e<T, BN, EP_GELU, ET, SEG>(g, persistent, s);
    else if (c == EP_MULG) launch_nt8_e<T, BN, EP_MULG, ET, SEG>(g, persistent, s);
    else if (c == EP_PLAIN) launch_nt8_e<T, BN, EP_PLAIN, ET, SEG>(g, persistent, s);
    else launch_nt8_e<T, BN, EP_ANY, ET, SEG>(g, persistent, s);
}
static void launch_nt8_d(const GemmNTPair& g, int dtype, int bn, int persistent, hipStream_t s) {
    static int ilv = -1;
    if (ilv < 0) { const char* e = getenv("MMHIP_X3_INTERLEAVE"); ilv = e ? atoi(e) : 1; }
    if (dtype == DT_F32 && g.p[0].a_pair && (ilv || g.p[0].nprod == 2)) {          // parity mode, operands as plane pairs: the three (two) products interleaved per k slice
        if (bn == 256) launch_nt8_t<bf16_t, 256, float, 2>(g, persistent, s);
        else if (bn == 192) launch_nt8_t<bf16_t, 192, float, 2>(g, persistent, s);
        else launch_nt8_t<bf16_t, 128, float, 2>(g, persistent, s);
    } else if (dtype == DT_F32 && g.p[0].a_pair) {          // ... as three serial K segments (MMHIP_X3_INTERLEAVE=0)
        if (bn == 256) launch_nt8_t<bf16_t, 256, float, 1>(g, persistent, s);
        else if (bn == 192) launch_nt8_t<bf16_t, 192, float, 1>(g, persistent, s);
        else launch_nt8_t<bf16_t, 128, float, 1>(g, persistent, s);
    } else if (dtype == DT_F32) {          // parity mode: bf16 split planes in (x3.hip scratch copies), fp32 epilogue
        if (bn == 256) launch_nt8_t<bf16_t, 256, float>(g, persistent, s);
        else if (bn == 192) launch_nt8_t<bf16_t, 192, float>(g, persistent, s);
        else launch_nt8_t<bf16_t, 128, float>(g, persistent, s);
    } else if (dtype == DT_BF16) {
        if (bn == 256) launch_nt8_t<bf16_t, 256>(g, persistent, s);
        else if (bn == 192) launch_nt8_t<bf16_t, 192>(g, persistent, s);
        else launch_nt8_t<bf16_t, 128>(g, persistent, s);
    } else {
        if (bn == 256) launch_nt8_t<f16_t, 256>(g, persistent, s);
        else if (bn == 192) launch_nt8_t<f16_t, 192>(g, persistent, s);
        else launch_nt8_t<f16_t, 128>(g, persistent, s);
    }
}

// the problems' bias vectors live in LDS behind the K-tile buffers; a piece's source is a 32-bit byte offset from the operand's base
static bool nt8_fits(const GemmNTArgs& a, int nprob, int bn) {
    const size_t lds = (bn == 256 ? P8<256>::LDS : (bn == 192 ? P8<192>::LDS : P8<128>::LDS)) + (size_t)a.N * 4 * nprob;
    return lds <= 160 * 1024 && (size_t)a.M * a.lda * 2 < (1ull << 32) && (size_t)a.N * a.ldb * 2 < (1ull << 32);
}
// bn: 256, 192 or 128.  Returns false when the shape rules of the kernel do not hold (caller falls back).
bool launch_gemm_nt8(const GemmNTArgs& a, int dtype, int bn, int persistent, hipStream_t s) {
    if ((bn != 256 && bn != 192 && bn != 128) || !nt8_ok(a, bn) || !nt8_fits(a, 1, bn)) return false;
    GemmNTPair g;
    g.p[0] = a; g.p[1] = a; g.count = 1;
    launch_nt8_d(g, dtype, bn, persistent, s);
    return true;
}

// two problems of equal N and K in one persistent launch; bn = 0 picks the tile that fills the rounds of 256 workgroups best.
// false = rules not met (caller launches them one by one)
bool launch_gemm_nt8_pair(const GemmNTArgs& a0, const GemmNTArgs& a1, int dtype, int bn, hipStream_t s) {
    if (dtype == DT_F32) {          // parity mode: both problems on plane pairs with the same plane offsets (the K loop reads them from problem 0), all three products
        if (!a0.a_pair || !a0.b_pair || !a1.a_pair || !a1.b_pair || a0.a_lo != a1.a_lo || a0.b_lo != a1.b_lo || a0.nprod != a1.nprod || a0.nprod == 1) return false;
        if (a0.M <= 128 || a1.M <= 128) return false;      // the <= 128-row problems of the parity mode take the direct kernel (x3.hip)
    } else if (dtype != DT_BF16 && dtype != DT_F16) return false;
    if (a0.N != a1.N || a0.K != a1.K) return false;
    if (bn == 0) {
        double best = 0;
        for (int cand : {256, 192, 128}) {
            if (a0.N % cand) continue;
            const long t = (long)((a0.M + 255) / 256 + (a1.M + 255) / 256) * (a0.N / cand);
            const double u = (double)t / (double)(((t + 255) / 256) * 256) + (cand == 256 ? 0.04 : (cand == 192 ? 0.02 : 0.0));   // ties -> wider
            if (u > best) { best = u; bn = cand; }
        }
    }
    if ((bn != 256 && bn != 192 && bn != 128) || !nt8_ok(a0, bn) || !nt8_ok(a1, bn) || !nt8_fits(a0, 2, bn)) return false;
    GemmNTPair g;
    g.p[0] = a0; g.p[1] = a1; g.count = 2;
    launch_nt8_d(g, dtype, bn, 1, s);
    return true;
}

}  // namespace mmhip
