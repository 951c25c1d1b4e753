// Internal launcher interface between the kernels (*.hip) and the engine / C ABI (engine.hip, capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mmhip_common.h"

namespace mmhip {

enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2,
       DT_PAIR = 3 };      // destination type of launch_cast_group / launch_cast_pad / launch_patchify only: bf16 plane pairs (parity mode)
enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

// ---------------------------------------------------------------- GEMM
enum {
    GEMM_BIAS = 1,            // + bias[n] (fp32)
    GEMM_GELU = 2,            // exact-erf GELU
    GEMM_RESIDUAL = 4,        // + residual[m][n] (16-bit), applied last
    GEMM_DROPOUT = 8,         // hash dropout on element index m*N + n, before the residual
    GEMM_OUT_F32 = 16,        // C is fp32
    GEMM_AUX_PRE = 32,        // aux[m][n] = value after bias, before the activation (16-bit)
    GEMM_MUL_GELU_GRAD = 64,  // value *= gelu'(mul_in[m][n])
    GEMM_TANH = 128,
    GEMM_QGELU = 256,         // quick-GELU x * sigmoid(1.702 x)   (HF CLIP hidden_act "quick_gelu")
    GEMM_OUT_PAIR = 512,      // parity mode: C is a plane pair (below) instead of fp32
    GEMM_OUT_PAIR_HI = 1024,  // ... of which only the hi plane is written: every reader takes one product (GemmNTArgs::nprod = 1), the lo plane is never read
};
// Parity mode (bf16x3), round 4: PLANE PAIRS.  A tensor that feeds a matrix product is stored by its PRODUCER as two bf16 planes,
// hi = bf16(x) and lo = bf16(x - hi), side by side in one row: element (r, c) has hi at base[r * ld + c] and lo at base[r * ld + lo_off + c]
// (ld, lo_off in 16-bit elements; the natural layout is ld = 2 W, lo_off = W: the same bytes as the fp32 tensor it replaces).  The matrix
// cores read the planes directly -- hi.hi + lo.hi + hi.lo as three K segments of ONE launch -- instead of every GEMM call splitting its fp32
// operands into scratch copies first (round 3: split3_kernel, 6.2 of the 31 ms step).  x = hi + lo carries 16 significant bits.
struct GemmNTArgs {
    const void* A; const void* B; void* C; void* aux; const float* bias; const void* residual; const void* mul_in;
    int M, N, K, lda, ldb, ldc, ldaux, ldres, ldmul;
    int flags;
    int force_slow;
    int tile;                 // 0 = auto, 1 = 128x128, 2 = 256x128, 3 = 256x256, ...
    int drop_row_mul;         // dropout element index uses row m * drop_row_mul (0 = 1): compact CLS-row GEMMs keep the
                              // masks of the full [posts*T, N] tensor
    DropCfg drop;
    int grid;                 // persistent deep-pipelined kernels (gemm8.hip): at most this many workgroups (0 = one per CU).  The forward
                              // partitions the chip between its two towers this way: each workgroup holds a CU's LDS, so 96 + 160
                              // resident workgroups of two concurrent launches ARE a 96 / 160 CU split, and 256-row tiles of the
                              // 8192-row text GEMMs come in multiples of 96
    void* x3_ws;              // parity mode (bf16x3) only: scratch for the operands' split planes (x3.hip, x3_nt_scratch_bytes); null = the direct kernel
    size_t x3_ws_bytes;
    float* splitk_ws;         // optional fp32 scratch [K/384][M][N]: a GEMM of <= 128 rows with K >= 1536 (the CLS-row GEMMs of the last
                              // text layer) is cut along K into slices that run side by side; a second kernel sums them and applies the epilogue
    int a_pair, b_pair;       // parity mode: A / B is a plane pair (lda / ldb in 16-bit elements), lo plane a_lo / b_lo elements behind the hi plane
    int a_lo, b_lo, c_lo;     // c_lo: lo-plane offset of C with GEMM_OUT_PAIR (ldc in 16-bit elements)
    int nprod;                // plane pairs only: MFMA products per k slice -- 0 / 3: hi.hi + lo.hi + hi.lo (parity on the result); 2: hi.hi + lo.hi (B rounded
                              // to its hi plane); 1: hi.hi only (both operands rounded to bf16, pair / fp32 epilogue kept).  The engine's backward policy
                              // (mmhip_set_backward_products): north_star's 1e-3 is on logits and loss, the gradient bound is stated in DESIGN.md 4c
};
struct GemmNTPair { GemmNTArgs p[2]; int count; int gw; };      // gw: N-tiles per column group of the tile walk (0 = 8)      // gemm8.hip: one or two problems of equal N and K per launch
static constexpr int GEMM_TN_MAX_GROUP = 8;
struct GemmTNProblem {
    const void* A; const void* B; float* C;
    int M, Nn, Nc, lda, ldb, ldc, tile_start;
    float* colsum;            // optional [Nn]: (+)= alpha * sum_m A[m][n] -- the bias gradient that goes with dW = dY^T X, taken
                              // from the same operand tiles by one extra MFMA column (B = ones), no extra pass, no atomics
    int colsum_rows;          // > 0: the column sums cover rows 0 .. colsum_rows-1 only (a multiple of 64)
    int pair;                 // parity mode: A and B are plane pairs (lda / ldb in 16-bit elements, lo planes a_lo / b_lo elements behind): the kernel
    int a_lo, b_lo;           // walks the M rows three times -- (A hi, B hi), (A lo, B hi), (A hi, B lo) -- and the column sums cover A hi + A lo
    int nprod;                // pairs: products per 64-row slice (0 / 3: all three; 2: (A hi, B hi), (A lo, B hi); 1: (A hi, B hi) -- GemmNTArgs::nprod)
};
struct GemmTNGroup {
    GemmTNProblem p[GEMM_TN_MAX_GROUP];
    int count, accumulate;
    float alpha;              // C (+)= alpha * A^T B   (un-does the f16 gradient scale)
    int pair_serial;          // plane pairs: 1 = three passes over all M rows one after the other (round-4 first form); 0 = the three products of a
                              // 64-row slice follow each other
};
struct SmallGemmArgs {
    const void* A; const float* W; const float* bias; float* out;
    int M, N, K, lda, ldw, ldo, act, accumulate;
    int sam, sak, sbk, sbn;   // element strides of A(m,k) and B(k,n); filled by the launch_small_* wrappers
};
hipError_t launch_gemm_nt(const GemmNTArgs& a, int dtype, hipStream_t s);
// Measurement hook (round 5): while a sink is set on the calling thread, launch_gemm_nt brackets every launch with HIP events recorded on the
// launch's own stream (the early-fusion engine enqueues its NT GEMMs from composite blocks on three streams: there is no single call site to wrap)
struct GemmTimingSink {
    struct Ev { hipEvent_t a, b; double flops; };
    Ev* evs; size_t capacity, used;
};
void gemm_timing_sink(GemmTimingSink* sink);      // nullptr = off
// deep-pipelined 256 x bn tiles (gemm8.hip); false = shape rules not met, nothing launched
bool launch_gemm_nt8(const GemmNTArgs& a, int dtype, int bn, int persistent, hipStream_t s);
// two problems of equal N and K in one persistent launch (the two towers' GEMMs of one layer); bn = 0: best-filling tile
// split-K finish: C = epilogue(sum over `slices` partial products in a.splitk_ws), same flags as the fused epilogue
hipError_t launch_splitk_finish(const GemmNTArgs& a, int dtype, int slices, hipStream_t s);
bool launch_gemm_nt8_pair(const GemmNTArgs& a0, const GemmNTArgs& a1, int dtype, int bn, hipStream_t s);
hipError_t launch_gemm_tn(const GemmTNProblem* probs, int count, int accumulate, int dtype, int force_slow, hipStream_t s, float alpha = 1.0f,
                          void* x3_ws = nullptr, size_t x3_ws_bytes = 0);
// parity mode (fp32 activations, three bf16 MFMA products of split operands / fp32 attention): csrc/x3.hip
hipError_t launch_gemm_nt_x3(const GemmNTArgs& a, hipStream_t s);
hipError_t launch_gemm_tn_x3(const GemmTNProblem* probs, int count, int accumulate, hipStream_t s, float alpha, void* ws = nullptr, size_t ws_bytes = 0);
size_t x3_nt_scratch_bytes(int M, int N, int K);      // split planes of one NT problem's operands / of one TN problem's
size_t x3_tn_scratch_bytes(int M, int Nn, int Nc);
hipError_t launch_small_nt(const SmallGemmArgs& a, int a_dtype, hipStream_t s);   // out[M,N] = act(A[M,K] W[N,K]^T + b)
hipError_t launch_small_nn(const SmallGemmArgs& a, hipStream_t s);                // out[M,N] = A[M,K] W[K,N]
hipError_t launch_small_tn(const SmallGemmArgs& a, int b_dtype, int n_rows, hipStream_t s);  // out[n_rows,N] = A[M,n_rows]^T B[M,N]

// ---------------------------------------------------------------- attention
struct AttnArgs {
    const void* qkv;      // [rows, ld_qkv] packed q | k | v, head h at columns h*64
    const float* maskbias;  // [posts, S] additive key bias (0 or -inf) or null
    void* ctx;            // [rows, ld_ctx]
    float* lse;           // [posts, heads, S] (natural-log units of the scaled scores) or null
    int posts, S, heads, ld_qkv, ld_ctx, hidden;
    float scale;
    DropCfg drop;
    int q_tiles;          // > 0: only the first q_tiles 32-row query tiles are computed / written (last layer: CLS row only)
    int pair;             // parity mode: qkv and ctx are plane pairs (ld_qkv / ld_ctx in 16-bit elements), lo planes lo_qkv / lo_ctx elements behind
    int lo_qkv, lo_ctx;
    int Sq_live, Sk_live; // > 0: a post has Sq_live queries and Sk_live keys (cross attention, Sq != Sk; 0 = S).  Nothing past them is read or written, key tiles
                          // past Sk_live are not computed.  The dropout element index, the lse row and the maskbias row stay those of the S x S layout
                          // (S = the larger of the two), so padded and compact callers drop the same probabilities.
    int q_rps, kv_rps, ctx_rps;   // rows per post of the Q columns / of the K and V columns of qkv / of ctx (0 = S): round 5 -- cross attention on COMPACT tensors:
                          // query row q of post p is row p * q_rps + q, key row k is row p * kv_rps + k; the two projections write Mq and Mc rows and nothing is
                          // remapped, cleared or copied
};
struct AttnBwdArgs {
    const void* qkv; const float* maskbias; const void* ctx; const void* dctx; const float* lse;
    void* dqkv;           // [rows, ld_qkv]
    int posts, S, heads, ld_qkv, ld_ctx, hidden;
    float scale;
    DropCfg drop;
    int q_tiles;          // > 0: d ctx is zero outside the first q_tiles query tiles; dQ of the other tiles is NOT written
    int pair;             // parity mode: qkv, ctx, dctx, dqkv are plane pairs (qkv and dqkv share ld_qkv / lo_qkv; ctx and dctx ld_ctx / lo_ctx)
    int lo_qkv, lo_ctx;
    int Sq_live, Sk_live; // as in AttnArgs: rows of ctx / dctx past Sq_live are never read (their gradient is taken as zero), dQ rows past Sq_live and
                          // dK / dV rows past Sk_live never written
    int q_rps, kv_rps, ctx_rps;   // as in AttnArgs (qkv and dqkv share q_rps / kv_rps; ctx and dctx share ctx_rps)
    int nprod;            // plane pairs: 1 = the scores keep three products (they are re-computed against the forward's log-sum-exp), the four other matrix
                          // products take one, d ctx is read from its hi plane only; 0 / 3: three products throughout (GemmNTArgs::nprod: the engine's policy)
};
hipError_t launch_attn_fwd(const AttnArgs& a, int dtype, hipStream_t s);
hipError_t launch_attn_fwd_f32(const AttnArgs& a, hipStream_t s);
hipError_t launch_attn_bwd_f32(const AttnBwdArgs& a, hipStream_t s);
hipError_t launch_attn_bwd(const AttnBwdArgs& a, int dtype, hipStream_t s);

// ---------------------------------------------------------------- row ops (LayerNorm, embeddings, elementwise)
struct LNArgs {
    const void* x; void* y; const float* gamma; const float* beta; float* mean; float* rstd;
    int rows, width, ldx, ldy; float eps;
    void* y_pair; int ld_pair, lo_pair;      // parity mode (optional): y also as a plane pair; y itself may then be null
};
struct LNBwdArgs {
    const void* dy; const void* x; const float* gamma; const float* mean; const float* rstd;
    void* dx;             // 16-bit
    const void* dres;     // optional 16-bit tensor added to dx (gradient arriving through the residual branch)
    float* dgamma; float* dbeta;   // fp32, accumulated (+=)
    int rows, width;
    float* partial;       // optional workspace of partial_floats_rows(rows, width, 2) floats: two-stage column reduction
    float alpha;          // dgamma / dbeta += alpha * (...)   (0 is read as 1)
    // fused tail (optional): dx_drop = dropout-backward(dx) with the forward's mask (drop.thresh16 != 0), and
    // colsum_out[c] += alpha * sum_rows (dx_drop or dx)[.][c]   -- the bias gradient of the Linear that produced the LN input
    void* dx_drop; float* colsum_out; DropCfg drop;
    int drop_row_mul;     // dropout index uses row * drop_row_mul (0 = 1)
    int defer_reduce;     // 1: only write the per-block partials; the caller runs launch_layernorm_bwd_reduce later
    void* pair_out; int ld_pair, lo_pair;      // parity mode (optional): the tensor the following GEMMs read (dx_drop where dropout is on, else dx) as a
                                               // plane pair; the fp32 dx_drop is then not written
    int pair_hi_only;                          // ... its hi plane only (every reader takes one product: GemmNTArgs::nprod = 1)
};
hipError_t launch_layernorm_bwd_reduce(const LNBwdArgs& a, hipStream_t s);
hipError_t launch_layernorm_fwd(const LNArgs& a, int dtype, hipStream_t s);
hipError_t launch_layernorm_bwd(const LNBwdArgs& a, int dtype, hipStream_t s);

struct EmbedArgs {
    const int64_t* ids; const int64_t* mask;
    const float* word; const float* pos; const float* type; const float* gamma; const float* beta;
    void* x;              // [rows, H] 16-bit output (after LN and dropout)
    void* xhat;           // [rows, H] 16-bit normalised value before gamma/beta (saved for backward) or null
    float* rstd;          // [rows]
    int* pos_ids;         // [rows] (written)
    float* maskbias;      // [rows] additive key bias (written)
    int posts, T, H, xlmr, pad_id; float eps;
    DropCfg drop;
    void* x_pair; int ld_pair, lo_pair;      // parity mode (optional): x also as a plane pair
    const int64_t* type_ids;                 // optional [rows] token type ids (null: every token takes row 0 of the type table)
};
struct EmbedBwdArgs {
    const void* dx; const void* xhat; const float* rstd; const float* gamma;
    const int64_t* ids; const int* pos_ids;
    float* dword; float* dpos; float* dtype; float* dgamma; float* dbeta;
    int posts, T, H, pad_id, pos_pad_id;
    DropCfg drop;
    float* partial;       // optional workspace of partial_floats_embed(posts, T, H) floats
    float alpha;          // every parameter gradient is multiplied by alpha (0 is read as 1)
    uint8_t* row_state;   // optional [vocab] row flags of the word table: bit0 is set on every row that receives a gradient
    float* det_rows;      // optional [posts*T, H] fp32 workspace: deterministic mode -- the per-slot gradient rows are stored here and the
                          // word / position rows are summed from them in slot order by a second kernel (no fp32 atomics)
    int max_pos;          // rows of the position table (deterministic mode)
    unsigned* status;     // optional device words {counter, skip flag} (mmhip_set_step_guard): a non-finite element of dx -- the END of the
                          // backward's 16-bit chain, so an overflow anywhere upstream arrives here as inf / NaN -- counts and raises the flag
    const int64_t* type_ids;   // optional [rows]: with it the token-type gradient goes to row 1 of dtype from the tokens of type 1 only -- the
                               // early-fusion LXMERT tables are nn.Embedding(padding_idx = 0): type 0 gets no gradient (two-row table)
};
bool deterministic();     // MMHIP_DETERMINISTIC=1
hipError_t launch_embed_fwd(const EmbedArgs& a, int dtype, hipStream_t s);
hipError_t launch_embed_bwd(const EmbedBwdArgs& a, int dtype, hipStream_t s);

hipError_t launch_patchify(const float* pixels, void* out, int B, int img, int patch, int ld, int dtype, hipStream_t s);   // rows of ld >= 3*patch^2 columns, zero-padded
hipError_t launch_cast_pad(const float* src, void* dst, int rows, int cols, int ld, int dtype, hipStream_t s);   // dst[r][0..ld) = cast(src[r][0..cols)), 0 beyond
hipError_t launch_vit_assemble(const void* patches, const float* cls, const float* pos, void* x, int B, int P, int H, int dtype, hipStream_t s);
hipError_t launch_colsum(const void* x, int rows, int cols, int ld, float* out, int dtype, hipStream_t s, float* partial = nullptr, float alpha = 1.0f);   // out[c] += sum_r x[r][c]
size_t partial_floats_rows(int rows, int width, int nvec);
size_t partial_floats_colsum(int rows, int cols);
size_t partial_floats_embed(int posts, int T, int width);
hipError_t launch_cast(const float* src, void* dst, size_t n, int dtype, hipStream_t s);
// out[i] = min(max(in[i], 0), hi - 1): the engines' private copies of caller-supplied indices (token ids, token types).  An id outside its table --
// a tokenizer that does not match the checkpoint -- would otherwise send the embedding gather / scatter outside the table (a GPU fault); the
// reference raises IndexError on the CPU, here the step goes on with the clamped row and `*bad` (optional device word) counts the offenders
hipError_t launch_copy_ids_clamped(const int64_t* in, int64_t* out, size_t n, int64_t hi, unsigned* bad, hipStream_t s);
static constexpr int CAST_MAX_GROUP = 4;
struct CastMat { const float* src; void* dst; void* dstT; int rows, cols, tile_start; };
struct CastGroup { CastMat m[CAST_MAX_GROUP]; int count; };
hipError_t launch_cast_group(const CastMat* mats, int count, int dtype, hipStream_t s);   // dst = cast(src), dstT = cast(src)^T (optional)
hipError_t launch_cast_transpose(const float* src, void* dst, int rows, int cols, int dtype, hipStream_t s);   // dst[c][r] = src[r][c]
hipError_t launch_gather_rows_f32(const void* src, size_t src_stride, float* out, int ldo, int rows, int H, int dtype, hipStream_t s);
hipError_t launch_scatter_rows16(const void* src, void* dst, int rows, size_t dst_stride, int H, int add, int dtype, hipStream_t s);   // dst[r*stride] (+)= src[r]
hipError_t launch_dropout16(const void* src, void* dst, size_t n, const DropCfg& d, int dtype, hipStream_t s);
hipError_t launch_scatter_cls_rows(const float* d, void* dx, int posts, int T, int H, int dtype, hipStream_t s, float scale = 1.0f);

// ---------------------------------------------------------------- heads
struct FusionAttnArgs {
    const float* qk;      // [Bt, H]  W_K^T q   (fp32)
    const void* xv;       // [B*P, H] 16-bit image tokens
    float* prob;          // [Bt, P]  softmax weights (saved)
    float* xbar;          // [Bt, H]  sum_j p_j x_v[j]
    int Bt, B, P, H; float scale;
};
struct FusionAttnBwdArgs {
    const float* dxbar; const float* prob; const void* xv; float* dqk; int Bt, B, P, H; float scale;
};
hipError_t launch_fusion_attn_fwd(const FusionAttnArgs& a, int dtype, hipStream_t s);
hipError_t launch_fusion_attn_bwd(const FusionAttnBwdArgs& a, int dtype, hipStream_t s);

struct ItcArgs {
    const float* txt_e; const float* img_e;   // [B, E] raw projections
    const float* logit_scale;                 // scalar parameter
    float* txt_n; float* img_n; float* txt_inv; float* img_inv;   // normalised rows and 1/norm (saved)
    float* logits;                            // [B, B] logits_per_text
    int B, E;
};
struct ItcBwdArgs {
    const float* dlogits; const float* logits; const float* txt_n; const float* img_n; const float* txt_inv; const float* img_inv;
    const float* logit_scale;
    float* dtxt_e; float* dimg_e; float* dlogit_scale;   // dlogit_scale accumulated
    int B, E;
};
hipError_t launch_itc_fwd(const ItcArgs& a, hipStream_t s);
hipError_t launch_itc_bwd(const ItcBwdArgs& a, hipStream_t s);

struct LossArgs {
    const float* out_cls; const int64_t* onehot; const float* class_w;   // [B,C], [B,C], [C] or null
    const float* logits_per_text;                                        // [B,B] or null
    const float* out_tim; const int64_t* lbl_tim;                        // [B,2], [B] or null
    float w_cls, w_itc, w_itm;
    float* loss;            // [4]: total, cls, itc, itm
    float* d_out_cls; float* d_logits; float* d_out_tim;                 // gradients of the total loss (may be null)
    int* n_correct;         // argmax(out_cls) == argmax(onehot) count (may be null)
    int B, C;
};
hipError_t launch_loss(const LossArgs& a, hipStream_t s);

hipError_t launch_elementwise(int op, const float* a, const float* b, float* out, size_t n, float alpha, const DropCfg& drop, hipStream_t s);
enum { EW_TANH_BWD = 0, EW_RELU_BWD = 1, EW_DROPOUT = 2, EW_ADD = 3, EW_COPY = 4 };
hipError_t launch_bias_grad_f32(const float* d, int rows, int cols, int ld, float* out, int accumulate, hipStream_t s);

// ---------------------------------------------------------------- optimizer
struct AdamWArgs {
    float* p; float* g; float* m; float* v; size_t n;
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
    int zero_grad;
    float grad_scale;     // gradients are multiplied by this before use (1/world for DP averaging)
    unsigned* nonfinite;  // optional device counter: a non-finite gradient element is treated as 0 (its moments are not poisoned) and counted
    const unsigned* skip; // optional device word: non-zero = this step's gradients are void (the backward met a non-finite value in its
                          // 16-bit chain): nothing is updated, the gradient is only cleared (zero_grad) -- the whole step is skipped
};
hipError_t launch_adamw(const AdamWArgs& a, hipStream_t s);
// row-lazy AdamW over a [rows, width] table (a.p .. a.v, a.n = rows*width): rows whose state byte is 0 (no gradient now,
// moments still exactly zero) only take the decoupled decay -- bit-identical to the dense update, a quarter of its traffic
enum { ROW_HAS_GRAD = 1, ROW_HAS_MOMENTS = 2 };
hipError_t launch_adamw_rows(const AdamWArgs& a, int rows, int width, uint8_t* row_state, hipStream_t s);

}  // namespace mmhip
