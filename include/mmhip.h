/* mmhip -- C ABI of the MI355X-native late-fusion fine-tuning path (libmmhip.so).
 *
 * The reference (danaesavi/SocialMedia-TextImage-Classification-AuxLosses) has no FFI: its hot path is the Python
 * object boundary  MMLate_Model -> MM_Model.forward  (models/mm_late.py:148-193, called at :464-468 / :572-576) plus
 * loss.backward() / optimizer.step() (models/mm_late.py:489-491).  This header is the boundary a binding for that path
 * binds: plain pointers and sizes only (device pointers unless said otherwise), no torch types.  Each entry point names
 * the reference interface it replaces.
 *
 * Conventions
 *   - return value: 0 = ok; < 0 = invalid argument / shape / state (MMHIP_E_*); > 0 = a hipError_t.
 *   - never throws, never exits, never allocates device memory: the caller owns every buffer (PyTorch's allocator in
 *     the shipped binding); the library borrows pointers for the duration of a call.
 *   - all work is enqueued asynchronously on the caller's stream (`void* stream` is a hipStream_t); no host sync.
 *   - one handle per device and per thread; not thread-safe.
 */
#ifndef MMHIP_H
#define MMHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMHIP_E_INVALID (-1)   /* bad argument / shape outside what the kernels take */
#define MMHIP_E_STATE (-2)     /* call order (not bound, forward not run, ...) */
#define MMHIP_E_CAPACITY (-3)  /* batch / sequence longer than the handle was created for, or workspace too small */

/* MMHIP_BF16X3 (as mmhip_config.dtype): the strict-parity mode -- activations stay fp32 in HBM (same value as MMHIP_F32, the
 * element type the op-level entry points then take) and every Linear runs as three bf16 MFMA products of hi/lo-split
 * operands (csrc/x3.hip); ~1e-5 on the logits against the fp32 reference, where bf16 gives 5e-3..2e-2 and f16 1e-3..3e-3 */
enum { MMHIP_BF16 = 0, MMHIP_F16 = 1, MMHIP_F32 = 2, MMHIP_BF16X3 = 2,
       MMHIP_PAIR = 3 };   /* op-level entry points that say so only: the parity mode's tensors as the engine stores them -- bf16 PLANE PAIRS, a row of W values =
                              [hi = bf16(x) (W) | lo = bf16(x - hi) (W)], the same bytes as the fp32 row (csrc/mmhip_kernels.h) */
enum { MMHIP_TXT_BERT = 0, MMHIP_TXT_XLMR = 1 };
enum { MMHIP_IMG_VIT = 0, MMHIP_IMG_CLIP = 1 };   /* HF ViTModel | HF CLIPVisionModel (pre-LN, quick-GELU, bias-free patch conv, pre_layrnorm) */
enum { MMHIP_FUSION_CONCAT = 0, MMHIP_FUSION_ATTENTION = 1 };
/* gradient groups: which parameters receive a gradient for a given flag set */
enum { MMHIP_G_NEVER = 0, MMHIP_G_ITC = 1, MMHIP_G_ITM = 2, MMHIP_G_FUSION_ATT = 3, MMHIP_G_ALWAYS = 4, MMHIP_G_FROZEN = 5 };

typedef struct mmhip_config {
    int hidden, heads, inter;            /* 768, 12, 3072 (models/config.py:82-84 hard-wires 768) */
    int layers_txt, layers_img;
    int vocab, max_pos, type_vocab;
    int txt_kind, pad_id;                /* MMHIP_TXT_*; XLM-R pad id 1, BERT 0 */
    float ln_eps_txt, ln_eps_img;
    int image, patch, proj_dim, num_labels;
    int fusion;                          /* MMHIP_FUSION_*  (--fusion_name, models/run_mm_late.py:23) */
    float p_hidden, p_attn, p_head;      /* text hidden / attention-prob dropout, --dropout */
    int dtype;                           /* MMHIP_BF16 | MMHIP_F16: storage + MFMA operand type of activations; MMHIP_BF16X3: parity mode */
    int max_posts, max_text_len;         /* capacity: B <= max_posts, T <= max_text_len (ITM doubles the text rows).  LIMIT: max_text_len <= 128 -- the
                                            text tower's attention backward keeps Q, dO, K of a head and a 64 x S dS tile in LDS and gives each wave one
                                            32-key tile (S <= 128); the reference tokenises to 128 (models/config.py:60).  Image tokens (forward only):
                                            <= 608 (CLIP-ViT-L/14 @336: 577), in every dtype since round 4. */
    float loss_scale;                    /* gradient scale inside the 16-bit text tower; 0 = default (1 for bf16, 1024 for f16:
                                            f16 has 5 exponent bits, deep-layer activation gradients ~1e-6 would be subnormal).
                                            train_grad is always in true units. */
    int img_kind;                        /* MMHIP_IMG_*: BASELINE config 4 = MMHIP_IMG_CLIP with hidden_img 1024, heads_img 16, inter_img 4096,
                                            patch 14, image 224 (257 tokens) or 336 (577), fusion MMHIP_FUSION_CONCAT */
    int hidden_img, heads_img, inter_img;/* image tower width / heads / MLP width; 0 = the text tower's (ViT-B/16) */
} mmhip_config;

typedef struct mmhip_param_info {
    char name[192];                      /* state_dict key of the reference checkpoint (transformers 4.25.1 naming) */
    int ndim;
    int64_t dims[4];
    int buffer;                          /* 0 = frozen flat buffer (vision tower, mm_late.py:67-69), 1 = trainable */
    int group;                           /* MMHIP_G_* */
    uint64_t offset, numel;              /* element offset inside the flat fp32 buffer */
} mmhip_param_info;

typedef struct mmhip_engine* mmhip_handle;

/* ---- lifetime; replaces MM_Model.__init__ (models/mm_late.py:50-89) minus weight loading */
int mmhip_create(const mmhip_config* cfg, mmhip_handle* out);
void mmhip_destroy(mmhip_handle h);
int mmhip_param_count(mmhip_handle h);
int mmhip_param_info_at(mmhip_handle h, int index, mmhip_param_info* out);
uint64_t mmhip_buffer_numel(mmhip_handle h, int buffer);        /* fp32 elements of flat buffer 0 / 1 */
uint64_t mmhip_workspace_bytes(mmhip_handle h);                 /* activations + 16-bit weight copies */
/* frozen / train / train_grad are flat fp32 device buffers laid out as mmhip_param_info_at describes */
int mmhip_bind(mmhip_handle h, float* frozen, float* train, float* train_grad, void* workspace, uint64_t workspace_bytes);
/* re-derive the 16-bit (and transposed) GEMM operand copies from the fp32 masters: which = 1 frozen, 2 train, 3 both.
 * Call after load_state_dict (models/mm_late.py:343-345) and after every optimizer step. */
int mmhip_refresh_weights(mmhip_handle h, int which, void* stream);

/* ---- MM_Model.forward (models/mm_late.py:148-193).  ids/mask/tim_* are int64 [B,T]; pixels fp32 [B,3,image,image];
 * outputs fp32: out_cls [B,num_labels], logits_per_text [B,B], out_tim [B,2] (only when tim_ids != NULL),
 * mm_features [B,hidden].  train != 0 applies dropout with masks derived from `seed`.
 * Token ids are clamped into [0, vocab) as they are copied into the engine (mmhip_early_forward: token types into [0, type_vocab), ITM source
 * rows into [0, B) as well): an id outside its table would send the embedding gather and the gradient scatter outside the table -- a GPU
 * fault, where the reference raises IndexError on the CPU.  Out-of-range ids are therefore a caller error that degrades to a wrong row, never a fault. */
/* (pixels may be NULL after mmhip_vision_import, see below) */
int mmhip_forward(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                  const int64_t* tim_mask, int B, int T, int train, uint64_t seed, float* out_cls, float* logits_per_text,
                  float* out_tim, float* mm_features, void* stream);

/* ---- loss mixing of MMLate_Model.train (models/mm_late.py:471-487) on the outputs of the last forward, fused with its
 * own backward: loss[4] = {total, cls, itc, itm}; the output gradients stay inside the handle for mmhip_backward(NULL...).
 * onehot int64 [B,C] (models/datasets.py one-hot labels), class_w fp32 [C] or NULL (run_mm_late.py:85),
 * lbl_tim int64 [B] or NULL.  weights: w_cls = 1 - betas, w_itc / w_itm = beta or 0 when the aux loss is off. */
int mmhip_loss(mmhip_handle h, const int64_t* onehot, const float* class_w, const int64_t* lbl_tim, float w_cls, float w_itc,
               float w_itm, float* loss, int* n_correct, void* stream);

/* ---- image-tower output cache (optional; sized for 288 GB of HBM: 306 KB per post, a 100 k-image data set is 31 GB).
 * The vision tower is frozen (models/mm_late.py:67-69) and has no dropout, so its outputs (last_hidden_state [197,768] in the
 * compute dtype + pooler_output [768] fp32) are a pure function of the pixels: a caller that sees the same image again (every
 * epoch after the first) may keep them.  `cache` is caller-owned device memory of cache_records * mmhip_vision_record_bytes(h);
 * slots = int64 [B] on the device, negative = skip that post.
 *   mmhip_vision_export: after mmhip_forward, copy the tower outputs of the B posts of that forward into their slots.
 *   mmhip_vision_import: before mmhip_forward, load the outputs of B posts from their slots; the next mmhip_forward with
 *   pixels == NULL and the same B uses them and skips the tower (MMHIP_E_STATE otherwise).  Results are bit-identical. */
uint64_t mmhip_vision_record_bytes(mmhip_handle h);
int mmhip_vision_export(mmhip_handle h, const int64_t* slots, void* cache, uint64_t cache_records, void* stream);
int mmhip_vision_import(mmhip_handle h, const int64_t* slots, const void* cache, uint64_t cache_records, int B, void* stream);

/* ---- loss.backward() (models/mm_late.py:489): gradients of every trainable parameter, written to train_grad.
 * CONTRACT: train_grad must be ZERO on entry over the ranges that receive gradients, and holds exactly this call's
 * gradient on exit.  (The head and embedding kernels add into it -- fp32 `+=` / row atomics, and they set bit0 of the
 * word-table row flags -- while the text-layer weight / bias gradients are plain stores: calling backward twice without
 * clearing in between is NOT gradient accumulation; a caller that wants micro-batches sums the flat buffers itself.)
 * mmhip_adamw / mmhip_adamw_rows with zero_grad = 1 re-establish the entry condition (and clear bit0) as they consume the
 * gradient; a caller that does not run them clears train_grad and bit0 of the row flags itself.  Pass NULL pointers to use the gradients mmhip_loss left in
 * the handle, or explicit fp32 output gradients (autograd binding).  Stages let a data-parallel caller start the
 * all-reduce of finished parameter ranges while later stages run: stage 0 = heads, 1..layers_txt = text layers
 * last -> first, layers_txt+1 = embeddings.  mmhip_backward runs all of them and the finish.
 * Internal concurrency: the engine forks an internal stream from the caller's stream (events) for work that is off the
 * critical path -- the frozen image tower beside the text tower in mmhip_forward, a layer's parameter gradients beside the
 * next layer's activation gradients in backward -- and joins it before mmhip_forward returns / in mmhip_backward_finish.
 * With stages: the gradient range of stage k is final in the caller's stream order once stage k+1 has been enqueued for
 * k = 0 and for the embeddings stage, and for a text-layer stage once mmhip_backward_finish (or two further stages) has
 * been enqueued; a caller that wants per-stage exchange calls mmhip_backward_join_stage(k) first. MMHIP_OVERLAP=0 disables. */
int mmhip_backward(mmhip_handle h, const float* d_out_cls, const float* d_logits_per_text, const float* d_out_tim,
                   const float* d_mm_features, void* stream);
int mmhip_backward_begin(mmhip_handle h, const float* d_out_cls, const float* d_logits_per_text, const float* d_out_tim,
                         const float* d_mm_features, void* stream);
int mmhip_backward_stage(mmhip_handle h, int stage, void* stream);
int mmhip_backward_finish(mmhip_handle h, void* stream);
/* make the caller's stream wait for the side-stream work of text-layer stage `stage` (no-op for other stages) */
int mmhip_backward_join_stage(mmhip_handle h, int stage, void* stream);
int mmhip_num_backward_stages(mmhip_handle h);
/* element range [begin, end) of the trainable flat buffer whose gradients are final after `stage` */
int mmhip_stage_grad_range(mmhip_handle h, int stage, uint64_t* begin, uint64_t* end);

/* ---- optimizer.step(): torch.optim.AdamW over a flat range (models/mm_late.py:420-422, models/utils.py:280-292).
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce); zero_grad clears it afterwards. */
int mmhip_adamw(float* p, float* g, float* m, float* v, uint64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int step, float grad_scale, int zero_grad, void* stream);

/* Row-lazy form of the same update for the word-embedding table [rows, width] (69 % of Bernice's parameters; at most B*T
 * of its rows receive a gradient per step).  `row_state` is a caller-owned byte per row, 4-byte aligned, allocated with
 * rows rounded up to a multiple of 4 and zero-initialised together with the moments: bit0 = the row holds a gradient now
 * (set by the backward pass once mmhip_set_row_state() registered the array, and by the caller for rows it adds itself,
 * e.g. after a data-parallel row exchange), bit1 = the row's moments are non-zero.  Rows with state 0 only take the
 * decoupled decay, which is bit-identical to what torch.optim.AdamW computes for g = m = v = 0 (models/mm_late.py:420-422
 * runs the dense optimizer over nn.Embedding's dense gradient). */
int mmhip_adamw_rows(float* p, float* g, float* m, float* v, int rows, int width, uint8_t* row_state, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream);
/* register (or with NULL remove) the word-table row flags the backward pass maintains */
int mmhip_set_row_state(mmhip_handle h, uint8_t* row_state);

/* ---- overflow guard (f16 mode carries the text tower's gradients multiplied by loss_scale; an inf / NaN there would poison the
 * optimizer state for good).  mmhip_set_nonfinite_counter: a uint32 on the device (zero it yourself); the AdamW entry points
 * read a non-finite gradient element as 0 -- its moments and parameter only take the decay -- and add to the counter once per
 * thread that met one.  The caller polls the counter when it reads the loss anyway and reacts: mmhip_set_loss_scale lowers
 * the scale (0 = back to the default); with bf16 (no scale) a count means a real divergence. */
int mmhip_set_nonfinite_counter(uint32_t* device_counter);
/* Whole-step form of the guard (round 3): TWO uint32 on the device, {counter, flag}.  mmhip_backward_begin clears the flag; the
 * embedding backward -- the end of the backward's 16-bit chain, where an overflow anywhere upstream arrives as inf / NaN -- raises
 * it and counts; every AdamW entry point that finds the flag set leaves parameters and moments alone and only clears the gradient,
 * so an overflowed step is skipped as a whole without the host looking (the host reads the counter a step late and lowers the
 * scale).  While armed, an f16 engine runs all its AdamW launches after the backward (no per-layer launches beside it).
 * NULL disarms.  Like the counter it is one registration per process; it serves the handle-less AdamW entry points above and
 * handles that have no guard of their own (below). */
int mmhip_set_step_guard(uint32_t* device_words2);
/* Per-handle guard (round 4; supersedes the process-wide registration for a model that owns a handle): the same two words
 * {counter, flag}, registered on the handle.  mmhip_backward_begin of THIS handle clears the flag, its embedding backward raises
 * it, and the AdamW launches inside mmhip_train_step[_dp] honour it; two models in one process (late + early fusion, or two
 * trainers) no longer share a flag.  A caller that runs the optimizer itself (staged data-parallel step) passes the same words
 * to mmhip_adamw_guarded / mmhip_adamw_rows_guarded (NULL = unguarded).  Under data parallelism the caller MAX-reduces word [1]
 * of ITS handle before the optimizer so that the replicas take the same decision.
 * What the guard guarantees: in f16 (the only dtype with a loss scale) a void step is skipped as a WHOLE -- while the guard is
 * armed every AdamW launch of an f16 engine follows the backward.  In bf16 / bf16x3 the per-layer AdamW launches run beside the
 * backward (before the flag can be raised), so there the guard is per element (a non-finite gradient element is read as 0 and
 * counted) and the host raises FloatingPointError one step later: a non-finite bf16 gradient is a divergence, not an overflow. */
int mmhip_set_guard(mmhip_handle h, uint32_t* device_words2);
int mmhip_adamw_guarded(float* p, float* g, float* m, float* v, uint64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int zero_grad, void* stream, uint32_t* guard_words2);
int mmhip_adamw_rows_guarded(float* p, float* g, float* m, float* v, int rows, int width, uint8_t* row_state, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream,
                             uint32_t* guard_words2);
int mmhip_set_loss_scale(mmhip_handle h, float loss_scale);
/* Clamped-index counter.  mmhip_forward / mmhip_train_step clamp token ids into [0, vocab) on their way into the engine (above: why); the reference
 * raises IndexError for such an id on the CPU (nn.Embedding).  With a device word registered here every clamped index adds 1 to it, so a caller can
 * turn the silent wrong row into the reference's error one step late and without a synchronisation (the shipped binding copies the word to pinned
 * memory with the overflow guard's and raises IndexError at the next step / at the end of an evaluation loop).  NULL = do not count.
 * mmhip_early_set_index_counter: the same for the early-fusion engine (token ids, token types). */
int mmhip_set_index_counter(mmhip_handle h, uint32_t* device_word);
/* Parity mode (MMHIP_BF16X3) only: how many bf16 MFMA products the BACKWARD's matrix products (activation gradients dX = dY . W, weight
 * gradients dW = dY^T . X of the text tower) take per reduction slice.  3 (default): hi.hi + lo.hi + hi.lo, as the forward -- gradients
 * within 1e-3 of the fp32 reference (measured 3e-5).  2: hi.hi + lo.hi -- the second operand (W, X) is read rounded to bf16.  1: hi.hi --
 * both operands rounded to bf16 at the matrix cores; tensors stay stored as plane pairs / fp32, accumulation, row operations and the attention
 * backward are unchanged.  The FORWARD (logits, loss: what the reference's tolerance is stated on, BASELINE.json north_star) always takes
 * three.  Measured gradient bounds per setting: DESIGN.md section 4c.  Env MMHIP_X3_BWD=n sets the default at mmhip_create.
 * Returns MMHIP_E_STATE for products != 3 on a handle of another dtype. */
int mmhip_set_backward_products(mmhip_handle h, int products);

/* ---- one whole training step of MMLate_Model.train (models/mm_late.py:452-491: zero_grad, forward, loss mix, backward,
 * optimizer.step) in ONE call: mmhip_forward(train) + mmhip_loss + mmhip_backward + AdamW over exactly the parameter ranges
 * that receive a gradient for this flag set (torch.optim.AdamW skips `grad is None` tensors: never-used heads always; the ITC
 * group unless use_itc; linear_tim unless use_itm; the fusion-attention group for 'concat') + mmhip_refresh_weights(train).
 * adam_m / adam_v: the caller's moment buffers, laid out like the trainable flat buffer (zero before the first step; the
 * word-table row flags of mmhip_set_row_state must then have bit1 clear).  loss[4] / n_correct as in mmhip_loss.  Everything
 * is enqueued on `stream`; the host returns without waiting.  Single-rank form: a data-parallel caller that exchanges
 * gradients between backward and the optimizer uses the staged calls above. */
int mmhip_train_step(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                     const int64_t* tim_mask, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T,
                     uint64_t seed, int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* loss,
                     int* n_correct, void* stream);

/* ---- the same step under data parallelism (reference: none -- SURVEY.md 2.1; semantics "the reference's step per rank, gradients averaged").
 * The gradient exchange belongs to the caller (torch.distributed / RCCL in smtc_amd/dist.py), the enqueue order to the library: `on_stage`
 * is called on the host, between launches,
 *   on_stage(user, st)  0 <= st < mmhip_num_backward_stages: the parameter gradients of backward stage st (mmhip_stage_grad_range) are final
 *                       in `stream` order -- called while the stages below st are already enqueued, so a collective started here travels
 *                       beside them;
 *   on_stage(user, MMHIP_CB_WAIT_DENSE)   after the last stage: make `stream` wait for the dense collectives (Work.wait());
 *   on_stage(user, MMHIP_CB_FINISH_ROWS)  after the dense AdamW and the 16-bit weight refresh were enqueued: finish the row-sparse
 *                       word-table exchange (it travelled meanwhile); the row-lazy AdamW of the table follows.
 * A non-zero return aborts the step with that code.  grad_scale = 1 / world.  Everything else as mmhip_train_step. */
enum { MMHIP_CB_WAIT_DENSE = -1, MMHIP_CB_FINISH_ROWS = -2, MMHIP_CB_WAIT_BUCKET = -3 };
/* Per-bucket optimizer (round 5).  on_stage(user, st) may return MMHIP_CB_BUCKET instead of 0: "the collective that carries the gradients of every
 * stage since my last MMHIP_CB_BUCKET answer, st included, has been started".  If the library runs the layer optimizers beside the backward (side
 * stream on), it then calls on_stage(user, MMHIP_CB_WAIT_BUCKET): the caller makes the library's SIDE stream (mmhip_side_stream) wait for that
 * collective -- with RCCL a stream-side wait, nothing blocks on the host -- and returns 0, upon which the AdamW and the operand refresh of the text
 * layers the bucket carries are enqueued on the side stream, beside the backward stages below, exactly as in the single-rank step; or it returns
 * MMHIP_CB_HANDLED after running the optimizer of the bucket's dense ranges itself ON THAT STREAM (sharded optimizer), and only the refresh follows.
 * Ranges that are not text layers, and stages after the last MMHIP_CB_BUCKET answer, stay behind MMHIP_CB_WAIT_DENSE.  A caller that never answers
 * MMHIP_CB_BUCKET gets round 4's single barrier. */
enum { MMHIP_CB_BUCKET = 2 };
void* mmhip_side_stream(mmhip_handle h);      /* hipStream_t of the library's side stream of the backward, NULL when MMHIP_OVERLAP=0 */
/* on_stage(user, MMHIP_CB_WAIT_DENSE) may return MMHIP_CB_HANDLED instead of 0: the caller has run the optimizer over the DENSE parameter
 * ranges itself (sharded: reduce-scatter of the gradients, AdamW on the rank's own shard with moments only it keeps, all-gather of the updated
 * parameters -- smtc_amd/dist.py ShardedBuckets); the library then skips its dense AdamW launches and goes on with the 16-bit weight refresh
 * and the word-table rows (adam_m / adam_v are then only dereferenced over the word table's range). */
enum { MMHIP_CB_HANDLED = 1 };
typedef int (*mmhip_exchange_cb)(void* user, int stage);
int mmhip_train_step_dp(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                        const int64_t* tim_mask, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T,
                        uint64_t seed, int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* loss,
                        int* n_correct, void* stream, mmhip_exchange_cb on_stage, void* user);

/* Debug: phase ends of the last forward / train step, in milliseconds after the forward's fork, from HIP events recorded on the
 * streams the phases run on (no profiler in the way): ms[0] image tower end, [1] text tower end, [2] forward end, [3] backward
 * end, [4] step end; -1 = not recorded.  enable != 0 arms the events for the steps that follow.  Synchronises the device.
 * No reference counterpart (run_mm_late.py times whole epochs only, models/mm_late.py:424-430). */
int mmhip_step_spans(mmhip_handle h, int enable, float* ms);

/* ---- timing of the dominant kernel for bench.py: HIP events recorded around every MFMA NT-GEMM launch issued by the
 * handle, on the stream the launch goes to, while enabled; returns accumulated milliseconds, launches and algorithmic FLOPs
 * since reset.  enable = 1: the engine keeps its internal side streams (the conditions of a normal step: a launch may share
 * the chip with the other tower / the weight-gradient GEMM); enable = 2: side streams off, every kernel alone on the chip. */
int mmhip_gemm_timing(mmhip_handle h, int enable, int reset, double* ms, uint64_t* launches, double* flops);
/* the same launches as a NUL-terminated text table, one line per (M, N, K, epilogue flags, forced tile): count, average
 * microseconds, TFLOP/s -- call before the reset (diagnostic: bench.py --gemm-shapes) */
int mmhip_gemm_timing_by_shape(mmhip_handle h, char* out, uint64_t capacity);

/* ---- input pipeline, image leg: decoded RGB bytes -> pixel_values [n,3,S,S] fp32.
 * Replaces the ViT feature extractor call made per item inside the reference's Dataset.__getitem__
 * (models/datasets.py:160-181: Image.open().convert("RGB") -> processor(images=...)): PIL Image.resize((S,S), BILINEAR)
 * (Pillow's two-pass 22-bit fixed-point resampler, 8-bit intermediate), rescale 1/255, normalize -- bit-identical.
 *   1. mmhip_image_plan_words / _build: HOST ONLY (no GPU call): per image the resampling windows and integer weights,
 *      computed in double with Pillow's operation order, written into a caller-owned int32 plan (pin it and copy it to
 *      the device asynchronously).  offsets[i] = byte offset of image i (uint8, HWC, RGB, rows packed) in `images`; offsets are
 *      multiples of 16 and `images` is allocated with 16 spare bytes at the end (rows are staged in 16-byte chunks).
 *   2. mmhip_image_preprocess: two kernel launches on `stream` for the whole batch.  `lut` = float[3][256] on the device:
 *      channel value -> normalized float (the caller builds it with the reference's float64/float32 arithmetic, so the
 *      device does integer work only); `tmp` = device scratch of mmhip_image_plan_tmp_bytes(plan) bytes;
 *      out_u8 (optional, [n,S,S,3]) receives the resized bytes. */
uint64_t mmhip_image_plan_words(int n, const int32_t* heights, const int32_t* widths, int out_size);   /* 0 = invalid sizes */
int mmhip_image_plan_build(int n, const uint64_t* offsets, const int32_t* heights, const int32_t* widths, int out_size,
                           int32_t* plan, uint64_t capacity_words);
uint64_t mmhip_image_plan_tmp_bytes(const int32_t* plan_host);
int mmhip_image_preprocess(const uint8_t* images, const int32_t* plan_host, const int32_t* plan_dev, const float* lut,
                           float* out_f32, uint8_t* out_u8, uint8_t* tmp, void* stream);

/* ---- operator-level entry points (parity tests; the engine calls the same launchers; the first version of the early-fusion
 * LXMERT path, reference models/mm_early.py:105-172, is built on them).  dtype = MMHIP_BF16 | MMHIP_F16 | MMHIP_F32 (= the
 * bf16x3 parity products on fp32 matrices).  Matrices are row-major with explicit leading dimensions (elements). */
/* C[M,N] = epilogue(A[M,K] . B[N,K]^T): + bias[N] (fp32, may be NULL); act: 0 none, 1 exact-erf GELU, 2 tanh;
 * aux_pre (may be NULL) receives the value before the activation; mul_gelu_grad_of (may be NULL): value *= gelu'(that);
 * dropout (p > 0) on element index m*N+n with (seed, stream_id); + residual (may be NULL); out_f32 selects fp32 C. */
int mmhip_op_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const float* bias, int act, void* aux_pre, int ldaux, const void* mul_gelu_grad_of, int ldmul,
                     float p_drop, uint64_t seed, uint32_t stream_id, const void* residual, int ldres, int out_f32,
                     int force_slow, void* stream);
/* C[Nn,Nc] (fp32) (+)= A[M,Nn]^T . B[M,Nc];  colsum (may be NULL): fp32 [Nn] (+)= sum_m A[m][n], the bias gradient that goes
 * with a weight gradient, from the same operand tiles */
int mmhip_op_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int Nn, int Nc,
                     int accumulate, int force_slow, float* colsum, void* stream);
/* many such products in few launches (the kernel takes up to 8 problems per launch and fills the chip with their tiles together:
 * a single 768 x 768 weight gradient is 18 tiles).  Used by the early-fusion path, which queues the weight gradients of a whole
 * backward pass. */
typedef struct mmhip_tn_problem {
    const void* A; const void* B; float* C;
    int32_t M, Nn, Nc, lda, ldb, ldc;
    float* colsum;
} mmhip_tn_problem;
int mmhip_op_gemm_tn_group(int dtype, const mmhip_tn_problem* problems, int count, int accumulate, void* stream);
/* refresh of many weight copies in few launches: dst = cast(src) [rows, cols] and (dst_t may be NULL) dst_t = cast(src)^T [cols, rows], every fp32
 * 64 x 64 tile read once; rows, cols multiples of 4 */
typedef struct mmhip_cast_mat { const float* src; void* dst; void* dst_t; int32_t rows, cols; } mmhip_cast_mat;
int mmhip_op_cast_group(int dtype, const mmhip_cast_mat* mats, int count, void* stream);
/* composite operators of the early-fusion path: one post-LN sub-block of a BERT-shaped stream per call, on caller-owned buffers
 * (activation-typed [rows, width] matrices; weights as the activation-typed copies, `*T` = transposed copies [K_in, N_out] for the
 * input gradients).  Self-attention block:  y = LayerNorm(dropout(att Wo^T + bo) + x), att = softmax(Q K^T / 8 + maskbias) V,
 * [Q|K|V] = x Wqkv^T + bqkv (64-wide heads, rows = posts * S); dropouts are the hash dropouts of the fused epilogues (p_att on the
 * probabilities, p_hid on the projection output), replayed in the backward from (seed).  Saved for the backward: qkv, att, lse, pre,
 * mean, rstd.  The backward returns dx (incl. the residual branch) and leaves the operands of the weight gradients in
 * dd (= d of the projection output; = dpre when p_hid == 0, then dd is not written) / att (Wo), dqkv / x (Wq, Wk, Wv): the caller
 * runs them through mmhip_op_gemm_tn_group.  LayerNorm weight / bias gradients are ADDED to dgamma / dbeta. */
int mmhip_op_self_att_block_fwd(int dtype, const void* x, const float* maskbias, const void* wqkv, const float* bqkv, const void* wo, const float* bo,
                                const float* gamma, const float* beta, float eps, int posts, int S, int heads, float p_att, float p_hid, uint64_t seed,
                                void* qkv, void* att, float* lse, void* pre, float* mean, float* rstd, void* y, void* stream);
int mmhip_op_self_att_block_bwd(int dtype, const void* dy, const float* maskbias, const void* wqkvT, const void* woT, const float* gamma, int posts, int S,
                                int heads, float p_att, float p_hid, uint64_t seed, const void* qkv, const void* att, const float* lse, const void* pre,
                                const float* mean, const float* rstd, float* dgamma, float* dbeta, void* dpre, void* dd, void* datt, void* dqkv, void* dx,
                                void* stream);
/* cross-attention block (LXMERT cross-modality layers, reference models/mm_early.py:121-127 via HF LxmertCrossAttentionLayer): queries from
 * xq [posts*Sq, H], keys / values from xc [posts*Sk, H];  y = LayerNorm(dropout(att Wo^T + bo) + xq).  wqkv / bqkv = the fused [Wq; Wk; Wv]
 * [3H, H] copy and bias; keybias [posts, S] additive key mask, S = max(Sq, Sk) (entries past Sk are not read).
 * Round 5 -- compact tensors, every dtype: qkv is [max(Mq, Mc), 3H] with Mq = posts*Sq, Mc = posts*Sk: the query projection fills rows [0, Mq) of
 * columns [0, H), the key / value projection rows [0, Mc) of columns [H, 3H) (row p*Sq + q = query q of post p, row p*Sk + k = its key k); att is
 * [Mq, H]; lse [posts, heads, S].  The attention kernels take the two row pitches and lengths, skip key tiles past Sk, and nothing is padded,
 * cleared, remapped or copied (rounds 3-4 did: tq, tkv, attq, dattq, dq, dkv were their scratch -- now unused, may be NULL).  The dropout masks keep
 * the element indices of the S x S layout.  Saved for the backward: qkv, att, lse, pre, mean, rstd.  Backward: dxq (incl. the residual branch), dxc;
 * datt [Mq, H]; dqkv laid out like qkv.  Weight-gradient operands, as the block leaves them: dd / att (Wo), dqkv[:Mq, :H] / xq (Wq),
 * dqkv[:Mc, H:] / xc (Wk, Wv) -- leading dimension 3H. */
int mmhip_op_cross_att_block_fwd(int dtype, const void* xq, const void* xc, const float* keybias, const void* wqkv, const float* bqkv, const void* wo,
                                 const float* bo, const float* gamma, const float* beta, float eps, int posts, int Sq, int Sk, int heads, float p_att,
                                 float p_hid, uint64_t seed, void* qkv, void* att, float* lse, void* tq, void* tkv, void* attq, void* pre, float* mean,
                                 float* rstd, void* y, void* stream);
int mmhip_op_cross_att_block_bwd(int dtype, const void* dy, const float* keybias, const void* wqkvT, const void* woT, const float* gamma, int posts, int Sq,
                                 int Sk, int heads, float p_att, float p_hid, uint64_t seed, const void* qkv, const void* att, const float* lse,
                                 const void* pre, const float* mean, const float* rstd, float* dgamma, float* dbeta, void* dpre, void* dd, void* dattq,
                                 void* datt, void* dqkv, void* dq, void* dkv, void* dxq, void* dxc, void* stream);
/* feed-forward block:  y = LayerNorm(dropout(GELU(x W1^T + b1) W2^T + b2) + x); saved: h (activation), u (pre-activation), pre, mean, rstd;
 * backward: du = (dd W2) * gelu'(u), dx = du W1 + dpre; weight-gradient operands: dd / h (W2), du / x (W1). */
int mmhip_op_ffn_block_fwd(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma, const float* beta,
                           float eps, int M, int H, int I, float p_hid, uint64_t seed, void* h, void* u, void* pre, float* mean, float* rstd, void* y,
                           void* stream);
int mmhip_op_ffn_block_bwd(int dtype, const void* dy, const void* w1T, const void* w2T, const float* gamma, int M, int H, int I, float p_hid, uint64_t seed,
                           const void* u, const void* pre, const float* mean, const float* rstd, float* dgamma, float* dbeta, void* dpre, void* dd, void* du,
                           void* dx, void* stream);
int mmhip_op_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                           int rows, int width, float eps, void* stream);
int mmhip_op_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                           void* dx, const void* dres, float* dgamma, float* dbeta, int rows, int width, void* stream);
/* qkv [posts*S, 3*hidden] packed q|k|v; maskbias fp32 [posts,S] additive (0 / -inf) or NULL; lse fp32 [posts,heads,S].
 * dtype MMHIP_PAIR: qkv / ctx / dctx / dqkv are plane pairs (rows of 2 * 3 * hidden resp. 2 * hidden bf16), three MFMA products per matrix product;
 * mmhip_op_layernorm_fwd with MMHIP_PAIR: x fp32, y a plane pair (rows of 2 * width bf16) */
int mmhip_op_attn_fwd(int dtype, const void* qkv, const float* maskbias, void* ctx, float* lse, int posts, int S, int heads,
                      float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int mmhip_op_attn_bwd(int dtype, const void* qkv, const float* maskbias, const void* ctx, const void* dctx, const float* lse,
                      void* dqkv, int posts, int S, int heads, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int mmhip_op_colsum(int dtype, const void* x, int rows, int cols, int ld, float* out, void* stream);
int mmhip_op_cast(int dtype, const float* src, void* dst, uint64_t n, int transpose_rows, int transpose_cols, void* stream);
/* hardware-layout probe: runs one MFMA of each shape and one transposing LDS read on index-coded data and writes what
 * each lane received (tests/test_hw_layouts.py checks the lane maps every kernel here is built on). out: int32[4096] */
int mmhip_op_probe_layouts(int32_t* out, void* stream);

/* ==== early-fusion engine (BASELINE config 5): the LXMERT step of the reference's models/mm_early.py as ONE native call path (round 4;
 * csrc/early.hip).  Boundary replaced: class Lxmert -- forward(input_ids, attention_mask, token_type_ids, features, normalized_boxes,
 * tim_inputs) -> (linear_output, max_embeddings_t, max_embeddings_v, out_tim), models/mm_early.py:121-163, get_logits_per_text :165-172 --
 * and the step body of MMEarly_Model.train :332-407 (loss mix :366-379, loss.backward(), optimizer.step()).  Below it: HF LxmertModel
 * (transformers 4.25.1: embeddings with padding_idx = 0 on all three tables, visual-feature encoder, 9 language + 5 relational + 5
 * cross-modality layers whose ONE cross-attention module serves both directions) on hand-written HIP kernels, two internal streams
 * (language = the caller's, vision = internal).  Same conventions as the late-fusion handle: caller-owned flat fp32 parameter / gradient
 * buffers laid out as mmhip_early_param_info_at describes (names = the reference module's state_dict keys: `model.*`, `linear_fusion.*`,
 * `linear.*`, `linear_tim.*`, `logit_scale`), caller-owned workspace, caller's stream, error codes. */
typedef struct mmhip_early_config {
    int hidden, heads, inter;            /* 768, 12, 3072 (heads * 64 == hidden) */
    int l_layers, r_layers, x_layers;    /* 9, 5, 5 */
    int vocab, max_pos, type_vocab;      /* 30522, 512, 2 */
    int feat_dim, pos_dim;               /* 2048 ROI feature width, 4 box coordinates (multiples of 4) */
    int num_labels;
    int max_posts, max_text_len, max_boxes;   /* capacity: B <= max_posts, T <= max_text_len <= 128, boxes <= max_boxes <= 128 (the ITM pass doubles the posts) */
    int dtype;                           /* MMHIP_BF16 | MMHIP_F16 | MMHIP_BF16X3 */
    float p_hidden, p_attn, p_head;      /* LXMERT hidden / attention dropout (0.1), the head's --dropout */
    float ln_eps;                        /* 1e-12 */
} mmhip_early_config;
typedef struct mmhip_early* mmhip_early_handle;
int mmhip_early_create(const mmhip_early_config* cfg, mmhip_early_handle* out);
void mmhip_early_destroy(mmhip_early_handle h);
int mmhip_early_param_count(mmhip_early_handle h);
int mmhip_early_param_info_at(mmhip_early_handle h, int index, mmhip_param_info* out);      /* buffer = 1 for every entry; group: MMHIP_G_NEVER = the pooler */
uint64_t mmhip_early_numel(mmhip_early_handle h);
uint64_t mmhip_early_workspace_bytes(mmhip_early_handle h);
/* params / grads: flat fp32 device buffers (256-byte aligned; grads may be NULL for inference); the call clears the boxes' key bias on `stream` */
int mmhip_early_bind(mmhip_early_handle h, float* params, float* grads, void* workspace, uint64_t workspace_bytes, void* stream);
/* 16-bit (parity mode: fp32) operand copies of the weights + transposed copies; call after the parameters changed */
int mmhip_early_refresh_weights(mmhip_early_handle h, void* stream);
int mmhip_early_set_index_counter(mmhip_early_handle h, uint32_t* device_word);      /* see mmhip_set_index_counter */
/* Measurement only, as mmhip_gemm_timing: HIP events around every NT GEMM launch of the following mmhip_early_train_step calls, on the stream each
 * launch goes to; sums over the launches timed since the last reset (synchronises).  No reference counterpart. */
int mmhip_early_gemm_timing(mmhip_early_handle h, int enable, int reset, double* ms, uint64_t* launches, double* flops);
/* Lxmert.forward.  ids / mask / token_type_ids (may be NULL = zeros): int64 [B, T]; feats fp32 [B, Nb, feat_dim]; boxes fp32 [B, Nb, pos_dim];
 * tim_*: the swapped texts of the ITM pass (NULL = no ITM) -- both passes run as ONE encoder pass of 2B posts.  Outputs (may be NULL) fp32:
 * out [B, num_labels], emb_t [B, H] (masked max over tokens, detached), emb_v [B, H] (max over boxes), out_tim [B, 2]. */
int mmhip_early_forward(mmhip_early_handle h, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, const float* feats, const float* boxes,
                        const int64_t* tim_ids, const int64_t* tim_mask, const int64_t* tim_token_type_ids, int B, int T, int Nb, int train, uint64_t seed,
                        float* out, float* emb_t, float* emb_v, float* out_tim, void* stream);
/* loss mix models/mm_early.py:366-379 on the last forward's outputs; loss[4] = total, cls, itc, itm; logits_per_text (may be NULL) [B, B].
 * Leaves the output gradients in the handle and ADDS d logit_scale to the gradient buffer. */
int mmhip_early_loss(mmhip_early_handle h, const int64_t* onehot, const float* class_w, const int64_t* lbl_tim, float w_cls, float w_itc, float w_itm, float* loss,
                     float* logits_per_text, void* stream);
/* loss.backward(): NULL pointers = the gradients mmhip_early_loss left; else explicit fp32 d_out [B, C], d_emb_v [B, H] | NULL, d_out_tim [B, 2] | NULL.
 * Entry condition: the gradient buffer is zero over the ranges that receive gradients (the step's AdamW re-establishes it). */
int mmhip_early_backward(mmhip_early_handle h, const float* d_out, const float* d_emb_v, const float* d_out_tim, void* stream);
/* backward stages for a data-parallel caller: 0 heads, then the cross-modality layers last -> first, then the language / relational layers by
 * depth, then the inputs (visual-feature encoder + embeddings); [begin, end) of the flat buffers whose gradients are final after `stage` */
int mmhip_early_num_stages(mmhip_early_handle h);
int mmhip_early_stage_grad_range(mmhip_early_handle h, int stage, uint64_t* begin, uint64_t* end);
/* one whole training step (forward, loss mix, backward, AdamW over the ranges that receive gradients for this flag set, operand refresh).
 * itm_src: device int64 [B], row b of the ITM pass = row itm_src[b] of this batch (MMEarly_Model.prepare_itm_inputs' draw); on_stage: the
 * exchange hook of mmhip_train_step_dp (may be NULL): on_stage(user, st) once stage st's gradient range is final in `stream` order,
 * on_stage(user, MMHIP_CB_WAIT_DENSE) before the optimizer; grad_scale = 1 / world. */
int mmhip_early_train_step(mmhip_early_handle h, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, const float* feats, const float* boxes,
                           const int64_t* itm_src, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T, int Nb, uint64_t seed,
                           int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int step, float grad_scale, float* loss, void* stream, mmhip_exchange_cb on_stage, void* user);

const char* mmhip_version(void);
#ifdef __cplusplus
}
#endif
#endif
