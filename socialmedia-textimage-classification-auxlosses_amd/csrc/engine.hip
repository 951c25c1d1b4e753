// mmhip engine: owns the flat parameter layout, the workspace carve-up and the launch sequence of one
// forward / backward of the late-fusion model (reference models/mm_late.py:148-193 + the HF dual encoder it calls).
// Everything is enqueued on the caller's stream from this single C++ call path: no per-op host round trip.
#include <string>
#include <vector>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include "mmhip_common.h"
#include "mmhip_kernels.h"
#include "../../include/mmhip.h"

using namespace mmhip;

namespace {

struct LayerOff {   // element offsets into a flat fp32 buffer
    size_t qkv_w, qkv_b, ao_w, ao_b, ln1_w, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_w, ln2_b, begin, end;
};
struct LayerW16 {   // 16-bit GEMM operand copies (byte offsets into the workspace)
    size_t qkv, ao, fc1, fc2, qkvT, aoT, fc1T, fc2T;
};
struct TextAct {    // saved activations of one text layer (byte offsets into the workspace)
    size_t qkv, ctx, pre1, a1, u, h, pre2, out, mean1, rstd1, mean2, rstd2, lse;
    size_t a1p, outp;      // parity mode with plane pairs: the LayerNorm outputs once more as pairs (the fp32 forms stay the residual inputs)
};
struct Stream16 { size_t off; };

#define CHECK_HIP(expr)                       \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

}  // namespace

struct mmhip_engine {
    mmhip_config cfg;
    std::vector<mmhip_param_info> params;
    size_t n_frozen = 0, n_train = 0;
    // offsets (elements) ---------------------------------------------------------------
    std::vector<LayerOff> vit, txt;
    size_t v_cls, v_pos, v_patch_w, v_patch_b, v_ln_w, v_ln_b, v_pool_w, v_pool_b;          // frozen
    size_t t_word, t_pos, t_type, t_eln_w, t_eln_b, t_pool_w, t_pool_b;                     // train
    size_t logit_scale, vproj_w, tproj_w;
    size_t fq_w, fq_b, fk_w, fk_b, fv_w, fv_b, fus_w, fus_b, cls_w, cls_b, tim_w, tim_b;
    size_t heads_begin, heads_end, emb_begin, emb_end;
    // bound buffers --------------------------------------------------------------------
    float* frozen = nullptr; float* train = nullptr; float* grad = nullptr;
    char* ws = nullptr; size_t ws_bytes = 0, ws_need = 0;
    // workspace offsets (bytes) --------------------------------------------------------
    std::vector<LayerW16> vit_w16, txt_w16;
    size_t x3_ws[3] = {0, 0, 0}, x3_bytes[3] = {0, 0, 0};      // parity mode: split-plane scratch of the caller's stream | side stream | image-tower stream
    size_t patch_w16;
    std::vector<TextAct> tact;
    size_t ids_all, mask_all, pos_ids, maskbias, x0, xhat_emb, rstd_emb, x0p = 0;
    // Parity mode (bf16x3), round 4: every tensor that feeds a matrix product is written by its producer as a PLANE PAIR (mmhip_kernels.h) and the
    // GEMMs read the planes directly (MMHIP_X3_PAIRS=0 at mmhip_create: round 3's form -- fp32 tensors, split into scratch copies per call).
    // Pairs: weights' operand copies, qkv, ctx, FC1 output h, the LayerNorm outputs (beside their fp32 forms), the image tower's LN output /
    // qkv / ctx / h / patches, and in the backward du, dqkv, d ctx and the LayerNorm-backward outputs the GEMMs read.  fp32: the residual
    // stream (pre1, pre2, x, d pre, dx), the GELU stash u, everything the heads touch.
    bool px = false;
    int bwd_np = 3;          // parity mode, plane pairs: products per k slice in the BACKWARD's matrix products (3 = as the forward; 2; 1) -- mmhip_set_backward_products
    size_t v_patches, v_pe, v_x, v_ln, v_qkv, v_ctx, v_h, v_out;                             // ViT ping-pong
    size_t g_partial, g_partial_side, g_det_rows = 0;
    size_t g_lnp[2][2];      // LN-backward partials per (layer parity, LN index): reduced on the side stream with the layer's dW
    uint8_t* word_row_state = nullptr;     // caller-owned row flags of the word table (mmhip_set_row_state)
    unsigned* bad_index = nullptr;         // caller-owned device word counting token ids that had to be clamped into the word table (mmhip_set_index_counter)
    unsigned* guard = nullptr;             // caller-owned overflow-guard words of THIS handle, {counter, void-step flag} (mmhip_set_guard); null: the
                                           // process-wide registration of mmhip_set_step_guard, if any
    size_t g_set[2][6];      // double-buffered backward temporaries read by the side stream: dpre2, ddrop2, du, dpre1, ddrop1, dqkv
    size_t g_dx, g_dx2, g_dpre, g_ddrop, g_dpre1, g_ddrop1, g_dqkv, g_dctx, g_du;                               // backward temporaries
    // heads (fp32) ----------------------------------------------------------------------
    size_t h_vpool, h_tpool, h_txt_e, h_img_e, h_txt_n, h_img_n, h_txt_inv, h_img_inv, h_logits;
    size_t h_q, h_qk, h_prob, h_xbar, h_z, h_feats, h_featd, h_out_cls, h_out_tim;
    size_t splitk_ws = 0, splitk_ws_vit = 0; bool has_splitk = false;      // fp32 scratch of the split-K path of the <= 128-row GEMMs: one per stream that
                                                                           // issues NT GEMMs (the caller's; the image tower's side stream) -- two towers' small
                                                                           // GEMMs may run side by side
    size_t h_d_out_cls, h_d_logits, h_d_out_tim, h_dfeats, h_dpre, h_dz, h_dxcls, h_dxbar, h_dqk, h_dq, h_dtxt_e, h_dimg_e,
        h_dtpool, h_dprepool, h_loss;
    // state of the last forward --------------------------------------------------------
    int B = 0, T = 0, Bt = 0; bool itm = false, train_mode = false, fwd_done = false, bwd_begun = false;
    bool skip_itc = false, itc_done = false;   // mmhip_train_step without the ITC loss: the dual-encoder similarity head (text pooler, both projections,
                                               // normalisation, logits: five launches on the critical path between towers and backward) is not run
    uint64_t seed = 0;
    const float *bd_out_cls = nullptr, *bd_logits = nullptr, *bd_out_tim = nullptr, *bd_feats = nullptr;
    // internal side stream (ViT forward beside the text forward; weight gradients beside the dX chain) -------------
    hipStream_t side = nullptr;          // weight-gradient work of the backward
    int vision_ready_B = 0;              // > 0: the image tower's outputs for that many posts were imported (mmhip_vision_import)
    bool vit_is_long = false;            // the image tower is the longer forward chain of this call (set by mmhip_forward)
    hipStream_t side_vit[2] = {nullptr, nullptr};     // image tower of the forward: [0] normal, [1] high priority
    hipEvent_t ev_fork = nullptr, ev_vit = nullptr, ev_ready[2] = {nullptr, nullptr}, ev_tn[2] = {nullptr, nullptr};
    hipEvent_t ev_layer[2] = {nullptr, nullptr}, ev_opt = nullptr;      // mmhip_train_step: per-layer AdamW on the side stream
    bool tn_pending[2] = {false, false};
    int cls_only = -1;         // -1 = read MMHIP_CLS_ONLY on first use; 1: the last text layer runs its post-attention part on CLS rows only
    bool cls_compact = false;  // state of the last forward
    int overlap = -1;          // -1 = read MMHIP_OVERLAP on first use
    // CU partition of the forward (MMHIP_PART="txt,vit", e.g. "96,160"; read once): while both towers run side by side their big GEMMs are
    // persistent 256 x 256-tile launches of at most part[0] (text) / part[1] (image) workgroups.  A workgroup of that kernel holds a CU
    // (128 KB of LDS), so the two launches split the chip 96 / 160 without CU masks, each tower's GEMMs see a fixed machine (8192 rows x
    // 2304 / 768 / 3072 columns = 288 / 96 / 384 tiles = 3 / 1 / 4 rounds of 96; 12608 rows = 450 / 150 / 600 tiles = 2.8 / 0.94 / 3.75 rounds
    // of 160) instead of racing for CUs launch by launch.  cur_part: the caps of the forward being enqueued (0 = off).
    int part[2] = {-1, -1}, cur_part[2] = {0, 0}; bool part_auto = false;
    int part_bwd = -1;         // MMHIP_PART_BWD=n: the backward's activation-gradient GEMMs as persistent 256 x 256-tile launches of at most n workgroups
                               // (the grouped weight-gradient GEMM of the layer above runs beside them on the side stream)
    // mmhip_step_spans: timing events at the phase ends of the last forward / train step (0 fork, 1 image tower end, 2 text tower end,
    // 3 forward end, 4 backward end, 5 step end); recorded only while enabled
    int spans_on = 0; hipEvent_t span_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool span_set[6] = {false, false, false, false, false, false};
    int span(int i, hipStream_t s) {
        if (!spans_on) return 0;
        if (!span_ev[i]) { hipError_t r = hipEventCreate(&span_ev[i]); if (r != hipSuccess) return (int)r; }
        hipError_t r = hipEventRecord(span_ev[i], s);
        span_set[i] = r == hipSuccess;
        return (int)r;
    }
    // GEMM timing ----------------------------------------------------------------------
    int timing = 0;            // 0 off, 1 = events around every NT GEMM with the side streams on, 2 = side streams off
    struct Ev { hipEvent_t a, b; double flops; int M, N, K, flags, tile, cus; };      // cus: workgroup cap of a partitioned launch (0 = whole chip)
    std::vector<Ev> evs; size_t ev_used = 0;

    template <typename U> U* wsp(size_t off) const { return reinterpret_cast<U*>(ws + off); }
    int dt() const { return cfg.dtype; }
    // image tower geometry: its own width for CLIP-ViT-L/14 (BASELINE config 4), the text tower's otherwise
    bool clip() const { return cfg.img_kind == MMHIP_IMG_CLIP; }
    int Hv() const { return cfg.hidden_img > 0 ? cfg.hidden_img : cfg.hidden; }
    int Iv() const { return cfg.inter_img > 0 ? cfg.inter_img : cfg.inter; }
    int heads_v() const { return cfg.heads_img > 0 ? cfg.heads_img : cfg.heads; }
    int P() const { return (cfg.image / cfg.patch) * (cfg.image / cfg.patch) + 1; }
    int Kp() const { return 3 * cfg.patch * cfg.patch; }                 // patch-embedding reduction length
    int Kpp() const { return (Kp() + 63) / 64 * 64; }                    // ... padded to the GEMM's k-step (588 -> 640 for 14 x 14)
    size_t v_pre_ln_w = 0, v_pre_ln_b = 0;                               // CLIP pre_layrnorm
    // f16 activations: gradients inside the text tower are carried multiplied by gscale() (range, not precision) and
    // every fp32 parameter gradient is written multiplied by 1 / gscale(); bf16 needs none
    float gscale() const { return cfg.loss_scale > 0.f ? cfg.loss_scale : (cfg.dtype == MMHIP_F16 ? 1024.f : 1.f); }
    size_t esz() const { return cfg.dtype == MMHIP_BF16X3 ? 4 : 2; }
};

namespace {

// ------------------------------------------------------------------------------------------------ layout
struct Builder {
    mmhip_engine& e;
    size_t off[2] = {0, 0};
    size_t add(const std::string& name, int buffer, int group, std::initializer_list<int64_t> dims) {
        mmhip_param_info p;
        memset(&p, 0, sizeof(p));
        strncpy(p.name, name.c_str(), sizeof(p.name) - 1);
        p.ndim = (int)dims.size();
        size_t n = 1;
        int i = 0;
        for (auto d : dims) { p.dims[i++] = d; n *= (size_t)d; }
        p.buffer = buffer;
        p.group = group;
        p.offset = off[buffer];
        p.numel = n;
        e.params.push_back(p);
        off[buffer] += (n + 3) & ~(size_t)3;      // keep every tensor 16-byte aligned
        return (size_t)p.offset;
    }
};

void add_text_layer(Builder& b, const mmhip_config& c, int l, LayerOff& o) {
    const int H = c.hidden, I = c.inter;
    const std::string p = "dual_encoder.text_model.encoder.layer." + std::to_string(l) + ".";
    const int g = MMHIP_G_ALWAYS;
    o.begin = b.off[1];
    o.qkv_w = b.add(p + "attention.self.query.weight", 1, g, {H, H});
    b.add(p + "attention.self.key.weight", 1, g, {H, H});
    b.add(p + "attention.self.value.weight", 1, g, {H, H});
    o.qkv_b = b.add(p + "attention.self.query.bias", 1, g, {H});
    b.add(p + "attention.self.key.bias", 1, g, {H});
    b.add(p + "attention.self.value.bias", 1, g, {H});
    o.ao_w = b.add(p + "attention.output.dense.weight", 1, g, {H, H});
    o.ao_b = b.add(p + "attention.output.dense.bias", 1, g, {H});
    o.ln1_w = b.add(p + "attention.output.LayerNorm.weight", 1, g, {H});
    o.ln1_b = b.add(p + "attention.output.LayerNorm.bias", 1, g, {H});
    o.fc1_w = b.add(p + "intermediate.dense.weight", 1, g, {I, H});
    o.fc1_b = b.add(p + "intermediate.dense.bias", 1, g, {I});
    o.fc2_w = b.add(p + "output.dense.weight", 1, g, {H, I});
    o.fc2_b = b.add(p + "output.dense.bias", 1, g, {H});
    o.ln2_w = b.add(p + "output.LayerNorm.weight", 1, g, {H});
    o.ln2_b = b.add(p + "output.LayerNorm.bias", 1, g, {H});
    o.end = b.off[1];
}
// CLIP vision layer, transformers 4.25.1 keys (CLIPVisionModel wraps the transformer as `.vision_model`):
// HF:models/clip/modeling_clip.py (CLIPEncoderLayer: pre-LN, quick-GELU MLP)
void add_clip_layer(Builder& b, int H, int I, int l, LayerOff& o) {
    const std::string p = "dual_encoder.vision_model.vision_model.encoder.layers." + std::to_string(l) + ".";
    const int g = MMHIP_G_FROZEN;
    o.begin = b.off[0];
    o.qkv_w = b.add(p + "self_attn.q_proj.weight", 0, g, {H, H});
    b.add(p + "self_attn.k_proj.weight", 0, g, {H, H});
    b.add(p + "self_attn.v_proj.weight", 0, g, {H, H});
    o.qkv_b = b.add(p + "self_attn.q_proj.bias", 0, g, {H});
    b.add(p + "self_attn.k_proj.bias", 0, g, {H});
    b.add(p + "self_attn.v_proj.bias", 0, g, {H});
    o.ao_w = b.add(p + "self_attn.out_proj.weight", 0, g, {H, H});
    o.ao_b = b.add(p + "self_attn.out_proj.bias", 0, g, {H});
    o.ln1_w = b.add(p + "layer_norm1.weight", 0, g, {H});
    o.ln1_b = b.add(p + "layer_norm1.bias", 0, g, {H});
    o.fc1_w = b.add(p + "mlp.fc1.weight", 0, g, {I, H});
    o.fc1_b = b.add(p + "mlp.fc1.bias", 0, g, {I});
    o.fc2_w = b.add(p + "mlp.fc2.weight", 0, g, {H, I});
    o.fc2_b = b.add(p + "mlp.fc2.bias", 0, g, {H});
    o.ln2_w = b.add(p + "layer_norm2.weight", 0, g, {H});
    o.ln2_b = b.add(p + "layer_norm2.bias", 0, g, {H});
    o.end = b.off[0];
}
void add_vit_layer(Builder& b, const mmhip_config& c, int l, LayerOff& o) {
    const int H = c.hidden_img > 0 ? c.hidden_img : c.hidden, I = c.inter_img > 0 ? c.inter_img : c.inter;
    const std::string p = "dual_encoder.vision_model.encoder.layer." + std::to_string(l) + ".";
    const int g = MMHIP_G_FROZEN;
    o.begin = b.off[0];
    o.qkv_w = b.add(p + "attention.attention.query.weight", 0, g, {H, H});
    b.add(p + "attention.attention.key.weight", 0, g, {H, H});
    b.add(p + "attention.attention.value.weight", 0, g, {H, H});
    o.qkv_b = b.add(p + "attention.attention.query.bias", 0, g, {H});
    b.add(p + "attention.attention.key.bias", 0, g, {H});
    b.add(p + "attention.attention.value.bias", 0, g, {H});
    o.ao_w = b.add(p + "attention.output.dense.weight", 0, g, {H, H});
    o.ao_b = b.add(p + "attention.output.dense.bias", 0, g, {H});
    o.fc1_w = b.add(p + "intermediate.dense.weight", 0, g, {I, H});
    o.fc1_b = b.add(p + "intermediate.dense.bias", 0, g, {I});
    o.fc2_w = b.add(p + "output.dense.weight", 0, g, {H, I});
    o.fc2_b = b.add(p + "output.dense.bias", 0, g, {H});
    o.ln1_w = b.add(p + "layernorm_before.weight", 0, g, {H});
    o.ln1_b = b.add(p + "layernorm_before.bias", 0, g, {H});
    o.ln2_w = b.add(p + "layernorm_after.weight", 0, g, {H});
    o.ln2_b = b.add(p + "layernorm_after.bias", 0, g, {H});
    o.end = b.off[0];
}

void build_layout(mmhip_engine& e) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, C = c.num_labels, E = c.proj_dim, P = e.P(), Hv = e.Hv();
    Builder b{e};
    // ---- frozen: vision tower (every dual_encoder parameter with 'vision' in its name, mm_late.py:67-69)
    e.vit.resize(c.layers_img);
    if (e.clip()) {
        // HF CLIPVisionTransformer (models/clip/modeling_clip.py): class_embedding [Hv], bias-free patch conv, learned positions,
        // pre_layrnorm, pre-LN layers, post_layernorm on the pooled CLS row only
        const std::string vm = "dual_encoder.vision_model.vision_model.";
        e.v_cls = b.add(vm + "embeddings.class_embedding", 0, MMHIP_G_FROZEN, {Hv});
        e.v_patch_w = b.add(vm + "embeddings.patch_embedding.weight", 0, MMHIP_G_FROZEN, {Hv, 3, c.patch, c.patch});
        e.v_patch_b = 0;
        e.v_pos = b.add(vm + "embeddings.position_embedding.weight", 0, MMHIP_G_FROZEN, {P, Hv});
        e.v_pre_ln_w = b.add(vm + "pre_layrnorm.weight", 0, MMHIP_G_FROZEN, {Hv});
        e.v_pre_ln_b = b.add(vm + "pre_layrnorm.bias", 0, MMHIP_G_FROZEN, {Hv});
        for (int l = 0; l < c.layers_img; ++l) add_clip_layer(b, Hv, e.Iv(), l, e.vit[l]);
        e.v_ln_w = b.add(vm + "post_layernorm.weight", 0, MMHIP_G_FROZEN, {Hv});
        e.v_ln_b = b.add(vm + "post_layernorm.bias", 0, MMHIP_G_FROZEN, {Hv});
        e.v_pool_w = e.v_pool_b = 0;
    } else {
        const std::string vm = "dual_encoder.vision_model.";
        e.v_cls = b.add(vm + "embeddings.cls_token", 0, MMHIP_G_FROZEN, {1, 1, Hv});
        e.v_pos = b.add(vm + "embeddings.position_embeddings", 0, MMHIP_G_FROZEN, {1, P, Hv});
        e.v_patch_w = b.add(vm + "embeddings.patch_embeddings.projection.weight", 0, MMHIP_G_FROZEN, {Hv, 3, c.patch, c.patch});
        e.v_patch_b = b.add(vm + "embeddings.patch_embeddings.projection.bias", 0, MMHIP_G_FROZEN, {Hv});
        for (int l = 0; l < c.layers_img; ++l) add_vit_layer(b, c, l, e.vit[l]);
        e.v_ln_w = b.add(vm + "layernorm.weight", 0, MMHIP_G_FROZEN, {Hv});
        e.v_ln_b = b.add(vm + "layernorm.bias", 0, MMHIP_G_FROZEN, {Hv});
        e.v_pool_w = b.add(vm + "pooler.dense.weight", 0, MMHIP_G_FROZEN, {Hv, Hv});
        e.v_pool_b = b.add(vm + "pooler.dense.bias", 0, MMHIP_G_FROZEN, {Hv});
    }
    // ---- trainable, ordered [never | ITC | ITM | fusion-attention | always: heads, layers last->first, embeddings]
    for (const char* n : {"aspectattention", "linear_iadds", "linear_gmu_t", "linear_gmu_v"}) {
        const int out = !strcmp(n, "aspectattention") ? 1 : (!strcmp(n, "linear_iadds") ? 2 : 2 * H);
        b.add(std::string(n) + ".weight", 1, MMHIP_G_NEVER, {out, H});
        b.add(std::string(n) + ".bias", 1, MMHIP_G_NEVER, {out});
    }
    e.heads_begin = b.off[1];
    e.logit_scale = b.add("dual_encoder.logit_scale", 1, MMHIP_G_ITC, {});
    e.vproj_w = b.add("dual_encoder.visual_projection.weight", 1, MMHIP_G_ITC, {E, Hv});
    e.tproj_w = b.add("dual_encoder.text_projection.weight", 1, MMHIP_G_ITC, {E, H});
    e.t_pool_w = b.add("dual_encoder.text_model.pooler.dense.weight", 1, MMHIP_G_ITC, {H, H});
    e.t_pool_b = b.add("dual_encoder.text_model.pooler.dense.bias", 1, MMHIP_G_ITC, {H});
    e.tim_w = b.add("linear_tim.weight", 1, MMHIP_G_ITM, {2, H});
    e.tim_b = b.add("linear_tim.bias", 1, MMHIP_G_ITM, {2});
    e.fq_w = b.add("fc_Q.weight", 1, MMHIP_G_FUSION_ATT, {H, H});
    e.fq_b = b.add("fc_Q.bias", 1, MMHIP_G_FUSION_ATT, {H});
    e.fk_w = b.add("fc_K.weight", 1, MMHIP_G_FUSION_ATT, {H, H});
    e.fk_b = b.add("fc_K.bias", 1, MMHIP_G_FUSION_ATT, {H});
    e.fv_w = b.add("fc_V.weight", 1, MMHIP_G_FUSION_ATT, {H, H});
    e.fv_b = b.add("fc_V.bias", 1, MMHIP_G_FUSION_ATT, {H});
    // 'concat' takes [x_t CLS | x_v CLS]: H + Hv wide (the reference hard-wires 768 + 768, models/config.py:82-84, mm_late.py:81)
    e.fus_w = b.add("linear_fusion.weight", 1, MMHIP_G_ALWAYS, {H, H + Hv});
    e.fus_b = b.add("linear_fusion.bias", 1, MMHIP_G_ALWAYS, {H});
    e.cls_w = b.add("linear_cls.weight", 1, MMHIP_G_ALWAYS, {C, H});
    e.cls_b = b.add("linear_cls.bias", 1, MMHIP_G_ALWAYS, {C});
    e.heads_end = b.off[1];
    e.txt.resize(c.layers_txt);
    for (int l = c.layers_txt - 1; l >= 0; --l) add_text_layer(b, c, l, e.txt[l]);
    const std::string tm = "dual_encoder.text_model.embeddings.";
    e.emb_begin = b.off[1];
    e.t_eln_w = b.add(tm + "LayerNorm.weight", 1, MMHIP_G_ALWAYS, {H});
    e.t_eln_b = b.add(tm + "LayerNorm.bias", 1, MMHIP_G_ALWAYS, {H});
    e.t_type = b.add(tm + "token_type_embeddings.weight", 1, MMHIP_G_ALWAYS, {c.type_vocab, H});
    e.t_pos = b.add(tm + "position_embeddings.weight", 1, MMHIP_G_ALWAYS, {c.max_pos, H});
    e.t_word = b.add(tm + "word_embeddings.weight", 1, MMHIP_G_ALWAYS, {c.vocab, H});
    e.emb_end = b.off[1];
    e.n_frozen = b.off[0];
    e.n_train = b.off[1];
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; }
};

void build_workspace(mmhip_engine& e) {
    const mmhip_config& c = e.cfg;
    const size_t H = c.hidden, I = c.inter, E = c.proj_dim, C = c.num_labels;
    const size_t Bm = c.max_posts, Tm = c.max_text_len, P = e.P(), Hv = e.Hv(), Iv = e.Iv(), Kpp = e.Kpp();
    const size_t Mt = 2 * Bm * Tm, Mv = Bm * P, Bt = 2 * Bm;
    const size_t Z = e.esz();      // bytes per activation / GEMM-operand element: 2 (bf16, f16) or 4 (bf16x3 parity mode)
    Carver w;
    auto w16 = [&](std::vector<LayerW16>& v, int n, bool transposed, size_t H, size_t I) {
        v.resize(n);
        for (auto& L : v) {
            L.qkv = w.take(3 * H * H * Z); L.ao = w.take(H * H * Z); L.fc1 = w.take(I * H * Z); L.fc2 = w.take(H * I * Z);
            if (transposed) { L.qkvT = w.take(3 * H * H * Z); L.aoT = w.take(H * H * Z); L.fc1T = w.take(I * H * Z); L.fc2T = w.take(H * I * Z); }
        }
    };
    w16(e.vit_w16, c.layers_img, false, Hv, Iv);
    w16(e.txt_w16, c.layers_txt, true, H, I);
    e.patch_w16 = w.take(Hv * Kpp * Z);
    e.ids_all = w.take(Bt * Tm * 8); e.mask_all = w.take(Bt * Tm * 8); e.pos_ids = w.take(Bt * Tm * 4); e.maskbias = w.take(Bt * Tm * 4);
    e.x0 = w.take(Mt * H * Z); e.xhat_emb = w.take(Mt * H * Z); e.rstd_emb = w.take(Mt * 4);
    if (e.px) e.x0p = w.take(Mt * H * Z);
    e.tact.resize(c.layers_txt);
    for (auto& a : e.tact) {
        a.a1p = a.outp = 0;
        if (e.px) { a.a1p = w.take(Mt * H * Z); a.outp = w.take(Mt * H * Z); }
        a.qkv = w.take(Mt * 3 * H * Z); a.ctx = w.take(Mt * H * Z); a.pre1 = w.take(Mt * H * Z); a.a1 = w.take(Mt * H * Z);
        a.u = w.take(Mt * I * Z); a.h = w.take(Mt * I * Z); a.pre2 = w.take(Mt * H * Z); a.out = w.take(Mt * H * Z);
        a.mean1 = w.take(Mt * 4); a.rstd1 = w.take(Mt * 4); a.mean2 = w.take(Mt * 4); a.rstd2 = w.take(Mt * 4);
        a.lse = w.take(Bt * c.heads * Tm * 4);
    }
    e.v_patches = w.take(Bm * (P - 1) * Kpp * Z); e.v_pe = w.take(Bm * (P - 1) * Hv * Z);
    e.v_x = w.take(Mv * Hv * Z); e.v_ln = w.take(Mv * Hv * Z); e.v_qkv = w.take(Mv * 3 * Hv * Z); e.v_ctx = w.take(Mv * Hv * Z);
    e.v_h = w.take(Mv * Iv * Z); e.v_out = w.take(Mv * Hv * Z);
    e.g_dx = w.take(Mt * H * Z); e.g_dx2 = w.take(Mt * H * Z); e.g_dpre = w.take(Mt * H * Z); e.g_ddrop = w.take(Mt * H * Z); e.g_dpre1 = w.take(Mt * H * Z); e.g_ddrop1 = w.take(Mt * H * Z);
    e.g_dqkv = w.take(Mt * 3 * H * Z); e.g_dctx = w.take(Mt * H * Z); e.g_du = w.take(Mt * I * Z);
    e.g_det_rows = w.take(Mt * H * 4);       // per-slot embedding gradient rows of the deterministic mode (MMHIP_DETERMINISTIC=1)
    e.g_set[0][0] = e.g_dpre; e.g_set[0][1] = e.g_ddrop; e.g_set[0][2] = e.g_du; e.g_set[0][3] = e.g_dpre1; e.g_set[0][4] = e.g_ddrop1; e.g_set[0][5] = e.g_dqkv;
    e.g_set[1][0] = w.take(Mt * H * Z); e.g_set[1][1] = w.take(Mt * H * Z); e.g_set[1][2] = w.take(Mt * I * Z);
    e.g_set[1][3] = w.take(Mt * H * Z); e.g_set[1][4] = w.take(Mt * H * Z); e.g_set[1][5] = w.take(Mt * 3 * H * Z);
    {
        size_t pf = partial_floats_rows((int)Mt, (int)H, 3), pc = partial_floats_colsum((int)Mt, (int)(3 * H > I ? 3 * H : I));
        const size_t pe = partial_floats_embed((int)Bt, (int)Tm, (int)H);
        if (pe > pf) pf = pe;
        e.g_partial = w.take((pf > pc ? pf : pc) * 4);
        e.g_partial_side = w.take((pf > pc ? pf : pc) * 4);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) e.g_lnp[i][j] = w.take(partial_floats_rows((int)Mt, (int)H, 2) * 4);
    }
    auto f = [&](size_t n) { return w.take(n * 4); };
    e.h_vpool = f(Bm * Hv); e.h_tpool = f(Bm * H); e.h_txt_e = f(Bm * E); e.h_img_e = f(Bm * E); e.h_txt_n = f(Bm * E); e.h_img_n = f(Bm * E);
    e.h_txt_inv = f(Bm); e.h_img_inv = f(Bm); e.h_logits = f(Bm * Bm);
    e.h_q = f(Bt * H); e.h_qk = f(Bt * H); e.h_prob = f(Bt * P); e.h_xbar = f(Bt * H); e.h_z = f(Bt * (H + Hv)); e.h_feats = f(Bt * H);
    e.h_featd = f(Bm * H); e.h_out_cls = f(Bm * C); e.h_out_tim = f(Bm * 2);
    e.splitk_ws = f((size_t)(I / 384 + 1) * 128 * (size_t)(I > 3 * H ? I : 3 * H)); e.has_splitk = true;
    e.splitk_ws_vit = f((size_t)(Iv / 384 + 1) * 128 * (size_t)(Iv > 3 * Hv ? Iv : 3 * Hv));
    if (c.dtype == MMHIP_BF16X3 && !e.px) {
        // split planes of the operands of one GEMM call at a time per stream (x3.hip): the widest NT problem of a tower, or the four
        // weight-gradient problems of a text layer together
        auto nt = [](size_t M, size_t H, size_t I) {
            size_t b = 0;
            const size_t nk[5][2] = {{3 * H, H}, {H, H}, {I, H}, {H, I}, {H, 3 * H}};
            for (auto& q : nk) { const size_t v = x3_nt_scratch_bytes((int)M, (int)q[0], (int)q[1]); if (v > b) b = v; }
            return b;
        };
        const size_t tn = x3_tn_scratch_bytes((int)Mt, (int)H, (int)I) + x3_tn_scratch_bytes((int)Mt, (int)I, (int)H) + x3_tn_scratch_bytes((int)Mt, (int)(3 * H), (int)H) +
                          x3_tn_scratch_bytes((int)Mt, (int)H, (int)H);
        size_t vit = nt(Mv, Hv, Iv);
        { const size_t pe = x3_nt_scratch_bytes((int)(Bm * (P - 1)), (int)Hv, (int)Kpp); if (pe > vit) vit = pe; }
        size_t txt = nt(Mt, H, I);
        if (tn > txt) txt = tn;
        e.x3_bytes[0] = txt > vit ? txt : vit; e.x3_bytes[1] = txt; e.x3_bytes[2] = vit;
        for (int i = 0; i < 3; ++i) e.x3_ws[i] = w.take(e.x3_bytes[i]);
    }
    e.h_d_out_cls = f(Bm * C); e.h_d_logits = f(Bm * Bm); e.h_d_out_tim = f(Bm * 2); e.h_dfeats = f(Bt * H); e.h_dpre = f(Bt * H);
    e.h_dz = f(Bt * (H + Hv)); e.h_dxcls = f(Bt * H); e.h_dxbar = f(Bt * H); e.h_dqk = f(Bt * H); e.h_dq = f(Bt * H);
    e.h_dtxt_e = f(Bm * E); e.h_dimg_e = f(Bm * E); e.h_dtpool = f(Bm * H); e.h_dprepool = f(Bm * H); e.h_loss = f(8);
    e.ws_need = w.off;
}

DropCfg make_drop(float p, uint64_t seed, uint32_t stream, bool on) {
    DropCfg d;
    d.seed = seed;
    d.stream = stream;
    uint32_t t = on ? (uint32_t)lrintf(p * 65536.0f) : 0u;
    if (t > 65535u) t = 65535u;
    d.thresh16 = t;
    d.keep_scale = 1.0f / (1.0f - (float)t / 65536.0f);
    return d;
}
enum { STREAM_EMBED = 1, STREAM_HEAD = 2 };
inline uint32_t stream_attn(int l) { return 16 + 4 * l; }
inline uint32_t stream_attn_out(int l) { return 16 + 4 * l + 1; }
inline uint32_t stream_ffn_out(int l) { return 16 + 4 * l + 2; }

// ------------------------------------------------------------------------------------------------ GEMM helper
struct G {
    GemmNTArgs a;
    G(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K) {
        memset(&a, 0, sizeof(a));
        a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    }
    G& bias(const float* b) { a.bias = b; a.flags |= GEMM_BIAS; return *this; }
    G& gelu() { a.flags |= GEMM_GELU; return *this; }
    G& qgelu() { a.flags |= GEMM_QGELU; return *this; }
    G& aux(void* p, int ld) { a.aux = p; a.ldaux = ld; a.flags |= GEMM_AUX_PRE; return *this; }
    G& residual(const void* p, int ld) { a.residual = p; a.ldres = ld; a.flags |= GEMM_RESIDUAL; return *this; }
    G& mul_gelu_grad(const void* p, int ld) { a.mul_in = p; a.ldmul = ld; a.flags |= GEMM_MUL_GELU_GRAD; return *this; }
    G& dropout(const DropCfg& d, int row_mul = 1) { a.drop = d; a.drop_row_mul = row_mul; if (d.thresh16) a.flags |= GEMM_DROPOUT; return *this; }
    // parity mode with plane pairs: both operands are pairs whose rows hold [hi(W) | lo(W)] -- the leading dimensions given in logical
    // elements double, the lo planes sit K elements behind; px_out: C as a pair, rows [hi(N) | lo(N)]
    G& px_in(bool px, int nprod = 0) { if (px) { a.a_pair = a.b_pair = 1; a.lda *= 2; a.ldb *= 2; a.a_lo = a.b_lo = a.K; a.nprod = nprod; } return *this; }
    G& px_out(bool px, bool hi_only = false) { if (px) { a.flags |= GEMM_OUT_PAIR | (hi_only ? GEMM_OUT_PAIR_HI : 0); a.ldc *= 2; a.c_lo = a.N; } return *this; }
};
// forward GEMM of tower `which` (0 text, 1 image) under the CU partition: persistent 256 x 256 tiles on at most cur_part[which] workgroups
inline void part_gemm(const mmhip_engine& e, G& g, int which) {
    const int n = e.cur_part[which];
    if (n > 0 && !g.a.tile && g.a.M >= 2048 && g.a.N % 256 == 0 && g.a.K % 64 == 0 && (e.dt() == DT_BF16 || e.dt() == DT_F16)) { g.a.tile = 15; g.a.grid = n; }
}
int run_gemm(mmhip_engine& e, G& g, hipStream_t s) {
    // GEMMs of <= 128 rows with K >= 1536 (the CLS-row GEMMs of the last text layer; a tiny image tower) are split along K
    // (gemm.hip launch_nt_splitk) through an fp32 scratch: the image tower's side stream has its own, every other stream shares the caller's
    {
        const bool on_vit = e.side_vit[0] != nullptr && (s == e.side_vit[0] || s == e.side_vit[1]);
        const int h = on_vit ? e.Hv() : e.cfg.hidden, in = on_vit ? e.Iv() : e.cfg.inter;
        if (e.has_splitk && g.a.M <= 128 && (size_t)(g.a.K / 384) * g.a.M * g.a.N <= (size_t)(in / 384 + 1) * 128 * (size_t)(in > 3 * h ? in : 3 * h))
            g.a.splitk_ws = e.wsp<float>(on_vit ? e.splitk_ws_vit : e.splitk_ws);
        if (e.x3_bytes[0]) {
            const int i = on_vit ? 2 : (e.side && s == e.side ? 1 : 0);
            g.a.x3_ws = e.ws + e.x3_ws[i];
            g.a.x3_ws_bytes = e.x3_bytes[i];
        }
    }
    if (e.timing) {
        if (e.ev_used == e.evs.size()) {
            mmhip_engine::Ev ev;
            CHECK_HIP(hipEventCreate(&ev.a));
            CHECK_HIP(hipEventCreate(&ev.b));
            e.evs.push_back(ev);
        }
        auto& ev = e.evs[e.ev_used++];
        ev.flops = 2.0 * g.a.M * (double)g.a.N * g.a.K;
        ev.M = g.a.M; ev.N = g.a.N; ev.K = g.a.K; ev.flags = g.a.flags; ev.tile = g.a.tile; ev.cus = g.a.grid;
        CHECK_HIP(hipEventRecord(ev.a, s));
        CHECK_HIP(launch_gemm_nt(g.a, e.dt(), s));
        CHECK_HIP(hipEventRecord(ev.b, s));
        return 0;
    }
    CHECK_HIP(launch_gemm_nt(g.a, e.dt(), s));
    return 0;
}
// two GEMMs of equal N and K (the two towers' same-named GEMM of one layer) in ONE persistent launch; falls back to two launches
int run_gemm_pair(mmhip_engine& e, G& g0, G& g1, hipStream_t s) {
    const bool pairable = (e.dt() == DT_BF16 || e.dt() == DT_F16 || (e.dt() == DT_F32 && g0.a.a_pair && g1.a.a_pair)) && g0.a.N == g1.a.N && g0.a.K == g1.a.K;
    mmhip_engine::Ev* ev = nullptr;
    if (e.timing && pairable) {
        if (e.ev_used == e.evs.size()) {
            mmhip_engine::Ev n;
            CHECK_HIP(hipEventCreate(&n.a));
            CHECK_HIP(hipEventCreate(&n.b));
            e.evs.push_back(n);
        }
        ev = &e.evs[e.ev_used];
        CHECK_HIP(hipEventRecord(ev->a, s));
    }
    if (pairable && launch_gemm_nt8_pair(g0.a, g1.a, e.dt(), 0, s)) {
        CHECK_HIP(hipGetLastError());
        if (ev) {
            ev->flops = 2.0 * ((double)g0.a.M + g1.a.M) * (double)g0.a.N * g0.a.K;
            ev->M = g0.a.M + g1.a.M; ev->N = g0.a.N; ev->K = g0.a.K; ev->flags = g0.a.flags | g1.a.flags; ev->tile = -2; ev->cus = 0;     // -2: a pair
            CHECK_HIP(hipEventRecord(ev->b, s));
            e.ev_used++;
        }
        return 0;
    }
    if (int r = run_gemm(e, g0, s)) return r;
    return run_gemm(e, g1, s);
}
SmallGemmArgs small(const void* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, int act = ACT_NONE, int acc = 0) {
    SmallGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.act = act; a.accumulate = acc;
    return a;
}

// ------------------------------------------------------------------------------------------------ side stream
int side_init(mmhip_engine& e) {
    if (e.overlap < 0) { const char* v = getenv("MMHIP_OVERLAP"); e.overlap = v ? atoi(v) : 1; }
    if (!e.overlap || e.side) return 0;
    // The side streams are borrowed from the process-wide pool (mmhip_common.h: pool_stream -- why they are not created per engine).
    // The forward's two towers race for the CUs: whichever chain is longer should not be the one that waits.  The image
    // tower gets the high-priority stream when it is the longer chain (plain batch: 34.9 vs 22.3 GF per post), the
    // normal one when the text pass is doubled by the ITM posts.  Measured same-box: -0.17 ms/step on config 2; the
    // same high priority on config 3 costs +0.38 ms.  MMHIP_VIT_PRIO=0/1 forces the choice.
    CHECK_HIP(pool_stream(POOL_SIDE, &e.side));
    CHECK_HIP(pool_stream(POOL_VIT, &e.side_vit[0]));
    CHECK_HIP(pool_stream(POOL_VIT_HI, &e.side_vit[1]));
    hipEvent_t* evs[9] = {&e.ev_fork, &e.ev_vit, &e.ev_ready[0], &e.ev_ready[1], &e.ev_tn[0], &e.ev_tn[1], &e.ev_layer[0], &e.ev_layer[1], &e.ev_opt};
    for (auto ev : evs) CHECK_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return 0;
}
inline bool use_side(const mmhip_engine& e) { return e.overlap > 0 && e.side && e.timing != 2; }

// ------------------------------------------------------------------------------------------------ weights
int refresh_layer(mmhip_engine& e, const float* base, const LayerOff& o, const LayerW16& w, bool transposed, hipStream_t s, int H, int I) {
    CastMat m[4] = {{base + o.qkv_w, e.ws + w.qkv, transposed ? e.ws + w.qkvT : nullptr, 3 * H, H, 0},
                    {base + o.ao_w, e.ws + w.ao, transposed ? e.ws + w.aoT : nullptr, H, H, 0},
                    {base + o.fc1_w, e.ws + w.fc1, transposed ? e.ws + w.fc1T : nullptr, I, H, 0},
                    {base + o.fc2_w, e.ws + w.fc2, transposed ? e.ws + w.fc2T : nullptr, H, I, 0}};
    CHECK_HIP(launch_cast_group(m, 4, e.px ? DT_PAIR : e.dt(), s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ forward pieces
int vit_forward(mmhip_engine& e, const float* pixels, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = e.Hv(), I = e.Iv(), B = e.B, P = e.P(), Kpp = e.Kpp(), dt = e.dt();
    const int Mv = B * P;
    const bool clip = e.clip();
    const float* F = e.frozen;
    const bool px = e.px;
    auto ln_to = [&](LNArgs& ln, size_t pair_buf, int W) {          // parity mode: a LayerNorm output that only GEMMs read goes out as a plane pair only
        if (px) { ln.y = nullptr; ln.y_pair = e.ws + pair_buf; ln.ld_pair = 2 * W; ln.lo_pair = W; }
    };
    CHECK_HIP(launch_patchify(pixels, e.ws + e.v_patches, B, c.image, c.patch, Kpp, px ? DT_PAIR : dt, s));
    {
        G g(e.ws + e.v_patches, Kpp, e.ws + e.patch_w16, Kpp, e.ws + e.v_pe, H, B * (P - 1), H, Kpp);
        if (!clip) g.bias(F + e.v_patch_b);           // CLIP's patch conv has no bias
        g.px_in(px);
        part_gemm(e, g, 1);
        if (int r = run_gemm(e, g, s)) return r;
    }
    CHECK_HIP(launch_vit_assemble(e.ws + e.v_pe, F + e.v_cls, F + e.v_pos, e.ws + e.v_x, B, P, H, dt, s));
    char* x = e.ws + e.v_x;
    if (clip) {      // pre_layrnorm, in place (HF:models/clip/modeling_clip.py CLIPVisionTransformer.forward)
        LNArgs ln{x, x, F + e.v_pre_ln_w, F + e.v_pre_ln_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
        CHECK_HIP(launch_layernorm_fwd(ln, dt, s));
    }
    for (int l = 0; l < c.layers_img; ++l) {
        const LayerOff& o = e.vit[l];
        const LayerW16& w = e.vit_w16[l];
        LNArgs ln{x, e.ws + e.v_ln, F + o.ln1_w, F + o.ln1_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
        ln_to(ln, e.v_ln, H);
        CHECK_HIP(launch_layernorm_fwd(ln, dt, s));
        { G g(e.ws + e.v_ln, H, e.ws + w.qkv, H, e.ws + e.v_qkv, 3 * H, Mv, 3 * H, H); g.bias(F + o.qkv_b).px_in(px).px_out(px); part_gemm(e, g, 1); if (int r = run_gemm(e, g, s)) return r; }
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.qkv = e.ws + e.v_qkv; at.ctx = e.ws + e.v_ctx; at.posts = B; at.S = P; at.heads = e.heads_v(); at.ld_qkv = 3 * H; at.ld_ctx = H; at.hidden = H;
        if (px) { at.pair = 1; at.ld_qkv = 6 * H; at.lo_qkv = 3 * H; at.ld_ctx = 2 * H; at.lo_ctx = H; }
        at.scale = 1.0f / sqrtf((float)(H / e.heads_v()));
        CHECK_HIP(launch_attn_fwd(at, dt, s));
        { G g(e.ws + e.v_ctx, H, e.ws + w.ao, H, x, H, Mv, H, H); g.bias(F + o.ao_b).residual(x, H).px_in(px); part_gemm(e, g, 1); if (int r = run_gemm(e, g, s)) return r; }
        LNArgs ln2{x, e.ws + e.v_ln, F + o.ln2_w, F + o.ln2_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
        ln_to(ln2, e.v_ln, H);
        CHECK_HIP(launch_layernorm_fwd(ln2, dt, s));
        { G g(e.ws + e.v_ln, H, e.ws + w.fc1, H, e.ws + e.v_h, I, Mv, I, H); g.bias(F + o.fc1_b); if (clip) g.qgelu(); else g.gelu(); g.px_in(px).px_out(px); part_gemm(e, g, 1); if (int r = run_gemm(e, g, s)) return r; }
        { G g(e.ws + e.v_h, I, e.ws + w.fc2, I, x, H, Mv, H, I); g.bias(F + o.fc2_b).residual(x, H).px_in(px); part_gemm(e, g, 1); if (int r = run_gemm(e, g, s)) return r; }
    }
    if (clip) {
        // last_hidden_state = the encoder output as it is; pooler_output = post_layernorm(CLS row)   (CLIPVisionTransformer.forward)
        CHECK_HIP(hipMemcpyAsync(e.ws + e.v_out, x, (size_t)Mv * H * e.esz(), hipMemcpyDeviceToDevice, s));
        CHECK_HIP(launch_gather_rows_f32(x, (size_t)P * H, e.wsp<float>(e.h_vpool), H, B, H, dt, s));
        LNArgs lnp{e.wsp<float>(e.h_vpool), e.wsp<float>(e.h_vpool), F + e.v_ln_w, F + e.v_ln_b, nullptr, nullptr, B, H, H, H, c.ln_eps_img};
        CHECK_HIP(launch_layernorm_fwd(lnp, DT_F32, s));
        return 0;
    }
    LNArgs lnf{x, e.ws + e.v_out, F + e.v_ln_w, F + e.v_ln_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
    CHECK_HIP(launch_layernorm_fwd(lnf, dt, s));
    // pooler on the CLS rows (row stride P*H): tanh(dense(x[:,0]))
    SmallGemmArgs sp = small(e.ws + e.v_out, P * H, F + e.v_pool_w, H, F + e.v_pool_b, e.wsp<float>(e.h_vpool), H, B, H, H, ACT_TANH);
    CHECK_HIP(launch_small_nt(sp, dt, s));
    return 0;
}

int text_forward(mmhip_engine& e, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, I = c.inter, T = e.T, Bt = e.Bt, Mt = Bt * T, dt = e.dt();
    const float* W = e.train;
    const bool tr = e.train_mode;
    EmbedArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.ids = e.wsp<int64_t>(e.ids_all); ea.mask = e.wsp<int64_t>(e.mask_all);
    ea.word = W + e.t_word; ea.pos = W + e.t_pos; ea.type = W + e.t_type; ea.gamma = W + e.t_eln_w; ea.beta = W + e.t_eln_b;
    ea.x = e.ws + e.x0; ea.xhat = e.ws + e.xhat_emb; ea.rstd = e.wsp<float>(e.rstd_emb); ea.pos_ids = e.wsp<int>(e.pos_ids);
    ea.maskbias = e.wsp<float>(e.maskbias);
    ea.posts = Bt; ea.T = T; ea.H = H; ea.xlmr = c.txt_kind == MMHIP_TXT_XLMR; ea.pad_id = c.pad_id; ea.eps = c.ln_eps_txt;
    ea.drop = make_drop(c.p_hidden, e.seed, STREAM_EMBED, tr);
    const bool px = e.px;
    if (px) { ea.x_pair = e.ws + e.x0p; ea.ld_pair = 2 * H; ea.lo_pair = H; }
    CHECK_HIP(launch_embed_fwd(ea, dt, s));
    const char* x = e.ws + e.x0;
    const char* xg = px ? e.ws + e.x0p : x;          // the layer input as the GEMMs read it (parity mode: its plane pair)
    if (e.cls_only < 0) { const char* v = getenv("MMHIP_CLS_ONLY"); e.cls_only = v ? atoi(v) : 1; }
    e.cls_compact = false;
    for (int l = 0; l < c.layers_txt; ++l) {
        const LayerOff& o = e.txt[l];
        const LayerW16& w = e.txt_w16[l];
        const TextAct& a = e.tact[l];
        { G g(xg, H, e.ws + w.qkv, H, e.ws + a.qkv, 3 * H, Mt, 3 * H, H); g.bias(W + o.qkv_b).px_in(px).px_out(px); part_gemm(e, g, 0); if (int r = run_gemm(e, g, s)) return r; }
        // Only the CLS row of the last layer's output is ever consumed (fusion query and pooler, mm_late.py:111,155-158):
        // its attention needs query tile 0 only and everything after it runs on Bt rows (row stride T*H in the full
        // tensors, compact [Bt, .] outputs).  Dropout indices keep the full-tensor numbering (row_mul = T).
        const bool compact = e.cls_only > 0 && l == c.layers_txt - 1;
        const int Mr = compact ? Bt : Mt, rs = compact ? T * H : H, rmul = compact ? T : 1;
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.qkv = e.ws + a.qkv; at.maskbias = e.wsp<float>(e.maskbias); at.ctx = e.ws + a.ctx; at.lse = e.wsp<float>(a.lse);
        at.posts = Bt; at.S = T; at.heads = c.heads; at.ld_qkv = 3 * H; at.ld_ctx = H; at.hidden = H;
        if (px) { at.pair = 1; at.ld_qkv = 6 * H; at.lo_qkv = 3 * H; at.ld_ctx = 2 * H; at.lo_ctx = H; }
        at.scale = 1.0f / sqrtf((float)(H / c.heads));
        at.drop = make_drop(c.p_attn, e.seed, stream_attn(l), tr);
        at.q_tiles = compact ? 1 : 0;
        CHECK_HIP(launch_attn_fwd(at, dt, s));
        { G g(e.ws + a.ctx, rs, e.ws + w.ao, H, e.ws + a.pre1, H, Mr, H, H);
          g.bias(W + o.ao_b).dropout(make_drop(c.p_hidden, e.seed, stream_attn_out(l), tr), rmul).residual(x, rs).px_in(px);
          part_gemm(e, g, 0);
          if (int r = run_gemm(e, g, s)) return r; }
        LNArgs ln1{e.ws + a.pre1, e.ws + a.a1, W + o.ln1_w, W + o.ln1_b, e.wsp<float>(a.mean1), e.wsp<float>(a.rstd1), Mr, H, H, H, c.ln_eps_txt};
        if (px) { ln1.y_pair = e.ws + a.a1p; ln1.ld_pair = 2 * H; ln1.lo_pair = H; }
        CHECK_HIP(launch_layernorm_fwd(ln1, dt, s));
        { G g(e.ws + (px ? a.a1p : a.a1), H, e.ws + w.fc1, H, e.ws + a.h, I, Mr, I, H); g.bias(W + o.fc1_b).aux(e.ws + a.u, I).gelu().px_in(px).px_out(px); part_gemm(e, g, 0); if (int r = run_gemm(e, g, s)) return r; }
        { G g(e.ws + a.h, I, e.ws + w.fc2, I, e.ws + a.pre2, H, Mr, H, I);
          g.bias(W + o.fc2_b).dropout(make_drop(c.p_hidden, e.seed, stream_ffn_out(l), tr), rmul).residual(e.ws + a.a1, H).px_in(px);
          part_gemm(e, g, 0);
          if (int r = run_gemm(e, g, s)) return r; }
        LNArgs ln2{e.ws + a.pre2, e.ws + a.out, W + o.ln2_w, W + o.ln2_b, e.wsp<float>(a.mean2), e.wsp<float>(a.rstd2), Mr, H, H, H, c.ln_eps_txt};
        if (px) { ln2.y_pair = e.ws + a.outp; ln2.ld_pair = 2 * H; ln2.lo_pair = H; }
        CHECK_HIP(launch_layernorm_fwd(ln2, dt, s));
        e.cls_compact = compact;
        x = e.ws + a.out;
        xg = px ? e.ws + a.outp : x;
    }
    return 0;
}

// Both towers layer by layer on ONE stream, each pair of same-named GEMMs (QKV, attention output, FC1, FC2: equal N and K in
// ViT-B/16 and the BERT-base-sized text tower) as one persistent launch (gemm8.hip, GemmNTPair): alone, the 256 x 256 tiles of
// M = 8192 x N = 2304 fill 288 of 512 workgroup slots in two rounds and those of M = 12608 fill 450 of 512; together 738 of 768.
// Same arithmetic as vit_forward + text_forward (same kernels on the same operands), another launch order.
// Measured (same box, BASELINE config 2): the paired GEMMs run at 0.276 of the MFMA peak against 0.254 for the separate ones
// (every kernel alone on the chip), and the step takes 11.2 ms against 12.2 ms with the side streams off -- but 10.4 ms with the
// image tower on its own stream, where kernels of the two towers share CUs.  Hence opt-in: MMHIP_LOCKSTEP=1.
bool lockstep_ok(const mmhip_engine& e) {
    // MMHIP_LOCKSTEP=1: opt-in, every dtype.  Round 5 tried it as the default of the parity mode with plane pairs, where every GEMM is three times
    // as long and the towers' own launches fill 78 % / 75 % of their rounds: the paired launches cut the forward's GEMM time by 14 % (12.4 -> 10.7 ms
    // measured alone) -- and the forward got 0.5 ms LONGER (13.5 vs 13.0 ms, same box, profiles/r05_x3_lockstep.txt), also with the towers' row
    // kernels forked onto two streams between the pairs: on two streams one tower's attention / LayerNorm kernels hide under the other's
    // GEMMs for free, in lockstep they are serial work between launches that wait for each other.  Off by default in every mode.
    static int on = -2;
    if (on == -2) { const char* v = getenv("MMHIP_LOCKSTEP"); on = v ? atoi(v) : 0; }
    return on > 0 && (e.dt() == DT_BF16 || e.dt() == DT_F16 || (e.dt() == DT_F32 && e.px)) && !e.clip() && e.Hv() == e.cfg.hidden && e.Iv() == e.cfg.inter &&
           e.cfg.layers_txt > 0 && e.cfg.layers_img > 0;
}
int towers_forward_lockstep(mmhip_engine& e, const float* pixels, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, I = c.inter, T = e.T, B = e.B, Bt = e.Bt, Mt = Bt * T, P = e.P(), Kpp = e.Kpp(), dt = e.dt();
    const int Mv = B * P;
    const float* W = e.train;
    const float* F = e.frozen;
    const bool tr = e.train_mode, px = e.px;
    auto ln_to = [&](LNArgs& ln, size_t pair_buf, int Wd) {          // parity mode: a LayerNorm output that only GEMMs read goes out as a plane pair only
        if (px) { ln.y = nullptr; ln.y_pair = e.ws + pair_buf; ln.ld_pair = 2 * Wd; ln.lo_pair = Wd; }
    };
    if (int r = e.span(0, s)) return r;
    // ---- embeddings of both towers
    CHECK_HIP(launch_patchify(pixels, e.ws + e.v_patches, B, c.image, c.patch, Kpp, px ? DT_PAIR : dt, s));
    { G g(e.ws + e.v_patches, Kpp, e.ws + e.patch_w16, Kpp, e.ws + e.v_pe, H, B * (P - 1), H, Kpp); g.bias(F + e.v_patch_b).px_in(px); if (int r = run_gemm(e, g, s)) return r; }
    CHECK_HIP(launch_vit_assemble(e.ws + e.v_pe, F + e.v_cls, F + e.v_pos, e.ws + e.v_x, B, P, H, dt, s));
    char* xv = e.ws + e.v_x;
    EmbedArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.ids = e.wsp<int64_t>(e.ids_all); ea.mask = e.wsp<int64_t>(e.mask_all);
    ea.word = W + e.t_word; ea.pos = W + e.t_pos; ea.type = W + e.t_type; ea.gamma = W + e.t_eln_w; ea.beta = W + e.t_eln_b;
    ea.x = e.ws + e.x0; ea.xhat = e.ws + e.xhat_emb; ea.rstd = e.wsp<float>(e.rstd_emb); ea.pos_ids = e.wsp<int>(e.pos_ids);
    ea.maskbias = e.wsp<float>(e.maskbias);
    ea.posts = Bt; ea.T = T; ea.H = H; ea.xlmr = c.txt_kind == MMHIP_TXT_XLMR; ea.pad_id = c.pad_id; ea.eps = c.ln_eps_txt;
    ea.drop = make_drop(c.p_hidden, e.seed, STREAM_EMBED, tr);
    if (px) { ea.x_pair = e.ws + e.x0p; ea.ld_pair = 2 * H; ea.lo_pair = H; }
    CHECK_HIP(launch_embed_fwd(ea, dt, s));
    const char* xt = e.ws + e.x0;
    const char* xtg = px ? e.ws + e.x0p : xt;          // the text layer input as the GEMMs read it
    if (e.cls_only < 0) { const char* v = getenv("MMHIP_CLS_ONLY"); e.cls_only = v ? atoi(v) : 1; }
    e.cls_compact = false;
    const int L = c.layers_txt > c.layers_img ? c.layers_txt : c.layers_img;
    // Between the paired GEMMs the towers' row kernels are independent: the image tower's attention / LayerNorm go to the side stream, beside the
    // text tower's on `s` (fork / join by events; 71 + 42 us of attention and 18 + 13 us of LayerNorm per layer, each alone on the chip, round 5)
    hipStream_t sv = use_side(e) ? e.side_vit[0] : s;
    auto fork = [&]() -> int { if (sv != s) { CHECK_HIP(hipEventRecord(e.ev_fork, s)); CHECK_HIP(hipStreamWaitEvent(sv, e.ev_fork, 0)); } return 0; };
    auto join = [&]() -> int { if (sv != s) { CHECK_HIP(hipEventRecord(e.ev_vit, sv)); CHECK_HIP(hipStreamWaitEvent(s, e.ev_vit, 0)); } return 0; };
    auto attn = [&](hipStream_t st, const char* qkv, const float* mb, char* ctx, float* lse, int posts, int S, int heads, const DropCfg* d, int q_tiles) -> int {
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.qkv = qkv; at.maskbias = mb; at.ctx = ctx; at.lse = lse; at.posts = posts; at.S = S; at.heads = heads; at.ld_qkv = 3 * H; at.ld_ctx = H; at.hidden = H;
        if (px) { at.pair = 1; at.ld_qkv = 6 * H; at.lo_qkv = 3 * H; at.ld_ctx = 2 * H; at.lo_ctx = H; }
        at.scale = 1.0f / sqrtf((float)(H / heads));
        if (d) at.drop = *d;
        at.q_tiles = q_tiles;
        CHECK_HIP(launch_attn_fwd(at, dt, st));
        return 0;
    };
    for (int l = 0; l < L; ++l) {
        const bool ht = l < c.layers_txt, hv = l < c.layers_img;
        const LayerOff& ot = e.txt[ht ? l : 0];
        const LayerW16& wt = e.txt_w16[ht ? l : 0];
        const TextAct& a = e.tact[ht ? l : 0];
        const LayerOff& ov = e.vit[hv ? l : 0];
        const LayerW16& wv = e.vit_w16[hv ? l : 0];
        const bool compact = ht && e.cls_only > 0 && l == c.layers_txt - 1;
        const int Mr = compact ? Bt : Mt, rs = compact ? T * H : H, rmul = compact ? T : 1;
        auto both = [&](G& gt, G& gv) -> int {
            if (ht && hv) return run_gemm_pair(e, gt, gv, s);
            return run_gemm(e, ht ? gt : gv, s);
        };
        // ---- QKV (image tower: pre-LN; layer 0's runs here, the later layers' beside the text tower's closing LayerNorm of the layer above)
        if (hv && l == 0) {
            LNArgs ln{xv, e.ws + e.v_ln, F + ov.ln1_w, F + ov.ln1_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
            ln_to(ln, e.v_ln, H);
            CHECK_HIP(launch_layernorm_fwd(ln, dt, s));
        }
        {
            G gt(xtg, H, e.ws + wt.qkv, H, e.ws + a.qkv, 3 * H, Mt, 3 * H, H); gt.bias(W + ot.qkv_b).px_in(px).px_out(px);
            G gv(e.ws + e.v_ln, H, e.ws + wv.qkv, H, e.ws + e.v_qkv, 3 * H, Mv, 3 * H, H); gv.bias(F + ov.qkv_b).px_in(px).px_out(px);
            if (int r = both(gt, gv)) return r;
        }
        // ---- attention
        if (hv) {
            if (int r = fork()) return r;
            if (int r = attn(sv, e.ws + e.v_qkv, nullptr, e.ws + e.v_ctx, nullptr, B, P, e.heads_v(), nullptr, 0)) return r;
        }
        if (ht) {
            const DropCfg d = make_drop(c.p_attn, e.seed, stream_attn(l), tr);
            if (int r = attn(s, e.ws + a.qkv, e.wsp<float>(e.maskbias), e.ws + a.ctx, e.wsp<float>(a.lse), Bt, T, c.heads, &d, compact ? 1 : 0)) return r;
        }
        if (hv) if (int r = join()) return r;
        // ---- attention output (+ residual)
        {
            G gt(e.ws + a.ctx, rs, e.ws + wt.ao, H, e.ws + a.pre1, H, Mr, H, H);
            gt.bias(W + ot.ao_b).dropout(make_drop(c.p_hidden, e.seed, stream_attn_out(l), tr), rmul).residual(xt, rs).px_in(px);
            G gv(e.ws + e.v_ctx, H, e.ws + wv.ao, H, xv, H, Mv, H, H); gv.bias(F + ov.ao_b).residual(xv, H).px_in(px);
            if (int r = both(gt, gv)) return r;
        }
        if (hv) {
            if (int r = fork()) return r;
            LNArgs ln2{xv, e.ws + e.v_ln, F + ov.ln2_w, F + ov.ln2_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
            ln_to(ln2, e.v_ln, H);
            CHECK_HIP(launch_layernorm_fwd(ln2, dt, sv));
        }
        if (ht) {
            LNArgs ln1{e.ws + a.pre1, e.ws + a.a1, W + ot.ln1_w, W + ot.ln1_b, e.wsp<float>(a.mean1), e.wsp<float>(a.rstd1), Mr, H, H, H, c.ln_eps_txt};
            if (px) { ln1.y_pair = e.ws + a.a1p; ln1.ld_pair = 2 * H; ln1.lo_pair = H; }
            CHECK_HIP(launch_layernorm_fwd(ln1, dt, s));
        }
        if (hv) if (int r = join()) return r;
        // ---- feed-forward
        {
            G gt(e.ws + (px ? a.a1p : a.a1), H, e.ws + wt.fc1, H, e.ws + a.h, I, Mr, I, H); gt.bias(W + ot.fc1_b).aux(e.ws + a.u, I).gelu().px_in(px).px_out(px);
            G gv(e.ws + e.v_ln, H, e.ws + wv.fc1, H, e.ws + e.v_h, I, Mv, I, H); gv.bias(F + ov.fc1_b).gelu().px_in(px).px_out(px);
            if (int r = both(gt, gv)) return r;
        }
        {
            G gt(e.ws + a.h, I, e.ws + wt.fc2, I, e.ws + a.pre2, H, Mr, H, I);
            gt.bias(W + ot.fc2_b).dropout(make_drop(c.p_hidden, e.seed, stream_ffn_out(l), tr), rmul).residual(e.ws + a.a1, H).px_in(px);
            G gv(e.ws + e.v_h, I, e.ws + wv.fc2, I, xv, H, Mv, H, I); gv.bias(F + ov.fc2_b).residual(xv, H).px_in(px);
            if (int r = both(gt, gv)) return r;
        }
        // ---- the text layer's closing LayerNorm beside the image tower's opening one of the next layer
        const bool vnext = l + 1 < c.layers_img;
        if (vnext) {
            if (int r = fork()) return r;
            const LayerOff& on = e.vit[l + 1];
            LNArgs ln{xv, e.ws + e.v_ln, F + on.ln1_w, F + on.ln1_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
            ln_to(ln, e.v_ln, H);
            CHECK_HIP(launch_layernorm_fwd(ln, dt, sv));
        }
        if (ht) {
            LNArgs ln2{e.ws + a.pre2, e.ws + a.out, W + ot.ln2_w, W + ot.ln2_b, e.wsp<float>(a.mean2), e.wsp<float>(a.rstd2), Mr, H, H, H, c.ln_eps_txt};
            if (px) { ln2.y_pair = e.ws + a.outp; ln2.ld_pair = 2 * H; ln2.lo_pair = H; }
            CHECK_HIP(launch_layernorm_fwd(ln2, dt, s));
            e.cls_compact = compact;
            xt = e.ws + a.out;
            xtg = px ? e.ws + a.outp : xt;
        }
        if (vnext) if (int r = join()) return r;
    }
    LNArgs lnf{xv, e.ws + e.v_out, F + e.v_ln_w, F + e.v_ln_b, nullptr, nullptr, Mv, H, H, H, c.ln_eps_img};
    CHECK_HIP(launch_layernorm_fwd(lnf, dt, s));
    SmallGemmArgs sp = small(e.ws + e.v_out, P * H, F + e.v_pool_w, H, F + e.v_pool_b, e.wsp<float>(e.h_vpool), H, B, H, H, ACT_TANH);
    CHECK_HIP(launch_small_nt(sp, dt, s));
    if (int r = e.span(1, s)) return r;
    return e.span(2, s);
}

// The split of the chip's 256 CUs between the two towers of this forward: both towers' big GEMMs run 256 x 256 tiles of K / 64 K-steps
// each, a launch of n tiles on c workgroups takes ceil(n / c) rounds, a round costs its K-steps plus a fixed prologue + epilogue share
// (in K-step units: 5, from the in-kernel stamps: 7.5 us of 29.5 at K = 768).  Pick the split (whole XCD-eighths: multiples of 8) that
// minimises the longer tower; partition only when the model says it beats both towers sharing every launch's tail.
void choose_partition(mmhip_engine& e) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, I = c.inter, Hv = e.Hv(), Iv = e.Iv();
    if (H % 256 || I % 256 || Hv % 256 || Iv % 256 || c.layers_txt < 1 || c.layers_img < 1) return;
    const long Mt = (long)e.Bt * e.T, Mv = (long)e.B * e.P();
    if (Mt < 2048 || Mv < 2048) return;
    const long rt = (Mt + 255) / 256, rv = (Mv + 255) / 256;
    auto tower = [](long rows, int h, int inter, int layers, int cap) -> double {
        auto gemm = [&](int n, int k) { const long tiles = rows * (n / 256); return (double)((tiles + cap - 1) / cap) * (k / 64 + 5); };
        return layers * (gemm(3 * h, h) + gemm(h, h) + gemm(inter, h) + gemm(h, inter));
    };
    // Same-box A/B of the step (profiles/r03_step_ab.txt): the split pays where the image tower is clearly the longer one (CLIP-L/14 beside
    // 32 posts of text: 12.8 vs 13.2 ms), is neutral to -3 % at config 2 (12608 image rows beside 8192 text rows: 10.24-10.30 either way on two boxes, 10.0 vs 10.3 on a third) and LOSES
    // where the text tower is the longer one (config 3, 16384 text rows with ITM: 15.95 vs 15.3 ms) -- free sharing then fills the tails of
    // the short tower's launches with the long tower's workgroups, which a fixed split forbids.
    // (model costs, image / text: config 2 1.12, config 3 0.90, config 4 > 3.  Config 2 stays unpartitioned: nothing to gain on three boxes of
    // four, and a capped launch is priced against the whole chip in bench.py's roofline line; MMHIP_PART=96,160 turns it on)
    if (tower(rv, Hv, Iv, c.layers_img, 256) < 1.3 * tower(rt, H, I, c.layers_txt, 256)) return;
    double best = 1e30; int bt = 0;
    for (int t = 32; t <= 224; t += 8) {
        const double ct = tower(rt, H, I, c.layers_txt, t), cv = tower(rv, Hv, Iv, c.layers_img, 256 - t);
        const double m = ct > cv ? ct : cv;
        if (m < best) { best = m; bt = t; }
    }
    if (bt) { e.cur_part[0] = bt; e.cur_part[1] = 256 - bt; }
}

const char* text_last(const mmhip_engine& e) { return e.cfg.layers_txt ? e.ws + e.tact[e.cfg.layers_txt - 1].out : e.ws + e.x0; }

int heads_forward(mmhip_engine& e, float* out_cls, float* logits, float* out_tim, float* feats_out, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, E = c.proj_dim, C = c.num_labels, T = e.T, B = e.B, Bt = e.Bt, dt = e.dt();
    const int P = e.P(), Hv = e.Hv(), ZW = H + Hv;     // z = [x_t CLS | image feature]: H + Hv wide
    const float* W = e.train;
    const char* xt = text_last(e);
    const int cs = e.cls_compact ? H : T * H;           // row stride of the CLS rows of the last hidden state
    // z = [x_t CLS | fused image feature]; the fp32 copy of the CLS rows in z[:, :H] is also the A operand of the text pooler and of
    // fc_Q (the 16-bit rows would send small_gemm down its scalar-load path: 21 us instead of 11)
    float* z = e.wsp<float>(e.h_z);
    CHECK_HIP(launch_gather_rows_f32(xt, (size_t)cs, z, ZW, Bt, H, dt, s));
    // text pooler (first B posts) and ITC similarity -- HF dual encoder :261-274
    e.itc_done = !(e.skip_itc && logits == nullptr);
    if (e.itc_done) {
        CHECK_HIP(launch_small_nt(small(z, ZW, W + e.t_pool_w, H, W + e.t_pool_b, e.wsp<float>(e.h_tpool), H, B, H, H, ACT_TANH), DT_F32, s));
        CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_tpool), H, W + e.tproj_w, H, nullptr, e.wsp<float>(e.h_txt_e), E, B, E, H), DT_F32, s));
        CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_vpool), Hv, W + e.vproj_w, Hv, nullptr, e.wsp<float>(e.h_img_e), E, B, E, Hv), DT_F32, s));
        ItcArgs it{e.wsp<float>(e.h_txt_e), e.wsp<float>(e.h_img_e), W + e.logit_scale, e.wsp<float>(e.h_txt_n), e.wsp<float>(e.h_img_n),
                   e.wsp<float>(e.h_txt_inv), e.wsp<float>(e.h_img_inv), e.wsp<float>(e.h_logits), B, E};
        CHECK_HIP(launch_itc_fwd(it, s));
    }
    if (c.fusion == MMHIP_FUSION_ATTENTION) {
        CHECK_HIP(launch_small_nt(small(z, ZW, W + e.fq_w, H, W + e.fq_b, e.wsp<float>(e.h_q), H, Bt, H, H), DT_F32, s));
        CHECK_HIP(launch_small_nn(small(e.wsp<float>(e.h_q), H, W + e.fk_w, H, nullptr, e.wsp<float>(e.h_qk), H, Bt, H, H), s));
        FusionAttnArgs fa{e.wsp<float>(e.h_qk), e.ws + e.v_out, e.wsp<float>(e.h_prob), e.wsp<float>(e.h_xbar), Bt, B, P, H, 1.0f / sqrtf((float)H)};
        CHECK_HIP(launch_fusion_attn_fwd(fa, dt, s));
        CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_xbar), H, W + e.fv_w, H, W + e.fv_b, z + H, ZW, Bt, H, H), DT_F32, s));
    } else {
        // concat: image CLS row of post bt % B -- two strided copies (original rows, ITM rows)
        CHECK_HIP(launch_gather_rows_f32(e.ws + e.v_out, (size_t)P * Hv, z + H, ZW, B, Hv, dt, s));
        if (Bt > B) CHECK_HIP(launch_gather_rows_f32(e.ws + e.v_out, (size_t)P * Hv, z + (size_t)B * ZW + H, ZW, B, Hv, dt, s));
    }
    float* feats = e.wsp<float>(e.h_feats);
    CHECK_HIP(launch_small_nt(small(z, ZW, W + e.fus_w, ZW, W + e.fus_b, feats, H, Bt, H, ZW, ACT_RELU), DT_F32, s));
    CHECK_HIP(launch_elementwise(EW_DROPOUT, feats, nullptr, e.wsp<float>(e.h_featd), (size_t)B * H, 0.f,
                                 make_drop(c.p_head, e.seed, STREAM_HEAD, e.train_mode), s));
    CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_featd), H, W + e.cls_w, H, W + e.cls_b, e.wsp<float>(e.h_out_cls), C, B, C, H), DT_F32, s));
    if (e.itm) CHECK_HIP(launch_small_nt(small(feats + (size_t)B * H, H, W + e.tim_w, H, W + e.tim_b, e.wsp<float>(e.h_out_tim), 2, B, 2, H), DT_F32, s));
    if (out_cls) CHECK_HIP(hipMemcpyAsync(out_cls, e.ws + e.h_out_cls, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s));
    if (logits && e.itc_done) CHECK_HIP(hipMemcpyAsync(logits, e.ws + e.h_logits, (size_t)B * B * 4, hipMemcpyDeviceToDevice, s));
    if (out_tim && e.itm) CHECK_HIP(hipMemcpyAsync(out_tim, e.ws + e.h_out_tim, (size_t)B * 2 * 4, hipMemcpyDeviceToDevice, s));
    if (feats_out) CHECK_HIP(hipMemcpyAsync(feats_out, feats, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ backward pieces
int heads_backward(mmhip_engine& e, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, E = c.proj_dim, C = c.num_labels, T = e.T, B = e.B, Bt = e.Bt, dt = e.dt();
    const int P = e.P(), Hv = e.Hv(), ZW = H + Hv;     // z = [x_t CLS | image feature]: H + Hv wide
    const float* W = e.train;
    float* Gd = e.grad;
    const float* d_out_cls = e.bd_out_cls ? e.bd_out_cls : e.wsp<float>(e.h_d_out_cls);
    const float* d_logits = e.bd_logits;
    const float* d_out_tim = e.bd_out_tim;
    float* dfeats = e.wsp<float>(e.h_dfeats);
    float* feats = e.wsp<float>(e.h_feats);
    float* z = e.wsp<float>(e.h_z);
    float* dxcls = e.wsp<float>(e.h_dxcls);
    const DropCfg nodrop = make_drop(0.f, 0, 0, false);
    // classifier: out_cls = linear_cls(dropout(feats[:B]))
    CHECK_HIP(launch_small_nn(small(d_out_cls, C, W + e.cls_w, H, nullptr, dfeats, H, B, H, C), s));
    CHECK_HIP(launch_elementwise(EW_DROPOUT, dfeats, nullptr, dfeats, (size_t)B * H, 0.f, make_drop(c.p_head, e.seed, STREAM_HEAD, e.train_mode), s));
    if (e.bd_feats) CHECK_HIP(launch_elementwise(EW_ADD, dfeats, e.bd_feats, dfeats, (size_t)B * H, 1.f, nodrop, s));
    CHECK_HIP(launch_small_tn(small(d_out_cls, C, e.wsp<float>(e.h_featd), H, nullptr, Gd + e.cls_w, H, B, H, 0, 0, 1), DT_F32, C, s));
    CHECK_HIP(launch_bias_grad_f32(d_out_cls, B, C, C, Gd + e.cls_b, 1, s));
    if (e.itm) {
        if (d_out_tim) {
            CHECK_HIP(launch_small_nn(small(d_out_tim, 2, W + e.tim_w, H, nullptr, dfeats + (size_t)B * H, H, B, H, 2), s));
            CHECK_HIP(launch_small_tn(small(d_out_tim, 2, feats + (size_t)B * H, H, nullptr, Gd + e.tim_w, H, B, H, 0, 0, 1), DT_F32, 2, s));
            CHECK_HIP(launch_bias_grad_f32(d_out_tim, B, 2, 2, Gd + e.tim_b, 1, s));
        } else {
            CHECK_HIP(hipMemsetAsync(dfeats + (size_t)B * H, 0, (size_t)B * H * 4, s));
        }
    }
    // feats = relu(linear_fusion(z))
    float* dpre = e.wsp<float>(e.h_dpre);
    CHECK_HIP(launch_elementwise(EW_RELU_BWD, dfeats, feats, dpre, (size_t)Bt * H, 0.f, nodrop, s));
    CHECK_HIP(launch_small_tn(small(dpre, H, z, ZW, nullptr, Gd + e.fus_w, ZW, Bt, ZW, 0, 0, 1), DT_F32, H, s));
    CHECK_HIP(launch_bias_grad_f32(dpre, Bt, H, H, Gd + e.fus_b, 1, s));
    float* dz = e.wsp<float>(e.h_dz);
    CHECK_HIP(launch_small_nn(small(dpre, H, W + e.fus_w, ZW, nullptr, dz, ZW, Bt, ZW, H), s));
    // d x_t[:,0] starts as dz[:, :H]
    CHECK_HIP(hipMemcpy2DAsync(dxcls, (size_t)H * 4, dz, (size_t)ZW * 4, (size_t)H * 4, Bt, hipMemcpyDeviceToDevice, s));
    if (c.fusion == MMHIP_FUSION_ATTENTION) {
        const float* dctx = dz + H;      // ld ZW
        float* q = e.wsp<float>(e.h_q);
        CHECK_HIP(launch_small_tn(small(dctx, ZW, e.wsp<float>(e.h_xbar), H, nullptr, Gd + e.fv_w, H, Bt, H, 0, 0, 1), DT_F32, H, s));
        CHECK_HIP(launch_bias_grad_f32(dctx, Bt, H, ZW, Gd + e.fv_b, 1, s));
        CHECK_HIP(launch_small_nn(small(dctx, ZW, W + e.fv_w, H, nullptr, e.wsp<float>(e.h_dxbar), H, Bt, H, H), s));
        FusionAttnBwdArgs fb{e.wsp<float>(e.h_dxbar), e.wsp<float>(e.h_prob), e.ws + e.v_out, e.wsp<float>(e.h_dqk), Bt, B, P, H, 1.0f / sqrtf((float)H)};
        CHECK_HIP(launch_fusion_attn_bwd(fb, dt, s));
        // qk = q . W_K  ->  dW_K[o][i] = sum q[:,o] dqk[:,i];  dq = dqk . W_K^T ;  fc_K.bias gets an exactly-zero gradient
        CHECK_HIP(launch_small_tn(small(q, H, e.wsp<float>(e.h_dqk), H, nullptr, Gd + e.fk_w, H, Bt, H, 0, 0, 1), DT_F32, H, s));
        CHECK_HIP(launch_small_nt(small(e.wsp<float>(e.h_dqk), H, W + e.fk_w, H, nullptr, e.wsp<float>(e.h_dq), H, Bt, H, H), DT_F32, s));
        CHECK_HIP(launch_small_tn(small(e.wsp<float>(e.h_dq), H, z, ZW, nullptr, Gd + e.fq_w, H, Bt, H, 0, 0, 1), DT_F32, H, s));
        CHECK_HIP(launch_bias_grad_f32(e.wsp<float>(e.h_dq), Bt, H, H, Gd + e.fq_b, 1, s));
        CHECK_HIP(launch_small_nn(small(e.wsp<float>(e.h_dq), H, W + e.fq_w, H, nullptr, dxcls, H, Bt, H, H, ACT_NONE, 1), s));
    }
    if (d_logits) {
        ItcBwdArgs ib{d_logits, e.wsp<float>(e.h_logits), e.wsp<float>(e.h_txt_n), e.wsp<float>(e.h_img_n), e.wsp<float>(e.h_txt_inv),
                      e.wsp<float>(e.h_img_inv), W + e.logit_scale, e.wsp<float>(e.h_dtxt_e), e.wsp<float>(e.h_dimg_e), Gd + e.logit_scale, B, E};
        CHECK_HIP(launch_itc_bwd(ib, s));
        CHECK_HIP(launch_small_tn(small(e.wsp<float>(e.h_dtxt_e), E, e.wsp<float>(e.h_tpool), H, nullptr, Gd + e.tproj_w, H, B, H, 0, 0, 1), DT_F32, E, s));
        CHECK_HIP(launch_small_tn(small(e.wsp<float>(e.h_dimg_e), E, e.wsp<float>(e.h_vpool), Hv, nullptr, Gd + e.vproj_w, Hv, B, Hv, 0, 0, 1), DT_F32, E, s));
        CHECK_HIP(launch_small_nn(small(e.wsp<float>(e.h_dtxt_e), E, W + e.tproj_w, H, nullptr, e.wsp<float>(e.h_dtpool), H, B, H, E), s));
        CHECK_HIP(launch_elementwise(EW_TANH_BWD, e.wsp<float>(e.h_dtpool), e.wsp<float>(e.h_tpool), e.wsp<float>(e.h_dprepool), (size_t)B * H, 0.f, nodrop, s));
        CHECK_HIP(launch_small_tn(small(e.wsp<float>(e.h_dprepool), H, z, ZW, nullptr, Gd + e.t_pool_w, H, B, H, 0, 0, 1), DT_F32, H, s));
        CHECK_HIP(launch_bias_grad_f32(e.wsp<float>(e.h_dprepool), B, H, H, Gd + e.t_pool_b, 1, s));
        CHECK_HIP(launch_small_nn(small(e.wsp<float>(e.h_dprepool), H, W + e.t_pool_w, H, nullptr, dxcls, H, B, H, H, ACT_NONE, 1), s));
    }
    // gradient of the last hidden state: CLS rows only
    // gradient of the last hidden state: CLS rows only (compact [Bt, H] when the last layer ran on CLS rows)
    CHECK_HIP(launch_scatter_cls_rows(dxcls, e.ws + e.g_dx, Bt, e.cls_compact ? 1 : T, H, dt, s, e.gscale()));
    return 0;
}

// one text layer; on entry g_dx holds the gradient of the layer's output, on exit of its input
int text_layer_backward(mmhip_engine& e, int l, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const int H = c.hidden, I = c.inter, T = e.T, Bt = e.Bt, Mt = Bt * T, dt = e.dt();
    const float* W = e.train;
    float* Gd = e.grad;
    const LayerOff& o = e.txt[l];
    const LayerW16& w = e.txt_w16[l];
    const TextAct& a = e.tact[l];
    const bool px = e.px;
    const int np = e.bwd_np;          // parity mode: MFMA products per k slice of the backward's GEMMs (mmhip_set_backward_products)
    const bool hi1 = px && np == 1;   // one product everywhere downstream: the lo planes of du / d ctx / d qkv / the dropped LayerNorm gradients are never read -- not written
    const char* x_in = px ? (l ? e.ws + e.tact[l - 1].outp : e.ws + e.x0p) : (l ? e.ws + e.tact[l - 1].out : e.ws + e.x0);      // as the dWqkv product reads it
    const bool tr = e.train_mode;
    // temporaries that the weight-gradient GEMMs read are double-buffered per layer parity so that the side stream may
    // still be consuming layer l's set while the main stream runs layer l-1 on the other one
    const int set = l & 1;
    const bool side = use_side(e);
    // last layer in CLS-only mode: its gradient arrives as compact [Bt, H] rows; everything up to d ctx runs on Bt rows
    const bool compact = e.cls_compact && l == c.layers_txt - 1;
    const int Mr = compact ? Bt : Mt, rs = compact ? T * H : H, rmul = compact ? T : 1;
    char *dx = e.ws + e.g_dx, *dx2 = e.ws + e.g_dx2, *dctx = e.ws + e.g_dctx;
    char *dpre2 = e.ws + e.g_set[set][0], *ddrop2 = e.ws + e.g_set[set][1], *du = e.ws + e.g_set[set][2];
    char *dpre1 = e.ws + e.g_set[set][3], *ddrop1 = e.ws + e.g_set[set][4], *dqkv = e.ws + e.g_set[set][5];
    if (side && e.tn_pending[set]) {           // the set was last read by layer l+2's side work
        CHECK_HIP(hipStreamWaitEvent(s, e.ev_tn[set], 0));
        e.tn_pending[set] = false;
    }
    // ---- x' = LN2(pre2), pre2 = drop(fc2(h)) + a1.  LN backward also emits the dropout-backward copy and the bias
    // gradient of the Linear that fed the LN
    const DropCfg d_ffn = make_drop(c.p_hidden, e.seed, stream_ffn_out(l), tr);
    // (the second stage of its dgamma / dbeta reduction is not on the dX chain: it runs with the layer's dW on the side stream)
    LNBwdArgs b2{dx, e.ws + a.pre2, W + o.ln2_w, e.wsp<float>(a.mean2), e.wsp<float>(a.rstd2), dpre2, nullptr, Gd + o.ln2_w, Gd + o.ln2_b, Mr, H,
                 e.wsp<float>(e.g_lnp[set][0]), 1.0f / e.gscale(), d_ffn.thresh16 ? ddrop2 : nullptr, nullptr, d_ffn, rmul, 1};
    if (px) { b2.pair_out = ddrop2; b2.ld_pair = 2 * H; b2.lo_pair = H; b2.pair_hi_only = hi1; }      // parity mode: the GEMMs' operand (dropped or not) as a plane pair in the ddrop buffer
    CHECK_HIP(launch_layernorm_bwd(b2, dt, s));
    const char* df = (px || d_ffn.thresh16) ? ddrop2 : dpre2;
    // du = (df . W2) * gelu'(u);  d_a1 = du . W1 + dpre2
    if (e.part_bwd < 0) { const char* v = getenv("MMHIP_PART_BWD"); e.part_bwd = v ? atoi(v) : 0; }
    static int bwd_mask = -1;
    if (bwd_mask < 0) { const char* v = getenv("MMHIP_PART_BWD_MASK"); bwd_mask = v ? atoi(v) : 15; }
    auto part_bwd = [&](G& g, int bit) {
        if (e.part_bwd > 0 && (bwd_mask & bit) && side && g.a.M >= 2048 && g.a.N % 256 == 0 && g.a.K % 64 == 0 && (dt == DT_BF16 || dt == DT_F16)) { g.a.tile = 15; g.a.grid = e.part_bwd; }
    };
    { G g(df, H, e.ws + w.fc2T, H, du, I, Mr, I, H); g.mul_gelu_grad(e.ws + a.u, I).px_in(px, np).px_out(px, hi1); part_bwd(g, 1); if (int r = run_gemm(e, g, s)) return r; }
    // the two long-K activation-gradient GEMMs of the layer (768 wide): one role-specialised 256x96 tile per CU when M gives
    // exactly <= 256 of them -- in isolation 7 % faster than the 192 tiles of 256x128, in the step -0.05 ms (same-box A/B; the
    // same tile in the FORWARD costs +0.3 ms: it leaves no CU to the image tower).  MMHIP_BWD_TILE12=0 turns it off.
    static int bt12 = -1;
    if (bt12 < 0) { const char* v = getenv("MMHIP_BWD_TILE12"); bt12 = v ? atoi(v) : 1; }
    const int nt = (bt12 && Mr >= 4096 && Mr <= 8192 && H % 96 == 0) ? 12 : 0;
    { G g(du, I, e.ws + w.fc1T, I, dx2, H, Mr, H, I); g.residual(dpre2, H).px_in(px, np); g.a.tile = nt; part_bwd(g, 2); if (int r = run_gemm(e, g, s)) return r; }
    // ---- a1 = LN1(pre1), pre1 = drop(ao(ctx)) + x_in        (dx2 = d_a1)
    const DropCfg d_ao = make_drop(c.p_hidden, e.seed, stream_attn_out(l), tr);
    LNBwdArgs b1{dx2, e.ws + a.pre1, W + o.ln1_w, e.wsp<float>(a.mean1), e.wsp<float>(a.rstd1), dpre1, nullptr, Gd + o.ln1_w, Gd + o.ln1_b, Mr, H,
                 e.wsp<float>(e.g_lnp[set][1]), 1.0f / e.gscale(), d_ao.thresh16 ? ddrop1 : nullptr, nullptr, d_ao, rmul, 1};
    if (px) { b1.pair_out = ddrop1; b1.ld_pair = 2 * H; b1.lo_pair = H; b1.pair_hi_only = hi1; }
    CHECK_HIP(launch_layernorm_bwd(b1, dt, s));
    const char* dout = (px || d_ao.thresh16) ? ddrop1 : dpre1;
    if (compact) {
        // d ctx for the CLS rows only, spread into an otherwise-zero full tensor for the attention backward (parity mode: pair rows of 2 H 16-bit
        // elements are moved as the H 4-byte words they occupy; an all-zero pair is zero)
        { G g(dout, H, e.ws + w.aoT, H, dx2, H, Mr, H, H); g.px_in(px, np).px_out(px, hi1); if (int r = run_gemm(e, g, s)) return r; }
        CHECK_HIP(hipMemsetAsync(dctx, 0, (size_t)Mt * H * e.esz(), s));
        CHECK_HIP(launch_scatter_rows16(dx2, dctx, Bt, (size_t)T * H, H, 0, dt, s));
        CHECK_HIP(hipMemsetAsync(dqkv, 0, (size_t)Mt * 3 * H * e.esz(), s));      // dQ of the skipped query tiles is zero
    } else {
        G g(dout, H, e.ws + w.aoT, H, dctx, H, Mt, H, H);
        g.px_in(px, np).px_out(px, hi1);
        g.a.tile = (bt12 & 2) ? nt : 0;
        part_bwd(g, 4);
        if (int r = run_gemm(e, g, s)) return r;
    }
    AttnBwdArgs ab;
    memset(&ab, 0, sizeof(ab));
    ab.qkv = e.ws + a.qkv; ab.maskbias = e.wsp<float>(e.maskbias); ab.ctx = e.ws + a.ctx; ab.dctx = dctx; ab.lse = e.wsp<float>(a.lse);
    ab.dqkv = dqkv; ab.posts = Bt; ab.S = T; ab.heads = c.heads; ab.ld_qkv = 3 * H; ab.ld_ctx = H; ab.hidden = H;
    if (px) { ab.pair = 1; ab.ld_qkv = 6 * H; ab.lo_qkv = 3 * H; ab.ld_ctx = 2 * H; ab.lo_ctx = H; ab.nprod = np; }
    ab.scale = 1.0f / sqrtf((float)(H / c.heads));
    ab.drop = make_drop(c.p_attn, e.seed, stream_attn(l), tr);
    ab.q_tiles = compact ? 1 : 0;
    CHECK_HIP(launch_attn_bwd(ab, dt, s));
    // ---- parameter gradients of the layer: off the critical path -> side stream: all four weight gradients AND their bias
    // gradients in one grouped launch, no split-K, plain stores
    hipStream_t ps = s;
    if (side) {
        CHECK_HIP(hipEventRecord(e.ev_ready[set], s));
        CHECK_HIP(hipStreamWaitEvent(e.side, e.ev_ready[set], 0));
        ps = e.side;
    }
    if (compact) {
        // dx_in = dqkv . Wqkv, plus d pre1 on the CLS rows (the residual branch of the CLS rows)
        { G g(dqkv, 3 * H, e.ws + w.qkvT, 3 * H, dx, H, Mt, H, 3 * H); g.px_in(px, np); if (int r = run_gemm(e, g, s)) return r; }
        CHECK_HIP(launch_scatter_rows16(dpre1, dx, Bt, (size_t)T * H, H, 1, dt, s));
    } else {
        G g(dqkv, 3 * H, e.ws + w.qkvT, 3 * H, dx, H, Mt, H, 3 * H);
        g.residual(dpre1, H).px_in(px, np);
        g.a.tile = nt;
        part_bwd(g, 8);
        if (int r = run_gemm(e, g, s)) return r;
    }
    GemmTNProblem pr[4];
    CHECK_HIP(launch_layernorm_bwd_reduce(b2, ps));
    CHECK_HIP(launch_layernorm_bwd_reduce(b1, ps));
    // each problem also yields its Linear's bias gradient (column sums of the dY operand) from the same tiles
    pr[0] = GemmTNProblem{df, e.ws + a.h, Gd + o.fc2_w, Mr, H, I, H, I, I, 0, Gd + o.fc2_b};           // dW2[H,I]   = df^T h
    pr[1] = GemmTNProblem{du, e.ws + a.a1, Gd + o.fc1_w, Mr, I, H, I, H, H, 0, Gd + o.fc1_b};          // dW1[I,H]   = du^T a1
    pr[2] = GemmTNProblem{dqkv, x_in, Gd + o.qkv_w, Mt, 3 * H, H, 3 * H, H, H, 0, Gd + o.qkv_b};       // dWqkv[3H,H] = dqkv^T x_in
    pr[3] = GemmTNProblem{dout, e.ws + a.ctx, Gd + o.ao_w, Mr, H, H, H, rs, H, 0, Gd + o.ao_b};        // dWo[H,H]   = dout^T ctx (CLS rows: stride T*H)
    if (px) {
        pr[1].B = e.ws + a.a1p;
        for (auto& q : pr) { q.pair = 1; q.lda *= 2; q.ldb *= 2; q.a_lo = q.Nn; q.b_lo = q.Nc; q.nprod = np; }
    }
    {
        const int i = e.side && ps == e.side ? 1 : 0;
        CHECK_HIP(launch_gemm_tn(pr, 4, 0, dt, 0, ps, 1.0f / e.gscale(), e.x3_bytes[i] ? e.ws + e.x3_ws[i] : nullptr, e.x3_bytes[i]));
    }
    if (side) {
        CHECK_HIP(hipEventRecord(e.ev_tn[set], e.side));
        e.tn_pending[set] = true;
    }
    return 0;
}

// join the side stream: after this, every gradient is final in the caller's stream order
int backward_finish(mmhip_engine& e, hipStream_t s) {
    for (int set = 0; set < 2; ++set)
        if (e.tn_pending[set]) {
            CHECK_HIP(hipStreamWaitEvent(s, e.ev_tn[set], 0));
            e.tn_pending[set] = false;
        }
    return 0;
}

// device words of the overflow guard: [0] counter of non-finite gradient sightings, [1] "this step is void" flag raised by the backward and
// honoured by the AdamW kernels.  Per handle (mmhip_set_guard; round 4: two models in one process no longer share a flag); the process-wide
// pair of mmhip_set_step_guard / mmhip_set_nonfinite_counter serves the handle-less AdamW entry points and handles without a guard of their own.
static unsigned* g_nonfinite = nullptr;
static unsigned* g_skip = nullptr;
inline unsigned* guard_counter(const mmhip_engine& e) { return e.guard ? e.guard : g_nonfinite; }
inline unsigned* guard_flag(const mmhip_engine& e) { return e.guard ? e.guard + 1 : g_skip; }

int embed_backward(mmhip_engine& e, hipStream_t s) {
    const mmhip_config& c = e.cfg;
    const float* W = e.train;
    float* Gd = e.grad;
    EmbedBwdArgs b;
    memset(&b, 0, sizeof(b));
    b.dx = e.ws + e.g_dx; b.xhat = e.ws + e.xhat_emb; b.rstd = e.wsp<float>(e.rstd_emb); b.gamma = W + e.t_eln_w;
    b.ids = e.wsp<int64_t>(e.ids_all); b.pos_ids = e.wsp<int>(e.pos_ids);
    b.dword = Gd + e.t_word; b.dpos = Gd + e.t_pos; b.dtype = Gd + e.t_type; b.dgamma = Gd + e.t_eln_w; b.dbeta = Gd + e.t_eln_b;
    b.posts = e.Bt; b.T = e.T; b.H = c.hidden; b.pad_id = c.pad_id;
    b.pos_pad_id = c.txt_kind == MMHIP_TXT_XLMR ? c.pad_id : -1;     // nn.Embedding(padding_idx=...) rows get no gradient
    b.drop = make_drop(c.p_hidden, e.seed, STREAM_EMBED, e.train_mode);
    b.partial = e.wsp<float>(e.g_partial);
    b.alpha = 1.0f / e.gscale();
    b.row_state = e.word_row_state;
    if (deterministic()) { b.det_rows = e.wsp<float>(e.g_det_rows); b.max_pos = c.max_pos; }
    b.status = guard_flag(e) ? guard_counter(e) : nullptr;
    CHECK_HIP(launch_embed_bwd(b, e.dt(), s));
    return 0;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

const char* mmhip_version(void) { return "mmhip 0.1 (gfx950)"; }

int mmhip_create(const mmhip_config* cfg, mmhip_handle* out) {
    if (!cfg || !out) return MMHIP_E_INVALID;
    const mmhip_config& c = *cfg;
    if (c.hidden <= 0 || c.hidden % 64 || c.hidden > 1024 || c.heads * 64 != c.hidden || c.inter % 128 || c.hidden % 128) return MMHIP_E_INVALID;
    {   // image tower: ViT (HF ViTModel) or CLIP vision (HF CLIPVisionModel), its own width allowed (CLIP-ViT-L/14: 1024 / 16 / 4096)
        const int hv = c.hidden_img > 0 ? c.hidden_img : c.hidden, iv = c.inter_img > 0 ? c.inter_img : c.inter, nh = c.heads_img > 0 ? c.heads_img : c.heads;
        if (c.img_kind != MMHIP_IMG_VIT && c.img_kind != MMHIP_IMG_CLIP) return MMHIP_E_INVALID;
        if (hv % 128 || hv > 1024 || nh * 64 != hv || iv % 128 || c.patch < 2 || c.image % c.patch) return MMHIP_E_INVALID;
        if (c.fusion == MMHIP_FUSION_ATTENTION && hv != c.hidden) return MMHIP_E_INVALID;       // fc_K / fc_V are hidden x hidden (mm_late.py:74-75)
        // attention keeps the K / V of a head in LDS: 608 keys of 64 x 16 bit fill the 160 KB of a CU (336 / 14 -> 577 tokens);
        // the fp32 parity-mode attention holds 4-byte K / V: 288 keys
        const int P = (c.image / c.patch) * (c.image / c.patch) + 1;
        if (P > 608) return MMHIP_E_INVALID;
    }
    if (c.max_text_len > 128 || c.max_text_len < 1 || c.max_posts < 1 || c.max_posts > 1024) return MMHIP_E_INVALID;
    if (c.dtype != MMHIP_BF16 && c.dtype != MMHIP_F16 && c.dtype != MMHIP_BF16X3) return MMHIP_E_INVALID;
    if (c.num_labels < 1 || c.num_labels > 64 || c.proj_dim < 1 || c.proj_dim > 1024) return MMHIP_E_INVALID;
    if (c.txt_kind == MMHIP_TXT_XLMR && c.max_pos < c.max_text_len + c.pad_id + 1) return MMHIP_E_INVALID;
    if (c.txt_kind == MMHIP_TXT_BERT && c.max_pos < c.max_text_len) return MMHIP_E_INVALID;
    mmhip_engine* e = new (std::nothrow) mmhip_engine();
    if (!e) return MMHIP_E_INVALID;
    e->cfg = c;
    if (c.dtype == MMHIP_BF16X3) { const char* v = getenv("MMHIP_X3_PAIRS"); e->px = v ? atoi(v) != 0 : true; }
    if (c.dtype == MMHIP_BF16X3) { const char* v = getenv("MMHIP_X3_BWD"); const int n = v ? atoi(v) : 3; e->bwd_np = n >= 1 && n <= 3 ? n : 3; }
    build_layout(*e);
    build_workspace(*e);
    *out = e;
    return 0;
}
void mmhip_destroy(mmhip_handle h) {
    if (!h) return;
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        for (hipEvent_t ev : {h->ev_fork, h->ev_vit, h->ev_ready[0], h->ev_ready[1], h->ev_tn[0], h->ev_tn[1], h->ev_layer[0], h->ev_layer[1], h->ev_opt})
            if (ev) (void)hipEventDestroy(ev);
        for (auto sv : h->side_vit) if (sv) (void)hipStreamSynchronize(sv);      // pooled streams: drained, not destroyed
        for (auto ev : h->span_ev) if (ev) (void)hipEventDestroy(ev);
    }
    for (auto& ev : h->evs) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    delete h;
}
int mmhip_param_count(mmhip_handle h) { return h ? (int)h->params.size() : MMHIP_E_INVALID; }
int mmhip_param_info_at(mmhip_handle h, int i, mmhip_param_info* out) {
    if (!h || !out || i < 0 || i >= (int)h->params.size()) return MMHIP_E_INVALID;
    *out = h->params[i];
    return 0;
}
uint64_t mmhip_buffer_numel(mmhip_handle h, int buffer) { return !h ? 0 : (buffer == 0 ? h->n_frozen : h->n_train); }
uint64_t mmhip_workspace_bytes(mmhip_handle h) { return h ? h->ws_need : 0; }
int mmhip_bind(mmhip_handle h, float* frozen, float* train, float* train_grad, void* workspace, uint64_t workspace_bytes) {
    if (!h || !frozen || !train || !workspace) return MMHIP_E_INVALID;
    if (workspace_bytes < h->ws_need) return MMHIP_E_CAPACITY;
    if (((uintptr_t)frozen | (uintptr_t)train | (uintptr_t)train_grad | (uintptr_t)workspace) & 255) return MMHIP_E_INVALID;
    h->frozen = frozen; h->train = train; h->grad = train_grad; h->ws = (char*)workspace; h->ws_bytes = workspace_bytes;
    h->fwd_done = false;
    return 0;
}
int mmhip_refresh_weights(mmhip_handle h, int which, void* stream) {
    if (!h || !h->ws) return MMHIP_E_STATE;
    hipStream_t s = (hipStream_t)stream;
    mmhip_engine& e = *h;
    if (which & 1) {
        for (int l = 0; l < e.cfg.layers_img; ++l)
            if (int r = refresh_layer(e, e.frozen, e.vit[l], e.vit_w16[l], false, s, e.Hv(), e.Iv())) return r;
        // patch-embedding weight [Hv, 3*p*p], rows zero-padded to the GEMM's k-step (14 x 14 patches: 588 -> 640)
        CHECK_HIP(launch_cast_pad(e.frozen + e.v_patch_w, e.ws + e.patch_w16, e.Hv(), e.Kp(), e.Kpp(), e.px ? DT_PAIR : e.dt(), s));
    }
    if (which & 2)
        for (int l = 0; l < e.cfg.layers_txt; ++l)
            if (int r = refresh_layer(e, e.train, e.txt[l], e.txt_w16[l], true, s, e.cfg.hidden, e.cfg.inter)) return r;
    return 0;
}

int mmhip_forward(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                  const int64_t* tim_mask, int B, int T, int train, uint64_t seed, float* out_cls, float* logits_per_text,
                  float* out_tim, float* mm_features, void* stream) {
    if (!h || !h->ws) return MMHIP_E_STATE;
    if (!ids || !mask || B < 1 || T < 1) return MMHIP_E_INVALID;
    if ((tim_ids == nullptr) != (tim_mask == nullptr)) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    const bool imported = pixels == nullptr;           // image tower outputs come from the caller's cache (mmhip_vision_import)
    if (imported && e.vision_ready_B != B) return MMHIP_E_STATE;
    e.vision_ready_B = 0;
    if (B > e.cfg.max_posts || T > e.cfg.max_text_len) return MMHIP_E_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    e.B = B; e.T = T; e.itm = tim_ids != nullptr; e.Bt = e.itm ? 2 * B : B; e.train_mode = train != 0; e.seed = seed;
    e.fwd_done = false; e.bwd_begun = false;
    const size_t nb = (size_t)B * T * 8;
    // token ids are clamped into the word table on their way into the engine's copy (launch_copy_ids_clamped: why)
    CHECK_HIP(launch_copy_ids_clamped(ids, e.wsp<int64_t>(e.ids_all), (size_t)B * T, e.cfg.vocab, e.bad_index, s));
    CHECK_HIP(hipMemcpyAsync(e.ws + e.mask_all, mask, nb, hipMemcpyDeviceToDevice, s));
    if (e.itm) {
        CHECK_HIP(launch_copy_ids_clamped(tim_ids, e.wsp<int64_t>(e.ids_all) + (size_t)B * T, (size_t)B * T, e.cfg.vocab, e.bad_index, s));
        CHECK_HIP(hipMemcpyAsync(e.ws + e.mask_all + nb, tim_mask, nb, hipMemcpyDeviceToDevice, s));
    }
    if (int r = side_init(e)) return r;
    if (e.part[0] < 0) {
        // MMHIP_PART: "txt,vit" forces a split, "0" turns the partition off; default: chosen per forward (choose_partition)
        e.part[0] = e.part[1] = 0; e.part_auto = true;
        if (const char* v = getenv("MMHIP_PART")) {
            int a = 0, b = 0;
            e.part_auto = false;
            if (sscanf(v, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { e.part[0] = a; e.part[1] = b; }
        }
    }
    const bool can_part = !imported && use_side(e) && !lockstep_ok(e) && (e.dt() == DT_BF16 || e.dt() == DT_F16);
    e.cur_part[0] = e.cur_part[1] = 0;
    if (can_part && e.part_auto) choose_partition(e);
    else if (can_part && e.part[0] > 0) { e.cur_part[0] = e.part[0]; e.cur_part[1] = e.part[1]; }
    if (!imported && lockstep_ok(e)) {
        e.vit_is_long = false;
        if (int r = towers_forward_lockstep(e, pixels, s)) return r;
    } else if (use_side(e)) {
        // the frozen image tower does not depend on the text tower: run it on the side stream, join before the heads
        CHECK_HIP(hipEventRecord(e.ev_fork, s));
        if (int r = e.span(0, s)) return r;
        static int force = -2;
        if (force == -2) { const char* v = getenv("MMHIP_VIT_PRIO"); force = v ? atoi(v) : -1; }
        const int P = (e.cfg.image / e.cfg.patch) * (e.cfg.image / e.cfg.patch) + 1;
        const bool vit_longer = (double)e.B * P * e.cfg.layers_img > (double)e.Bt * e.T * e.cfg.layers_txt;     // rows x layers of equal width
        e.vit_is_long = vit_longer;
        hipStream_t sv = e.side_vit[force >= 0 ? (force ? 1 : 0) : (vit_longer ? 1 : 0)];
        if (!imported) {
            CHECK_HIP(hipStreamWaitEvent(sv, e.ev_fork, 0));
            if (int r = vit_forward(e, pixels, sv)) return r;
            if (int r = e.span(1, sv)) return r;
            CHECK_HIP(hipEventRecord(e.ev_vit, sv));
        }
        if (int r = text_forward(e, s)) return r;
        if (int r = e.span(2, s)) return r;
        if (!imported) CHECK_HIP(hipStreamWaitEvent(s, e.ev_vit, 0));
    } else {
        if (int r = e.span(0, s)) return r;
        if (!imported) if (int r = vit_forward(e, pixels, s)) return r;
        if (int r = e.span(1, s)) return r;
        if (int r = text_forward(e, s)) return r;
        if (int r = e.span(2, s)) return r;
    }
    if (int r = heads_forward(e, out_cls, logits_per_text, out_tim, mm_features, s)) return r;
    if (int r = e.span(3, s)) return r;
    e.fwd_done = true;
    return 0;
}

// ---- image-tower output cache (the tower is frozen and dropout-free: its output is a pure function of the pixels)
namespace {
__global__ __launch_bounds__(256) void vision_copy_kernel(int to_cache, const int64_t* __restrict__ slots, char* cache, uint64_t rec_bytes,
                                                          uint64_t cache_records, char* v_out, char* vpool, uint32_t tok_bytes, uint32_t pool_bytes) {
    const int64_t slot = slots[blockIdx.y];
    if (slot < 0 || (uint64_t)slot >= cache_records) return;
    char* rec = cache + (uint64_t)slot * rec_bytes;
    char* tok = v_out + (size_t)blockIdx.y * tok_bytes;
    char* pool = vpool + (size_t)blockIdx.y * pool_bytes;
    const uint32_t n16 = (tok_bytes + pool_bytes) / 16, t16 = tok_bytes / 16;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        uint4* live = reinterpret_cast<uint4*>(i < t16 ? tok + (size_t)i * 16 : pool + (size_t)(i - t16) * 16);
        uint4* kept = reinterpret_cast<uint4*>(rec + (size_t)i * 16);
        if (to_cache) *kept = *live;
        else *live = *kept;
    }
}
int vision_copy(mmhip_engine& e, int to_cache, const int64_t* slots, void* cache, uint64_t cache_records, int B, hipStream_t s) {
    const int P = e.P(), H = e.Hv();
    const uint32_t tok_bytes = (uint32_t)(P * H * e.esz()), pool_bytes = (uint32_t)H * 4;
    if (tok_bytes % 16 || pool_bytes % 16) return MMHIP_E_INVALID;
    hipLaunchKernelGGL(vision_copy_kernel, dim3(16, B), dim3(256), 0, s, to_cache, slots, (char*)cache, mmhip_vision_record_bytes(&e), cache_records,
                       e.ws + e.v_out, (char*)e.wsp<float>(e.h_vpool), tok_bytes, pool_bytes);
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : (int)err;
}
}  // namespace
uint64_t mmhip_vision_record_bytes(mmhip_handle h) {
    if (!h) return 0;
    const uint64_t P = (uint64_t)(h->cfg.image / h->cfg.patch) * (h->cfg.image / h->cfg.patch) + 1;
    return (P * h->Hv() * h->esz() + (uint64_t)h->Hv() * 4 + 255) & ~255ull;
}
int mmhip_vision_export(mmhip_handle h, const int64_t* slots, void* cache, uint64_t cache_records, void* stream) {
    if (!h || !h->ws || !h->fwd_done) return MMHIP_E_STATE;
    if (!slots || !cache || ((uintptr_t)cache & 15)) return MMHIP_E_INVALID;
    return vision_copy(*h, 1, slots, cache, cache_records, h->B, (hipStream_t)stream);
}
int mmhip_vision_import(mmhip_handle h, const int64_t* slots, const void* cache, uint64_t cache_records, int B, void* stream) {
    if (!h || !h->ws) return MMHIP_E_STATE;
    if (!slots || !cache || ((uintptr_t)cache & 15) || B < 1) return MMHIP_E_INVALID;
    if (B > h->cfg.max_posts) return MMHIP_E_CAPACITY;
    if (int r = vision_copy(*h, 0, slots, const_cast<void*>(cache), cache_records, B, (hipStream_t)stream)) return r;
    h->vision_ready_B = B;
    return 0;
}

int mmhip_loss(mmhip_handle h, const int64_t* onehot, const float* class_w, const int64_t* lbl_tim, float w_cls, float w_itc,
               float w_itm, float* loss, int* n_correct, void* stream) {
    if (!h || !h->fwd_done) return MMHIP_E_STATE;
    if (!onehot) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    if (w_itm != 0.f && (!e.itm || !lbl_tim)) return MMHIP_E_INVALID;
    if (w_itc != 0.f && !e.itc_done) return MMHIP_E_STATE;
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.out_cls = e.wsp<float>(e.h_out_cls); a.onehot = onehot; a.class_w = class_w;
    a.logits_per_text = w_itc != 0.f ? e.wsp<float>(e.h_logits) : nullptr;
    a.out_tim = w_itm != 0.f ? e.wsp<float>(e.h_out_tim) : nullptr; a.lbl_tim = lbl_tim;
    a.w_cls = w_cls; a.w_itc = w_itc; a.w_itm = w_itm;
    a.loss = e.wsp<float>(e.h_loss);
    a.d_out_cls = e.wsp<float>(e.h_d_out_cls);
    a.d_logits = w_itc != 0.f ? e.wsp<float>(e.h_d_logits) : nullptr;
    a.d_out_tim = w_itm != 0.f ? e.wsp<float>(e.h_d_out_tim) : nullptr;
    a.n_correct = n_correct;
    a.B = e.B; a.C = e.cfg.num_labels;
    hipStream_t s = (hipStream_t)stream;
    CHECK_HIP(launch_loss(a, s));
    if (loss) CHECK_HIP(hipMemcpyAsync(loss, a.loss, 16, hipMemcpyDeviceToDevice, s));
    e.bd_out_cls = a.d_out_cls; e.bd_logits = a.d_logits; e.bd_out_tim = a.d_out_tim; e.bd_feats = nullptr;
    return 0;
}

int mmhip_num_backward_stages(mmhip_handle h) { return h ? h->cfg.layers_txt + 2 : MMHIP_E_INVALID; }

int mmhip_stage_grad_range(mmhip_handle h, int stage, uint64_t* begin, uint64_t* end) {
    if (!h || !begin || !end || stage < 0 || stage > h->cfg.layers_txt + 1) return MMHIP_E_INVALID;
    const mmhip_engine& e = *h;
    if (stage == 0) { *begin = e.heads_begin; *end = e.heads_end; }
    else if (stage <= e.cfg.layers_txt) { const LayerOff& o = e.txt[e.cfg.layers_txt - stage]; *begin = o.begin; *end = o.end; }
    else { *begin = e.emb_begin; *end = e.emb_end; }
    return 0;
}

int mmhip_backward_begin(mmhip_handle h, const float* d_out_cls, const float* d_logits, const float* d_out_tim, const float* d_feats, void* stream) {
    if (!h || !h->fwd_done || !h->grad) return MMHIP_E_STATE;
    mmhip_engine& e = *h;
    if (unsigned* f = guard_flag(e)) CHECK_HIP(hipMemsetAsync(f, 0, 4, (hipStream_t)stream));      // the step guard's flag belongs to one backward
    if (d_out_cls || d_logits || d_out_tim || d_feats) {
        if (!d_out_cls) return MMHIP_E_INVALID;
        e.bd_out_cls = d_out_cls; e.bd_logits = d_logits; e.bd_out_tim = d_out_tim; e.bd_feats = d_feats;
    } else if (!e.bd_out_cls) {
        return MMHIP_E_STATE;      // no mmhip_loss before, and no explicit gradients
    }
    e.bwd_begun = true;
    e.tn_pending[0] = e.tn_pending[1] = false;
    return side_init(e);
}
int mmhip_backward_stage(mmhip_handle h, int stage, void* stream) {
    if (!h || !h->bwd_begun) return MMHIP_E_STATE;
    mmhip_engine& e = *h;
    hipStream_t s = (hipStream_t)stream;
    const int L = e.cfg.layers_txt;
    if (stage == 0) return heads_backward(e, s);
    if (stage >= 1 && stage <= L) return text_layer_backward(e, L - stage, s);
    if (stage == L + 1) return embed_backward(e, s);
    return MMHIP_E_INVALID;
}
int mmhip_backward_join_stage(mmhip_handle h, int stage, void* stream) {
    if (!h || !h->bwd_begun) return MMHIP_E_STATE;
    mmhip_engine& e = *h;
    const int L = e.cfg.layers_txt;
    if (stage < 1 || stage > L) return 0;
    const int set = (L - stage) & 1;
    if (e.tn_pending[set]) {
        CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, e.ev_tn[set], 0));
        e.tn_pending[set] = false;
    }
    return 0;
}
int mmhip_backward_finish(mmhip_handle h, void* stream) {
    if (!h || !h->bwd_begun) return MMHIP_E_STATE;
    return backward_finish(*h, (hipStream_t)stream);
}
int mmhip_backward(mmhip_handle h, const float* d_out_cls, const float* d_logits, const float* d_out_tim, const float* d_feats, void* stream) {
    if (int r = mmhip_backward_begin(h, d_out_cls, d_logits, d_out_tim, d_feats, stream)) return r;
    const int n = mmhip_num_backward_stages(h);
    for (int st = 0; st < n; ++st)
        if (int r = mmhip_backward_stage(h, st, stream)) return r;
    return mmhip_backward_finish(h, stream);
}

int mmhip_set_nonfinite_counter(uint32_t* device_counter) {
    if ((uintptr_t)device_counter & 3) return MMHIP_E_INVALID;
    g_nonfinite = device_counter;
    g_skip = nullptr;
    return 0;
}
int mmhip_set_step_guard(uint32_t* device_words2) {
    if ((uintptr_t)device_words2 & 3) return MMHIP_E_INVALID;
    g_nonfinite = device_words2;
    g_skip = device_words2 ? device_words2 + 1 : nullptr;
    return 0;
}
int mmhip_set_guard(mmhip_handle h, uint32_t* device_words2) {
    if (!h || ((uintptr_t)device_words2 & 3)) return MMHIP_E_INVALID;
    h->guard = device_words2;
    return 0;
}
void* mmhip_side_stream(mmhip_handle h) {
    if (!h) return nullptr;
    if (side_init(*h)) return nullptr;
    return use_side(*h) ? (void*)h->side : nullptr;
}
int mmhip_set_index_counter(mmhip_handle h, uint32_t* device_word) {
    if (!h || ((uintptr_t)device_word & 3)) return MMHIP_E_INVALID;
    h->bad_index = device_word;
    return 0;
}
int mmhip_set_backward_products(mmhip_handle h, int products) {
    if (!h || products < 1 || products > 3) return MMHIP_E_INVALID;
    if (h->cfg.dtype != MMHIP_BF16X3 || !h->px) return products == 3 ? 0 : MMHIP_E_STATE;      // the 16-bit modes have one product; round 3's copy form has three
    h->bwd_np = products;
    return 0;
}
int mmhip_set_loss_scale(mmhip_handle h, float loss_scale) {
    if (!h || !(loss_scale >= 0.f)) return MMHIP_E_INVALID;
    h->cfg.loss_scale = loss_scale;
    return 0;
}

static int adamw_impl(float* p, float* g, float* m, float* v, uint64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float grad_scale, int zero_grad, void* stream, unsigned* g_nonfinite, const unsigned* g_skip) {
    if (!p || !g || !m || !v || step < 1) return MMHIP_E_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MMHIP_E_INVALID;
    AdamWArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    a.zero_grad = zero_grad; a.grad_scale = grad_scale; a.nonfinite = g_nonfinite; a.skip = g_skip;
    CHECK_HIP(launch_adamw(a, (hipStream_t)stream));
    return 0;
}
int mmhip_adamw(float* p, float* g, float* m, float* v, uint64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
    return adamw_impl(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, stream, g_nonfinite, g_skip);
}
int mmhip_adamw_guarded(float* p, float* g, float* m, float* v, uint64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int zero_grad, void* stream, uint32_t* guard_words2) {
    if ((uintptr_t)guard_words2 & 3) return MMHIP_E_INVALID;
    return adamw_impl(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, stream, guard_words2, guard_words2 ? guard_words2 + 1 : nullptr);
}

int mmhip_set_row_state(mmhip_handle h, uint8_t* row_state) {
    if (!h) return MMHIP_E_INVALID;
    if ((uintptr_t)row_state & 3) return MMHIP_E_INVALID;
    h->word_row_state = row_state;
    return 0;
}

static int adamw_rows_impl(float* p, float* g, float* m, float* v, int rows, int width, uint8_t* row_state, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream, unsigned* g_nonfinite, const unsigned* g_skip) {
    if (!p || !g || !m || !v || !row_state || step < 1 || rows < 0 || width <= 0 || width % 4) return MMHIP_E_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MMHIP_E_INVALID;
    AdamWArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = (size_t)rows * width; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    a.zero_grad = zero_grad; a.grad_scale = grad_scale; a.nonfinite = g_nonfinite; a.skip = g_skip;
    CHECK_HIP(launch_adamw_rows(a, rows, width, row_state, (hipStream_t)stream));
    return 0;
}
int mmhip_adamw_rows(float* p, float* g, float* m, float* v, int rows, int width, uint8_t* row_state, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
    return adamw_rows_impl(p, g, m, v, rows, width, row_state, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, stream, g_nonfinite, g_skip);
}
int mmhip_adamw_rows_guarded(float* p, float* g, float* m, float* v, int rows, int width, uint8_t* row_state, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream, uint32_t* guard_words2) {
    if ((uintptr_t)guard_words2 & 3) return MMHIP_E_INVALID;
    return adamw_rows_impl(p, g, m, v, rows, width, row_state, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, stream, guard_words2,
                           guard_words2 ? guard_words2 + 1 : nullptr);
}

// One fused training step, enqueued natively (no per-stage host round trips): forward (train mode) -> loss -> backward ->
// AdamW over the ranges that receive gradients for this flag set (torch skips `grad is None` tensors) ->
// 16-bit weight refresh.  Single-rank form of MMLate_Model.train_step; a data-parallel caller keeps the staged calls.
static int train_step_impl(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                           const int64_t* tim_mask, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T,
                           uint64_t seed, int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* loss,
                           int* n_correct, void* stream, mmhip_exchange_cb cb, void* user) {
    if (!h || !h->ws || !h->grad) return MMHIP_E_STATE;
    if (!adam_m || !adam_v || !onehot || step < 1) return MMHIP_E_INVALID;
    if (use_itm && (!tim_ids || !lbl_tim)) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    e.skip_itc = !use_itc;
    const int rf = mmhip_forward(h, ids, mask, pixels, use_itm ? tim_ids : nullptr, use_itm ? tim_mask : nullptr, B, T, 1, seed, nullptr, nullptr,
                                 nullptr, nullptr, stream);
    e.skip_itc = false;
    if (rf) return rf;
    if (int r = mmhip_loss(h, onehot, class_w, use_itm ? lbl_tim : nullptr, w_cls, use_itc ? w_itc : 0.f, use_itm ? w_itm : 0.f, loss, n_correct, stream)) return r;
    // Backward.  With the side stream on, each text layer's AdamW and 16-bit weight refresh follow its weight-gradient GEMM on the
    // SIDE stream, beside the activation-gradient chain of the layers below (an HBM-bound kernel next to MFMA-bound ones) instead
    // of after the whole backward; the layer's fp32 LayerNorm weights and transposed 16-bit copies are read by its own backward
    // kernels on the caller's stream, so the side stream first waits for the event recorded behind them.
    hipStream_t s = (hipStream_t)stream;
    const char* early_env = getenv("MMHIP_EARLY_ADAMW");          // read per step: tests compare both orders in one process
    // f16 with the step guard armed: a void step is only known when the backward has reached the embeddings, so every AdamW waits for it
    unsigned* gc = guard_counter(e);
    unsigned* gf = guard_flag(e);
    const int early = (e.dt() == DT_F16 && gf) ? 0 : (early_env ? atoi(early_env) : 1);
    if (e.dt() == DT_F16 && gf && !early_env) {
        static bool told = false;
        if (!told && getenv("MMHIP_VERBOSE")) { fprintf(stderr, "[mmhip] f16 with the step guard armed: every AdamW launch follows the backward (no per-layer optimizer beside it)\n"); told = true; }
    }
    if (int r = mmhip_backward_begin(h, nullptr, nullptr, nullptr, nullptr, stream)) return r;
    const int L = e.cfg.layers_txt;
    // data-parallel form (cb): the callback is told about stage st-1 once stage st is enqueued and st-1's weight gradients are ordered in the
    // caller's stream -- its collective travels while the stages below compute.  The layer optimizers wait for the gradient exchange: PER BUCKET
    // (round 5).  When the callback answers MMHIP_CB_BUCKET -- "the collective that carries every stage since the last such answer has been
    // started" -- the side stream is made to wait for that collective (on_stage(MMHIP_CB_WAIT_BUCKET): the caller waits on mmhip_side_stream) and
    // the AdamW + operand refresh of the text layers it carries follow on the side stream, beside the stages below, as in the single-rank step;
    // before round 5 every layer's optimizer sat behind ONE barrier after the last all-reduce (MMHIP_CB_WAIT_DENSE), which made the N > 1 step
    // longer than the N = 1 step before any wire time.  Ranges that are not text layers (heads, embeddings) and the last, unflushed bucket keep
    // that barrier.
    const bool layer_opt = early && use_side(e) && !cb;
    const bool bucket_opt = early && use_side(e) && cb;
    bool opt_pending = false, dense_by_caller = false;
    std::vector<char> layer_done((size_t)(L > 0 ? L : 1), 0);
    std::vector<int> open_layers;          // text layers whose gradient stage lies in the bucket that is still open
    for (int st = 0; st < L + 2; ++st) {
        if (int r = mmhip_backward_stage(h, st, stream)) return r;
        if (cb && st >= 1) {
            if (int r = mmhip_backward_join_stage(h, st - 1, stream)) return r;
            const int r = cb(user, st - 1);
            if (r != 0 && r != MMHIP_CB_BUCKET) return r;
            if (st - 1 >= 1 && st - 1 <= L) open_layers.push_back(L - (st - 1));
            if (r == MMHIP_CB_BUCKET) {
                if (bucket_opt && !open_layers.empty()) {
                    // the side stream waits for the layers' backward kernels on `s` (they read the fp32 LayerNorm weights and the transposed
                    // operand copies that the optimizer and the refresh are about to overwrite) and, through the callback, for the collective
                    CHECK_HIP(hipEventRecord(e.ev_layer[0], s));
                    CHECK_HIP(hipStreamWaitEvent(e.side, e.ev_layer[0], 0));
                    const int w = cb(user, MMHIP_CB_WAIT_BUCKET);
                    if (w != 0 && w != MMHIP_CB_HANDLED) return w;
                    for (int l : open_layers) {
                        const LayerOff& o = e.txt[l];
                        if (w == 0)
                            if (int r2 = adamw_impl(e.train + o.begin, e.grad + o.begin, adam_m + o.begin, adam_v + o.begin, o.end - o.begin, lr, beta1, beta2, eps,
                                                    weight_decay, step, grad_scale, 1, e.side, gc, gf)) return r2;
                        if (int r2 = refresh_layer(e, e.train, o, e.txt_w16[l], true, e.side, e.cfg.hidden, e.cfg.inter)) return r2;
                        layer_done[l] = w == 0 ? 1 : 2;          // 2: stepped by the caller (MMHIP_CB_HANDLED), refreshed here
                    }
                    CHECK_HIP(hipEventRecord(e.ev_opt, e.side));
                    opt_pending = true;
                }
                open_layers.clear();
            }
        }
        if (layer_opt && st >= 1 && st <= L) {
            const int l = L - st, set = l & 1;
            const LayerOff& o = e.txt[l];
            CHECK_HIP(hipEventRecord(e.ev_layer[set], s));
            CHECK_HIP(hipStreamWaitEvent(e.side, e.ev_layer[set], 0));
            if (int r = adamw_impl(e.train + o.begin, e.grad + o.begin, adam_m + o.begin, adam_v + o.begin, o.end - o.begin, lr, beta1, beta2, eps,
                                   weight_decay, step, grad_scale, 1, e.side, gc, gf)) return r;
            if (int r = refresh_layer(e, e.train, o, e.txt_w16[l], true, e.side, e.cfg.hidden, e.cfg.inter)) return r;
            layer_done[l] = 1;
            CHECK_HIP(hipEventRecord(e.ev_opt, e.side));
            opt_pending = true;
        }
    }
    if (int r = mmhip_backward_finish(h, stream)) return r;
    if (opt_pending) CHECK_HIP(hipStreamWaitEvent(s, e.ev_opt, 0));
    if (cb) {
        if (int r = cb(user, L + 1)) return r;                    // embedding stage: dense part + the row-sparse word-table exchange start
        // the dense exchanges are ordered before what follows in `stream`; MMHIP_CB_HANDLED: the caller also ran the dense optimizer itself
        // (reduce-scatter -> AdamW on its shard -> all-gather of the parameters: dist.ShardedBuckets) -- the dense AdamW launches below are skipped
        const int r = cb(user, MMHIP_CB_WAIT_DENSE);
        if (r == MMHIP_CB_HANDLED) dense_by_caller = true;
        else if (r) return r;
    }
    if (int r = e.span(4, s)) return r;
    // merged [begin, end) ranges of the active gradient groups, in address order (text layers already stepped: skipped)
    bool act[6] = {false, use_itc != 0, use_itm != 0, e.cfg.fusion == MMHIP_FUSION_ATTENTION, true, false};
    const uint64_t w0 = e.t_word, V = (uint64_t)e.cfg.vocab, H = (uint64_t)e.cfg.hidden;
    uint64_t rb = 0, re = 0;
    bool open = false;
    bool rows_due = false;
    auto flush = [&]() -> int {
        if (!open || re <= rb) return 0;
        const uint64_t dense_end = re < w0 ? re : w0;
        if (dense_end > rb && !dense_by_caller)
            if (int r = adamw_impl(e.train + rb, e.grad + rb, adam_m + rb, adam_v + rb, dense_end - rb, lr, beta1, beta2, eps, weight_decay, step,
                                   grad_scale, 1, stream, gc, gf)) return r;
        if (re > w0) rows_due = true;          // the word table goes last: under data parallelism its rows are still travelling
        return 0;
    };
    auto in_layer = [&](uint64_t off) {          // a text layer whose optimizer has already run (beside the backward)
        for (int l = 0; l < L; ++l) if (layer_done[l] && off >= e.txt[l].begin && off < e.txt[l].end) return true;
        return false;
    };
    for (const auto& p : e.params) {
        if (p.buffer != 1 || !act[p.group] || in_layer(p.offset)) continue;
        const uint64_t b = p.offset, en = p.offset + ((p.numel + 3) & ~(uint64_t)3);
        if (open && b == re) { re = en; continue; }
        if (int r = flush()) return r;
        rb = b; re = en; open = true;
    }
    if (int r = flush()) return r;
    for (int l = 0; l < L; ++l)          // 16-bit GEMM operand copies of the layers stepped just now: no word-table dependence
        if (!layer_done[l]) if (int r = refresh_layer(e, e.train, e.txt[l], e.txt_w16[l], true, s, e.cfg.hidden, e.cfg.inter)) return r;
    if (rows_due) {
        if (cb) if (int r = cb(user, MMHIP_CB_FINISH_ROWS)) return r;                 // the exchanged word rows are summed into the gradient
        if (!e.word_row_state) return MMHIP_E_STATE;
        if (int r = adamw_rows_impl(e.train + w0, e.grad + w0, adam_m + w0, adam_v + w0, (int)V, (int)H, e.word_row_state, lr, beta1, beta2, eps,
                                    weight_decay, step, grad_scale, 1, stream, gc, gf)) return r;
    }
    return e.span(5, s);
}

int mmhip_train_step(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                     const int64_t* tim_mask, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T,
                     uint64_t seed, int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* loss,
                     int* n_correct, void* stream) {
    return train_step_impl(h, ids, mask, pixels, tim_ids, tim_mask, lbl_tim, onehot, class_w, B, T, seed, use_itc, use_itm, w_cls, w_itc, w_itm,
                           adam_m, adam_v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, loss, n_correct, stream, nullptr, nullptr);
}
int mmhip_train_step_dp(mmhip_handle h, const int64_t* ids, const int64_t* mask, const float* pixels, const int64_t* tim_ids,
                        const int64_t* tim_mask, const int64_t* lbl_tim, const int64_t* onehot, const float* class_w, int B, int T,
                        uint64_t seed, int use_itc, int use_itm, float w_cls, float w_itc, float w_itm, float* adam_m, float* adam_v,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* loss,
                        int* n_correct, void* stream, mmhip_exchange_cb on_stage, void* user) {
    if (!on_stage) return MMHIP_E_INVALID;
    return train_step_impl(h, ids, mask, pixels, tim_ids, tim_mask, lbl_tim, onehot, class_w, B, T, seed, use_itc, use_itm, w_cls, w_itc, w_itm,
                           adam_m, adam_v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, loss, n_correct, stream, on_stage, user);
}

// phase ends of the last step as milliseconds after the forward's fork (hipEvents on the streams the phases run on; no profiler):
// ms[0] image tower end, [1] text tower end, [2] forward end (heads), [3] backward end (all gradients and layer optimizers joined),
// [4] step end.  enable != 0 arms the events for the following steps; a phase that was not recorded reads -1.  Synchronises.
int mmhip_step_spans(mmhip_handle h, int enable, float* ms) {
    if (!h) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    if (ms) {
        for (int i = 0; i < 5; ++i) ms[i] = -1.f;
        if (e.span_set[0]) {
            for (int i = 1; i < 6; ++i) {
                if (!e.span_set[i]) continue;
                CHECK_HIP(hipEventSynchronize(e.span_ev[i]));
                float t = 0.f;
                if (hipEventElapsedTime(&t, e.span_ev[0], e.span_ev[i]) == hipSuccess) ms[i - 1] = t;
            }
        }
    }
    e.spans_on = enable;
    if (!enable) for (int i = 0; i < 6; ++i) e.span_set[i] = false;
    return 0;
}
int mmhip_gemm_timing(mmhip_handle h, int enable, int reset, double* ms, uint64_t* launches, double* flops) {
    if (!h) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    if (ms || launches || flops) {
        double tms = 0, tf = 0;
        for (size_t i = 0; i < e.ev_used; ++i) {
            CHECK_HIP(hipEventSynchronize(e.evs[i].b));
            float t = 0;
            CHECK_HIP(hipEventElapsedTime(&t, e.evs[i].a, e.evs[i].b));
            tms += t; tf += e.evs[i].flops;
        }
        if (ms) *ms = tms;
        if (launches) *launches = e.ev_used;
        if (flops) *flops = tf;
    }
    if (reset) e.ev_used = 0;
    e.timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return 0;
}

// per-shape table of the launches timed since the last reset (text, one line per (M, N, K, epilogue flags))
int mmhip_gemm_timing_by_shape(mmhip_handle h, char* out, uint64_t capacity) {
    if (!h || !out || capacity < 2) return MMHIP_E_INVALID;
    mmhip_engine& e = *h;
    struct Row { int M, N, K, flags, tile, cus; int n; double ms, flops; };
    std::vector<Row> rows;
    for (size_t i = 0; i < e.ev_used; ++i) {
        const auto& ev = e.evs[i];
        CHECK_HIP(hipEventSynchronize(ev.b));
        float t = 0;
        CHECK_HIP(hipEventElapsedTime(&t, ev.a, ev.b));
        Row* r = nullptr;
        for (auto& x : rows) if (x.M == ev.M && x.N == ev.N && x.K == ev.K && x.flags == ev.flags && x.tile == ev.tile && x.cus == ev.cus) { r = &x; break; }
        if (!r) { rows.push_back(Row{ev.M, ev.N, ev.K, ev.flags, ev.tile, ev.cus, 0, 0.0, 0.0}); r = &rows.back(); }
        r->n++; r->ms += t; r->flops += ev.flops;
    }
    size_t pos = 0;
    auto put = [&](const char* fmt, auto... a) {
        if (pos + 1 >= capacity) return;
        int w = snprintf(out + pos, capacity - pos, fmt, a...);
        if (w > 0) pos += (size_t)w < capacity - pos ? (size_t)w : capacity - pos - 1;
    };
    // cus: the launch's workgroup cap under the forward's CU partition (256 = the whole chip); TF/CU-share: TFLOP/s scaled to a whole chip of such CUs
    put("%7s %6s %6s %6s %5s %4s %6s %9s %9s %8s %11s\n", "M", "N", "K", "flags", "tile", "cus", "n", "avg_us", "total_ms", "TFLOP/s", "TF/CU-share");
    for (const auto& r : rows) {
        const int cus = r.cus > 0 && r.cus < 256 ? r.cus : 256;
        const double tf = r.flops / (r.ms * 1e-3) * 1e-12;
        put("%7d %6d %6d %6d %5d %4d %6d %9.2f %9.3f %8.1f %11.1f\n", r.M, r.N, r.K, r.flags, r.tile, cus, r.n, 1e3 * r.ms / r.n, r.ms, tf, tf * 256.0 / cus);
    }
    out[pos < capacity ? pos : capacity - 1] = 0;
    return 0;
}

}  // extern "C"
