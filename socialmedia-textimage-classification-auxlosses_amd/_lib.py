"""ctypes binding of libmmhip.so (include/mmhip.h).  No fallback: if the library is missing or a call fails,
this raises -- the product path never routes around the HIP kernels."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMHIP_LIB_PATH") or os.path.join(HERE, "libmmhip.so")      # override: A/B of two builds on one box

BF16, F16, F32 = 0, 1, 2
BF16X3 = 2          # as an engine dtype: the strict-parity mode (fp32 activations, bf16x3 matrix products; include/mmhip.h)
TXT_BERT, TXT_XLMR = 0, 1
FUSION_CONCAT, FUSION_ATTENTION = 0, 1
IMG_VIT, IMG_CLIP = 0, 1
G_NEVER, G_ITC, G_ITM, G_FUSION_ATT, G_ALWAYS, G_FROZEN = range(6)


class Config(C.Structure):
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("layers_txt", C.c_int), ("layers_img", C.c_int),
                ("vocab", C.c_int), ("max_pos", C.c_int), ("type_vocab", C.c_int), ("txt_kind", C.c_int), ("pad_id", C.c_int),
                ("ln_eps_txt", C.c_float), ("ln_eps_img", C.c_float), ("image", C.c_int), ("patch", C.c_int),
                ("proj_dim", C.c_int), ("num_labels", C.c_int), ("fusion", C.c_int), ("p_hidden", C.c_float),
                ("p_attn", C.c_float), ("p_head", C.c_float), ("dtype", C.c_int), ("max_posts", C.c_int),
                ("max_text_len", C.c_int), ("loss_scale", C.c_float), ("img_kind", C.c_int), ("hidden_img", C.c_int),
                ("heads_img", C.c_int), ("inter_img", C.c_int)]


class EarlyConfig(C.Structure):      # include/mmhip.h: mmhip_early_config
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("l_layers", C.c_int), ("r_layers", C.c_int), ("x_layers", C.c_int),
                ("vocab", C.c_int), ("max_pos", C.c_int), ("type_vocab", C.c_int), ("feat_dim", C.c_int), ("pos_dim", C.c_int), ("num_labels", C.c_int),
                ("max_posts", C.c_int), ("max_text_len", C.c_int), ("max_boxes", C.c_int), ("dtype", C.c_int), ("p_hidden", C.c_float), ("p_attn", C.c_float),
                ("p_head", C.c_float), ("ln_eps", C.c_float)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 192), ("ndim", C.c_int), ("dims", C.c_int64 * 4), ("buffer", C.c_int), ("group", C.c_int),
                ("offset", C.c_uint64), ("numel", C.c_uint64)]


class TNProblem(C.Structure):        # include/mmhip.h: mmhip_tn_problem
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int32), ("Nn", C.c_int32), ("Nc", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("colsum", C.c_void_p)]


class CastMat(C.Structure):          # include/mmhip.h: mmhip_cast_mat
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class MMHipError(RuntimeError):
    pass


EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)      # include/mmhip.h: mmhip_exchange_cb
CB_WAIT_DENSE, CB_FINISH_ROWS, CB_WAIT_BUCKET = -1, -2, -3
CB_HANDLED = 1               # include/mmhip.h MMHIP_CB_HANDLED: the caller ran the dense optimizer itself
CB_BUCKET = 2                # include/mmhip.h MMHIP_CB_BUCKET: a collective carrying every stage since the last such answer has been started
_lib = None
P, I, F, U64, U32, I64P = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p

_SIGS = {
    "mmhip_create": (I, [C.POINTER(Config), C.POINTER(P)]),
    "mmhip_destroy": (None, [P]),
    "mmhip_param_count": (I, [P]),
    "mmhip_param_info_at": (I, [P, I, C.POINTER(ParamInfo)]),
    "mmhip_buffer_numel": (U64, [P, I]),
    "mmhip_workspace_bytes": (U64, [P]),
    "mmhip_bind": (I, [P, P, P, P, P, U64]),
    "mmhip_refresh_weights": (I, [P, I, P]),
    "mmhip_forward": (I, [P, P, P, P, P, P, I, I, I, U64, P, P, P, P, P]),
    "mmhip_loss": (I, [P, P, P, P, F, F, F, P, P, P]),
    "mmhip_vision_record_bytes": (U64, [P]),
    "mmhip_vision_export": (I, [P, P, P, U64, P]),
    "mmhip_vision_import": (I, [P, P, P, U64, I, P]),
    "mmhip_backward": (I, [P, P, P, P, P, P]),
    "mmhip_backward_begin": (I, [P, P, P, P, P, P]),
    "mmhip_backward_stage": (I, [P, I, P]),
    "mmhip_backward_finish": (I, [P, P]),
    "mmhip_backward_join_stage": (I, [P, I, P]),
    "mmhip_num_backward_stages": (I, [P]),
    "mmhip_stage_grad_range": (I, [P, I, C.POINTER(U64), C.POINTER(U64)]),
    "mmhip_adamw": (I, [P, P, P, P, U64, F, F, F, F, F, I, F, I, P]),
    "mmhip_adamw_rows": (I, [P, P, P, P, I, I, P, F, F, F, F, F, I, F, I, P]),
    "mmhip_set_row_state": (I, [P, P]),
    "mmhip_set_nonfinite_counter": (I, [P]),
    "mmhip_set_step_guard": (I, [P]),
    "mmhip_set_guard": (I, [P, P]),
    "mmhip_adamw_guarded": (I, [P, P, P, P, U64, F, F, F, F, F, I, F, I, P, P]),
    "mmhip_adamw_rows_guarded": (I, [P, P, P, P, I, I, P, F, F, F, F, F, I, F, I, P, P]),
    "mmhip_set_loss_scale": (I, [P, F]),
    "mmhip_set_backward_products": (I, [P, I]),
    "mmhip_side_stream": (P, [P]),
    "mmhip_set_index_counter": (I, [P, P]),
    "mmhip_early_set_index_counter": (I, [P, P]),
    "mmhip_early_gemm_timing": (I, [P, I, I, C.POINTER(C.c_double), C.POINTER(U64), C.POINTER(C.c_double)]),
    "mmhip_train_step": (I, [P, P, P, P, P, P, P, P, P, I, I, U64, I, I, F, F, F, P, P, F, F, F, F, F, I, F, P, P, P]),
    "mmhip_train_step_dp": (I, [P, P, P, P, P, P, P, P, P, I, I, U64, I, I, F, F, F, P, P, F, F, F, F, F, I, F, P, P, P, P, P]),
    "mmhip_image_plan_words": (U64, [I, P, P, I]),
    "mmhip_image_plan_build": (I, [I, P, P, P, I, P, U64]),
    "mmhip_image_plan_tmp_bytes": (U64, [P]),
    "mmhip_image_preprocess": (I, [P, P, P, P, P, P, P, P]),
    "mmhip_step_spans": (I, [P, I, C.POINTER(C.c_float)]),
    "mmhip_gemm_timing": (I, [P, I, I, C.POINTER(C.c_double), C.POINTER(U64), C.POINTER(C.c_double)]),
    "mmhip_gemm_timing_by_shape": (I, [P, C.c_char_p, U64]),
    "mmhip_op_gemm_nt": (I, [I, P, I, P, I, P, I, I, I, I, P, I, P, I, P, I, F, U64, U32, P, I, I, I, P]),
    "mmhip_op_gemm_tn": (I, [I, P, I, P, I, P, I, I, I, I, I, I, P, P]),
    "mmhip_op_gemm_tn_group": (I, [I, P, I, I, P]),
    "mmhip_op_cast_group": (I, [I, P, I, P]),
    "mmhip_op_self_att_block_fwd": (I, [I, P, P, P, P, P, P, P, P, F, I, I, I, F, F, U64, P, P, P, P, P, P, P, P]),
    "mmhip_op_cross_att_block_fwd": (I, [I, P, P, P, P, P, P, P, P, P, F, I, I, I, I, F, F, U64, P, P, P, P, P, P, P, P, P, P, P]),
    "mmhip_op_cross_att_block_bwd": (I, [I, P, P, P, P, P, I, I, I, I, F, F, U64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mmhip_op_self_att_block_bwd": (I, [I, P, P, P, P, P, I, I, I, F, F, U64, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mmhip_op_ffn_block_fwd": (I, [I, P, P, P, P, P, P, P, F, I, I, I, F, U64, P, P, P, P, P, P, P]),
    "mmhip_op_ffn_block_bwd": (I, [I, P, P, P, P, I, I, I, F, U64, P, P, P, P, P, P, P, P, P, P, P]),
    "mmhip_op_layernorm_fwd": (I, [I, P, P, P, P, P, P, I, I, F, P]),
    "mmhip_op_layernorm_bwd": (I, [I, P, P, P, P, P, P, P, P, P, I, I, P]),
    "mmhip_op_attn_fwd": (I, [I, P, P, P, P, I, I, I, F, U64, U32, P]),
    "mmhip_op_attn_bwd": (I, [I, P, P, P, P, P, P, I, I, I, F, U64, U32, P]),
    "mmhip_op_colsum": (I, [I, P, I, I, I, P, P]),
    "mmhip_op_cast": (I, [I, P, P, U64, I, I, P]),
    "mmhip_op_probe_layouts": (I, [P, P]),
    "mmhip_early_create": (I, [C.POINTER(EarlyConfig), C.POINTER(P)]),
    "mmhip_early_destroy": (None, [P]),
    "mmhip_early_param_count": (I, [P]),
    "mmhip_early_param_info_at": (I, [P, I, C.POINTER(ParamInfo)]),
    "mmhip_early_numel": (U64, [P]),
    "mmhip_early_workspace_bytes": (U64, [P]),
    "mmhip_early_bind": (I, [P, P, P, P, U64, P]),
    "mmhip_early_refresh_weights": (I, [P, P]),
    "mmhip_early_forward": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, U64, P, P, P, P, P]),
    "mmhip_early_loss": (I, [P, P, P, P, F, F, F, P, P, P]),
    "mmhip_early_backward": (I, [P, P, P, P, P]),
    "mmhip_early_num_stages": (I, [P]),
    "mmhip_early_stage_grad_range": (I, [P, I, C.POINTER(U64), C.POINTER(U64)]),
    "mmhip_early_train_step": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, U64, I, I, F, F, F, P, P, F, F, F, F, F, I, F, P, P, P, P]),
    "mmhip_version": (C.c_char_p, []),
}
EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library; raises MMHipError when it has not been built (python -m smtc_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc, gfx950). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "invalid state / call order", -3: "capacity exceeded"}.get(rc, f"hipError_t {rc}" if rc > 0 else "error")
        raise MMHipError(f"mmhip {what} failed: {kind} ({rc})")


def ptr(t):
    """device (or host) pointer of a torch tensor, or NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
