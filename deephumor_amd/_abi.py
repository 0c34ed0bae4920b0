"""The C-ABI as ctypes sees it: the plain structs and the argument list of every entry point of ``include/deephumor_hip.h``, line by line
(``tests/test_abi_cpu.py`` checks header <-> library <-> this table).  ``deephumor_amd.hip`` binds them."""
import ctypes

import torch

F32, BF16, BF16_OUT_F32, F16, F16_OUT_F32 = 0, 1, 2, 3, 4
ABI_VERSION = 31
HALF_DTYPES = (torch.bfloat16, torch.float16)       # the two 16-bit storage / MFMA operand types
ERR_ALL_FILTERED, ERR_OVERFLOW, ERR_TOO_FEW, ERR_NONFINITE = 1, 2, 4, 8
MAX_BEAMS = 64

_c = ctypes
_P, _I, _F, _U64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_uint64

class TrLayer(_c.Structure):
    _fields_ = ([(n, _P) for n in ("wqkv", "wo", "w1", "w2", "wq", "weo", "bqkv", "bo", "b1", "b2", "bq", "beo",
                                   "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b")]
                + [(n, _F) for n in ("ln1_eps", "ln2_eps", "ln3_eps", "sa_scale", "ea_scale")] + [("_pad", _I)]
                + [(n, _P) for n in ("kcache", "vcache", "kv")]
                + [(n, _P) for n in ("wqkv_f", "wq_f", "w1_f", "bqkv_f", "bq_f", "b1_f", "cs_qkv", "cs_q", "cs_1", "kp", "vt")]
                + [("kp_dperm", _I), ("_pad2", _I)]
                + [(n, _P) for n in ("wqkv_pk", "wo_pk", "weo_pk", "w1_pk", "w2_pk", "wq_pk")]
                + [(n, _P) for n in ("wqkv_x", "wo_x", "w1_x", "w2_x", "wq_x", "weo_x")]
                + [(n, _P) for n in ("wqkv_xp", "wo_xp", "w1_xp", "w2_xp", "wq_xp", "weo_xp")])


class TrModel(_c.Structure):
    _fields_ = ([(n, _I) for n in ("n_layers", "D", "n_heads", "pf_dim", "V", "pad_index", "cross", "S", "dtype")]
                + [("emb_scale", _F), ("layers", _c.POINTER(TrLayer))]
                + [(n, _P) for n in ("tok_emb", "pos_emb", "cls_w", "cls_b", "keymask", "cls_w_pk", "cls_b_pad", "cls_w_x", "layers_table", "layers_sync")])


class TrScratch(_c.Structure):
    _fields_ = [(n, _P) for n in ("x", "qkv", "att", "o", "q", "ff", "y2", "st0", "st1", "st2", "xp", "attp", "ffp")]


class LnFold(_c.Structure):
    """``dh_ln_fold_t``: deferred-LayerNorm options of ``dh_linear_ln``."""
    _fields_ = [("a_stats", _P), ("a_tiles", _I), ("a_eps", _F), ("a_colsum", _P),
                ("r_stats", _P), ("r_tiles", _I), ("r_eps", _F), ("r_gamma", _P), ("r_beta", _P), ("o_stats", _P)]


class LstmLayer(_c.Structure):
    _fields_ = [("w", _P), ("b", _P), ("w_il", _P), ("b_il", _P), ("w_pk", _P), ("w_x", _P), ("w_xp", _P)]


class LstmModel(_c.Structure):
    _fields_ = ([(n, _I) for n in ("n_layers", "E", "Hh", "V", "dtype", "_pad")] + [("layers", _c.POINTER(LstmLayer))]
                + [(n, _P) for n in ("emb", "cls_w", "cls_b", "h", "c", "h_alt", "c_alt", "cls_w_pk", "cls_b_pad", "cls_w_x")])


class LstmScratch(_c.Structure):
    _fields_ = [(n, _P) for n in ("xcat0", "xcatl", "c_cur", "gates", "hout", "topp", "xcat0p", "xcatlp")]


# name -> argtypes, mirrors include/deephumor_hip.h line by line
SIGNATURES = {
    "dh_abi_version": [],
    "dh_conv2d_bn_act": [_P, _P, _P, _P, _P, _P] + [_I] * 11 + [_P],
    "dh_stem_conv_nhwc": [_P, _P, _P, _P, _P] + [_I] * 10 + [_P],
    "dh_pack_nchw_to_nhwc8": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dh_round16_keep_nonzero": [_P, _P, _c.c_longlong, _I, _P],
    "dh_conv3x3_direct_supported": [_I, _I, _I, _I],
    "dh_conv3x3_direct_nhwc": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dh_bottleneck_tail_nhwc": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_bottleneck_tail_s3_supported": [_I, _I, _I],
    "dh_bottleneck_tail_s1_supported": [_I, _I, _I, _I],
    "dh_bottleneck_tail_s1_nhwc": [_P] * 13 + [_I] * 6 + [_P],
    "dh_bottleneck_tail_s2_supported": [_I, _I, _I],
    "dh_bottleneck_tail_s2_nhwc": [_P] * 13 + [_I] * 6 + [_P],
    "dh_conv3x3_s4_supported": [_I, _I, _I],
    "dh_conv3x3_s4_nhwc": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_bottleneck_tail_s3_nhwc": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_pack_mfma_fragments": [_P, _P, _I, _I, _P],
    "dh_conv1x1_wreg_supported": [_c.c_longlong, _I, _I],
    "dh_conv1x1_wreg_nhwc": [_P, _P, _P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _P],
    "dh_stem_conv7_bn_relu_maxpool": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dh_conv1x1_dual_nhwc": [_P, _P, _P, _P, _P] + [_I] * 11 + [_P],
    "dh_conv1x1_dual_wreg_supported": [_I] * 8,
    "dh_conv1x1_dual_wreg_nhwc": [_P, _P, _P, _P, _P] + [_I] * 10 + [_P, _P, _P, _P, _I, _I, _P],
    "dh_normalize_u8_hwc": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dh_resize_u8_hwc": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dh_normalize_pack_u8": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_maxpool3x3s2_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dh_avgpool_nhwc": [_P, _P, _I, _I, _I, _I, _P],
    "dh_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dh_avgpool_rows": [_P, _P, _I, _I, _I, _P],
    "dh_nchw_to_rows": [_P, _P, _I, _I, _I, _I, _P],
    "dh_label_mean": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_linear": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "dh_linear_ln": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _c.POINTER(LnFold), _I, _P],
    "dh_linear_ln_wreg_supported": [_I, _I, _I],
    "dh_linear_ln_wreg": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _c.POINTER(LnFold), _I, _P],
    "dh_attn_cross_pack": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dh_attn_cross_decode_packed": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dh_attn_cross_prefill_packed": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dh_conv2d_nhwc_bn_act": [_P, _P, _P, _P, _P, _P] + [_I] * 10 + [_P],
    "dh_conv2d_nhwc_bn_relu_maxpool": [_P, _P, _P, _P, _P] + [_I] * 9 + [_P],
    "dh_embed_rows": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dh_add_layernorm": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "dh_attn_self_decode": [_P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dh_attn_cross_decode": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dh_enc_key_mask": [_P, _P, _I, _I, _I, _P],
    "dh_pad_mask": [_P, _P, _I, _I, _I, _c.c_longlong, _P],
    "dh_autoregressive_mask": [_P, _I, _I, _P],
    "dh_mask_or": [_P, _P, _c.c_longlong, _P],
    "dh_enc_nonzero_rows": [_P, _P, _I, _I, _I, _P],
    "dh_attn_masked": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "dh_lstm_prepare": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dh_lstm_cell": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_lstm_prepare_f32x": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dh_lstm_cell_f32x": [_P, _P, _P, _P, _P, _I, _P, _c.c_longlong, _I, _I, _I, _I, _P],
    "dh_embed_prefill": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _F, _I, _P],
    "dh_attn_self_prefill": [_P, _P, _I, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "dh_attn_cross_prefill": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dh_lstm_layer_fused": [_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_lstm_layer_wreg_supported": [_I, _I],
    "dh_lstm_layer_wreg": [_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_beam_row_sample": [_P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _U64, _P, _I, _I, _P, _P, _P, _P],
    "dh_beam_row_sample_exact": [_P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _U64, _P, _I, _I, _P, _P, _P, _P],
    "dh_beam_select": [_P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P,
                       _U64, _P, _I, _P],
    "dh_decode_layers_supported": [_c.POINTER(TrModel), _I, _I],
    "dh_decode_layers_table_bytes": [_I],
    "dh_decode_layers_table": [_c.POINTER(TrModel), _P, _P],
    "dh_decode_layers": [_c.POINTER(TrModel), _c.POINTER(TrScratch), _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "dh_transformer_decode_position": [_c.POINTER(TrModel), _c.POINTER(TrScratch), _P, _P, _I, _P, _I, _I, _I, _I, _I, _I,
                                       _P, _P, _I, _P, _I, _P],
    "dh_lstm_decode_step": [_c.POINTER(LstmModel), _c.POINTER(LstmScratch), _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P,
                            _I, _P, _I, _P, _I, _P],
    "dh_vocab_logprob": [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dh_vocab_logits": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "dh_vocab_logits_wreg_supported": [_I] * 5,
    "dh_vocab_logits_wreg": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "dh_beam_row_sample_groups": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _U64, _P, _I, _I, _P, _P, _P, _P],
    "dh_token_logprob": [_P, _I, _I, _P, _P, _I, _P],
    "dh_seq_perplexity": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dh_prof_begin": [_c.c_char_p],
    "dh_prof_set_stride": [_I],
    "dh_prof_end": [],
    "dh_prof_num": [],
    "dh_prof_get": [_I, _c.c_char_p, _I, _c.POINTER(_I), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double),
                    _c.POINTER(_c.c_double)],
    "dh_beam_finalize": [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _U64, _P, _I, _P],
    "dh_beam_filter_top_k": [_P, _I, _I, _I, _I, _I, _P],
    "dh_beam_sample_k": [_P, _I, _I, _I, _I, _F, _P, _I, _U64, _P, _I, _I, _P, _P, _P],
    "dh_beam_gather": [_P, _I, _I, _P, _I, _P, _I, _P, _P],
    "dh_beam_expand": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "dh_split_f32x": [_P, _I, _P, _I, _I, _I, _P],
    "dh_f32x_take_overflow": [_P, _P],
    "dh_linear_f32x_wreg_supported": [_I, _I, _I],
    "dh_linear_f32x_wreg": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "dh_linear_f32xp_wreg": [_P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "dh_add_layernorm_f32x": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "dh_linear_f32x": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "dh_conv2d_nhwc_f32x": [_P, _P, _I, _P, _P, _P, _P] + [_I] * 9 + [_P],
    "dh_nchw_to_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dh_split_act_f32x": [_P, _I, _P, _I, _I, _I, _P],
    "dh_conv2d_nhwc_f32x_planes_out": [_P, _P, _I, _P, _P, _P] + [_I] * 9 + [_P],
    "dh_conv2d_nhwc_f32xp": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 9 + [_P],
    "dh_conv1x1_f32x_stream_supported": [_I, _I, _I],
    "dh_conv1x1_f32x_stream": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dh_linear_f32xp": [_P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "dh_maxpool3x3s2_nhwc_f32": [_P, _P, _I, _I, _I, _I, _P],
    "dh_avgpool_nhwc_f32": [_P, _P, _I, _I, _I, _P],
    "dh_option_count": [],
    "dh_get_option": [_c.c_char_p, _c.POINTER(_I)],
    "dh_set_option": [_c.c_char_p, _I],
}
