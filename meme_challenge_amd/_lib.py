"""ctypes binding of libuniter_hip.so (include/uniter_hip.h).

Fails loudly if the shared library is missing or a call returns non-zero:
there is no fallback path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libuniter_hip.so')
if os.environ.get('UNITER_LIB_VARIANT'):      # experimental build for A/B kernel measurements (build.py)
    LIB_PATH = os.path.join(_HERE, 'libuniter_hip_%s.so' % os.environ['UNITER_LIB_VARIANT'])

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p


class UniterConfigC(C.Structure):
    _fields_ = [('hidden_size', C.c_int32), ('num_hidden_layers', C.c_int32),
                ('num_attention_heads', C.c_int32), ('intermediate_size', C.c_int32),
                ('vocab_size', C.c_int32), ('max_position_embeddings', C.c_int32),
                ('type_vocab_size', C.c_int32), ('img_dim', C.c_int32),
                ('hidden_dropout_prob', C.c_float),
                ('attention_probs_dropout_prob', C.c_float)]


class X3RidersC(C.Structure):
    """uniter_x3_riders_t (include/uniter_hip.h): side work riding on a grouped x3 weight-gradient launch"""
    _fields_ = [('ssq', C.c_void_p), ('colsum_out', C.c_void_p), ('grid', C.c_int), ('njobs', C.c_int), ('nred', C.c_int),
                ('part', C.c_void_p * 4), ('nparts', C.c_int * 4), ('stride', C.c_int * 4), ('n', C.c_int * 4), ('seg', C.c_int * 4),
                ('out', (C.c_void_p * 3) * 4), ('first_item', C.c_int * 5)]


class UniterBatchC(C.Structure):
    _fields_ = [('input_ids', C.c_void_p), ('position_ids', C.c_void_p),
                ('txt_type_ids', C.c_void_p), ('img_feat', C.c_void_p),
                ('img_pos_feat', C.c_void_p), ('img_type_ids', C.c_void_p),
                ('img_masks', C.c_void_p), ('attention_mask', C.c_void_p),
                ('gather_index', C.c_void_p),
                ('B', C.c_int32), ('T', C.c_int32), ('R', C.c_int32), ('L', C.c_int32),
                ('pos_bcast', C.c_int32),
                ('cu_seqlens', C.c_void_p), ('pack_src', C.c_void_p), ('pack_dst', C.c_void_p),
                ('Mp', C.c_int32)]


_I, _F, _P, _SZ, _U64, _U32 = C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32

# name -> (restype, argtypes)
_SIGS = {
    'uniter_abi_version': (_I, []),
    'uniter_last_error': (C.c_char_p, []),
    'uniter_build_info': (C.c_char_p, []),
    'uniter_gemm_f32': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    'uniter_gemm_f32_cfg': (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    'uniter_gemm_bf16_cfg': (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    'uniter_colsum_f32': (_I, [_P, _I, _I, _I, _P, _I, _P, _SZ, _P]),
    'uniter_colsum_ws_bytes': (_SZ, [_I, _I]),
    'uniter_ln_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_ln_fwd_b16': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_ln_fwd_slabs': (_I, [_P, _I, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_ln_bwd_rows_slabs': (_I, [_P, _I, _SZ, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_slab_reduce_add': (_I, [_P, _I, _SZ, _P, _SZ, _P]),
    'uniter_colsum_bf16_add': (_I, [_P, _I, _I, _I, _P, _P]),
    'uniter_wgrad_bf16_group': (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _P]),
    'uniter_wgrad_f32_group': (_I, [_I, _P, _P, _I, _P, _P, _P, _I, _P]),
    'uniter_ln_bwd_b16': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_ln_bwd_rows': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_ln_bwd_finalize': (_I, [_P, _SZ, _I, _I, _P, _P, _P, _P]),
    'uniter_ln_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_ln_bwd_ws_bytes': (_SZ, [_I, _I]),
    'uniter_attn_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_gemm_bf16res_cfg': (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    'uniter_gemm_bf16v2_cfg': (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, C.c_long, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    'uniter_split3': (_I, [_P, _I, _I, _I, _P, _SZ, _SZ, _P]),
    'uniter_join3': (_I, [_P, _I, _I, _SZ, _SZ, _P, _I, _P]),
    'uniter_gemm_x3_cfg': (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, C.c_long, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    'uniter_ln_fwd_slabs_x3': (_I, [_P, _I, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_ln_bwd_rows_slabs_x3': (_I, [_P, _I, _SZ, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_wgrad_x3_group': (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    'uniter_gemm_x3_plan': (_I, [_I, _I, _I, _I, _I, _P, _P]),
    'uniter_gemm_x3_plan_fwd32': (_I, [_I, _I, _I, _I, _I, _P, _P]),
    'uniter_gemm_x3_colpart': (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _P, _P]),
    'uniter_wgrad_x3_group_riders': (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    'uniter_wgrad_x3_group_slots': (_I, [_I, _I, _P, _P, _I]),
    'uniter_gemm_x3_balanced_ws_bytes': (_SZ, []),
    'uniter_gemm_x3_cfg_ws': (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, C.c_long, _P, _I, _I, _I, _P, _P, _P, _I, _P, _SZ, _P]),
    'uniter_wgrad_x3_group_ws': (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _SZ, _P]),
    'uniter_wgrad_x3_group_slots_ws': (_I, [_I, _I, _P, _P, _I, _I, _SZ]),
    'uniter_wgrad_bf16_group_riders': (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    'uniter_wgrad_bf16_group_slots': (_I, [_I, _P, _P, _I]),
    'uniter_wgrad_bf16_group_slots_cfg': (_I, [_I, _I, _P, _P, _I]),
    'uniter_gemm_bf16p_plan': (_I, [_I, _I, _I, _I, _I, _I, _P, _P]),
    'uniter_hidden_keep_bits_bytes': (_SZ, [_SZ]),
    'uniter_hidden_keep_bits_gen': (_I, [_P, _SZ, _I, _U32, _U32, _U32, _SZ, _F, _U64, _U32, _P]),
    'uniter_ln_set_next_keep_bits': (_I, [_P]),
    'uniter_model_set_norm_partials': (_I, [_P, _P, _SZ]),
    'uniter_model_set_aux_stream': (_I, [_P, _P]),
    'uniter_model_set_cu_reserve': (_I, [_P, _I]),
    'uniter_model_norm_partials_per_layer': (_I, [_P]),
    'uniter_colsum_x3_add': (_I, [_P, _I, _I, _I, _P, _P]),
    'uniter_cast_bf16': (_I, [_P, _P, _SZ, _P]),
    'uniter_attn_bwd_ws_bytes': (_SZ, [_I, _I, _I]),
    'uniter_attn_varlen_max_len': (_I, []),
    'uniter_attn_bf16_bwd_ws_bytes': (_SZ, [_I, _I, _I]),
    'uniter_attn_bf16_fwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_bf16_bwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_attn_keep_bits_bytes': (_SZ, [_I, _I, _I]),
    'uniter_attn_keep_bits_gen': (_I, [_P, _SZ, _I, _I, _I, _I, _F, _U64, _U32, _U32, _U32, _P]),
    'uniter_attn_fwd_pre': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_bf16_fwd_pre': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_fwd_ex': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_bwd_ex': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_attn_fwd_pre_x3': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_bwd_ex_x3': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_attn_x3_max_len': (_I, []),
    'uniter_attn_x3_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'uniter_attn_x3_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _SZ, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'uniter_attn_b16x_fwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'uniter_attn_b16x_bwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'uniter_ot_dist_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    'uniter_ot_dist_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'uniter_attn_fwd_varlen': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P]),
    'uniter_attn_bwd_varlen': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U32, _U32, _P, _SZ, _P]),
    'uniter_txt_embed_fwd': (_I, [_P] * 9 + [_I] * 8 + [_F, _U64, _U32, _P]),
    'uniter_img_embed_fwd': (_I, [_P] * 14 + [_I] * 6 + [_F, _U64, _U32, _P]),
    'uniter_gather_rows': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'uniter_gather_rows_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'uniter_gather_rows_ex': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'uniter_img_mask_add': (_I, [_P, _P, _P, _P, _I, _I, _P]),
    'uniter_bias_rows': (_I, [_P, _P, _I, _I, _P]),
    'uniter_txt_embed_bwd': (_I, [_P] * 13 + [_I] * 8 + [_F, _U64, _U32, _P, _SZ, _P]),
    'uniter_img_embed_bwd': (_I, [_P] * 24 + [_I] * 6 + [_F, _U64, _U32, _P, _SZ, _P]),
    'uniter_embed_bwd_ws_bytes': (_SZ, [_I, _I]),
    'uniter_pooler_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_pooler_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'uniter_linear_small_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_linear_small_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_bce_logits': (_I, [_P, _P, _F, _P, _P, _P, _F, _I, _P]),
    'uniter_pool_head_fwd': (_I, [_P] * 8 + [_I] * 4 + [_P]),
    'uniter_pool_head_bwd': (_I, [_P] * 10 + [_I] * 5 + [_P]),
    'uniter_row_gather': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'uniter_row_scatter_add': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'uniter_cross_entropy_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_cross_entropy_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_kl_div_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_kl_div_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'uniter_row_argmax': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'uniter_mse_fwd': (_I, [_P, _P, _P, _SZ, _P]),
    'uniter_mse_bwd': (_I, [_P, _P, _P, _P, _SZ, _P]),
    'uniter_dgelu_mul': (_I, [_P, _P, _P, _SZ, _P]),
    'uniter_grad_sumsq': (_I, [_P, _P, _SZ, _P, _P, _SZ, _P]),
    'uniter_grad_sumsq_ws_bytes': (_SZ, [_SZ]),
    'uniter_adam_step': (_I, [_P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P]),
    'uniter_adam_step_mirror': (_I, [_P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _P]),
    'uniter_adam_step_ex': (_I, [_P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _I, _P]),
    'uniter_adam_step_g16': (_I, [_P, _P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _I, _P]),
    'uniter_adam_step_x3': (_I, [_P, _P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _SZ, _I, _P]),
    'uniter_adam_step_x3p': (_I, [_P, _P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _SZ, _P, _SZ, _I, _P]),
    'uniter_mirror_refresh_x3': (_I, [_P, _SZ, _SZ, _P, _SZ, _P, _P]),
    'uniter_model_set_weight_pairing': (_I, [_P, _I]),
    'uniter_adam_step_rows': (_I, [_P, _P, _P, _P, _P, _SZ, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _I, _I, _I, _P]),
    'uniter_grad_sumsq_bf16': (_I, [_P, _P, _SZ, _P, _P, _SZ, _P]),
    'uniter_sumsq_combine': (_I, [_P, _I, _P, _P]),
    'uniter_grad_sumsq_part': (_I, [_P, _P, _SZ, _P, _I, _P]),
    'uniter_num_params': (_I, [C.POINTER(UniterConfigC)]),
    'uniter_param_name': (C.c_char_p, [C.POINTER(UniterConfigC), _I]),
    'uniter_param_shape': (_I, [C.POINTER(UniterConfigC), _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'uniter_model_create': (_I, [C.POINTER(UniterConfigC), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, C.POINTER(C.c_void_p)]),
    'uniter_model_destroy': (None, [_P]),
    'uniter_model_generation': (_U64, [_P]),
    'uniter_model_set_ready_events': (_I, [_P, _P, _I]),
    'uniter_model_set_weight_mirror': (_I, [_P, _P, _P, _SZ]),
    'uniter_model_set_precision': (_I, [_P, _I]),
    'uniter_model_set_wgrad_overwrite': (_I, [_P, _I]),
    'uniter_model_ws_bytes': (_SZ, [_P, _I, _I, _I, _I, _I]),
    'uniter_model_forward': (_I, [_P, C.POINTER(UniterBatchC), _P, _I, _I, _U64, _U32, _P, _SZ, _P]),
    'uniter_model_backward_begin': (_I, [_P, C.POINTER(UniterBatchC), _P, _I, _U64, _U32, _P, _SZ, _P, _P]),
    'uniter_model_backward_layer': (_I, [_P, _I]),
    'uniter_model_backward_embed': (_I, [_P]),
    'uniter_model_backward': (_I, [_P, C.POINTER(UniterBatchC), _P, _I, _U64, _U32, _P, _SZ, _P, _P]),
    'uniter_prof_enable': (_I, [_P, _I]),
    'uniter_prof_enable_stamps': (_I, [_P, _I, _P]),
    'uniter_prof_collect_stamps': (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double), _I]),
    'uniter_prof_stamps_union': (_I, [_P, C.c_uint, C.POINTER(C.c_double)]),
    'uniter_prof_stamp_spans': (_I, [_P, _P, _P, _P, _I, _P]),
    'uniter_prof_collect_kinds': (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double), _I]),
    'uniter_prof_collect': (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


class UniterHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UniterHipError(
                'libuniter_hip.so not found at %s -- build it with '
                '`python -m meme_challenge_amd.build` (there is no fallback path)' % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(h, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing and not os.environ.get('UNITER_DEV_PARTIAL_LIB'):
            raise UniterHipError('libuniter_hip.so lacks declared symbols: %s' % ', '.join(missing))
        if h.uniter_abi_version() != 1:
            raise UniterHipError('libuniter_hip.so ABI version mismatch')
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().uniter_last_error().decode('utf-8', 'replace')
        raise UniterHipError('%s failed (rc=%d): %s' % (what or 'libuniter_hip call', rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), 'libuniter_hip needs contiguous tensors'
    return C.c_void_p(t.data_ptr())


def cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------- #
# streams shared by everything this package builds in a process
# --------------------------------------------------------------------------- #
# HIP maps every stream that has been USED onto one of GPU_MAX_HW_QUEUES hardware queues, round-robin.  A side stream per
# model and a copy stream per DataLoader (six loaders and a model per cross-validation fold) walk through those queues, and
# the stream that lands on the main stream's queue runs BEHIND it instead of beside it: the fp32 step 16.0 instead of 13.5 ms
# with seven other streams in use (tests/tools/stream_queue_probe.py).  So: ONE side stream and ONE copy stream per device
# and process, and the side stream is chosen by measurement -- the first candidate whose spin kernel overlaps the current
# stream's.
_SHARED_STREAMS = {}


def _overlaps(stream, device, cycles=400000):
    """True when a spin on `stream` runs BESIDE a spin on the current stream (different hardware queues)."""
    main = torch.cuda.current_stream(device)
    for attempt in range(2):                 # the first pass pays the stream's first use (its hardware queue is created then)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize(device)
        ev[0].record(main)
        torch.cuda._sleep(cycles)
        ev[1].record(main)                   # one spin alone
        torch.cuda.synchronize(device)
        ev[2].record(main)
        stream.wait_event(ev[2])
        torch.cuda._sleep(cycles)
        with torch.cuda.stream(stream):
            torch.cuda._sleep(cycles)
        main.wait_stream(stream)
        ev[3].record(main)                   # two spins, one per stream: 1.15 x one spin beside each other, 2.1 x in a row
        torch.cuda.synchronize(device)
    return ev[2].elapsed_time(ev[3]) < 1.6 * ev[0].elapsed_time(ev[1])


def shared_stream(device, kind='side'):
    """The process-wide stream of `kind` ('side': weight gradients / optimizer blocks / collectives; 'copy': host -> device
    prefetch) on `device`."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), kind)
    s = _SHARED_STREAMS.get(key)
    if s is None:
        if kind == 'side' and hasattr(torch.cuda, '_sleep') and os.environ.get('UNITER_STREAM_PROBE') != '0':
            tried = []
            for _ in range(8):
                s = torch.cuda.Stream(device=dev)
                if _overlaps(s, dev):
                    break
                tried.append(s)               # (kept alive: the next candidate is another stream)
            else:
                s = tried[0]
        else:
            s = torch.cuda.Stream(device=dev)
        _SHARED_STREAMS[key] = s
    return s


def require_gpu_tensor(t, dtype=None, name='tensor'):
    if t is None:
        return
    if not t.is_cuda:
        raise UniterHipError('%s must live on the GPU (cuda/HIP device); got %s' % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise UniterHipError('%s must be %s; got %s' % (name, dtype, t.dtype))
