"""ctypes binding of libt2onet_hip.so (C ABI: include/t2onet_hip.h).

There is deliberately NO fallback: if the library is missing or fails to load the package
raises, and every wrapper raises RuntimeError with the library's error text on a non-zero
status.  The library is built in-tree by `python -m t2onet_amd.build` (hipcc, gfx950).
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; ours resolves against the same runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libt2onet_hip.so')

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int)
_P = ctypes.c_void_p
_I = ctypes.c_int
_Z = ctypes.c_size_t
_F = ctypes.c_float

# name -> (restype, argtypes); must list every function declared in include/t2onet_hip.h
SIGNATURES = {
    't2o_abi_version': (_I, []),
    't2o_last_error': (ctypes.c_char_p, []),
    't2o_source_digest': (ctypes.c_char_p, []),
    't2o_op_num_params': (_I, [_I]),
    't2o_workspace_bytes': (_Z, [_I, _I, _I]),
    't2o_op_fwd': (_I, [_I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _P]),
    't2o_op_bwd': (_I, [_I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    't2o_apply_fwd': (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _P]),
    't2o_apply_bwd': (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    't2o_l1_fwd': (_I, [_P, _P, _P, _Z, _P, _Z, _P]),
    't2o_l1_bwd': (_I, [_P, _P, _P, _P, _Z, _P]),
    't2o_op_fwd_l1': (_I, [_I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_op_bwd_l1': (_I, [_I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    't2o_sequence_fwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_sequence_bwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_fused_sequence_buffers': (_I, [_P, _I]),
    't2o_fused_sequence_fwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_fused_sequence_bwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_fused_sequence_l1_value_grad': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    't2o_candidates_workspace_bytes': (_Z, [_I, _I, _I]),
    't2o_op_candidates_l1': (_I, [_I, _P, _P, _P, _I, _I, _P, _P, _Z, _I, _I, _P]),
    't2o_candidates_multi_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    't2o_op_candidates_multi_l1': (_I, [c_i, c_i, _I, _P, _I, _P, _P, _I, _I, _P, _P, _Z, _I, _I, _P]),
    't2o_ssim_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    't2o_ssim_fwd': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    't2o_ssim_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_wino_fused_wgrad_supported': (_I, [_I, _I, _I, _I, _I]),
    't2o_wino_fused_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_wino_fused_wgrad_nhwc': (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    't2o_comm_available': (_I, []),
    't2o_comm_unique_id': (_I, [_P]),
    't2o_comm_init_rank': (_I, [_P, _I, _P, _I]),
    't2o_comm_destroy': (_I, [_P]),
    't2o_allreduce': (_I, [_P, _Z, _P, _P]),
    't2o_allreduce_mean': (_I, [_P, _Z, _P, _P]),
    't2o_choose_op': (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _P]),
    't2o_attn_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    't2o_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    't2o_bn_workspace_bytes': (_Z, [_I, _I]),
    't2o_bn_relu_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _Z, _I, _I, _I, _P]),
    't2o_bn_relu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    't2o_bn_nhwc_workspace_bytes': (_Z, [_I, _I]),
    't2o_bn_relu_nhwc_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _Z, _I, _I, _P]),
    't2o_bn_relu_nhwc_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _Z, _I, _I, _P]),
    't2o_bn_relu_nhwc_fwd_partials': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _I, _P, _Z, _I, _I, _P]),
    't2o_param_heads_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _I, _P]),
    't2o_param_heads_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _I, _P]),
    't2o_param_heads_bwd_acc': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _I, _I, _P]),
    't2o_adam_step': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _I, _P]),
    't2o_graph_memsets_to_kernels': (_I, [_P, _P]),
    't2o_conv3x3_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3_wgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_fwd_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3_fwd_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_dgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3_dgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv_set_zero_region': (_I, [_I, _P, _Z]),
    't2o_conv3x3s2_fwd_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3s2_fwd_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_fwd_stats_rows': (_I, [_I, _I, _I, _I, _I]),
    't2o_conv3x3_fwd_stats_nhwc': (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    't2o_stem_fwd_stats_rows': (_I, [_I, _I, _I]),
    't2o_stem_fwd_nhwc': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_stem_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    't2o_stem_wgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    't2o_conv3x3s2_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3s2_wgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3s2_dgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv3x3s2_dgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_bn_relu_nhwc_bwd_acc': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _I, _I, _P]),
    't2o_bn_relu_nhwc_bwd_partials_acc': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _Z, _I, _I, _P]),
    't2o_conv3x3_wgrad_acc_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P]),
    't2o_conv_weight_transform': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    't2o_conv_weight_transform_batch': (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    't2o_wino_weight_transform_batch': (_I, [_P, _P, _P, _P, _I, _P]),
    't2o_wino_weight_transform_chunked_batch': (_I, [_P, _P, _P, _P, _I, _P]),
    't2o_end_select_l1_fwd': (_I, [_P, _I, _P, _P, _P, _I, _Z, _P, _Z, _P]),
    't2o_end_select_l1_bwd': (_I, [_P, _P, _I, _P, _P, _P, _I, _Z, _P]),
    't2o_wino_padded_tiles': (_I, [_I, _I, _I]),
    't2o_wino_weight_transform': (_I, [_P, _P, _I, _I, _P]),
    't2o_gemm_nt_batched': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_gemm': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    't2o_colsum': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    't2o_gemm_tn_splits': (_I, [_I, _I, _I, _I]),
    't2o_gemm_tn_batched': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_gemm_tn_batched_ld': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    't2o_wino_input_transform': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    't2o_wino_input_transform_ld': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_wino_stats_rows': (_I, [_I, _I, _I, _I]),
    't2o_wino_output_transform': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_wino_dy_transform': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    't2o_wino_dy_transforms': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_wino_dy_transforms_ld': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_wino_dy_transform_ld': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_wino_dw_transform': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    't2o_conv3x3_dgrad_pre_nhwc': (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_dgrad_bnsums_rows': (_I, [_I, _I, _I, _I, _I]),
    't2o_wino_fused_supported': (_I, [_I, _I, _I, _I, _I]),
    't2o_wino_fused_stats_rows': (_I, [_I, _I, _I]),
    't2o_wino_u_chunked': (_I, [_P, _P, _I, _I, _P]),
    't2o_wino_fused_conv_nhwc': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_wino_fused_conv_bnsums_nhwc': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_dgrad_pre_bnsums_nhwc': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3s2_dgrad_pre_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    't2o_stem_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_stem_fwd_any': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_stem_wgrad': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    't2o_stem_dgrad': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    't2o_conv1x1s2_fwd_nhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_conv1x1s2_dgrad_acc_nhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    't2o_conv1x1s2_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    't2o_conv1x1s2_wgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    't2o_fused_sequence_prepare': (_I, [_P, _I]),
    't2o_jit_set_cache_dir': (_I, [ctypes.c_char_p]),
    't2o_jit_specialisations': (_I, []),
    't2o_conv3x3_any_fwd_nhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_any_dgrad_nhwc': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    't2o_conv3x3_any_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I, _I]),
    't2o_conv3x3_any_wgrad_nhwc': (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P]),
    't2o_lstm_layer_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_lstm_layer_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    't2o_bn_dual_nhwc_workspace_bytes': (_Z, [_I, _I]),
    't2o_bn_dual_relu_nhwc_fwd': (_I, [_P, _P, _I, _P] + [_P] * 13 + [_F, _F, _F, _F, _P, _Z, _I, _I, _P]),
    't2o_bn_dual_relu_nhwc_bwd_acc': (_I, [_P] * 18 + [_I, _P, _Z, _I, _I, _P]),
    't2o_image_feature_fwd': (_I, [_P, _P]),
    't2o_image_feature_bwd': (_I, [_P, _P]),
    't2o_decoder_step_fwd': (_I, [_P, _P]),
    't2o_decoder_step_bwd': (_I, [_P, _P]),
}


class ImageFeatureArgs(ctypes.Structure):
    """t2o_image_feature_t of include/t2onet_hip.h (field for field)."""
    _fields_ = ([(n, _P) for n in ('fc_w', 'fc_b', 'bn_w', 'bn_b', 'running_mean', 'running_var', 'num_batches_tracked', 'pooled',
                                   'pooled_copy', 'fc_out', 'stats', 'feat', 'g_feat', 'd_fc', 'd_bn', 'd_pooled')]
                + [('momentum', _F), ('eps', _F), ('training', _I), ('B', _I), ('K', _I), ('D', _I)])


class DecoderStepArgs(ctypes.Structure):
    """t2o_decoder_step_t of include/t2onet_hip.h (field for field)."""
    _fields_ = ([(n, _P) for n in ('emb', 'vis_w', 'vis_b', 'w_ih0', 'w_hh0', 'b_ih0', 'b_hh0', 'w_ih1', 'w_hh1', 'b_ih1', 'b_hh1',
                                   'lo_w', 'lo_b', 'out_w', 'out_b', 'prev_op', 'feat', 'h0', 'c0', 'h1', 'c1', 'enc', 'step_in',
                                   'feat_copy', 'hp0', 'hp1', 'prev_op_copy', 'gates0', 'gates1', 'h0n', 'c0n', 'h1n', 'c1n', 'attn', 'mix', 'ctx', 'logp',
                                   'g_logp', 'g_ctx', 'g_h0n', 'g_c0n', 'g_h1n', 'g_c1n', 'd_logits', 'd_lin', 'd_gates1',
                                   'd_gates0', 'd_step_in', 'd_vis', 'd_ctx', 'd_mix', 'd_qa', 'd_q', 'd_x1', 'd_feat', 'd_h0',
                                   'd_c0', 'd_h1', 'd_c1', 'd_enc')]
                + [(n, _I) for n in ('B', 'L', 'D', 'E', 'V')])

_lib = None


ABI_VERSION = 4


def load():
    """Load (once) and return the ctypes library.  The library must have been built from exactly the sources
    in this tree (digest compiled into it): a stale one is rebuilt when hipcc is here -- compiler errors
    propagate -- and refused otherwise.  Nothing is swallowed and nothing falls back."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    want = _build.source_digest()
    have = _build.library_digest(LIB_PATH)
    if have != want:
        if os.path.exists(_build.hipcc_path()):
            _build.build()                      # raises on failure; atomic rename under a file lock
            have = _build.library_digest(LIB_PATH)
        if have != want:
            raise RuntimeError(
                'libt2onet_hip.so at %s %s; build it with `python -m t2onet_amd.build` (needs hipcc; there is no '
                'CPU fallback)' % (LIB_PATH, 'is missing' if have is None and not os.path.exists(LIB_PATH)
                                   else 'was built from other sources (digest %s, tree %s)' % (have, want)))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.t2o_abi_version() != ABI_VERSION or lib.t2o_source_digest().decode() != want:
        raise RuntimeError('libt2onet_hip.so ABI version / source digest mismatch after load')
    jit_dir = os.environ.get('T2O_JIT_CACHE', os.path.join(_HERE, 'lib', 'jit'))      # code objects of run-time specialised chains
    try:
        os.makedirs(jit_dir, exist_ok=True)
        lib.t2o_jit_set_cache_dir(jit_dir.encode())
    except OSError:
        pass
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().t2o_last_error().decode('utf-8', 'replace')
        raise RuntimeError('%s failed (status %d): %s' % (what, rc, msg))
