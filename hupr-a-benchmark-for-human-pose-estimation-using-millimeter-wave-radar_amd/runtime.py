"""ctypes binding of the C ABI (include/hupr.h) — the only door to compute in this package.

There is deliberately no fallback: if ``lib/libhupr_hip.so`` is missing, or a kernel is asked
to run on a non-GPU tensor, this raises.  PyTorch is used for device memory, streams and
``torch.distributed`` only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhupr_hip.so")

c_void_p, c_int, c_size_t, c_float, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                                ctypes.c_float, ctypes.c_char_p)
c_long = ctypes.c_long

# name -> (restype, argtypes); must list every symbol declared in include/hupr.h and include/hupr_debug.h
class AttnItem(ctypes.Structure):
    """hupr_attn_item (include/hupr.h): one attention of a batched single-sample launch."""
    _fields_ = [("K", ctypes.c_void_p), ("Q", ctypes.c_void_p), ("V", ctypes.c_void_p), ("Vres", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("lse", ctypes.c_void_p), ("out16", ctypes.c_void_p)]


class SumItem(ctypes.Structure):
    """hupr_sum_item (include/hupr.h)."""
    _fields_ = [("partial", ctypes.c_void_p), ("n", ctypes.c_int), ("out", ctypes.c_void_p)]


class AttnBwdItem(ctypes.Structure):
    """hupr_attn_bwd_item (include/hupr.h): one attention of a batched backward pass."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("K", "Q", "V", "dO", "V32", "out", "lse", "dK", "dQ", "dV", "Dq")] + \
               [("residual", ctypes.c_int), ("accumulate", ctypes.c_int)]


SIGNATURES = {
    "hupr_version": (c_int, []),
    "hupr_last_error": (c_char_p, []),
    "hupr_launch_count": (ctypes.c_ulonglong, []),
    "hupr_fft_chain_ws_bytes": (c_size_t, [c_int]),
    "hupr_fft_chain_c64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_fft_chain_loader_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_fft_chain_loader_means_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_mnet_fwd_means_f32": (c_int, [c_void_p] * 5 + [c_long, c_int, c_void_p]),
    "hupr_mnet_fwd_means_bf16act": (c_int, [c_void_p] * 5 + [c_long, c_int, c_void_p]),
    "hupr_fft_chain_opts": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_loader_normalize_c64": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "hupr_dca1000_deinterleave": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "hupr_gemm_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long,
                              c_long, c_int, c_long, c_long, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "hupr_conv_fwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 19 + [c_void_p]),
    "hupr_conv_wgrad_ws_bytes": (c_size_t, [c_int] * 9),
    "hupr_conv_wgrad_f32": (c_int, [c_void_p] * 3 + [c_int] * 17 + [c_void_p, c_size_t, c_void_p]),
    "hupr_gemm_bf16": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long,
                               c_long, c_int, c_long, c_long, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "hupr_conv_fwd_bf16": (c_int, [c_void_p] * 5 + [c_int] * 19 + [c_void_p]),
    "hupr_conv_wgrad_bf16": (c_int, [c_void_p] * 3 + [c_int] * 17 + [c_void_p, c_size_t, c_void_p]),
    "hupr_conv_fwd_bf16_mixed": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 17 + [c_void_p]),
    "hupr_conv_wgrad_bf16_mixed": (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 17
                                   + [c_void_p, c_size_t, c_void_p]),
    "hupr_tmerge_dgrad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 5 + [c_void_p]),
    "hupr_pack_conv_weights_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "hupr_conv3x3_halo_stats_supported": (c_int, [c_int] * 7),
    "hupr_conv3x3_halo_stats_rows": (c_int, []),
    "hupr_conv3x3_halo_bf16act_stats": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
    "hupr_bn_train_finalize_f32": (c_int, [c_void_p, c_int, c_long, c_int] + [c_void_p] * 4 + [c_float, c_float] + [c_void_p] * 5),
    "hupr_bn_train_finalize2_f32": (c_int, ([c_void_p, c_int] + [c_void_p] * 4 + [c_float, c_float] + [c_void_p] * 4) * 2
                                    + [c_long, c_int, c_void_p]),
    "hupr_pack_conv_weights_table": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "hupr_debug_attn_trace": (None, [c_void_p]),
    "hupr_debug_halo_ablate": (None, [c_int]),
    "hupr_debug_halo_variant": (None, [c_int]),
    "hupr_debug_halo_tiles": (None, [c_int]),
    "hupr_debug_halo_trace": (None, [c_void_p]),
    "hupr_debug_wgrad_ci32": (None, [c_int]),
    "hupr_debug_wgrad_m16": (None, [c_int]),
    "hupr_debug_splitk_slices": (None, [c_int]),
    "hupr_debug_halo_res_prefetch": (None, [c_int]),
    "hupr_conv3x3_halo_supported": (c_int, [c_int] * 10),
    "hupr_conv3x3_halo_bf16": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    "hupr_conv3x3_wgrad_halo_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "hupr_conv3x3_wgrad_halo_bf16": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "hupr_pack_conv_weights_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "hupr_bn_ws_bytes": (c_size_t, [c_int]),
    "hupr_bn_train_stats_f32": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                        c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_eval_params_f32": (c_int, [c_void_p] * 4 + [c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "hupr_scale_shift_act_f32": (c_int, [c_void_p] * 7 + [c_long, c_int, c_int, c_void_p]),
    "hupr_bn_eval_act_f32": (c_int, [c_void_p] * 5 + [c_float] + [c_void_p] * 5 + [c_float, c_void_p, c_long, c_int, c_int, c_void_p]),
    "hupr_bn_eval_act_bf16act": (c_int, [c_void_p] * 5 + [c_float] + [c_void_p] * 5 + [c_float, c_void_p, c_long, c_int, c_int, c_void_p]),
    "hupr_bn_bwd_f32": (c_int, [c_void_p] * 9 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd2_f32": (c_int, [c_void_p] * 16 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd_remask_f32": (c_int, [c_void_p] * 10 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd_remask_bf16act": (c_int, [c_void_p] * 10 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd2_remask_f32": (c_int, [c_void_p] * 19 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd2_remask_bf16act": (c_int, [c_void_p] * 19 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_bn_bwd2_bf16act": (c_int, [c_void_p] * 16 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_colsum_f32": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_prelu_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_prelu_ws_bytes": (c_size_t, []),
    "hupr_prelu_bwd_f32": (c_int, [c_void_p] * 5 + [c_long, c_void_p, c_size_t, c_void_p]),
    "hupr_mnet_fwd_f32": (c_int, [c_void_p] * 5 + [c_long, c_int, c_void_p]),
    "hupr_mnet_bwd_ws_bytes": (c_size_t, []),
    "hupr_mnet_bwd_f32": (c_int, [c_void_p] * 7 + [c_long, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_interp_linear_fwd_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_interp_linear_bwd_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_attn_flash_supported": (c_int, [c_int, c_int]),
    "hupr_attn_fwd_bf16": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "hupr_attn_bwd_bf16": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p]),
    "hupr_attn_fwd_bf16in": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p]),
    "hupr_attn_bwd_bf16in": (c_int, [c_void_p] * 12 + [c_int] * 4 + [c_void_p]),
    "hupr_attn_fwd_bf16in_ld": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_void_p, c_int] + [c_int] * 3
                                + [c_void_p]),
    "hupr_attn_fwd_split_ws_bytes": (c_size_t, [c_int] * 3),
    "hupr_attn_fwd_bf16in_ld_ws_batch": (c_int, [ctypes.POINTER(AttnItem)] + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "hupr_debug_attn_split": (None, [c_int]),
    "hupr_attn_fwd_bf16in_ld_ws": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_void_p, c_int] + [c_int] * 3
                                   + [c_void_p, c_size_t, c_void_p]),
    "hupr_attn_bwd_bf16in_ld": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 4
                                + [c_void_p, c_int, c_void_p, c_int] + [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "hupr_attn_fwd_bf16in_ld_ws_qs": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_void_p, c_int] + [c_int] * 3
                                      + [c_void_p, c_size_t, c_void_p]),
    "hupr_attn_fwd_bf16in_ld_ws_batch_qs": (c_int, [ctypes.POINTER(AttnItem)] + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "hupr_attn_bwd_bf16in_ld_qs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 4
                                   + [c_void_p, c_int, c_void_p, c_int] + [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "hupr_attn_bwd_bf16in_ld_batch": (c_int, [ctypes.POINTER(AttnBwdItem)] + [c_int] * 9 + [c_void_p]),
    "hupr_attn_bwd_bf16in_ld_batch_qs": (c_int, [ctypes.POINTER(AttnBwdItem)] + [c_int] * 9 + [c_void_p]),
    "hupr_softmax_rows_f32": (c_int, [c_void_p, c_long, c_int, c_void_p]),
    "hupr_softmax_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "hupr_head1x1_ws_bytes": (c_size_t, []),
    "hupr_head1x1_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_head1x1_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_size_t, c_void_p]),
    "hupr_head1x1_bwd_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_size_t, c_void_p]),
    "hupr_gcn_adj_fwd_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "hupr_gcn_adj_fwd_sliced_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "hupr_gcn_adj_bwd_f32": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "hupr_gcn_wx_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "hupr_gcn_dw_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "hupr_sigmoid_to_nchw_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "hupr_sigmoid_to_nchw_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "hupr_bce_ws_bytes": (c_size_t, []),
    "hupr_bce_fwd_f32": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_bce_bwd_f32": (c_int, [c_void_p] * 4 + [c_long, c_void_p]),
    "hupr_bce_pair_fwd_f32": (c_int, [c_void_p] * 3 + [c_long, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_bce_pair_bwd_f32": (c_int, [c_void_p] * 5 + [c_float, c_float, c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_gaussian_targets_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_float, c_void_p]),
    "hupr_argmax_rows_f32": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "hupr_adam_step_f32": (c_int, [c_void_p] * 4 + [c_long] + [c_float] * 5 + [c_int, c_float, c_void_p]),
    "hupr_adam_step_dev_f32": (c_int, [c_void_p] * 4 + [c_long, c_void_p] + [c_float] * 5 + [c_void_p]),
    # bf16-activation variants (same argument lists as their fp32-activation counterparts)
    "hupr_conv3x3_halo_bf16act": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    "hupr_conv3x3_halo_splitk_ws_bytes": (c_size_t, [c_int] * 7),
    "hupr_conv3x3_halo_bf16act_ws": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "hupr_conv3x3_halo_bf16act_partial": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "hupr_infer_tail_bf16act": (c_int, [c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_float, c_void_p, c_int] + [c_void_p] * 4 +
                                [c_float, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p]),
    "hupr_debug_halo_split_k": (None, [c_int]),
    "hupr_tmerge_stream_supported": (c_int, [c_int] * 4),
    "hupr_tmerge_fwd_stream_bf16": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "hupr_tmerge_dgrad_stream_bf16": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "hupr_tmerge_wgrad_stream_supported": (c_int, [c_int] * 4),
    "hupr_tmerge_wgrad_stream_ws_bytes": (c_size_t, [c_int] * 5),
    "hupr_tmerge_wgrad_stream_bf16": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "hupr_conv3x3_wgrad_halo_bf16act": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "hupr_conv3x3_wgrad_halo_dual_supported": (c_int, [c_int] * 7),
    "hupr_conv3x3_wgrad_halo_bf16act_dual": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "hupr_bn_train_stats_bf16act": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_scale_shift_act_bf16act": (c_int, [c_void_p] * 7 + [c_long, c_int, c_int, c_void_p]),
    "hupr_bn_bwd_bf16act": (c_int, [c_void_p] * 9 + [c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_colsum_bf16act": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hupr_prelu_fwd_bf16act": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_prelu_bwd_bf16act": (c_int, [c_void_p] * 5 + [c_long, c_void_p, c_size_t, c_void_p]),
    "hupr_prelu_bwd_partials_f32": (c_int, [c_void_p] * 4 + [c_long, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "hupr_prelu_bwd_partials_bf16act": (c_int, [c_void_p] * 4 + [c_long, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "hupr_sum_partials_multi": (c_int, [ctypes.POINTER(SumItem), c_int, c_void_p]),
    "hupr_mnet_fwd_bf16act": (c_int, [c_void_p] * 5 + [c_long, c_int, c_void_p]),
    "hupr_mnet_bwd_bf16act": (c_int, [c_void_p] * 7 + [c_long, c_int, c_void_p, c_size_t, c_void_p]),
    "hupr_interp_linear_fwd_bf16act": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_interp_linear_bwd_bf16act": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_interp_linear_bwd_acc_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_interp_linear_bwd_acc_bf16act": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "hupr_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "hupr_mscsa_proj_supported": (c_int, [c_long, c_int]),
    "hupr_mscsa_proj_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "hupr_mscsa_proj_dgrad_supported": (c_int, [c_long, c_int]),
    "hupr_mscsa_proj_dgrad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    # (e) RCCL exchange step
    "hupr_comm_load": (c_int, [c_char_p]),
    "hupr_comm_unique_id": (c_int, [c_void_p]),
    "hupr_comm_init_rank": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_int, c_int]),
    "hupr_comm_destroy": (c_int, [c_void_p]),
    "hupr_comm_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "hupr_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "hupr_broadcast_bucket": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
}

_lib = None


class HuprError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HuprError(
                "HIP extension %s not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        override = os.environ.get("HUPR_LIB_PATH")      # A/B aid: an older build of the library (missing symbols are skipped)
        L = ctypes.CDLL(override or LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if override and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    """The calling thread's last error message (hupr_last_error)."""
    return lib().hupr_last_error().decode()


def check(rc):
    if rc != 0:
        raise HuprError("hupr error %d: %s" % (rc, lib().hupr_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Current torch HIP stream (of the current device) as a raw hipStream_t.  Called once per kernel launch
    (~1400 times per training step), hence the private fast accessor when this torch build has it."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous GPU tensor (loud failure otherwise)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HuprError("tensor is on %s; the HIP path needs a GPU tensor (no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise HuprError("tensor must be contiguous")
    return t.data_ptr()


def ptr_ld(t):
    """Device pointer of a GPU tensor that is a channel slice of a wider contiguous channels-last tensor (the caller passes its row
    stride as the kernel's leading dimension and has checked the layout: functional._ld_view_ok)."""
    if not t.is_cuda:
        raise HuprError("tensor is on %s; the HIP path needs a GPU tensor (no CPU fallback)" % t.device)
    return t.data_ptr()
