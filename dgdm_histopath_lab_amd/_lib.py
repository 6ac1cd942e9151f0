"""ctypes binding of libdgdm_hip.so (the C ABI declared in include/dgdm_hip.h).

There is no CPU fallback: every op in this package goes through this library and raises
``DGDMKernelError`` if it is missing or if a kernel reports an error.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libdgdm_hip.so")


class DGDMKernelError(RuntimeError):
    """Raised when libdgdm_hip.so is unavailable or a kernel entry point fails."""


_i32, _i64, _sz, _p = C.c_int32, C.c_int64, C.c_size_t, C.c_void_p

ABI_VERSION = 2      # DGDM_ABI_VERSION of include/dgdm_hip.h that SIGNATURES below was written for

# name -> (restype, argtypes); mirrors include/dgdm_hip.h (tests check the two stay in sync)
SIGNATURES = {
    "dgdm_abi_version": (C.c_int, []),
    "dgdm_error_string": (C.c_char_p, [C.c_int]),
    "dgdm_validate_inputs": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _p]),
    "dgdm_csr_build_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "dgdm_csr_build": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "dgdm_csr_build_pair_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "dgdm_csr_build_pair_status_offset": (_sz, [_i64, _i32, _i32]),
    "dgdm_csr_build_pair": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p, _i32, _p]),
    "dgdm_csr_extend": (C.c_int, [_p] * 10 + [_i32, _i64, _i32, _i32, _i64] + [_p] * 11),
    "dgdm_spmm_long_item_cap": (_i32, [_i64]),
    "dgdm_spmm_long_slot_cap": (_i32, [_i64]),
    "dgdm_spmm_long_table_words": (_sz, [_i64]),
    "dgdm_gcn_dinv": (C.c_int, [_p, _i32, _p, _p]),
    "dgdm_csr_edge_weights": (C.c_int, [_p, _p, _p, _i32, _p, _p]),
    "dgdm_spatial_attn_q_tile_rows": (_i32, []),
    "dgdm_spatial_attn_fwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_uint32, _p, _i64, _p, _p]),
    "dgdm_spatial_attn_fwd_variant": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_uint32, _p, _i64, _p, _i32, _p]),
    "dgdm_spatial_attn_bwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float,
                                        _p, C.c_float, C.c_uint32, _p, _p, _p, _i64, _p, _p]),
    "dgdm_spatial_attn_gen_fwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_uint32, _p, _i64, _p, _p]),
    "dgdm_spatial_attn_gen_bwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                            _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "dgdm_spatial_attn_gen_mean_weights": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _p, _p, _p, _p]),
    "dgdm_attn_dense_fwd": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, C.c_float, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p,
                                      C.c_float, C.c_float, C.c_uint32, _p, _i64, _p, _p]),
    "dgdm_attn_dense_bwd": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, C.c_float, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p,
                                      C.c_float, C.c_float, C.c_uint32, _p, _p, _i64, _p, _p, _p, _i64, _p, _p, _i64, _p]),
    "dgdm_attn_dense_weights": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _i32, C.c_float, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p,
                                          C.c_float, C.c_float, C.c_uint32, _p, _i32, _p, _p]),
    "dgdm_add_posenc": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _i64, _p, _p]),
    "dgdm_spatial_attn_mean_weights": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _p, _p, _p, _p]),
    "dgdm_rownorm_fwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, C.c_float, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p, _p]),
    "dgdm_rownorm_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dgdm_rownorm_bwd_slots": (_i64, [_i32, _i32, _i32]),
    "dgdm_rownorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p, _sz, _p, _p]),
    "dgdm_act_dropout_fwd": (C.c_int, [_p, _i64, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p]),
    "dgdm_act_dropout_bwd": (C.c_int, [_p, _p, _i64, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p]),
    "dgdm_spatial_attn_bwd_dq": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float,
                                           _p, C.c_float, C.c_uint32, _p, _i64, _p, _p]),
    "dgdm_spatial_attn_bwd_dkv": (C.c_int, [_p, _p, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_float,
                                            _p, _p, C.c_float, C.c_uint32, _p, _p, _i64, _p]),
    "dgdm_segment_bcast_add": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "dgdm_segment_sum_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_segment_sum": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _sz, _p]),
    "dgdm_colnorm_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_colnorm_fwd": (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _i32, C.c_float, C.c_float, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p, _sz, _p, _p]),
    "dgdm_colnorm_bwd": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p, _p, _i32, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p, _sz, _p, _p]),
    "dgdm_segment_max_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_segment_max_fwd": (C.c_int, [_p, _i64, _p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "dgdm_segment_max_bwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "dgdm_attn_pool_fwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dgdm_attn_pool_fwd": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, C.c_float, C.c_uint32, _p, _p, _p, _sz, _p]),
    "dgdm_attn_pool_bwd": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _i32, C.c_float, C.c_uint32, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "dgdm_gemm_nt": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_nn": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_tn_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dgdm_gemm_tn": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "dgdm_adamw_step": (C.c_int, [_p, _i32, _p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p, _p, _p]),
    "dgdm_seed_epoch_advance": (C.c_int, [_p]),
    "dgdm_seed_epoch_set": (C.c_int, [C.c_uint32, _p]),
    "dgdm_linear_small_fwd": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "dgdm_linear_small_bwd": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _p]),
    "dgdm_qsample": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "dgdm_segment_mse_workspace_bytes": (_sz, [_i32]),
    "dgdm_segment_mse_fwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "dgdm_segment_mse_bwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "dgdm_mask_rows": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p, _p]),
    "dgdm_ddpm_step": (C.c_int, [_p, _p, _p, _i64, C.c_float, C.c_float, C.c_float, C.c_float, _i32, _p, _p]),
    "dgdm_denoise_ddpm_step_supported": (_i32, [_i32]),
    "dgdm_denoise_ddpm_step": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _i32, _p, _i32, _p, _p, _p, C.c_float, _p, _p, _p, C.c_float, _p,
                                         C.c_float, C.c_float, C.c_float, C.c_float, _i32, _p, _i64, _p]),
    "dgdm_pool_score_fwd": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, _p, _p, _i32, _p]),
    "dgdm_vec_softmax_fwd": (C.c_int, [_p, _i32, _p, _p]),
    "dgdm_vec_softmax_bwd": (C.c_int, [_p, _p, _i32, _p, _p]),
    "dgdm_count_ge": (C.c_int, [_p, _i32, C.c_float, _p, _p]),
    "dgdm_pool_score_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_pool_score_bwd": (C.c_int, [_p, _i64, _p, _p, _p, _i32, _i32, _p, _i64, _p, _p, _p, _i32, _p, _sz, _p, _p]),
    "dgdm_topk_perm_workspace_bytes": (_sz, [_i32]),
    "dgdm_topk_perm": (C.c_int, [_p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "dgdm_pool_gather_fwd": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, C.c_float, _p, _i64, _p]),
    "dgdm_pool_gather_bwd": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _i32, C.c_float, _p, _i64, _p, _p]),
    "dgdm_edge_relabel": (C.c_int, [_p, _i64, _p, _i32, _p, _p]),
    "dgdm_unpool_add_relu_fwd": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _i32, _p, _i64, _p, _p]),
    "dgdm_unpool_add_relu_bwd": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _i32, _p, _i64, _p, _i64, _p, _p]),
    "dgdm_knn2d": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "dgdm_row_sqnorm": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "dgdm_knn_gram_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_knn_gram": (C.c_int, [_p, _i64, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _sz, _p]),
    "dgdm_pair_cosine": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, _i32, _p, _p]),
    "dgdm_edge_dedup_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dgdm_edge_dedup_count": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _i32, C.c_float, _p, _sz, _p, _p]),
    "dgdm_edge_emit": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _i32, C.c_float, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "dgdm_gemm_nt_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_nt_split_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_nn_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_tn_bf16x3_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dgdm_gemm_tn_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "dgdm_gemm_tn_split": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "dgdm_gemm_tn_split_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "dgdm_fill_u32": (C.c_int, [_p, _i64, C.c_uint32, _p]),
    "dgdm_amax_bits": (C.c_int, [_p, _i64, _i64, _i32, _p, _p]),
    "dgdm_amax_table": (C.c_int, [_p, _i32, _p, _p]),
    "dgdm_gemm_nt_f16x2": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "dgdm_gemm_nt_split_f16x2": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "dgdm_gemm_nn_f16x2": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "dgdm_gemm_tn_f16x2_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dgdm_gemm_tn_f16x2": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p, _p, _p]),
    "dgdm_gemm_tn_split_f16x2": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _i32, _p, _sz, _p, _p, _p]),
    "dgdm_gemm_tn_chunks": (_i32, [_i32, _i32, _i32]),
    "dgdm_gemm_tn_partial_bf16x3": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    "dgdm_gemm_tn_partial_f16x2": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _sz, _p, _p, _p]),
    "dgdm_gemm_tn_reduce_many": (C.c_int, [_p, _i32, _p]),
    "dgdm_gemm_tn_partial_many_f16x2": (C.c_int, [_p, _i32, _p]),
    "dgdm_gemm_tn_chunks_grouped": (_i32, [_i32, _i32, _i32]),
    "dgdm_attn_pack_bytes": (_sz, [_i32, _i32, _i32]),
    "dgdm_amax_scale_workspace_bytes": (_sz, []),
    "dgdm_amax_pow2_scale": (C.c_int, [_p, _i64, C.c_float, _p, _p, _sz, _p]),
    "dgdm_attn_pack": (C.c_int, [_p, _i64, _i32, _i32, _i32, C.c_float, _p, _p, _i32, _i32, _i32, _p, _p, C.c_float, _p, _p, _i64,
                                 _p, _p, _p, _p]),
    "dgdm_spatial_attn_h_fwd": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float, C.c_uint32, _p, _i64, _p, _i32, _p]),
    "dgdm_spatial_attn_h_bwd_dq": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float, C.c_float,
                                             C.c_uint32, _p, _p, _i64, _i32, _p]),
    "dgdm_spatial_attn_h_bwd_dkv": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float,
                                              C.c_uint32, _p, _p, _p, _i64, _i32, _p]),
    "dgdm_spatial_attn_h_bwd_fused_superblocks": (_i32, [_p, _i32]),
    "dgdm_spatial_attn_h_bwd_fused_workspace_bytes": (_sz, [_p, _i32, _i32, _i32, _i32]),
    "dgdm_spatial_attn_h_bwd_fused": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float, C.c_uint32,
                                                _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p]),
    "dgdm_spatial_attn_h_bwd_fused_reduce": (C.c_int, [_p, _p, _i32, _i32, _i32, C.c_float, _p, _p, _i64, _i32, _i32, _p, _sz, _p]),
    "dgdm_attn_skip_map_bytes": (_sz, [_i32, _i32]),
    "dgdm_attn_skip_map_workspace_bytes": (_sz, [_i32, _i32]),
    "dgdm_attn_skip_map_build": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _sz, _p, _sz, _p]),
    "dgdm_attn_skip_map_count": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "dgdm_spatial_attn_h_fwd_sparse": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float, C.c_uint32, _p, _i64, _p, _i32, _p, _p, _p]),
    "dgdm_spatial_attn_h_bwd_fused_sparse": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, C.c_float, C.c_uint32,
                                                       _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p, _p, _p]),
    "dgdm_spatial_attn_h_bwd_fused_reduce_sparse": (C.c_int, [_p, _p, _i32, _i32, _i32, C.c_float, _p, _p, _i64, _i32, _i32, _p, _sz, _p, _p, _p]),
    "dgdm_gemm_image_bytes": (_sz, [_i32, _i32]),
    "dgdm_gemm_image_blocks": (_i32, [_i32, _i32]),
    "dgdm_gemm_image_build_many": (C.c_int, [_p, _i32, _i32, _p]),
    "dgdm_gemm_image_build": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dgdm_gemm_rows_img": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _p, _i64, _i32, _p, _p]),
    "dgdm_gemm_rows_img_act": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _p, _i64, _p, _i64, _i32, C.c_float, C.c_uint32, _p, _p, _p]),
    "dgdm_gemm_rows_img_act_bwd": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _i64, _p, _i64, _i32, C.c_float, C.c_uint32, _p, _p, _p]),
    "dgdm_gemm_rows_img_norm_supported": (_i32, [_i32, _i32]),
    "dgdm_gemm_rows_img_norm": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, C.c_float, C.c_uint32, _p, _i64, _p, _i32, _p, _p, _i32,
                                          C.c_float, _p, _i64, _p, _i64, _p, _p, _i32, C.c_float, C.c_uint32, _p, _p, _p]),
    "dgdm_spmm": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i64, _i32, _i32, _p, _i32, _p, _p]),
    "dgdm_spmm_add": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _i64, _i32, _i32, _p, _p]),
    "dgdm_spmm_concat": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i64, _i32, _p, _i64, _i32, _i32, _p, _p, _p]),
}

class TnReduce(C.Structure):
    """struct DgdmTnReduce of include/dgdm_hip.h"""
    _fields_ = [("partial", _p), ("dW0", _p), ("dW1", _p), ("db", _p), ("ld0", _i64), ("ld1", _i64), ("slots", _i32), ("N", _i32), ("K", _i32),
                ("K0", _i32)]


TN_REDUCE_MAX = 24


class TnPartial(C.Structure):
    """struct DgdmTnPartial of include/dgdm_hip.h"""
    _fields_ = [("dY", _p), ("X", _p), ("workspace", _p), ("amax_dy", _p), ("amax_x", _p), ("ldy", _i64), ("ldx", _i64),
                ("workspace_bytes", C.c_size_t), ("M", _i32), ("N", _i32), ("K", _i32), ("with_bias", _i32)]


TN_PARTIAL_MAX = 24


class LongRows(C.Structure):
    """struct DgdmLongRows of include/dgdm_hip.h (a HOST struct holding device pointers)"""
    _fields_ = [("table", _p), ("partial", _p), ("ld", _i64), ("item_cap", _i32), ("slot_cap", _i32)]

class AdamTensor(C.Structure):
    """struct DgdmAdamTensor of include/dgdm_hip.h (a HOST struct holding device pointers)"""
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("numel", _i64)]


_lib: Optional[C.CDLL] = None


def load(build_if_missing: bool = False) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import _build
            _build.build()
        else:
            raise DGDMKernelError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m dgdm_histopath_lab_amd._build`.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. missing libamdhip64
        raise DGDMKernelError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DGDMKernelError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype, fn.argtypes = res, args
    _check_version(lib, LIB_PATH)
    _lib = lib
    return lib


def _check_version(lib: C.CDLL, path: str) -> None:
    got = lib.dgdm_abi_version()
    if got != ABI_VERSION:
        raise DGDMKernelError(f"{path} reports C-ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the extension "
                              "(python -m dgdm_histopath_lab_amd._build); calling through a shifted argument list is not an error code")


def open_library(path: str) -> C.CDLL:
    """Another build of the library (diagnostic twins: lib/canary, lib/stamps) with the same signature table; not cached."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _check_version(lib, path)
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().dgdm_error_string(code).decode()
        raise DGDMKernelError(f"{what} failed: {msg} (code {code})")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


@functools.lru_cache(maxsize=4096)
def _workspace_bytes(name: str, args: tuple) -> int:
    return getattr(load(), name)(*args)


def workspace_bytes(name: str, *args) -> int:
    """``lib.<name>(*args)`` for the ``*_workspace_bytes`` entry points, memoised (bounded LRU: a mixed-size stream asks about new
    shapes at every step): they are pure functions of the shape, and a training step asks the same ~100 questions every time."""
    return _workspace_bytes(name, args)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None) -> int:
    """Handle of torch's current stream on ``device`` (called once per kernel launch: the raw accessor is ~10x cheaper
    than building a torch.cuda.Stream object)."""
    if _raw_stream is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index if device.index is not None else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DGDMKernelError("dgdm_histopath_lab_amd ops run on the GPU only (HIP kernels, no CPU fallback); "
                                  f"got a tensor on {t.device}")
