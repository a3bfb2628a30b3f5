"""ctypes binding of ``libmorgana_hip.so`` (C ABI declared in ``include/morgana_hip.h``).

This is the stub a maintainer of the reference would add: the reference has no FFI of its own, its
hot path is PyTorch eager.  There is NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Nothing here imports the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmorgana_hip.so')

c_void_p, c_int, c_int64, c_float, c_size_t, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p)

# name -> (restype, argtypes); must list every symbol of include/morgana_hip.h (checked by tests/test_abi.py).
SIGNATURES = {
    'mg_last_error': (c_char_p, []),
    'mg_set_tuning': (c_int, [c_int, c_int]),
    'mg_version': (c_int, []),
    'mg_build_arch': (c_char_p, []),
    'mg_upsample_lengths': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_upsample_index': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_gather_rows_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'mg_gather_rows_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mg_segment_index': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_scatter_rows_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'mg_frame_layout': (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mg_pad_rows_colsum_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mg_pad_rows_colsum_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gather_concat_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'mg_gather_concat_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'mg_upsample_backward_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'mg_sequence_mask': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'mg_masked_mse_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mg_masked_mse_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    'mg_masked_bce_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    'mg_stream_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mg_stream_loss_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_pad_normalise_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p]),
    'mg_pad_normalise_bf16_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p, c_int, c_int, c_void_p]),
    'mg_normalise_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mg_linear_fwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                  c_int, c_int, c_void_p]),
    'mg_linear_dgrad_f32': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_linear_wgrad_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'mg_linear_wgrad_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                    c_int, c_void_p, c_size_t, c_void_p]),
    'mg_linear_fwd_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_int, c_int, c_int, c_void_p]),
    'mg_linear_dgrad_bf16': (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_int, c_int, c_void_p]),
    'mg_linear_wgrad_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                     c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_linear_wgrad_rows_bf16': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                          c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_linear_bwd_fused_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'mg_linear_bwd_fused_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                         c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_linear_wgrad_slabs_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_size_t,
                                           c_void_p, c_void_p, c_void_p]),
    'mg_linear_wgrad_dgrad_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                           c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'mg_linear_wgrad_dgrad_expand_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                                  c_void_p, c_size_t, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_int64,
                                                  c_int, c_void_p, c_int, c_void_p]),
    'mg_slab_reduce_f32': (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    'mg_adam_step_plan_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p,
                                      c_float, c_void_p, c_void_p]),
    'mg_cast_pad_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'mg_cast_transpose_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'mg_cast_params_bf16': (c_int, [c_void_p, c_int, c_void_p]),
    'mg_cast_bf16_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'mg_copy_many': (c_int, [c_void_p, c_int, c_void_p]),
    'mg_host_pack': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int]),
    'mg_split3_bf16': (c_int, [c_void_p, c_int, c_void_p]),
    'mg_phone_front_linear_fwd_x3': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int64, c_int, c_void_p,
                                             c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'mg_linear_fwd_x3_f32': (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'mg_f0_tail_rows_x3_workspace_bytes': (c_size_t, [c_int64]),
    'mg_f0_tail_rows_x3': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                   c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'mg_f0_l2tail_x3_workspace_bytes': (c_size_t, [c_int64]),
    'mg_f0_l2tail_x3': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'mg_linear_wgrad_dgrad_x3_colsum_floats': (c_size_t, [c_int64, c_int]),
    'mg_linear_wgrad_dgrad_x3': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'mg_linear_wgrad_slabs_x3': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p,
                                         c_void_p]),
    'mg_dropout': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, ctypes.c_uint64, ctypes.c_uint32, c_void_p, c_void_p]),
    'mg_dropout_advance': (c_int, [c_void_p, c_void_p, c_void_p]),
    'mg_philox4x32_10': (None, [c_void_p, c_void_p, c_void_p]),
    'mg_calib_mfma_bf16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'mg_sigmoid_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'mg_sigmoid_grad_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'mg_f0_tail_workspace_bytes': (c_size_t, [c_int64]),
    'mg_f0_tail_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                c_void_p]),
    'mg_linear_bwd_fused_slabs_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                               c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'mg_linear_bwd_fused2_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'mg_linear_bwd_fused2_slabs_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    'mg_f0_l2tail_slabs_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p,
                                        c_void_p]),
    'mg_f0_l2tail_rows_slabs_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int, c_void_p,
                                             c_size_t, c_void_p, c_void_p]),
    'mg_expand_column_reduce_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_int64,
                                            c_int, c_void_p, c_void_p]),
    'mg_f0_l2tail_workspace_bytes': (c_size_t, [c_int64]),
    'mg_f0_l2tail_slab_stride': (c_int64, []),
    'mg_f0_l2tail_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_size_t, c_void_p]),
    'mg_f0_l2tail_rows_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_gru_fwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    'mg_gru_fwd_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'mg_gru_bwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_persist_f32_supported': (c_int, [c_int, c_int, c_int]),
    'mg_lstm_fwd_persist_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    'mg_lstm_bwd_persist_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_persist_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mg_gru_persist_supported': (c_int, [c_int, c_int, c_int]),
    'mg_gru_persist_status': (c_int, [c_void_p, c_void_p]),
    'mg_gru_fwd_persist_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_fwd_persist_rows_bf16': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_fwd_persist_out_bf16': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_bwd_persist_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_persist_supported': (c_int, [c_int, c_int, c_int]),
    'mg_lstm_fwd_persist_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_bwd_persist_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_metric_workspace_bytes': (c_size_t, []),
    'mg_metric_accumulate_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    'mg_store_pair_f32': (c_int, [c_void_p, c_float, c_float, c_void_p]),
    'mg_store_pairs_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'mg_adam_scalars': (None, [c_float, c_float, c_float, c_int64, c_void_p]),
    'mg_adam_step_dev_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p,
                                     c_float, c_void_p]),
    'mg_f0_tail_rows_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_expand_column_loss_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'mg_expand_column_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'mg_phone_loss_const_add': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'mg_phone_target_stats_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mg_phone_target_stats': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_upsample_index_maps': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_phone_front': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_phone_front_linear_fwd_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int64, c_int, c_void_p,
                                               c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'mg_segment_bounds': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'mg_linear_dgrad_gathered_bf16': (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                              c_int, c_int, c_void_p]),
    'mg_segment_sum': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                               c_void_p]),
    'mg_segment_sum_feat_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mg_segment_sum_feat_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                         c_int, c_void_p, c_size_t, c_void_p]),
    'mg_feat_wgrad_reduce': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'mg_f0_tail_rows_f32_workspace_bytes': (c_size_t, [c_int64]),
    'mg_f0_tail_rows_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                    c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_phone_mse_rows_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'mg_phone_concat_layer_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                           c_void_p, c_int, c_int, c_void_p]),
    'mg_mlpg_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mg_mlpg_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                            c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_gru_small_supported': (c_int, [c_int]),
    'mg_gru_fwd_small_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mg_gru_bwd_small_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    'mg_gru_persist_f32_supported': (c_int, [c_int, c_int, c_int]),
    'mg_gru_fwd_persist_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    'mg_gru_bwd_persist_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_bwd_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mg_gru_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_fwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'mg_lstm_bwd_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mg_lstm_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mg_gru_stack_small_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'mg_gru_stack_small_workspace_bytes': (c_size_t, []),
    'mg_gru_stack_fwd_small_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_gru_stack_fwd_small_fast_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_gru_stack_bwd_small_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_gru_stack_bwd_small_fast_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_pstack_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'mg_lstm_pstack_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mg_lstm_pstack_fwd_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_pstack_bwd_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'mg_lstm_pstack_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mg_lstm_pstack_bwd_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mg_lstm_stack_fwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'mg_lstm_stack_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'mg_adam_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                 c_float, c_int64, c_float, c_void_p]),
    'mg_ema_update_f32': (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
}



class CastDesc(ctypes.Structure):
    """mg_cast_desc of include/morgana_hip.h."""
    _fields_ = [('src', c_void_p), ('rows', c_int), ('cols', c_int), ('dst', c_void_p), ('ldd', c_int),
                ('dst_t', c_void_p), ('ldt', c_int)]


class CopyDesc(ctypes.Structure):
    """mg_copy_desc of include/morgana_hip.h."""
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('bytes', c_int64)]


COPY_MAX = 16


class Split3Desc(ctypes.Structure):
    """mg_split3_desc of include/morgana_hip.h."""
    _fields_ = [('src', c_void_p), ('rows', c_int64), ('cols', c_int), ('lds', c_int), ('dst', c_void_p), ('ldp', c_int),
                ('order', c_int), ('transpose', c_int), ('plane_rows', c_int64), ('sig', c_void_p), ('ldsig', c_int), ('colsum', c_void_p),
                ('colsum_blocks', c_int)]


class StreamDesc(ctypes.Structure):
    """mg_stream_desc of include/morgana_hip.h."""
    _fields_ = [('target', c_void_p), ('ldt', c_int), ('col0', c_int), ('width', c_int), ('kind', c_int)]


class LstmFwdLayer(ctypes.Structure):
    """mg_lstm_fwd_layer of include/morgana_hip.h."""
    _fields_ = [('xproj', c_void_p), ('x_T', c_int), ('x_t0', c_int), ('w_hh', c_void_p), ('b_hh', c_void_p),
                ('hstate', c_void_p), ('cstate', c_void_p), ('out', c_void_p), ('saved', c_void_p)]


class LstmPStackLayer(ctypes.Structure):
    """mg_lstm_pstack_layer of include/morgana_hip.h."""
    _fields_ = [('xproj', c_void_p), ('w_ih_bf', c_void_p), ('b_ih', c_void_p), ('w_hh_bf', c_void_p), ('b_hh', c_void_p),
                ('hstate', c_void_p), ('cstate', c_void_p), ('hstate_bf', c_void_p), ('out', c_void_p), ('saved', c_void_p),
                ('ldwi', c_int), ('ldwh', c_int)]


class GruStackLayer(ctypes.Structure):
    """mg_gru_stack_layer of include/morgana_hip.h."""
    _fields_ = [('xproj', c_void_p), ('w_ih', c_void_p), ('b_ih', c_void_p), ('w_hh', c_void_p), ('b_hh', c_void_p),
                ('hstate', c_void_p), ('out', c_void_p), ('saved', c_void_p), ('grad_out', c_void_p), ('grad_hn', c_void_p),
                ('dxin', c_void_p), ('dxproj', c_void_p), ('dhproj', c_void_p), ('dh0', c_void_p)]


class LstmPStackBwdLayer(ctypes.Structure):
    """mg_lstm_pstack_bwd_layer of include/morgana_hip.h."""
    _fields_ = [('grad_out', c_void_p), ('grad_hn', c_void_p), ('grad_cn', c_void_p), ('cstate', c_void_p), ('saved', c_void_p),
                ('w_hh_t_bf', c_void_p), ('w_ih_up_t_bf', c_void_p), ('dgates', c_void_p), ('dgates_bf', c_void_p),
                ('dh0', c_void_p), ('dc0', c_void_p), ('ldt', c_int), ('ldt_up', c_int)]


class LstmBwdLayer(ctypes.Structure):
    """mg_lstm_bwd_layer of include/morgana_hip.h."""
    _fields_ = [('grad_out', c_void_p), ('g_T', c_int), ('g_t0', c_int), ('cstate', c_void_p), ('saved', c_void_p),
                ('w_hh', c_void_p), ('dgates', c_void_p), ('carry_h', c_void_p), ('carry_c', c_void_p), ('dh0', c_void_p),
                ('dc0', c_void_p)]


class AdamSlabSrc(ctypes.Structure):
    """mg_adam_slab_src of include/morgana_hip.h."""
    _fields_ = [('begin', c_int64), ('count', c_int64), ('slab', c_void_p), ('n_slabs', c_int), ('stride', c_int64)]


class AdamShadow(ctypes.Structure):
    """mg_adam_shadow of include/morgana_hip.h."""
    _fields_ = [('offset', c_int64), ('rows', c_int), ('cols', c_int), ('dst', c_void_p), ('ldd', c_int), ('dst_t', c_void_p),
                ('ldt', c_int), ('pair', c_int)]


ADAM_MAX_SLABS, ADAM_MAX_SHADOWS = 4, 8


class AdamTail(ctypes.Structure):
    """mg_adam_tail of include/morgana_hip.h."""
    _fields_ = [('table', c_void_p), ('rows', c_void_p), ('frames', c_int64), ('out', c_void_p), ('partial', c_void_p),
                ('n_partial', c_int), ('slab', c_void_p), ('n', c_int64), ('stride', c_int64), ('n_slabs', c_int), ('dst', c_void_p)]


class AdamPlan(ctypes.Structure):
    """mg_adam_plan of include/morgana_hip.h."""
    _fields_ = [('n_slab_srcs', c_int), ('slabs', AdamSlabSrc * ADAM_MAX_SLABS), ('n_shadows', c_int),
                ('shadows', AdamShadow * ADAM_MAX_SHADOWS), ('clear_grad', c_int), ('tail', AdamTail)]


LSTM_MAX_LAYERS = 8
CAST_MAX = 16
SPLIT3_MAX = 16
STREAMS_MAX = 8
LOSS_MSE, LOSS_SIGMOID_BCE = 0, 1
_lib = None


class MorganaHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library once and attach the signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('MORGANA_HIP_LIB', LIB_PATH)      # the diagnostic build (make diag) for scripts/stamps.py
    if not os.path.exists(path):
        raise MorganaHipError(
            'libmorgana_hip.so is missing (%s): build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C morgana_amd/csrc`.  There is no CPU fallback.' % path)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    for item in filter(None, os.environ.get('MG_TUNE', '').split(',')):      # experiments: MG_TUNE=key:value[,key:value]
        key, value = item.split(':')
        lib.mg_set_tuning(int(key), int(value))
    _lib = lib
    return lib


def last_error():
    return load().mg_last_error().decode('utf-8', 'replace')


CALL_LOG = None      # debugging aid: set to a list and every C-ABI call that went through check() appends its entry-point name


def check(rc, what):
    if CALL_LOG is not None:
        CALL_LOG.append(what)
    if rc != 0:
        msg = last_error()
        if rc == -1:
            raise ValueError('%s: %s' % (what, msg))
        raise MorganaHipError('%s failed (code %d): %s' % (what, rc, msg))
