"""ctypes binding of libprodsearch_hip.so (the C ABI in include/prodsearch_hip.h).

This is the reference-side stub INTEGRATION.md shows: plain pointers and sizes,
no torch types cross the boundary.  There is NO CPU fallback: if the library is
missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

from . import build as _build

PS_MAX_LAYERS = 8
PS_MODEL_TEM, PS_MODEL_QEM = 0, 1
PS_QENC_FS, PS_QENC_AVG = 0, 1

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)


class PsTemDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'K', 'L', 'Q', 'W', 'C', 'd', 'H', 'F', 'n_layers')] + \
               [('product_size', C.c_int64), ('vocab_size', C.c_int64)] + \
               [(n, C.c_int32) for n in ('model', 'query_encoder', 'use_pos_emb', 'use_item_pos',
                                         'bias_product', 'pos_weight', 'sep_prod_emb', 'training')] + \
               [('dropout', C.c_float), ('seed', C.c_uint64), ('step', C.c_uint64)]


LAYER_FIELDS = ('wk', 'bk', 'wv', 'bv', 'wq', 'bq', 'wo', 'bo', 'w1', 'b1', 'w2', 'b2',
                'ff_ln_g', 'ff_ln_b', 'ln_g', 'ln_b')
TOP_FIELDS = ('product_emb', 'hist_product_emb', 'word_emb', 'product_bias', 'word_bias',
              'fs_w', 'fs_b', 'pe', 'final_ln_g', 'final_ln_b')


class PsLayerTensors(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LAYER_FIELDS]


class PsTemTensors(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TOP_FIELDS] + [('layer', PsLayerTensors * PS_MAX_LAYERS)]


class PsTemBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs',
                                          'pos_iword_idxs', 'neg_item_idxs', 'neg_word_idxs',
                                          'candi_prod_idxs')]


class PsTemWsLayout(C.Structure):
    _fields_ = [('total_floats', C.c_int64), ('R', C.c_int32), ('S', C.c_int32)] + \
               [(n, C.c_int64) for n in ('qmean', 'query_emb', 'x', 'kp', 'vp', 'qp', 'attn', 'ctx',
                                         'y1', 'ln1', 'a1', 'h1', 'y2', 'enc',
                                         'item_scores', 'word_scores', 'loss_parts', 'denc', 'dx')]


class PsAdamHyper(C.Structure):
    _fields_ = [('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
                ('weight_decay', C.c_float), ('max_grad_norm', C.c_float), ('noam', C.c_int32),
                ('warmup_steps', C.c_int32), ('grad_scale', C.c_float), ('zero_grads', C.c_int32), ('method', C.c_int32),
                ('pad_', C.c_int32)]


class PsIdxList(C.Structure):
    _fields_ = [('idx', C.c_void_p), ('n', C.c_int64)]


class PsRowTable(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p),
                ('rows', C.c_void_p), ('count', C.c_void_p), ('cap', C.c_int64), ('d', C.c_int32),
                ('pad_', C.c_int32)]


class PsRtmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'K', 'R', 'Q', 'W', 'WL', 'C', 'd', 'H', 'F', 'n_layers')] + \
               [('vocab_size', C.c_int64), ('review_count', C.c_int64)] + \
               [(n, C.c_int32) for n in ('review_encoder', 'query_encoder', 'use_pos_emb', 'use_seg_emb',
                                         'pos_weight', 'train_pv', 'training')] + \
               [('dropout', C.c_float), ('corrupt_rate', C.c_float), ('seed', C.c_uint64), ('step', C.c_uint64)] + \
               [('use_user_emb', C.c_int32), ('use_item_emb', C.c_int32), ('user_size', C.c_int64),
                ('product_size', C.c_int64)]


RTM_TOP_FIELDS = ('word_emb', 'review_emb', 'seg_emb', 'fs_w', 'fs_b', 'pe', 'final_ln_g', 'final_ln_b',
                  'wo_w', 'wo_b', 'user_emb', 'product_emb', 'rev_fs_w', 'rev_fs_b')
RTM_BATCH_FIELDS = ('query_word_idxs', 'pos_prod_ridxs', 'pos_seg_idxs', 'pos_prod_rword_idxs',
                    'pos_prod_rword_masks', 'neg_prod_ridxs', 'neg_seg_idxs', 'neg_prod_rword_idxs',
                    'pos_prod_rword_idxs_pvc', 'neg_prod_rword_idxs_pvc', 'neg_word_idxs',
                    'candi_prod_ridxs', 'candi_seg_idxs', 'review_embeddings', 'pos_user_idxs', 'neg_user_idxs',
                    'pos_item_idxs', 'neg_item_idxs', 'candi_seq_user_idxs', 'candi_seq_item_idxs', 'neg_prod_rword_masks')


class PsRtmTensors(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in RTM_TOP_FIELDS] + [('layer', PsLayerTensors * PS_MAX_LAYERS)]


class PsRtmBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in RTM_BATCH_FIELDS]


class PsRtmWsLayout(C.Structure):
    _fields_ = [('total_floats', C.c_int64)] + [(n, C.c_int32) for n in ('Bseq', 'S', 'J', 'pad_')] + \
               [(n, C.c_int64) for n in ('query_emb', 'valid', 'x', 'vec', 'cnt', 'enc', 'scores', 'weight',
                                         'pv_scores', 'dx')]


PS_RENC_PV, PS_RENC_PVC, PS_RENC_FS, PS_RENC_AVG = 0, 1, 2, 3

# every symbol include/prodsearch_hip.h declares: (restype, argtypes)
SYMBOLS = {
    'ps_version': (C.c_char_p, []),
    'ps_last_error': (C.c_char_p, []),
    'ps_arith_info': (C.c_char_p, []),
    'ps_set_fuse_bwd_min': (C.c_int, [C.c_int]),
    'ps_set_side_mode': (C.c_int, [C.c_int]),
    'ps_side_values_in_use': (C.c_int, []),
    'ps_set_deterministic': (C.c_int, [C.c_int]),
    'ps_side_abort': (None, []),
    'ps_debug_fail_fork': (None, [C.c_int]),
    'ps_gemm_x3_config': (C.c_int, [C.c_int, C.c_int]),
    'ps_gemm_f32_weight': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    'ps_tem_workspace_layout': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemWsLayout)]),
    'ps_tem_forward': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_tem_forward_sampled': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    'ps_tem_backward': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                                  C.c_void_p, C.POINTER(PsTemTensors), C.c_float, C.c_void_p, C.c_void_p]),
    'ps_gather_score': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                                  C.c_void_p, C.c_void_p]),
    'ps_tem_score': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rtm_workspace_floats': (C.c_int, [C.POINTER(PsRtmDesc), C.c_int32, C.POINTER(C.c_int64)]),
    'ps_rtm_workspace_layout': (C.c_int, [C.POINTER(PsRtmDesc), C.c_int32, C.POINTER(PsRtmWsLayout)]),
    'ps_rtm_forward': (C.c_int, [C.POINTER(PsRtmDesc), C.POINTER(PsRtmTensors), C.POINTER(PsRtmBatch),
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rtm_backward': (C.c_int, [C.POINTER(PsRtmDesc), C.POINTER(PsRtmTensors), C.POINTER(PsRtmBatch),
                                  C.c_void_p, C.POINTER(PsRtmTensors), C.c_float, C.c_void_p, C.c_void_p]),
    'ps_rtm_score': (C.c_int, [C.POINTER(PsRtmDesc), C.POINTER(PsRtmTensors), C.POINTER(PsRtmBatch),
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rtm_review_embeddings': (C.c_int, [C.POINTER(PsRtmDesc), C.POINTER(PsRtmTensors), C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_sample_negatives': (C.c_int, [C.POINTER(PsTemDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    'ps_build_alias_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'ps_adam_plan_bytes': (C.c_int64, [C.c_int32, C.c_void_p]),
    'ps_adam_plan_write_host': (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    'ps_adam_plan_chunks_host': (C.c_int32, [C.c_void_p]),
    'ps_clip_adam_dense': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsAdamHyper), C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'ps_adam_sumsq': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsAdamHyper), C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_adam_update_ext': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsAdamHyper), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    'ps_dropout_mult_host': (C.c_float, [C.POINTER(PsTemDesc), C.c_uint32, C.c_uint32, C.c_uint32]),
    'ps_zero_floats': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_sum_slices': (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_graph_replay_enabled': (C.c_int, []),
    'ps_tem_staged_batch': (C.c_int, [C.POINTER(PsTemDesc), C.c_void_p, C.POINTER(PsTemBatch)]),
    'ps_tem_forward_step': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_tem_backward_step': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.c_void_p, C.POINTER(PsTemTensors),
                                       C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_tem_encode': (C.c_int, [C.POINTER(PsTemDesc), C.POINTER(PsTemTensors), C.POINTER(PsTemBatch),
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rank_scratch_bytes': (C.c_int64, [C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    'ps_rank_all': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_rank_shard': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_void_p]),
    'ps_coalesce_ws_bytes': (C.c_int64, [C.c_int64]),
    'ps_coalesce_rows': (C.c_int, [C.POINTER(PsIdxList), C.c_int32, C.c_int64, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'ps_gather_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'ps_scatter_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'ps_zero_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_coalesce_bad_flag': (C.c_void_p, [C.c_void_p, C.c_int64]),
    'ps_pack_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    'ps_merge_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_adam_rowsparse_state_floats': (C.c_int64, [C.c_int32, C.POINTER(PsRowTable), C.c_int32]),
    'ps_clip_adam_rowsparse': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsRowTable), C.c_int32,
                                         C.POINTER(PsAdamHyper), C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_shard_bucket': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_shard_remap': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    'ps_rowsparse_sumsq': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsRowTable), C.c_int32, C.c_int32, C.POINTER(PsAdamHyper),
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rowsparse_update_ext': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PsRowTable), C.c_int32, C.POINTER(PsAdamHyper),
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ps_rowsparse_catchup': (C.c_int, [C.POINTER(PsRowTable), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                       C.POINTER(PsAdamHyper), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'ps_ktimer_arm': (C.c_int, [C.c_char_p, C.c_int32]),
    'ps_ktimer_read': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    'ps_gemm_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                              C.c_void_p]),
}

class PsCorpusView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('n_reviews', 'n_users', 'n_products', 'n_queries')] + \
               [(n, C.c_void_p) for n in ('review_u_p', 'u_seq_ptr', 'u_seq', 'train_review', 'review_loc',
                                          'pq_ptr', 'pq_idx', 'query_words')] + \
               [('Q', C.c_int32), ('pad_', C.c_int32)]


class PsCollateArgs(C.Structure):
    _fields_ = [('uprev_review_limit', C.c_int32), ('do_seq', C.c_int32), ('fix', C.c_int32), ('pad_', C.c_int32),
                ('prod_pad', C.c_int64)]


class PsTrainSlot(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('query_words', 'target', 'u_items', 'pos_words', 'query_idx', 'user_idx', 'hist_len')]


class PsRtmCorpusView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('n_reviews', 'n_users', 'n_products', 'n_queries')] + \
               [(n, C.c_void_p) for n in ('review_u_p', 'u_seq_ptr', 'u_seq', 'i_seq_ptr', 'i_seq', 'ut_seq_ptr', 'ut_seq',
                                          'it_seq_ptr', 'it_seq', 'loc_time', 'pq_ptr', 'pq_idx', 'query_words')] + \
               [('Q', C.c_int32), ('pad_', C.c_int32)]


class PsRtmCollateArgs(C.Structure):
    _fields_ = [('uprev_review_limit', C.c_int32), ('iprev_review_limit', C.c_int32), ('do_seq', C.c_int32),
                ('neg_per_pos', C.c_int32), ('user_pad', C.c_int64), ('prod_pad', C.c_int64), ('review_pad', C.c_int64)]


# include/prodsearch_data.h (host-only library)
DATA_SYMBOLS = {
    'ps_rtm_collate_train': (C.c_int, [C.POINTER(PsRtmCorpusView), C.POINTER(PsRtmCollateArgs), C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_int64] + [C.c_void_p] * 11),
    'ps_rtm_collate_test': (C.c_int, [C.POINTER(PsRtmCorpusView), C.POINTER(PsRtmCollateArgs), C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p] + [C.c_void_p] * 7),
    'ps_rtm_pv_windows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'ps_rtm_word_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'ps_rng_get_state': (None, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    'ps_rng_set_state': (None, [C.c_void_p, C.c_void_p, C.c_int32]),
    'ps_rng_np_interval': (C.c_uint64, [C.c_void_p, C.c_uint64]),
    'ps_rng_create': (C.c_void_p, [C.c_uint64]),
    'ps_rng_destroy': (None, [C.c_void_p]),
    'ps_rng_seed': (None, [C.c_void_p, C.c_uint64]),
    'ps_rng_randbelow': (C.c_uint32, [C.c_void_p, C.c_uint32]),
    'ps_rng_random': (C.c_double, [C.c_void_p]),
    'ps_collate_train': (C.c_int, [C.POINTER(PsCorpusView), C.POINTER(PsCollateArgs), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 8),
    'ps_collate_test': (C.c_int, [C.POINTER(PsCorpusView), C.POINTER(PsCollateArgs), C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6),
    'ps_epoch_start': (C.c_void_p, [C.POINTER(PsCorpusView), C.POINTER(PsCollateArgs), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    'ps_epoch_next': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'ps_epoch_release': (C.c_int, [C.c_void_p, C.c_int32]),
    'ps_epoch_stop': (None, [C.c_void_p]),
    'ps_rng_shuffle': (None, [C.c_void_p, C.c_void_p, C.c_int64]),
    'ps_collect_train_samples': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p]),
    'ps_data_last_error': (C.c_char_p, []),
}

_lib = None
_data_lib = None


def load_data():
    """Load the host-side batch builder (builds it with g++ on first use; needs no GPU)."""
    global _data_lib
    if _data_lib is not None:
        return _data_lib
    from . import build as _build
    path = _build.build_data()
    lib = C.CDLL(path)
    for name, (res, args) in DATA_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _data_lib = lib
    return lib


def check_data(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, load_data().ps_data_last_error().decode('utf-8', 'replace')))


def lib_path():
    """The shipped library; with PS_DIAG_LIB=1 the diagnostic build (`python -m prodsearch_amd.build --diag`: the same sources
    with -DPS_DIAG, which alone contains the tuning knobs, in-kernel stamps and timing-only kernel variants — tools/ only)."""
    return _build.DIAG_LIB if os.environ.get('PS_DIAG_LIB') == '1' else _build.LIB


def load():
    """dlopen the in-tree library (never builds implicitly on a GPU box: the .so travels)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libprodsearch_hip.so not found at %s: run `python -m prodsearch_amd.build` "
                           "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ps_last_error().decode('utf-8', 'replace')
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device/host address of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()
