"""ctypes binding of lib2ggcn_hip.so (C ABI declared in include/twog_gcn.h).

The product path has NO fallback: if the shared library is missing or an entry point is absent, importing/using the
kernels raises. Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C 2g-gcn_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TWOG_LIB_PATH: another build of the same library (measurement builds of tools/x3_ablate.sh); never a fallback -- a path
# that does not load raises exactly like a missing default library
LIB_PATH = os.environ.get('TWOG_LIB_PATH') or os.path.join(_HERE, 'lib2ggcn_hip.so')

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class Rows(C.Structure):  # twog_rows_t
    _fields_ = [('ptr', C.c_void_p), ('ld_outer', C.c_int64), ('ld_inner', C.c_int64), ('inner', C.c_int32),
                ('pad_', C.c_int32)]


class Gemm(C.Structure):  # twog_gemm_t
    _fields_ = [('A', Rows), ('B', Rows), ('C', Rows), ('bias', C.c_void_p), ('M', C.c_int32), ('N', C.c_int32),
                ('K', C.c_int32), ('act', C.c_int32), ('accumulate', C.c_int32), ('batch', C.c_int32),
                ('a_batch_stride', C.c_int64), ('b_batch_stride', C.c_int64), ('c_batch_stride', C.c_int64),
                ('a_colsum', C.c_void_p), ('a_colsum_accumulate', C.c_int32), ('pad2_', C.c_int32)]


GUARD_MAX = 16


class Guard(C.Structure):  # twog_guard_t
    _fields_ = [('words', C.c_void_p * GUARD_MAX), ('out', C.c_void_p * GUARD_MAX), ('n', C.c_int64 * GUARD_MAX),
                ('n_words', C.c_int32), ('n_out', C.c_int32)]


class GruStep(C.Structure):  # twog_gru_step_t
    _fields_ = [('gi', Rows), ('gi2', Rows), ('gh', Rows), ('h_prev', Rows), ('h_out', Rows), ('save', Rows),
                ('u', C.c_void_p), ('u_ld_outer', C.c_int64), ('u_ld_inner', C.c_int64), ('u_inner', C.c_int32),
                ('rows', C.c_int32), ('hidden', C.c_int32), ('pad_', C.c_int32)]


class GruStepBwd(C.Structure):  # twog_gru_step_bwd_t
    _fields_ = [('dh', Rows), ('dh2', Rows), ('save', Rows), ('h_prev', Rows), ('dgi', Rows), ('dgh', Rows),
                ('dh_prev', Rows), ('u', C.c_void_p), ('du', C.c_void_p), ('u_ld_outer', C.c_int64),
                ('u_ld_inner', C.c_int64), ('u_inner', C.c_int32), ('rows', C.c_int32), ('hidden', C.c_int32),
                ('dh_prev_accumulate', C.c_int32)]


class BiGru(C.Structure):  # twog_bigru_t
    _fields_ = [('gi', C.c_void_p), ('w_hh_f', C.c_void_p), ('b_hh_f', C.c_void_p), ('w_hh_r', C.c_void_p),
                ('b_hh_r', C.c_void_p), ('out', C.c_void_p), ('save', C.c_void_p), ('tmp_gh', C.c_void_p),
                ('zeros', C.c_void_p), ('E', C.c_int32), ('pad_', C.c_int32)]


class BiGruBwd(C.Structure):  # twog_bigru_bwd_t
    _fields_ = [('d_out', C.c_void_p), ('save', C.c_void_p), ('out', C.c_void_p), ('w_hh_f', C.c_void_p),
                ('w_hh_r', C.c_void_p), ('d_gi', C.c_void_p), ('d_gh', C.c_void_p), ('carry', C.c_void_p),
                ('E', C.c_int32), ('pad_', C.c_int32)]


class Attn(C.Structure):  # twog_attn_t
    _fields_ = [('feat_h', Rows), ('feat_o', Rows), ('msg_hh', Rows), ('msg_ho', Rows), ('msg_oh', Rows),
                ('msg_oo', Rows), ('msg_so', Rows), ('msg_sh', Rows), ('out_hh', Rows), ('out_oh', Rows),
                ('out_sh', Rows), ('out_ho', Rows), ('out_so', Rows), ('out_oo', Rows), ('obj_mask', C.c_void_p),
                ('att', C.c_void_p), ('n_inst', C.c_int32), ('inst_per_clip', C.c_int32), ('H', C.c_int32),
                ('O', C.c_int32), ('D', C.c_int32), ('hidden', C.c_int32), ('scale', C.c_float),
                ('recv_mask_ho', C.c_int32)]


class AttnBwd(C.Structure):  # twog_attn_bwd_t
    _fields_ = [('f', Attn), ('dout_hh', Rows), ('dout_oh', Rows), ('dout_sh', Rows), ('dout_ho', Rows),
                ('dout_so', Rows), ('dout_oo', Rows), ('dmsg_hh', Rows), ('dmsg_ho', Rows), ('dmsg_oh', Rows),
                ('dmsg_oo', Rows), ('dmsg_so', Rows), ('dmsg_sh', Rows), ('dfeat_h', Rows), ('dfeat_o', Rows),
                ('dw_extra', C.c_void_p), ('dfeat_accumulate', C.c_int32), ('relu_mask_dmsg', C.c_int32)]


class SegRnn(C.Structure):  # twog_segrnn_t
    _fields_ = [('bs', C.c_int32), ('T', C.c_int32), ('H', C.c_int32), ('O', C.c_int32), ('hidden', C.c_int32),
                ('msg_segment', C.c_int32), ('rel_hh', C.c_int32), ('rel_ho', C.c_int32), ('rel_oh', C.c_int32),
                ('rel_oo', C.c_int32), ('att_scale', C.c_float), ('pad_', C.c_int32),
                ('gi_h', C.c_void_p), ('gi_o', C.c_void_p), ('u_h', C.c_void_p), ('u_o', C.c_void_p),
                ('obj_mask', C.c_void_p),
                ('w_hh_h', C.c_void_p * 2), ('b_hh_h', C.c_void_p * 2), ('w_hh_o', C.c_void_p * 2),
                ('b_hh_o', C.c_void_p * 2), ('w_ihm_h', C.c_void_p * 2), ('w_ihm_o', C.c_void_p * 2),
                ('ld_ih_h', C.c_int64), ('ld_ih_o', C.c_int64),
                ('w_smsg_h', C.c_void_p), ('b_smsg_h', C.c_void_p), ('w_smsg_o', C.c_void_p), ('b_smsg_o', C.c_void_p),
                ('hs_h', C.c_void_p), ('hs_o', C.c_void_p), ('save_h', C.c_void_p), ('save_o', C.c_void_p),
                ('msrc_h', C.c_void_p), ('msrc_o', C.c_void_p), ('mg_h', C.c_void_p), ('mg_o', C.c_void_p),
                ('att', C.c_void_p), ('tmp_gim_h', C.c_void_p), ('tmp_gim_o', C.c_void_p), ('tmp_gh_h', C.c_void_p),
                ('tmp_gh_o', C.c_void_p), ('zeros', C.c_void_p)]


class SegRnnBwd(C.Structure):  # twog_segrnn_bwd_t
    _fields_ = [('d_hs_h', C.c_void_p), ('d_hs_o', C.c_void_p), ('d_gi_h', C.c_void_p), ('d_gi_o', C.c_void_p),
                ('d_gh_h', C.c_void_p), ('d_gh_o', C.c_void_p), ('d_u_h', C.c_void_p), ('d_u_o', C.c_void_p),
                ('d_pre_h', C.c_void_p), ('d_pre_o', C.c_void_p), ('carry_h', C.c_void_p), ('carry_o', C.c_void_p),
                ('tmp_dmg_h', C.c_void_p), ('tmp_dmg_o', C.c_void_p), ('trash', C.c_void_p), ('du_part_h', C.c_void_p),
                ('du_part_o', C.c_void_p)]


class Gate(C.Structure):  # twog_gate_t
    _fields_ = [('x', Rows), ('seg_col', C.c_int32 * 8), ('n_seg', C.c_int32), ('hidden', C.c_int32),
                ('w', C.c_void_p), ('b', C.c_void_p), ('noise', C.c_void_p), ('hard', C.c_void_p),
                ('soft', C.c_void_p), ('p_save', C.c_void_p), ('bs', C.c_int32), ('T', C.c_int32), ('E', C.c_int32),
                ('noise_entities', C.c_int32), ('noise_offset', C.c_int32), ('force_last', C.c_int32),
                ('threshold', C.c_float), ('pad_', C.c_int32)]


class Loss(C.Structure):  # twog_loss_t
    _fields_ = [('kind', C.c_int32), ('n_classes', C.c_int32), ('outer', C.c_int64), ('inner', C.c_int64),
                ('input', C.c_void_p), ('target', C.c_void_p), ('dinput', C.c_void_p), ('weight', C.c_float),
                ('ignore_value', C.c_float)]


class Relation(C.Structure):  # twog_relation_t
    _fields_ = [('q', Rows), ('k', Rows), ('msg', Rows), ('p_r', Rows), ('p_s', Rows), ('out', Rows),
                ('a_r', C.c_void_p), ('c_s', C.c_void_p), ('dist', C.c_void_p), ('dist_ld_inst', C.c_int64),
                ('dist_ld_r', C.c_int64), ('dist_ld_s', C.c_int64), ('send_mask', C.c_void_p),
                ('recv_mask', C.c_void_p), ('att', C.c_void_p), ('score_bias', C.c_void_p), ('scale', C.c_float),
                ('score_mode', C.c_int32), ('msg_mode', C.c_int32), ('relu_scores', C.c_int32),
                ('exclude_self', C.c_int32), ('n_inst', C.c_int32), ('inst_per_clip', C.c_int32), ('R', C.c_int32),
                ('S', C.c_int32), ('D', C.c_int32), ('hidden', C.c_int32), ('pad_', C.c_int32)]


class RowOp(C.Structure):  # twog_rowop_t
    _fields_ = [('a', Rows), ('b', Rows), ('dst', Rows), ('s', C.c_void_p), ('v', C.c_void_p), ('kind', C.c_int32),
                ('rows', C.c_int32), ('cols', C.c_int32), ('pad_', C.c_int32)]


class ColSum(C.Structure):  # twog_colsum_t
    _fields_ = [('x', Rows), ('rowscale', C.c_void_p), ('out', C.c_void_p), ('rows', C.c_int32), ('cols', C.c_int32),
                ('accumulate', C.c_int32), ('pad_', C.c_int32)]


COLSUM_MAX = 16   # TWOG_COLSUM_MAX


class Copy(C.Structure):  # twog_copy_t
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('n', C.c_int64)]


COPY_MAX = 16   # TWOG_COPY_MAX
PERSIST_NOT_RESIDENT = -3   # TWOG_PERSIST_NOT_RESIDENT

TAPE_GEMM, TAPE_RELATION_FWD, TAPE_RELATION_BWD, TAPE_GRU_STEP_FWD, TAPE_GRU_STEP_BWD, TAPE_ROWOPS = range(6)   # TWOG_TAPE_*


class TapeEntry(C.Structure):  # twog_tape_entry_t
    _fields_ = [('kind', C.c_int32), ('n', C.c_int32), ('flags', C.c_int32), ('pad_', C.c_int32), ('desc', C.c_void_p)]


class RelationBwd(C.Structure):  # twog_relation_bwd_t
    _fields_ = [('f', Relation), ('dout', Rows), ('dmsg', Rows), ('dp_r', Rows), ('dp_s', Rows), ('dq', Rows),
                ('dk', Rows), ('da_r', C.c_void_p), ('dc_s', C.c_void_p), ('dscore_sum', C.c_void_p),
                ('dq_accumulate', C.c_int32),
                ('dk_accumulate', C.c_int32), ('relu_mask_dmsg', C.c_int32), ('pad_', C.c_int32)]


REL_SUM, REL_DOT, REL_ADDITIVE, REL_DISTANCE, REL_MEAN = 0, 1, 2, 3, 4   # TWOG_REL_*
REL_MSG_SENDER, REL_MSG_PAIR = 0, 1

LOSS_MAX_TERMS, LOSS_BLOCKS = 16, 64  # TWOG_LOSS_MAX_TERMS, TWOG_LOSS_BLOCKS

# name -> (argtypes) ; every function returns int except twog_version
_I, _L, _F, _P = C.c_int, C.c_int64, C.c_float, C.c_void_p
SIGNATURES = {
    'twog_gemm_f32': [C.POINTER(Gemm), _I, _I, _I, _P, C.c_size_t, _P],
    'twog_guard_outputs': [C.POINTER(Guard), _P],
    'twog_stream_create_masked': [_I, C.POINTER(C.c_void_p)],
    'twog_stream_create_low_priority': [C.POINTER(C.c_void_p)],
    'twog_stream_destroy': [_P],
    'twog_gemm_colsum_fused': [C.POINTER(Gemm), _I, _I, _I, _P, C.c_size_t],
    'twog_gemm_last_class': [],
    'twog_chain_workspace_bytes': [],
    'twog_gemm_f32_chain': [C.POINTER(Gemm), _I, _I, _I, _P, C.c_size_t, _P],
    'twog_gcn_max_nodes': [],
    'twog_bn_stats': [_P, _L, _I, _I, _P, _I, _P],
    'twog_bn_finalize': [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    'twog_gcn_embed1_fwd': [_P, _L, _I, _I, _P, _P, _P, _P, _P],
    'twog_gcn_fused_fwd': [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'twog_gcn_embed1_bwd': [_P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    'twog_gcn_attn_fwd': [_P, _P, _I, _I, _P, _P, _P],
    'twog_gcn_attn_bwd': [_P, _P, _P, _P, _I, _I, _P, _P, _P],
    'twog_gcn_attn2_fwd': [_P, _P, _I, _I, _P, _P, _P],
    'twog_gcn_attn2_bwd_blocks': [_I],
    'twog_gcn_attn2_bwd': [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P],
    'twog_gru_step_fwd': [C.POINTER(GruStep), _I, _P],
    'twog_gru_step_bwd': [C.POINTER(GruStepBwd), _I, _P],
    'twog_bigru_fwd': [C.POINTER(BiGru), _I, _I, _I, _I, _P, C.c_size_t, _P],
    'twog_bigru_persistent_supported': [C.POINTER(BiGru), _I, _I, _I],
    'twog_bigru_fwd_persistent': [C.POINTER(BiGru), _I, _I, _I, _I, _P, _P],
    'twog_bigru_bwd_persistent_supported': [C.POINTER(BiGruBwd), _I, _I, _I],
    'twog_bigru_bwd_persistent': [C.POINTER(BiGruBwd), _I, _I, _I, _I, _P, _P],
    'twog_bigru_bwd': [C.POINTER(BiGruBwd), _I, _I, _I, _I, _P, C.c_size_t, _P],
    'twog_attn_fwd': [C.POINTER(Attn), _I, _P],
    'twog_attn_limits': [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    'twog_attn_bwd': [C.POINTER(AttnBwd), _I, _P],
    'twog_segrnn_fwd': [C.POINTER(SegRnn), _P, C.c_size_t, _P],
    'twog_segrnn_bwd': [C.POINTER(SegRnn), C.POINTER(SegRnnBwd), _P, C.c_size_t, _P],
    'twog_segrnn_persistent_supported': [C.POINTER(SegRnn)],
    'twog_segrnn_persistent_sync_bytes': [],
    'twog_segrnn_fwd_persistent': [C.POINTER(SegRnn), _P, _P],
    'twog_segrnn_bwd_persistent_scratch_bytes': [C.POINTER(SegRnn)],
    'twog_segrnn_bwd_persistent': [C.POINTER(SegRnn), C.POINTER(SegRnnBwd), _P, C.c_size_t, _P, _P],
    'twog_graph_cache_stats': [c_int64_p, c_int64_p],
    'twog_pos_embed_fwd': [_P, _P, _I, _I, _I, _I, _P, _P, _I, _I, Rows, _P, _P],
    'twog_periodic_embed_bwd': [Rows, _P, _I, _I, _P, _P],
    'twog_seglen_fwd': [_P, _P, _I, _I, _I, _I, _P, _P],
    'twog_seglen_bwd': [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    'twog_mul': [_P, _P, _P, _L, _I, _P],
    'twog_scale_rows': [Rows, _P, _I, _I, _P],
    'twog_relation_limits': [],
    'twog_relation_fwd': [C.POINTER(Relation), _P],
    'twog_relation_bwd': [C.POINTER(RelationBwd), _P],
    'twog_relation_fwd_n': [C.POINTER(Relation), _I, _P],
    'twog_relation_bwd_n': [C.POINTER(RelationBwd), _I, _P],
    'twog_ssp_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'twog_ssp_gather': [_P, _L, _P, _L, _L, _I, _P, _I, _I, _I, _I, _I, _P],
    'twog_ssp_bwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'twog_gate_fwd': [C.POINTER(Gate), _P],
    'twog_gate_bwd': [C.POINTER(Gate), _P, _P, _P, _P, _P],
    'twog_rank1_update': [Rows, _P, _P, _I, _I, _P],
    'twog_colsum': [Rows, _P, _I, _I, _P, _I, _P, _I, _P],
    'twog_colsum_n_partial_floats': [C.POINTER(ColSum), _I],
    'twog_colsum_n': [C.POINTER(ColSum), _I, _P, C.c_size_t, _P],
    'twog_filter_fwd': [_P, _P, _P, _I, _I, _I, _F, _P],
    'twog_reorder_fwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'twog_reorder_bwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'twog_logsoftmax_permute_fwd': [_P, _P, _I, _I, _I, _I, _P],
    'twog_logsoftmax_permute_bwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'twog_relu_bwd': [Rows, Rows, Rows, _I, _I, _P],
    'twog_add_rows': [Rows, Rows, _I, _I, _P],
    'twog_rowops': [C.POINTER(RowOp), _I, _P],
    'twog_tape_run': [C.POINTER(TapeEntry), C.POINTER(TapeEntry), _I, _I, _I, _P, C.c_size_t, _P],
    'twog_fill_zero': [_P, C.c_size_t, _P],
    'twog_copy_blocks': [C.POINTER(Copy), _I, _P],
    'twog_debug_occupy': [_I, _I, _I, _P],
    'twog_adam_step': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    'twog_multitask_loss_fwd': [C.POINTER(Loss), _I, _P, _P, _P, _P],
    'twog_multitask_loss_bwd': [C.POINTER(Loss), _I, _P, _P, _P],
    'twog_predict_labels': [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    'twog_f1_at_k': [_P, _P, _I, _I, _I, C.c_double, _L, _I, _P, _P, _P, _P],
}

_lib = None


def load():
    """Load lib2ggcn_hip.so and bind every entry point of include/twog_gcn.h. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: the HIP kernels are not built (run __graft_entry__.build() or '
                           f'`make -C 2g-gcn_amd/csrc`). There is no fallback path.')
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.twog_chain_workspace_bytes.restype = C.c_size_t
    lib.twog_colsum_n_partial_floats.restype = C.c_size_t
    lib.twog_segrnn_persistent_sync_bytes.restype = C.c_size_t
    lib.twog_segrnn_bwd_persistent_scratch_bytes.restype = C.c_size_t
    lib.twog_version.restype = C.c_char_p
    lib.twog_version.argtypes = []
    _lib = lib
    return lib


def exported_symbols():
    return ['twog_version'] + list(SIGNATURES.keys())
