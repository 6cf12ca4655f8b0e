"""ctypes binding of libptt_hip.so (the C ABI declared in include/ptt_hip.h).

There is no fallback: if the shared library is missing or fails to load, every op raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libptt_hip.so")

PTT_SA_MAX_LAYERS = 4
ABI_VERSION = 21            # PTT_ABI_VERSION of include/ptt_hip.h these structures mirror

# every symbol include/ptt_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "ptt_version", "ptt_error_name", "ptt_last_error_string",
    "ptt_fps_f32", "ptt_fps_ws_f32", "ptt_spatial_order_f32", "ptt_ball_query_grid_workspace", "ptt_ball_query_grid_f32",
    "ptt_centres_ball_query_grid_f32", "ptt_gather_f32", "ptt_gather_grad_f32", "ptt_select_centres_f32", "ptt_ball_query_f32",
    "ptt_group_f32", "ptt_group_grad_f32", "ptt_scatter_add_det_workspace", "ptt_scatter_add_det_f32",
    "ptt_knn_f32", "ptt_knn_rel_f32",
    "ptt_packed_weight_elems", "ptt_pack_weight_f32", "ptt_pack_weight_rot_f32", "ptt_linear_f32",
    "ptt_sa_fused_fwd_f32", "ptt_xcorr_fused_fwd_f32", "ptt_cosine_map_f32", "ptt_pt_attn_pair_f32",
    "ptt_crop_compact_f32", "ptt_regularize_f32", "ptt_mt19937_fill", "ptt_select_box_f32",
    "ptt_track_crop_bounds", "ptt_track_box_by_offset",
    "ptt_bn_stats_workspace", "ptt_bn_stats_f32", "ptt_bn_apply_f32", "ptt_bn_bwd_f32", "ptt_pool_rows_f32",
    "ptt_pool_rows_bwd_f32", "ptt_linear_wgrad_workspace", "ptt_linear_wgrad_f32",
    "ptt_pack_weight_strided_f32", "ptt_linear_batched_f32", "ptt_softmax_rows_f32",
    "ptt_gather_rows_f32", "ptt_scatter_csr_i32", "ptt_scatter_rows_csr_f32",
    "ptt_pt_pair_input_f32", "ptt_pt_attn_train_fwd_f32", "ptt_pt_attn_train_bwd_f32", "ptt_linear_act_in_f32",
    "ptt_centres_ball_query_f32",
    "ptt_bn_sums_f64", "ptt_bn_finish_f64", "ptt_bn_bwd_sums_f64", "ptt_bn_bwd_apply_f32",
    "ptt_rows_mlp_f32",
    "ptt_rows_gemm_supported", "ptt_rows_gemm_stat_chunks", "ptt_rows_gemm_f32", "ptt_rows_gemm_masked_f32", "ptt_bn_finish_partials_f32",
    "ptt_bn_sums_partials_f64", "ptt_linear_wgrad2_workspace", "ptt_linear_wgrad2_f32",
    "ptt_bn_bwd_pooled_f32", "ptt_bn_bwd_pooled_sums_f64", "ptt_bn_bwd_pooled_apply_f32",
    "ptt_pt_pair_input_ld_f32", "ptt_pt_attn_fwd_ld_f32",
    "ptt_rows_gemm_bnbwd_f32", "ptt_bn_bwd_from_partials_f32", "ptt_bn_bwd_sums_partials_f64",
    "ptt_bn_bwd_consts_f32", "ptt_bn_bwd_pooled_consts_f32", "ptt_rows_gemm_bnbwd_fused_supported", "ptt_rows_gemm_bnbwd_fused_f32",
    "ptt_bn_update_running_f32", "ptt_xcorr_z0_f32", "ptt_xcorr_z0_stat_chunks", "ptt_xcorr_z0_stats_f32", "ptt_xcorr_z0_bnbwd_f32", "ptt_xcorr_z0_bwd_workspace", "ptt_xcorr_z0_bwd_f32",
    "ptt_bn_stats_train_f32", "ptt_bn_finish_partials_train_f32", "ptt_pack_weights_f32",
    "ptt_sa_z0_rows_f32",
    "ptt_row_jobs_f32", "ptt_point_jobs_f32", "ptt_fps_ball_knn_f32", "ptt_crop_compact_host_f32", "ptt_crop_regularize_f32", "ptt_colsum_workspace", "ptt_colsum_f32", "ptt_rows_gemm_pool_supported", "ptt_rows_gemm_pool_f32", "ptt_pool_select_f32", "ptt_sa_z0_rows_stat_chunks", "ptt_sa_z0_rows_stats_f32",
    "ptt_track_losses_f32", "ptt_track_losses_bwd_f32", "ptt_adam_chunk_elems", "ptt_adam_clip_step_f32", "ptt_adam_clip_step_dev_f32",
    "ptt_linear_wgrad_partials_f32", "ptt_linear_wgrad2_partials_f32", "ptt_colsum_partials_f32", "ptt_grad_finish_f32",
    "ptt_rows_gemm_rsum16_supported", "ptt_rows_gemm_rsum16_f32", "ptt_scatter_rows_csr_sub_f32",
    "ptt_unit_rows_f32", "ptt_cos_bwd_rows_f32", "ptt_track_select_update", "ptt_sa_z0_bnbwd_workspace", "ptt_sa_z0_bnbwd_f32",
]
PTT_MAX_SEGMENTS = 4


class CropJob(Structure):
    """ptt_crop_job (include/ptt_hip.h): one crop_center_pc; arrays of these are uploaded to the device."""
    _fields_ = [("points", c_void_p), ("ld", c_int64),
                ("lo1", c_double * 3), ("hi1", c_double * 3), ("trans", c_double * 3), ("rot", c_double * 9),
                ("lo2", c_double * 3), ("hi2", c_double * 3),
                ("out", c_void_p), ("count", c_void_p), ("n_points", c_int32), ("capacity", c_int32),
                ("label_out", c_void_p), ("ltrans", c_double * 3), ("lrot", c_double * 9), ("llo", c_double * 3), ("lhi", c_double * 3)]


class RegularizeJob(Structure):
    """ptt_regularize_job: regularize_pc over the concatenation of up to 4 compacted crops."""
    _fields_ = [("seg", c_void_p * PTT_MAX_SEGMENTS), ("seg_count", c_void_p * PTT_MAX_SEGMENTS),
                ("seg_capacity", c_int32 * PTT_MAX_SEGMENTS), ("out", c_void_p), ("info", c_void_p),
                ("n_seg", c_int32), ("input_size", c_int32)]



class BnTrainTail(Structure):
    """ptt_bn_train_tail: a training-mode BatchNorm's bookkeeping, done by the launch that forms the statistics."""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("act_a", c_void_p), ("act_b", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p), ("momentum", c_float)]


class BnBwdInput(Structure):
    """ptt_bn_bwd_input: a layer's BatchNorm + ReLU backward as the A operand of ptt_rows_gemm_bnbwd_fused_f32."""
    _fields_ = [("g", c_void_p), ("ldg", c_int), ("arg", c_void_p), ("ns", c_int), ("z", c_void_p), ("ldz", c_int),
                ("k1", c_void_p), ("c0", c_void_p), ("c1", c_void_p), ("mean", c_void_p), ("act_a", c_void_p), ("act_b", c_void_p),
                ("dz_out", c_void_p), ("ldd", c_int)]


class PackJob(Structure):
    """ptt_pack_job: one weight (view) of ptt_pack_weights_f32; arrays of these are uploaded to the device."""
    _fields_ = [("W", c_void_p), ("out_offset", c_int64), ("stride_out", c_int64), ("stride_k", c_int64),
                ("Cout", c_int32), ("K", c_int32)]


class RowJob(Structure):
    """ptt_row_job: one row-wise layer of ptt_row_jobs_f32 (passed by value, host memory)."""
    _fields_ = [("X", c_void_p), ("X2", c_void_p), ("Xmax", c_void_p), ("Wpacked", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("res", c_void_p), ("res2", c_void_p), ("out", c_void_p), ("out2", c_void_p), ("raw", c_void_p),
                ("rel", c_void_p), ("w1", c_void_p), ("qkv", c_void_p), ("knn", c_void_p), ("pos", c_void_p),
                ("rows", c_int32), ("K", c_int32), ("K1", c_int32), ("ldx", c_int32), ("ldx2", c_int32), ("Cout", c_int32),
                ("act", c_int32), ("res_split", c_int32), ("ldr", c_int32), ("ldr2", c_int32), ("out_split", c_int32),
                ("out_col0", c_int32), ("ldo", c_int32), ("ldo2", c_int32), ("ldraw", c_int32),
                ("prologue", c_int32), ("epilogue", c_int32), ("ldq", c_int32), ("q_off", c_int32), ("k_off", c_int32),
                ("v_off", c_int32), ("ldp", c_int32), ("N", c_int32), ("sm_scale", c_float), ("col_tiles", c_int32),
                ("idx", c_void_p), ("xyz", c_void_p), ("centres", c_void_p), ("wx", c_void_p), ("radius", c_float),
                ("ns", c_int32), ("M", c_int32), ("normalize_xyz", c_int32), ("pro_relu", c_int32)]


class PointJob(Structure):
    """ptt_point_job: one ball-query level or kNN of ptt_point_jobs_f32."""
    _fields_ = [("xyz", c_void_p), ("centre_sel", c_void_p), ("point_sel", c_void_p), ("new_xyz", c_void_p), ("idx64_out", c_void_p),
                ("idx_out", c_void_p), ("rel_out", c_void_p), ("kind", c_int32), ("sel_ld", c_int32), ("B", c_int32),
                ("Nraw", c_int32), ("Npts", c_int32), ("M", c_int32), ("nsample", c_int32), ("radius", c_float)]


class TrackLossDesc(Structure):
    """ptt_track_loss_desc: the inputs of the four tracking losses (passed by value, host memory)."""
    _fields_ = [("seed_cls", c_void_p), ("cls_label", c_void_p), ("search_inds", c_void_p), ("votes", c_void_p), ("reg_label", c_void_p),
                ("box_data", c_void_p), ("centres", c_void_p), ("pos_weight_seed", c_void_p), ("pos_weight_box", c_void_p),
                ("B", c_int32), ("N", c_int32), ("Ns", c_int32), ("M", c_int32), ("ld_reg", c_int32),
                ("w_seed_cls", c_float), ("w_seed_reg", c_float), ("w_box_cls", c_float), ("w_box_reg", c_float)]


class AdamTensor(Structure):
    """ptt_adam_tensor: one parameter with its gradient and moments; arrays of these are uploaded to the device."""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_int64)]


class GradJob(Structure):
    """ptt_grad_job: one contribution to a parameter gradient (row-chunk partial sums); arrays of these are uploaded to the device."""
    _fields_ = [("partial", c_void_p), ("nchunks", c_int32), ("reserved", c_int32)]


class GradSegment(Structure):
    """ptt_grad_segment: one destination inside the flat gradient buffer with its jobs."""
    _fields_ = [("dst", c_int64), ("n", c_int32), ("cols", c_int32), ("ld", c_int32), ("job0", c_int32), ("njobs", c_int32), ("out", c_int32),
                ("vec", c_int32), ("reserved", c_int32)]


class AdamHyper(Structure):
    """ptt_adam_hyper (passed by value, host memory)."""
    _fields_ = [("beta1", c_float), ("beta2", c_float), ("one_minus_beta1", c_float), ("one_minus_beta2", c_float), ("eps", c_float), ("step_size", c_float), ("bias2_sqrt", c_float),
                ("weight_decay", c_float), ("max_norm", c_float), ("write_clipped", c_int32)]


class SaLayer(Structure):
    _fields_ = [("Wpacked", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("Cin", c_int), ("Cout", c_int), ("relu", c_int)]


class SaDesc(Structure):
    _fields_ = [("xyz", c_void_p), ("new_xyz", c_void_p), ("idx", c_void_p), ("feat", c_void_p),
                ("feat_sb", c_int64), ("feat_sc", c_int64), ("feat_sn", c_int64),
                ("out", c_void_p), ("out_sb", c_int64), ("out_sc", c_int64), ("out_sm", c_int64),
                ("B", c_int), ("N", c_int), ("M", c_int), ("nsample", c_int), ("C", c_int),
                ("radius", c_float), ("use_xyz", c_int), ("normalize_xyz", c_int), ("n_layers", c_int),
                ("layers", SaLayer * PTT_SA_MAX_LAYERS),
                ("l0_point_term", c_void_p), ("l0_xyz_weight", c_void_p), ("l0_channels", c_int), ("l0_relu", c_int)]


class XcorrDesc(Structure):
    _fields_ = [("cos_t", c_void_p), ("P", c_void_p), ("w_sim", c_void_p), ("scale0", c_void_p), ("shift0", c_void_p),
                ("out", c_void_p), ("out_sb", c_int64), ("out_sc", c_int64), ("out_sn", c_int64),
                ("sim_out", c_void_p),
                ("B", c_int), ("Ns", c_int), ("Nt", c_int), ("C0", c_int),
                ("n_layers", c_int), ("layers", SaLayer * PTT_SA_MAX_LAYERS),
                ("split", c_int32), ("out_sh", c_int64), ("search_feat", c_void_p), ("templ_feat", c_void_p),
                ("s_sb", c_int64), ("s_sn", c_int64), ("t_sb", c_int64), ("t_sn", c_int64), ("C", c_int), ("eps", c_float)]


class AttnDesc(Structure):
    _fields_ = [("xyz", c_void_p), ("rel", c_void_p), ("knn", c_void_p), ("qkv", c_void_p),
                ("Wd1p", c_void_p), ("Wd2p", c_void_p), ("bd2", c_void_p),
                ("Wg1p", c_void_p), ("bg1", c_void_p), ("Wg2p", c_void_p), ("bg2", c_void_p),
                ("res", c_void_p), ("attn", c_void_p),
                ("B", c_int), ("N", c_int), ("k", c_int), ("D", c_int), ("order", c_void_p)]


_lib = None


def _declare(lib):
    vp, i, f = c_void_p, c_int, c_float
    lib.ptt_version.restype = c_int
    lib.ptt_error_name.restype = c_char_p
    lib.ptt_error_name.argtypes = [i]
    lib.ptt_last_error_string.restype = c_char_p
    sigs = {
        "ptt_fps_f32": [vp, i, i, i, vp, vp],
        "ptt_fps_ws_f32": [vp, i, i, i, vp, vp, c_size_t, vp],
        "ptt_spatial_order_f32": [vp, i, i, vp, vp],
        "ptt_ball_query_grid_f32": [vp, vp, i, i, i, f, i, vp, vp, c_size_t, vp],
        "ptt_centres_ball_query_grid_f32": [vp, vp, i, i, i, f, i, vp, vp, vp, vp, c_size_t, vp],
        "ptt_gather_f32": [vp, vp, i, i, i, i, vp, vp],
        "ptt_gather_grad_f32": [vp, vp, i, i, i, i, vp, vp],
        "ptt_select_centres_f32": [vp, vp, i, i, i, vp, vp, vp],
        "ptt_ball_query_f32": [vp, vp, i, i, i, f, i, vp, vp],
        "ptt_centres_ball_query_f32": [vp, vp, i, i, i, f, i, vp, vp, vp, vp],
        "ptt_group_f32": [vp, vp, i, i, i, i, i, vp, vp],
        "ptt_group_grad_f32": [vp, vp, i, i, i, i, i, vp, vp],
        "ptt_scatter_add_det_f32": [vp, vp, i, i, i, i, vp, vp, c_size_t, vp],
        "ptt_knn_f32": [vp, i, i, i, vp, vp],
        "ptt_knn_rel_f32": [vp, i, i, i, vp, vp, vp],
        "ptt_pack_weight_f32": [vp, i, i, vp, vp],
        "ptt_pack_weight_rot_f32": [vp, i, i, i, vp, vp],
        "ptt_linear_f32": [vp, i, i, i, vp, i, vp, vp, i, vp, i, vp, i, vp],
        "ptt_sa_fused_fwd_f32": [POINTER(SaDesc), vp],
        "ptt_rows_mlp_f32": [vp, i, i, i, POINTER(SaLayer), i, vp, i, vp, i, vp],
        "ptt_xcorr_fused_fwd_f32": [POINTER(XcorrDesc), vp],
        "ptt_cosine_map_f32": [vp, c_int64, c_int64, c_int64, vp, c_int64, c_int64, c_int64, i, i, i, i, f, vp, vp],
        "ptt_pt_attn_pair_f32": [POINTER(AttnDesc), vp],
        "ptt_crop_compact_f32": [vp, i, vp],
        "ptt_crop_compact_host_f32": [vp, i, vp],
        "ptt_crop_regularize_f32": [vp, vp, i, vp, i, vp],
        "ptt_colsum_f32": [vp, i, i, i, vp, vp, c_size_t, vp],
        "ptt_rows_gemm_pool_supported": [i, i, i, i, i],
        "ptt_rows_gemm_pool_f32": [vp, i, i, i, vp, vp, vp, i, vp, i, vp, c_size_t, i, vp, vp, vp, vp, vp],
        "ptt_pool_select_f32": [vp, vp, vp, vp, vp, vp, i, i, vp, vp, vp],
        "ptt_sa_z0_rows_stat_chunks": [i, i, i, i],
        "ptt_sa_z0_rows_stats_f32": [vp, vp, vp, vp, vp, i, i, i, i, i, i, f, i, vp, vp, vp, c_size_t, vp],
        "ptt_track_losses_f32": [vp, vp, vp, vp],
        "ptt_track_losses_bwd_f32": [vp, vp, vp, vp, vp, vp, vp],
        "ptt_adam_clip_step_f32": [vp, vp, vp, i, vp, vp, c_size_t, vp, vp],
        "ptt_adam_clip_step_dev_f32": [vp, vp, vp, i, vp, i, vp, c_size_t, vp, vp],
        "ptt_unit_rows_f32": [vp, c_longlong, c_longlong, c_longlong, i, i, i, f, vp, vp, vp],
        "ptt_cos_bwd_rows_f32": [vp, vp, vp, vp, vp, c_longlong, c_longlong, c_longlong, i, i, i, i, vp, c_longlong, c_longlong, c_longlong, vp],
        "ptt_track_select_update": [vp, i, vp, vp, i, i, vp, vp, vp],
        "ptt_regularize_f32": [vp, i, vp, i, vp],
        "ptt_mt19937_fill": [c_uint32, vp, i],
        "ptt_select_box_f32": [vp, i, i, vp, vp, vp],
        "ptt_track_crop_bounds": [vp, i, c_double, c_double, vp, vp, i],
        "ptt_track_box_by_offset": [vp, i, vp, i, i, vp, vp],
        "ptt_bn_stats_f32": [vp, i, i, i, f, vp, vp, vp, vp, c_size_t, vp],
        "ptt_bn_apply_f32": [vp, i, vp, vp, vp, vp, i, i, i, vp, i, vp],
        "ptt_bn_bwd_f32": [vp, i, vp, i, vp, i, vp, vp, vp, i, i, i, vp, i, vp, vp, vp, c_size_t, vp, vp, vp],
        "ptt_bn_sums_f64": [vp, i, i, i, vp, vp, c_size_t, vp],
        "ptt_bn_finish_f64": [vp, i, f, vp, vp, vp, vp],
        "ptt_bn_bwd_sums_f64": [vp, i, vp, i, vp, i, vp, vp, i, i, vp, vp, c_size_t, vp, vp, vp],
        "ptt_bn_bwd_apply_f32": [vp, i, vp, i, vp, i, vp, vp, vp, vp, vp, vp, i, i, vp, i, vp, vp, vp],
        "ptt_pool_rows_f32": [vp, i, i, i, i, vp, i, vp, vp, vp, vp],
        "ptt_linear_act_in_f32": [vp, i, i, i, vp, vp, vp, i, vp, i, vp],
        "ptt_pool_rows_bwd_f32": [vp, i, vp, i, i, i, vp, i, vp],
        "ptt_linear_wgrad_f32": [vp, i, vp, i, i, i, i, vp, i, vp, c_size_t, vp, vp, vp],
        "ptt_pack_weight_strided_f32": [vp, i, i, c_int64, c_int64, i, c_int64, vp, vp],
        "ptt_linear_batched_f32": [vp, i, i, i, c_int64, vp, c_int64, i, vp, vp, i, vp, i, c_int64, vp, i, c_int64, i, vp],
        "ptt_softmax_rows_f32": [vp, c_int64, i, i, f, vp],
        "ptt_gather_rows_f32": [vp, vp, i, i, i, i, vp, vp],
        "ptt_scatter_csr_i32": [vp, i, i, i, vp, vp, vp],
        "ptt_scatter_rows_csr_f32": [vp, vp, vp, i, i, i, i, vp, vp],
        "ptt_pt_pair_input_f32": [vp, vp, vp, vp, i, i, i, i, vp, vp],
        "ptt_pt_attn_train_fwd_f32": [vp, vp, vp, vp, i, i, i, i, f, vp, vp, vp],
        "ptt_pt_attn_train_bwd_f32": [vp, vp, vp, vp, vp, i, i, i, i, f, vp, vp, vp],
        "ptt_rows_gemm_supported": [i, i, i, i, i],
        "ptt_rows_gemm_stat_chunks": [i, i, i],
        "ptt_rows_gemm_f32": [vp, i, i, i, vp, vp, vp, i, vp, i, vp, i, vp, i, vp, c_size_t, vp],
        "ptt_rows_gemm_masked_f32": [vp, i, i, i, vp, i, vp, i, vp, i, vp, c_size_t, vp],
        "ptt_bn_finish_partials_f32": [vp, i, i, i, f, vp, vp, vp, vp],
        "ptt_bn_sums_partials_f64": [vp, i, i, i, vp, vp],
        "ptt_linear_wgrad2_f32": [vp, i, vp, i, i, i, i, vp, i, vp, c_size_t, vp, vp, vp],
        "ptt_linear_wgrad_partials_f32": [vp, i, vp, i, i, i, i, vp, c_size_t, vp, vp, vp, vp],
        "ptt_linear_wgrad2_partials_f32": [vp, i, vp, i, i, i, i, vp, c_size_t, vp, vp, vp, vp],
        "ptt_colsum_partials_f32": [vp, i, i, i, vp, c_size_t, vp, vp],
        "ptt_grad_finish_f32": [vp, vp, vp, i, vp, vp],
        "ptt_rows_gemm_rsum16_supported": [i, i, i, i],
        "ptt_rows_gemm_rsum16_f32": [vp, i, i, i, vp, i, vp, i, vp, i, vp, i, vp, i, vp],
        "ptt_scatter_rows_csr_sub_f32": [vp, vp, vp, i, i, i, i, vp, vp, vp],
        "ptt_rows_gemm_bnbwd_f32": [vp, i, i, i, vp, i, vp, i, vp, vp, vp, vp, vp, i, vp, c_size_t, vp],
        "ptt_bn_bwd_from_partials_f32": [vp, i, vp, i, vp, i, vp, vp, vp, i, i, vp, i, vp, vp, vp, vp, vp],
        "ptt_bn_bwd_consts_f32": [vp, i, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp],
        "ptt_bn_bwd_pooled_consts_f32": [vp, i, vp, i, vp, i, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp, c_size_t, vp, vp, vp],
        "ptt_rows_gemm_bnbwd_fused_supported": [i, i, i, i],
        "ptt_rows_gemm_bnbwd_fused_f32": [POINTER(BnBwdInput), i, i, vp, i, vp, i, vp, vp, vp, vp, vp, i, vp, c_size_t, vp],
        "ptt_bn_bwd_sums_partials_f64": [vp, i, i, vp, vp],
        "ptt_pt_pair_input_ld_f32": [vp, i, vp, i, vp, vp, i, i, i, i, vp, vp],
        "ptt_pt_attn_fwd_ld_f32": [vp, vp, i, vp, vp, i, i, i, i, f, vp, vp, vp],
        "ptt_bn_update_running_f32": [vp, vp, vp, f, i, vp, vp, vp, vp],
        "ptt_bn_stats_train_f32": [vp, i, i, i, f, vp, vp, vp, vp, c_size_t, vp, vp],
        "ptt_bn_finish_partials_train_f32": [vp, i, i, i, f, vp, vp, vp, vp, vp],
        "ptt_pack_weights_f32": [vp, i, vp, vp],
        "ptt_sa_z0_rows_f32": [vp, vp, vp, vp, vp, i, i, i, i, i, i, f, i, vp, vp, vp],
        "ptt_xcorr_z0_f32": [vp, vp, vp, i, i, i, i, vp, vp],
        "ptt_xcorr_z0_stat_chunks": [i, i, i, i],
        "ptt_sa_z0_bnbwd_f32": [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, c_longlong, i, vp, vp, vp, vp, vp, c_size_t, vp],
        "ptt_xcorr_z0_bnbwd_f32": [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, c_size_t, vp],
        "ptt_xcorr_z0_stats_f32": [vp, vp, vp, i, i, i, i, vp, vp, c_size_t, vp],
        "ptt_xcorr_z0_bwd_f32": [vp, vp, vp, i, i, i, i, vp, vp, vp, vp, c_size_t, vp],
        "ptt_bn_bwd_pooled_f32": [vp, i, vp, i, vp, i, vp, vp, vp, i, i, vp, i, vp, vp, vp, c_size_t, vp, vp, vp],
        "ptt_bn_bwd_pooled_sums_f64": [vp, i, vp, i, vp, i, vp, vp, i, i, vp, vp, c_size_t, vp, vp, vp],
        "ptt_bn_bwd_pooled_apply_f32": [vp, i, vp, i, vp, i, vp, vp, vp, vp, vp, vp, i, i, vp, i, vp, vp, vp],
        "ptt_row_jobs_f32": [POINTER(RowJob), i, vp],
        "ptt_point_jobs_f32": [POINTER(PointJob), i, vp],
        "ptt_fps_ball_knn_f32": [vp, i, i, i, f, i, i, vp, vp, vp, vp, vp, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    lib.ptt_packed_weight_elems.restype = c_size_t
    lib.ptt_packed_weight_elems.argtypes = [i, i]
    lib.ptt_scatter_add_det_workspace.restype = c_size_t
    lib.ptt_scatter_add_det_workspace.argtypes = [i, i, i]
    lib.ptt_ball_query_grid_workspace.restype = c_size_t
    lib.ptt_ball_query_grid_workspace.argtypes = [i, i]
    lib.ptt_bn_stats_workspace.restype = c_size_t
    lib.ptt_bn_stats_workspace.argtypes = [i, i]
    lib.ptt_linear_wgrad_workspace.restype = c_size_t
    lib.ptt_linear_wgrad_workspace.argtypes = [i, i, i]
    lib.ptt_xcorr_z0_bwd_workspace.restype = c_size_t
    lib.ptt_xcorr_z0_bwd_workspace.argtypes = [i, i, i]
    lib.ptt_sa_z0_bnbwd_workspace.restype = c_size_t
    lib.ptt_sa_z0_bnbwd_workspace.argtypes = [c_longlong, i]
    lib.ptt_adam_chunk_elems.restype = c_int
    lib.ptt_adam_chunk_elems.argtypes = []
    lib.ptt_colsum_workspace.restype = c_size_t
    lib.ptt_colsum_workspace.argtypes = [i, i]
    lib.ptt_linear_wgrad2_workspace.restype = c_size_t
    lib.ptt_linear_wgrad2_workspace.argtypes = [i, i, i]


def lib():
    """Load libptt_hip.so once. Raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ptt_amd: %s is missing — build it with `python -m ptt_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch must initialise ITS bundled HIP runtime first: libptt_hip.so then binds to the
        # libamdhip64 already in the process instead of pulling a second copy from /opt/rocm.
        import torch  # noqa: F401
        try:
            loaded = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. no ROCm runtime on this host
            raise RuntimeError("ptt_amd: cannot load %s: %s" % (LIB_PATH, e))
        _declare(loaded)
        if loaded.ptt_version() != ABI_VERSION:
            raise RuntimeError("ptt_amd: %s has ABI version %d, this package expects %d — rebuild it with "
                               "`python -m ptt_amd.build`" % (LIB_PATH, loaded.ptt_version(), ABI_VERSION))
        _lib = loaded
    return _lib


def check(rc, what):
    if rc != 0:
        l = lib()
        raise RuntimeError("%s failed: %s (%s)" % (what, l.ptt_error_name(rc).decode(),
                                                   l.ptt_last_error_string().decode()))
