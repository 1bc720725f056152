"""ctypes binding of libdetmatch_hip.so (include/detmatch_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  Loading
fails loudly when the .so is missing, and every op raises when handed a tensor
that is not on a HIP device.
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DM_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libdetmatch_hip.so')  # env: A/B builds
_lib = None

cd = ctypes.c_double
c_int_p = ctypes.POINTER(ctypes.c_int)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_f32_p = ctypes.POINTER(ctypes.c_float)
vp = ctypes.c_void_p
ci = ctypes.c_int
sz = ctypes.c_size_t
cf = ctypes.c_float

class SpconvWgradJob(ctypes.Structure):
    """dm_spconv_wgrad_job (include/detmatch_hip.h)"""
    _fields_ = [('feat', vp), ('out_grad', vp), ('indice_pairs', vp), ('indice_num', vp), ('filt_grad', vp),
                ('pair_stride', ci), ('kvol', ci), ('cin', ci), ('cout', ci)]


# name -> (restype, argtypes); must list EVERY symbol include/detmatch_hip.h declares
SIGNATURES = {
    'dm_version': (ctypes.c_char_p, []),
    'dm_error_string': (ctypes.c_char_p, [ci]),
    'dm_hard_voxelize_workspace_bytes': (sz, [ci, ci]),
    'dm_hard_voxelize': (ci, [vp, ci, ci, c_i32_p, ci, c_f32_p, c_f32_p, ci, ci, ci, vp, vp, vp,
                              vp, vp, vp, sz, vp]),
    'dm_rulebook_workspace_bytes': (sz, [ci, ci]),
    'dm_rulebook_subm': (ci, [vp, ci, ci, c_int_p, c_int_p, vp, vp, vp, vp, sz, vp]),
    'dm_rulebook_conv_count': (ci, [vp, ci, ci, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, vp,
                                    vp, sz, vp]),
    'dm_rulebook_conv_fill': (ci, [vp, ci, ci, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, ci,
                                   vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_rulebook_subm_cap': (ci, [vp, vp, ci, ci, c_int_p, c_int_p, vp, vp, vp, vp, sz, vp]),
    'dm_rulebook_conv_cap': (ci, [vp, vp, ci, ci, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, ci, vp, vp, vp, vp, vp,
                                  vp, vp, sz, vp]),
    'dm_rulebook_set_mode': (ci, [ci]),
    'dm_pairs_to_table': (ci, [vp, vp, ci, ci, ci, vp, ci, vp]),
    'dm_spconv_workspace_bytes': (sz, [ci, ci, ci]),
    'dm_spconv_gather_gemm': (ci, [vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]),
    'dm_spconv16_workspace_bytes': (sz, [ci, ci, ci]),
    'dm_spconv_gather_gemm16': (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]),
    'dm_spconv_pack_rows_workspace_bytes': (sz, [ci]),
    'dm_spconv_pack_rows': (ci, [vp, ci, ci, vp, vp, vp, vp, sz, vp]),
    'dm_spconv_tile_order_workspace_bytes': (sz, []),
    'dm_spconv_tile_order': (ci, [vp, ci, ci, vp, vp, sz, vp]),
    'dm_spconv_wgrad_workspace_bytes': (sz, [ci, ci, ci, ci]),
    'dm_spconv_wgrad': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, sz, vp]),
    'dm_spconv_wgrad_batch_workspace_bytes': (sz, [ctypes.POINTER(SpconvWgradJob), ci]),
    'dm_spconv_wgrad_batch': (ci, [ctypes.POINTER(SpconvWgradJob), ci, ci, vp, sz, vp]),
    'dm_iou3d_workspace_bytes': (sz, [ci, ci]),
    'dm_boxes_overlap_bev': (ci, [vp, ci, vp, ci, vp, vp, sz, vp]),
    'dm_boxes_overlap_bev_exact': (ci, [vp, ci, vp, ci, vp, vp]),
    'dm_boxes_iou_bev': (ci, [vp, ci, vp, ci, vp, vp, sz, vp]),
    'dm_nms_workspace_bytes': (sz, [ci]),
    'dm_nms': (ci, [vp, ci, cf, ci, vp, vp, vp, sz, vp]),
    'dm_nms_normal': (ci, [vp, ci, cf, ci, vp, vp, vp, sz, vp]),
    'dm_nms_2d': (ci, [vp, ci, cf, ci, vp, vp, vp, sz, vp]),
    'dm_nms_batch': (ci, [vp, ci, ci, cf, ci, ci, vp, ctypes.c_longlong, vp, vp, sz, vp]),
    'dm_nms_2d_batch': (ci, [vp, ci, ci, cf, ci, vp, ctypes.c_longlong, vp, vp, sz, vp]),
    'dm_box3d_project_forward': (ci, [vp, ci, c_f32_p, c_f32_p, cf, cf, vp, vp, vp]),
    'dm_box3d_project_backward': (ci, [vp, ci, c_f32_p, c_f32_p, cf, cf, vp, vp, vp]),
    'dm_consistency_loss_forward': (ci, [vp, vp, vp, vp, ci, ci, cf, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp]),
    'dm_consistency_loss_backward': (ci, [vp, vp, vp, vp, ci, ci, vp, vp, vp]),
    'dm_bbox2d_transform': (ci, [vp, ci, cf, cf, cf, cf, cf, ci, ci, ci, vp, vp]),
    'dm_sort_rows_max': (ci, []),
    'dm_sort_rows_f32': (ci, [vp, ci, ci, ctypes.c_longlong, ci, vp, vp, vp]),
    'dm_fc_gemm_workspace_bytes': (sz, [ci, ci, ci]),
    'dm_fc_gemm': (ci, [ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp]),
    'dm_rowgemm_supported': (ci, [ci, ci]),
    'dm_rowgemm': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, vp]),
    'dm_rowgemm_parts': (ci, [ctypes.c_longlong, ci, ci]),
    'dm_rowgemm_stats': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, vp, vp, vp]),
    'dm_rowgemm_strided': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, ci, ci, vp]),
    'dm_rowgemm_wt': (ci, [vp, vp, ci, vp, ctypes.c_longlong, ci, ci, ci, ci, vp]),
    'dm_tall_wgrad_supported': (ci, [ci, ci]),
    'dm_tall_wgrad_workspace_bytes': (sz, [ctypes.c_longlong, ci, ci]),
    'dm_tall_wgrad': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, ci, vp, sz, vp]),
    'dm_anchor_decode': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, cf, cf, cf, vp, vp]),
    'dm_bn_rows_workspace_bytes': (sz, [ctypes.c_longlong, ci]),
    'dm_bn_rows_forward': (ci, [vp, ctypes.c_longlong, ci, vp, vp, cf, cf, vp, vp, ci, vp, vp, vp, vp, sz, vp]),
    'dm_bn_rows_max_forward': (ci, [vp, ctypes.c_longlong, ci, ci, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_bn_rows_max_backward': (ci, [vp, vp, vp, ctypes.c_longlong, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_bn_rows_max_backward_ld': (ci, [vp, ctypes.c_longlong, vp, vp, ctypes.c_longlong, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_bn_rows_forward_pre': (ci, [vp, ctypes.c_longlong, ci, vp, vp, cf, cf, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp]),
    'dm_bn_rows_max_forward_pre': (ci, [vp, ctypes.c_longlong, ci, ci, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    'dm_bn_rows_eval_max': (ci, [vp, ctypes.c_longlong, ci, ci, vp, vp, vp, vp, cf, vp, vp]),
    'dm_bn_rows_eval': (ci, [vp, ctypes.c_longlong, ci, vp, vp, vp, vp, cf, ci, vp, vp]),
    'dm_bn_rows_backward': (ci, [vp, vp, ctypes.c_longlong, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, sz, vp]),
    'dm_roi_align_forward': (ci, [vp, c_i32_p, c_i32_p, c_f32_p, ci, ci, vp, vp, ci, ci, ci, ci, ci, ci,
                                  vp, vp]),
    'dm_roi_align_backward': (ci, [vp, c_i32_p, c_i32_p, c_f32_p, ci, ci, vp, vp, ci, ci, ci, ci, ci, ci,
                                   vp, vp]),
    'dm_roi_align_forward_nhwc': (ci, [vp, c_i32_p, c_i32_p, c_f32_p, ci, ci, vp, vp, ci, ci, ci, ci, ci,
                                       vp, vp]),
    'dm_roi_align_backward_nhwc': (ci, [vp, c_i32_p, c_i32_p, c_f32_p, ci, ci, vp, vp, ci, ci, ci, ci, ci,
                                        vp, vp]),
    'dm_ema_update_f32': (ci, [vp, vp, sz, ctypes.c_double, vp]),
    'dm_ema_update_i64': (ci, [vp, vp, sz, ctypes.c_double, vp]),
    'dm_adamw_step_f32': (ci, [vp, vp, vp, vp, sz, cd, cd, cd, cd, cd, ctypes.c_longlong, vp, vp]),
    'dm_sgd_step_f32': (ci, [vp, vp, vp, sz, cd, cd, cd, cd, ci, vp, vp]),
    'dm_adamw_step_masked_f32': (ci, [vp, vp, vp, vp, sz, cd, cd, cd, cd, cd, ctypes.c_longlong, vp, vp, vp]),
    'dm_sgd_step_masked_f32': (ci, [vp, vp, vp, sz, cd, cd, cd, cd, ci, vp, vp, vp]),
    'dm_adamw_step_blocks_f32': (ci, [vp, vp, vp, vp, sz, cd, cd, cd, cd, cd, ctypes.c_longlong, vp, vp, vp, vp]),
    'dm_sgd_step_blocks_f32': (ci, [vp, vp, vp, sz, cd, cd, cd, cd, ctypes.c_longlong, vp, vp, vp, vp]),
    'dm_anchor_assign_workspace_bytes': (sz, [ci, ci, ci]),
    'dm_anchor_assign': (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]),
    'dm_bev_interpolate_forward': (ci, [vp, ci, ci, ci, ci, vp, ci, ci, c_f32_p, vp, vp, vp, vp]),
    'dm_bev_interpolate_backward': (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    'dm_roi_targets_workspace_bytes': (sz, [ci, ci]),
    'dm_roi_targets': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_rcnn_loss_forward': (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, c_f32_p, c_f32_p, cf, ci, vp, vp, vp, vp, vp]),
    'dm_rcnn_loss_backward': (ci, [vp, vp, vp, vp, ci, vp, vp, vp]),
    'dm_point_targets': (ci, [vp, ci, vp, ci, ci, ci, ci, c_f32_p, ci, vp, vp]),
    'dm_point_focal_loss': (ci, [vp, vp, ci, ci, cf, cf, vp, vp, vp]),
    'dm_det2d_assign_workspace_bytes': (sz, [ci, ci]),
    'dm_rpn_loss_forward': (ci, [vp, c_int_p, ci, ci, ci, vp, vp, ci, vp, c_int_p, ci, vp, cf, cf, cf, ci, ci, ci, c_f32_p, c_f32_p, cf, cf, vp, vp, vp, vp, vp, sz, vp]),
    'dm_rpn_loss_backward': (ci, [vp, vp, vp, ci, vp, vp]),
    'dm_rpn_proposals_workspace_bytes': (sz, [ci, ci]),
    'dm_rpn_proposals_pre_nms': (ci, [vp, c_int_p, ci, ci, ci, vp, ci, ci, c_f32_p, ci, c_f32_p, c_f32_p, cf, ci, cf, ci, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_roi2d_targets': (ci, [vp, vp, ci, ci, vp, vp, c_int_p, ci, ci, vp, ci, cf, cf, cf, ci, ci, ci, ci, c_f32_p, c_f32_p, vp, vp, vp, vp, vp, vp, sz, vp]),
    'dm_bbox_head_loss': (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, cf, vp, vp, vp, vp]),
    'dm_dconv_pack': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, vp]),
    'dm_dconv_pack_batch': (ci, [vp, ci, ci, vp]),
    'dm_dconv_set_math': (ci, [ci]),
    'dm_dconv_get_math': (ci, []),
    'dm_dconv_gemm_workspace_bytes': (sz, [c_int_p]),
    'dm_dconv_gemm': (ci, [vp, vp, vp, vp, c_int_p, ctypes.POINTER(ctypes.c_short), vp, sz, vp]),
    'dm_dconv_gemm_residual': (ci, [vp, vp, vp, vp, vp, c_int_p, ctypes.POINTER(ctypes.c_short), vp, sz, vp]),
    'dm_dconv_planes_bytes': (sz, [ci, ci, ci]),
    'dm_dconv_pack_planes': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, vp]),
    'dm_dconv_gemm_planes': (ci, [vp, vp, vp, vp, vp, vp, c_int_p, ctypes.POINTER(ctypes.c_short), vp, sz, vp]),
    'dm_dconv_wgrad_workspace_bytes': (sz, [c_int_p]),
    'dm_dconv_wgrad': (ci, [vp, vp, vp, vp, c_int_p, ctypes.POINTER(ctypes.c_short), ci, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ci, vp, sz, vp]),
    'dm_fusion_match_cost': (ci, [vp, vp, vp, ci, vp, vp, ci, ci, c_f32_p, cf, cf, cf, cf, cf, cf, cf, cf, vp, vp, vp]),
    'dm_lap_host': (ci, [c_f32_p, ci, ci, c_int_p, c_int_p]),
    'dm_anchor_head_loss_workspace_bytes': (sz, [ci, ci]),
    'dm_anchor_head_loss_forward': (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, cf, c_f32_p, c_f32_p, vp, vp, sz, vp]),
    'dm_anchor_head_loss_backward': (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, cf, c_f32_p, c_f32_p, vp, vp, vp, vp, vp]),
    'dm_kitti_tp_scores_host': (ctypes.c_longlong, [vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, cd, vp]),
    'dm_kitti_pr_host': (ci, [vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, cd, vp, ci, ci, vp]),
    'dm_points_augment_workspace_bytes': (sz, [ci, c_int_p]),
    'dm_points_augment': (ci, [vp, ci, ci, c_int_p, c_int_p, c_int_p, vp, vp, vp, vp, vp, sz, vp]),
    'dm_ball_query_stack': (ci, [ci, ci, cf, ci, vp, vp, vp, vp, ci, vp, vp, vp]),
    'dm_ball_query_stack2': (ci, [ci, ci, cf, ci, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'dm_group_points_stack': (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    'dm_group_points_grad_stack': (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    'dm_query_group_rows': (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'dm_group_rows_grad': (ci, [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    'dm_furthest_point_sampling': (ci, [ci, ci, ci, vp, vp, vp, vp]),
    'dm_furthest_point_sampling_stack': (ci, [ci, c_i32_p, ci, vp, vp, vp, vp]),
    'dm_fps_set_variant': (ci, [ci]),
    'dm_voxel_centers': (ci, [vp, ci, ci, cf, cf, cf, cf, cf, cf, vp, vp, vp]),
    'dm_points_in_boxes': (ci, [ci, ci, ci, vp, vp, vp, vp]),
    'dm_profile_enable': (ci, [ci]),
    'dm_spconv_debug_stamps': (ci, [vp]),
    'dm_spconv_set_variant': (ci, [ci]),
    'dm_spconv_set_wgrad_chunk': (ci, [ci]),
    'dm_profile_count': (ci, []),
    'dm_profile_get': (ci, [ci, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
                            ctypes.POINTER(ctypes.c_ulonglong), c_f32_p]),
    'dm_roi_grid_points': (ci, [vp, ci, ci, ci, vp, vp]),
    'dm_roi_decode_forward': (ci, [vp, vp, ci, vp, vp]),
    'dm_roi_decode_backward': (ci, [vp, vp, vp, ci, vp, vp]),
    'dm_chain_fn_count': (ci, []),
    'dm_chain_fn_index': (ci, [ctypes.c_char_p]),
    'dm_chain_fn_name': (ctypes.c_char_p, [ci]),
    'dm_chain_fn_signature': (ctypes.c_char_p, [ci]),
    'dm_chain_run': (ci, [vp, ci, vp, ci, c_int_p]),
    'dm_height_compress_forward': (ci, [vp, vp, ctypes.c_longlong, ci, ci, ci, ci, ci, vp, vp]),
    'dm_height_compress_backward': (ci, [vp, vp, ctypes.c_longlong, ci, ci, ci, ci, vp, vp]),
    'dm_relu_mask_f32': (ci, [vp, vp, vp, ctypes.c_longlong, vp]),
    'dm_add_mask_f32': (ci, [vp, vp, vp, vp, ctypes.c_longlong, vp]),
    'dm_colsum_workspace_bytes': (sz, [ctypes.c_longlong, ci]),
    'dm_colsum_f32': (ci, [vp, ctypes.c_longlong, ci, vp, ci, vp, sz, vp]),
    'dm_resize_nearest_nhwc': (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    'dm_resize_nearest_nhwc_backward': (ci, [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp]),
    'dm_maxpool_nhwc': (ci, [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    'dm_subsample_nhwc_backward': (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    'dm_copy2d_f32': (ci, [vp, ctypes.c_longlong, vp, ctypes.c_longlong, ctypes.c_longlong, ci, vp]),
    'dm_fill_bytes': (ci, [vp, ci, sz, vp]),
    'dm_bn_fold_batch': (ci, [vp, ci, ci, vp]),
    'dm_multi_add_f32': (ci, [vp, ci, ctypes.c_longlong, ci, vp]),
}


class DetMatchHipError(RuntimeError):
    pass


# Callables run ONCE with the freshly loaded library (process-wide defaults such as the arithmetic of the
# dense convolutions): importing a module never has to load the library, and the default is applied whenever
# and by whomever the library is first loaded.
_POST_LOAD = []


def on_load(fn):
    """Register fn(lib) to run right after libdetmatch_hip.so is loaded (at once if it already is)."""
    _POST_LOAD.append(fn)
    if _lib is not None:
        fn(_lib)


def lib():
    """Load the C-ABI library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DetMatchHipError(
                '%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no fallback path.' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
        for hook in _POST_LOAD:
            hook(l)
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().dm_error_string(int(code)).decode()
        raise DetMatchHipError('%s failed: %s (code %d)' % (what, msg, code))


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise DetMatchHipError(
                'detmatch_amd ops run on the MI355X only (got a %s tensor); there is no CPU path'
                % t.device)
        if not t.is_contiguous():
            raise DetMatchHipError('tensor must be contiguous')


def ptr(t):
    """Device address of a tensor as a plain int (the argtypes convert it; wrapping it in a c_void_p object first costs
    0.2 us more per argument, ~3 000 arguments per iteration), None for None."""
    return None if t is None else t.data_ptr()


def raw_stream():
    """The current HIP stream handle of the current device as an int (torch.cuda.current_stream()
    builds a Stream object through several Python layers: ~8 us per call, thousands of calls a step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    return raw_stream()


def ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def floats(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


# ---- vendor BLAS: one GEMM at a time per process ------------------------------------------------------------------
# Every fp32 GEMM torch hands to hipBLASLt on gfx950 is a Tensile STREAM-K kernel (`Cijk_..._SK3_...`): workgroups that
# finish a partial tile publish it through flags in a workspace of the library handle and the owner of the tile spins in
# `label_SK_Fixup` until they arrive.  Two such kernels in flight at the same time from one handle — two HIP streams of
# one host thread — use the SAME flags: every recorded "three-lane device dead-lock" of round 5 was one of them (the
# RoI head's 27 648 -> 256 FC) spinning for ever with the device 100 % busy (profiles/r06_deadlock/, profiles/r06_lane_soak.txt;
# tools/streamk_two_streams_repro.py reproduces it with nothing but torch).  Rule of this package: a vendor GEMM is only
# issued inside `blas_turn()`, which orders it behind the previous vendor GEMM of the process with an event edge when that
# one went to another stream — in stream order nothing changes, across streams no two of them ever overlap.
_BLAS_LOCK = threading.RLock()
_BLAS_LAST = [None, 0]          # (event recorded behind the last vendor GEMM, raw handle of the stream it went to)
_BLAS_RING = []                 # events are reused round-robin (a wait captures the record it was issued after)
_BLAS_RING_POS = [0]
BLAS_TURNS = [0, 0]             # turns taken, cross-stream edges inserted (census / tests)
BLAS_TRACE = [False]            # tests: every turn is a `dm_blas_turn` range in torch's profiler (tests/test_blas_turn_gpu.py)


class blas_turn(object):
    """with blas_turn(): <torch GEMM(s) on the current stream>"""
    __slots__ = ('cuda', 'rng')

    def __enter__(self):
        _BLAS_LOCK.acquire()
        self.rng = None
        if BLAS_TRACE[0]:
            self.rng = torch.autograd.profiler.record_function('dm_blas_turn')
            self.rng.__enter__()
        self.cuda = torch.cuda.is_available()
        if self.cuda:
            ev, last_stream = _BLAS_LAST
            if ev is not None and last_stream != raw_stream():
                torch.cuda.current_stream().wait_event(ev)
                BLAS_TURNS[1] += 1
        BLAS_TURNS[0] += 1
        return self

    def __exit__(self, *exc):
        try:
            if self.cuda:
                if len(_BLAS_RING) < 64:
                    _BLAS_RING.append(torch.cuda.Event())
                i = _BLAS_RING_POS[0] = (_BLAS_RING_POS[0] + 1) % 64
                ev = _BLAS_RING[min(i, len(_BLAS_RING) - 1)]
                ev.record()
                _BLAS_LAST[0], _BLAS_LAST[1] = ev, raw_stream()
        finally:
            if self.rng is not None:
                self.rng.__exit__(None, None, None)
            _BLAS_LOCK.release()
        return False


class _BlasLinear(torch.autograd.Function):
    """F.linear on the vendor library, forward AND backward inside a turn (autograd's own matmul backward would issue
    its GEMMs from the engine's thread outside any turn)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        with blas_turn():
            return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        g2, x2 = gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1])
        with blas_turn():
            if ctx.needs_input_grad[0]:
                gx = (g2 @ w).view(x.shape)
            if ctx.needs_input_grad[1]:
                gw = g2.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(dim=0)
        return gx, gw, gb


# The raw handle of the MAIN lane of the iteration (mm3d/ssl.py:_Lanes sets it; None: no lanes).  Vendor GEMMs stay on
# that lane: a turn orders a GEMM behind the previous one wherever that went, and behind the main lane's deep queue (the
# host issues the student's backward 7-12 ms ahead of the device) a teacher-lane GEMM would wait for ALL of it — the
# teacher's read-back moved from 27 to 35 ms into the iteration that way (profiles/r06_phase_timeline_token_only.txt).
# FC layers issued on another lane therefore run on the library's own GEMM (csrc/conv2d.hip on (M, K, 1, 1) views:
# forward within 0.7-1.3x of the vendor kernels, profiles/r05_fc_blas_vs_own.txt) and no turn ever inserts an edge.
MAIN_STREAM = [None]
OWN_LINEAR_CALLS = [0]


def off_main_lane():
    return MAIN_STREAM[0] is not None and torch.cuda.is_available() and raw_stream() != MAIN_STREAM[0]


def own_linear(x, w, b=None, relu=False):
    """[relu](F.linear(x, w, b)) on the library's GEMM (differentiable: dense_conv's autograd Function; the ReLU rides in
    the GEMM's epilogue)."""
    from . import dense_conv
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    m, k = x2.shape
    n = w.shape[0]
    OWN_LINEAR_CALLS[0] += 1
    w4 = w.__dict__.get('_dm_view4') if isinstance(w, torch.nn.Parameter) else None
    if w4 is None or w4.data_ptr() != w.data_ptr() or w4.shape[:2] != (n, k) or w4.requires_grad != w.requires_grad:
        with torch.enable_grad():                  # (a view made under no_grad would never carry a gradient)
            w4 = w.view(n, k, 1, 1)
        if isinstance(w, torch.nn.Parameter):      # ONE view object per weight: the packed-weight cache is keyed by it
            w4.dm_cacheable = True
            w.__dict__['_dm_view4'] = w4
    y = dense_conv.conv2d(x2.contiguous().view(m, k, 1, 1), w4, b, relu=relu)
    return y.reshape(m, n).view(*shp[:-1], n)


def _own_linear_takes(x, w):
    if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.numel() > 0
            and w.dim() == 2 and w.is_contiguous()):
        return False
    # the own backward reads gradient rows 16 bytes at a time: an output width that is not a multiple of 4 is fine in
    # inference (the teacher's 1- and 7-wide RoI heads), with gradients it stays on the vendor kernel
    return w.shape[0] % 4 == 0 or not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))


# ---- stable sort of short score rows (csrc/sort_rows.hip) -----------------------------------------------------------
SORT_ROWS = True          # module switch of the equality test (False: torch.sort)
SORT_ROWS_CALLS = [0]


def sort_rows(keys, descending=True, dim=-1):
    """torch.sort(keys, dim, descending, stable=True)[1] for float32 keys of up to 16 384 elements per row on the device
    (one launch, one workgroup per row); anything else goes to torch.sort."""
    if not (SORT_ROWS and keys.is_cuda and keys.dtype == torch.float32 and keys.dim() in (1, 2) and
            dim in (-1, keys.dim() - 1) and 0 < keys.shape[-1] <= 16384 and keys.numel() > 0):
        return torch.sort(keys, dim=dim, descending=descending, stable=True)[1]
    k = keys.detach()
    if k.stride(-1) != 1:
        k = k.contiguous()
    rows = 1 if k.dim() == 1 else k.shape[0]
    n = k.shape[-1]
    stride = n if k.dim() == 1 else k.stride(0)
    if stride < n:
        k = k.contiguous()
        stride = n
    idx = torch.empty(k.shape, dtype=torch.int64, device=k.device)
    check(lib().dm_sort_rows_f32(ptr(k), rows, n, stride, int(bool(descending)), ptr(idx), None, stream()), 'dm_sort_rows_f32')
    SORT_ROWS_CALLS[0] += 1
    return idx


# ---- fully connected layers on the library's own exact-fp32 GEMM (csrc/fc_gemm.hip) -------------------------------------
FC_GEMM = True            # module switch of the equality tests / tools (False: vendor GEMM inside a turn on the main lane,
                          # the convolution GEMM elsewhere): 60.6 / 60.7 against 63.4 / 63.7 ms per iteration, profiles/r06_fc_blas_vs_own.txt
FC_GEMM_CALLS = [0]


def _fc_gemm(form, a, b, bias, out, m, n, k, lda, ldb, relu=False):
    L = lib()
    wsb = int(L.dm_fc_gemm_workspace_bytes(m, n, k))
    ws = workspace(wsb, out.device, 'fc_gemm') if wsb else None
    check(L.dm_fc_gemm(form, ptr(a), ptr(b), ptr(bias), ptr(out), m, n, k, lda, ldb, n, int(relu), ptr(ws),
                       ws.numel() if ws is not None else 0, stream()), 'dm_fc_gemm')
    FC_GEMM_CALLS[0] += 1
    return out


class _FcLinear(torch.autograd.Function):
    """[relu](x w^T + b) for x (rows, in), w (out, in): forward, input gradient and weight gradient on dm_fc_gemm
    (forms 0 / 1 / 2), the bias gradient as a column sum."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x, w = x.detach().contiguous(), w.detach().contiguous()
        m, k = x.shape
        n = w.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        _fc_gemm(0, x, w, None if b is None else b.detach(), y, m, n, k, k, k, relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.has_bias = bool(relu), b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        m, k = x.shape
        n = w.shape[0]
        gy = gy.contiguous()
        if ctx.relu:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _fc_gemm(1, gy, w, None, torch.empty_like(x), m, k, n, n, k)
        if ctx.needs_input_grad[1]:
            gw = _fc_gemm(2, gy, x, None, torch.empty_like(w), n, k, m, n, k)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=0)
        return gx, gw, gb, None


def _fc_takes(x, w, b):
    return FC_GEMM and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 2 and \
        x.numel() > 0 and (b is None or b.dtype == torch.float32)


def fc_linear(x, w, b=None, relu=False):
    """[relu](F.linear(x, w, b)) on csrc/fc_gemm.hip; x (..., in)."""
    shp = x.shape
    y = _FcLinear.apply(x.reshape(-1, shp[-1]), w, b, bool(relu))
    return y.view(*shp[:-1], w.shape[0])


def blas_linear(x, w, b=None, relu=False):
    """[relu](torch.nn.functional.linear(x, w, b)) for the FC stacks of the step.  Main lane: the library's exact-fp32 FC
    GEMM (csrc/fc_gemm.hip; `FC_GEMM = False`: the vendor GEMM under the one-GEMM-at-a-time rule, see blas_turn).  Other
    lanes: the same, except that the one layer above 8 GFLOP (the 2D head's 12 544 -> 1 024 at 1 000-2 000 rows) takes the
    library's convolution GEMM on (M, K, 1, 1) views (bf16-split arithmetic: faster on GEMMs of that size) — never a
    vendor kernel, so no turn ever inserts a cross-lane edge."""
    if _fc_takes(x, w, b):
        rows = x.numel() // max(1, x.shape[-1])
        big = 2.0 * rows * w.shape[0] * w.shape[1] > 8e9        # the 2D head's 12 544 -> 1 024 layer: 26-51 GFLOP
        if not (big and _own_linear_takes(x, w)):
            return fc_linear(x, w, b, relu)
        return own_linear(x, w, b, relu)
    if off_main_lane() and _own_linear_takes(x, w):
        return own_linear(x, w, b, relu)
    if relu:
        return torch.relu_(blas_linear(x, w, b))
    if not torch.is_grad_enabled() or not (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        with blas_turn():
            return torch.nn.functional.linear(x, w, b)
    return _BlasLinear.apply(x, w, b)


def blas_mm(a, b):
    """a @ b (2-D, no autograd) under the rule."""
    with blas_turn():
        return torch.mm(a, b)


# events behind gradient kernels that were issued on a side stream (chain.SIDE_WGRAD): whoever reads the gradients
# next (FlatGradDDP.collect / finish) makes its stream wait for them first
PENDING_GRAD_EVENTS = []


def wait_pending_grads():
    while PENDING_GRAD_EVENTS:
        torch.cuda.current_stream().wait_event(PENDING_GRAD_EVENTS.pop())


_AUX = {}


def aux_stream(device):
    """THE side stream of a device for work that rides beside the iteration (the weight-gradient halves of the chained
    backward passes, the key-point encoder beside the BEV backbone, the key-point FPS when it is not issued on the teacher
    lane).  One, on purpose: the runtime multiplexes HIP streams onto 4 hardware queues (GPU_MAX_HW_QUEUES; more cost
    30 ms per iteration once an RCCL communicator exists, profiles/r06_hw_queues_with_rccl.txt), and with the main stream and
    the two stream lanes of SSL.forward_train this is the fourth."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _AUX.get(idx)
    if s is None:
        s = _AUX[idx] = torch.cuda.Stream(device=device)
    return s


_ws_cache = {}


def workspace(nbytes, device, tag='default'):
    """A reusable byte workspace per (device, HIP stream, tag); grows geometrically.  Keyed by the
    current stream: work queued on different streams (2D / 3D lanes, FPS side stream) never shares
    scratch memory."""
    if device.type == 'cuda':
        idx = device.index if device.index is not None else torch._C._cuda_getDevice()
        # (+ the issuing thread: a backward pass issued by autograd's device thread while the main thread keeps
        # issuing forward work on the SAME stream must not share multi-launch scratch — partials + reduce — with it)
        key = (idx, torch._C._cuda_getCurrentRawStream(idx), tag, threading.get_ident())
    else:
        key = (-1, 0, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def profile_records():
    """[(kind, a, b, c, rows, kvol, table_ptr, ms)] of the launches recorded since
    dm_profile_enable(1) (synchronises on each record's stop event)."""
    L = lib()
    out = []
    k, a, b, c, rows, kv = (ctypes.c_int() for _ in range(6))
    tab = ctypes.c_ulonglong()
    ms = ctypes.c_float()
    for i in range(L.dm_profile_count()):
        check(L.dm_profile_get(i, k, a, b, c, rows, kv, tab, ms), 'dm_profile_get')
        out.append((k.value, a.value, b.value, c.value, rows.value, kv.value, tab.value, ms.value))
    return out
