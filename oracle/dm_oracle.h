/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's algorithms for the DetMatch hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this; the product (detmatch_amd/) never
 * does.  Each function cites the reference file:line it follows
 * (paths relative to the reference tree; pcdet/ = thirdparty/Spconv-OpenPCDet/pcdet/).
 *
 * Pinning status (details in DESIGN.md §Oracle):
 *   voxelize        pinned: oracle/_ref (reference voxelization_cpu.cpp compiled
 *                   here) + tests/golden/voxelize_*.npz + SURVEY K1 KAT
 *   rulebook/conv   pinned: SURVEY K2/K3 KATs produced by the compiled
 *                   reference CPU functors (reference not buildable under the
 *                   no-stand-in rule: every spconv source includes
 *                   <cuda_runtime_api.h>)
 *   iou3d / nms     pinned: SURVEY K4 KAT (compiled reference boxes_iou_bev_cpu)
 *                   + reference tests/test_utils/test_box3d.py:939 KAT
 *   pointnet2 / points_in_boxes   pinned: reference KATs of the batch-layout
 *                   siblings (tests/test_models/test_common_modules/)
 */
#ifndef DM_ORACLE_H_
#define DM_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-103 (hard_voxelize_kernel) and
 * :105-141 (hard_voxelize_cpu).  points (N,C) f32; voxels (max_voxels,max_points,C)
 * must be zero-filled by the caller (voxelize.py:46-50 does), coors (max_voxels,3)
 * [z,y,x], num_points (max_voxels).  Returns voxel_num. */
int orc_hard_voxelize(const float *points, int n, int c, const float *voxel_size,
                      const float *coors_range, int max_points, int max_voxels,
                      float *voxels, int32_t *coors, int32_t *num_points);

/* mmdet3d/ops/spconv/include/spconv/spconv_ops.h:28-141 (getIndicePair, CPU branch)
 * + geometry.h:25-86,145-192,248-296.  indices (N,4) [b,z,y,x].
 * out_ids must hold kvol*N rows of 4; indice_pairs (kvol,2,N) is filled with -1
 * first; indice_num (kvol).  If sort_out != 0 the strided-conv outputs are
 * re-ordered by ascending flat cell id (the reference's GPU order,
 * indice_cuda.cu:79 + torch::_unique) and the pair lists remapped.
 * Returns the number of active outputs, or -1 if batch*volume overflows int32. */
int orc_get_indice_pairs(const int32_t *indices, int n, int batch_size,
                         const int *out_shape, const int *spatial_shape,
                         const int *ksize, const int *stride, const int *padding,
                         const int *dilation, int subm, int sort_out,
                         int32_t *out_ids, int32_t *indice_pairs,
                         int32_t *indice_num);

/* spconv_ops.h:260-360 (indiceConv<float>, CPU branch) with reordering.cc:21-50.
 * filters (kvol,Cin,Cout).  out (n_out,Cout) is zeroed here. */
void orc_indice_conv(const float *features, int n_in, const float *filters,
                     const int32_t *indice_pairs, const int32_t *indice_num,
                     int pair_stride, int kvol, int cin, int cout, int n_out,
                     int subm, float *out);

/* spconv_ops.h:363-456 (indiceConvBackward<float>).  in_grad (n_in,Cin) and
 * filt_grad (kvol,Cin,Cout) are zeroed here. */
void orc_indice_conv_backward(const float *features, int n_in, const float *filters,
                              const float *out_grad, int n_out,
                              const int32_t *indice_pairs, const int32_t *indice_num,
                              int pair_stride, int kvol, int cin, int cout, int subm,
                              float *in_grad, float *filt_grad);

/* pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-234 (box_overlap), :226-233 (iou_bev).
 * boxes [x,y,z,dx,dy,dz,heading]. */
float orc_box_overlap(const float *a, const float *b);
float orc_iou_bev(const float *a, const float *b);
void orc_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out);
/* 0 (default): cos/sin/atan2 correctly rounded (double libm -> float), as the HIP pre-pass;
 * 1: float libm (cosf ...), what the reference's iou3d_cpu.cpp compiles to under g++. */
void orc_set_trig_mode(int mode);
void orc_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out);
/* iou3d_nms_kernel.cu:267-311 (mask) + iou3d_nms.cpp:91-137 (greedy).  boxes must
 * already be sorted by descending score.  keep (n) int64.  Returns num_to_keep. */
int orc_nms(const float *boxes, int n, float thresh, int64_t *keep);
/* iou3d_nms_kernel.cu:314-359 + iou3d_nms.cpp:140-185 */
int orc_nms_normal(const float *boxes, int n, float thresh, int64_t *keep);

/* pcdet/ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:16-66 */
void orc_ball_query_stack(int b, int m, float radius, int nsample,
                          const float *new_xyz, const int32_t *new_xyz_batch_cnt,
                          const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idx);
/* group_points_gpu.cu:71-102 (fwd), :15-46 (grad) */
void orc_group_points_stack(int b, int m, int c, int nsample, const float *features,
                            const int32_t *features_batch_cnt, const int32_t *idx,
                            const int32_t *idx_batch_cnt, float *out);
void orc_group_points_grad_stack(int b, int m, int c, int n, int nsample,
                                 const float *grad_out, const int32_t *idx,
                                 const int32_t *idx_batch_cnt,
                                 const int32_t *features_batch_cnt, float *grad_features);
/* sampling_gpu.cu:25-141 incl. the block-size dependent tie rule (:9-13, :16-21) */
void orc_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                 int32_t *idxs);
/* pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:17-37,313-337 (MARGIN 1e-5) */
void orc_points_in_boxes(int batch, int nboxes, int npts, const float *boxes,
                         const float *pts, int32_t *box_idx);

/* mmcv.ops.RoIAlign, pool_mode 'avg' (mmcv-full 1.3.16 is NOT under /root/reference: restated from
 * the published Detectron2/mmcv algorithm — PARITY UNPINNED).  One feature map (N,C,H,W);
 * rois (R,5) [batch, x1, y1, x2, y2]; out (R,C,ph,pw).  Accumulates in double. */
void orc_roi_align_forward(const float *feat, int c, int h, int w, const float *rois, int r,
                           float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                           float *out);
/* grad_feat (N,C,H,W) double accumulators, caller-zeroed */
void orc_roi_align_backward(const float *grad_out, int c, int h, int w, const float *rois, int r,
                            float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                            double *grad_feat);

/* 3D augmentation of one view of a point cloud, the reference's host chain restated:
 * lidar_points.py:28-33 (flip), base_points.py:139-179 (rotate: p @ M, k-ordered fused multiply-adds as
 * the BLAS behind torch's (N,3)@(3,3)), :263-269 (scale), :186-205 (translate), :207-229 (in_range_3d,
 * strict), then the rows are kept in slot order; slot j reads source row perm[j] (or j).
 * params: 24 floats, layout of include/detmatch_hip.h dm_points_augment.  Returns the kept count. */
int orc_points_augment(const float *points, int n, int n_feat, const float *params,
                       const int32_t *perm, float *out);

#ifdef __cplusplus
}
#endif
#endif
