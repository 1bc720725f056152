/*
 * detmatch_hip.h — C-ABI of libdetmatch_hip.so, the MI355X (gfx950) native
 * operator library behind the DetMatch training step.
 *
 * This is the drop-in boundary "B2" of SURVEY.md §8(b): every entry point
 * replaces one pybind11/at::Tensor function of the reference's compiled
 * extensions (file:line cited per function).  Signatures use plain pointers
 * and sizes only — no torch types — so the same library binds from ctypes,
 * pybind, cgo or JNI alike.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - tensors are dense row-major ("contiguous"), fp32 / int32 unless noted;
 *   - `stream` is a hipStream_t (passed as void*); all work is stream-ordered,
 *     nothing here calls hipMalloc / hipFree / hipDeviceSynchronize;
 *   - scratch memory is caller-provided: ask `dm_*_workspace_bytes` first;
 *   - return value: 0 = DM_OK, otherwise a DM_ERR_* code (never exit()):
 *     see dm_error_string().  The reference reports errors by TORCH_CHECK /
 *     TV_ASSERT_RT_ERR exceptions or fprintf+exit(-1) (iou3d_nms.cpp:14-25,
 *     ball_query_gpu.cu:85-89); the Python host layer turns a non-zero code
 *     into RuntimeError.
 *   - ops whose OUTPUT SIZE is data dependent write the size to a device int32
 *     and never sync; the caller decides when to read it back.
 *
 * Reference paths below are relative to the reference tree; `pcdet/` stands
 * for thirdparty/Spconv-OpenPCDet/pcdet/ and `spconv/` for mmdet3d/ops/spconv/.
 */
#ifndef DETMATCH_HIP_H_
#define DETMATCH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dm_stream_t; /* hipStream_t */

enum {
  DM_OK = 0,
  DM_ERR_INVALID_ARG = 1,   /* bad size / null pointer / unsupported shape   */
  DM_ERR_WORKSPACE = 2,     /* workspace smaller than dm_*_workspace_bytes   */
  DM_ERR_INT32_RANGE = 3,   /* batch * volume does not fit the int32 cell id */
  DM_ERR_UNSUPPORTED = 4,   /* channel count / kernel volume not compiled in */
  DM_ERR_LAUNCH = 5         /* hipGetLastError() != hipSuccess after launch  */
};

const char *dm_version(void);
const char *dm_error_string(int code);

/* ------------------------------------------------------------------------ */
/* A. Hard voxelization  (+ fused MeanVFE)                                    */
/* ------------------------------------------------------------------------ */
/* Replaces voxel_layer.hard_voxelize
 *   mmdet3d/ops/voxel/src/voxelization.h:51-69 (dispatch),
 *   voxelization_cuda.cu:184-326 (hard_voxelize_gpu), semantics ==
 *   voxelization_cpu.cpp:44-141,
 * for a whole batch at once (the reference loops samples in Python,
 * mmdet3d/models/detectors/openpcdet.py:61-76) and MeanVFE.forward
 * (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-28).
 *
 * points          (n_total, c) all samples stacked; sample b owns rows
 *                 [offsets_host[b], offsets_host[b+1])
 * voxels          (batch*max_voxels, max_points, c)  zero-filled by the callee
 * coors           (batch*max_voxels, coor_dim): coor_dim 3 -> [z,y,x]
 *                 (reference layout), 4 -> [b,z,y,x] (already batch-padded as
 *                 openpcdet.py:69-72 does)
 * num_points      (batch*max_voxels)
 * mean_feats      (batch*max_voxels, c) or NULL: sum over the voxel's points /
 *                 max(num,1)
 * voxel_counts    (batch+1) int32: [0..batch) = voxels kept per sample,
 *                 [batch] = total.  Rows of all outputs are compacted: sample
 *                 b's voxels start at sum(voxel_counts[0..b)).
 * Voxel order = order of each voxel's first point; points beyond max_points in
 * a voxel, and every point of a voxel first seen after max_voxels voxels of its
 * sample exist, are dropped (bit-exact with the reference CPU path).
 */
size_t dm_hard_voxelize_workspace_bytes(int n_total, int batch);
int dm_hard_voxelize(const float *points, int n_total, int c,
                     const int32_t *offsets_host, int batch,
                     const float *voxel_size_host /*[3] x,y,z*/,
                     const float *coors_range_host /*[6]*/, int max_points,
                     int max_voxels, int coor_dim, float *voxels, int32_t *coors,
                     int32_t *num_points, float *mean_feats, int32_t *voxel_counts,
                     void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* B. Sparse convolution: rulebook                                            */
/* ------------------------------------------------------------------------ */
/* Replaces sparse_conv_ext.get_indice_pairs_3d
 *   spconv/src/all.cc:24, spconv/include/spconv/spconv_ops.h:28-141,
 *   kernels spconv/include/spconv/indice.cu.h:24-204, geometry.h:25-86.
 *
 * The native rulebook is a pair of dense GATHER TABLES instead of 27
 * atomically-filled pair lists:
 *   nbr_out (kvol, n_out): nbr_out[k][o] = input row feeding output row o
 *                          through kernel offset k, or -1
 *   nbr_in  (kvol, n_in) : nbr_in[k][i]  = output row fed by input row i
 *                          through offset k, or -1  (strided conv only; for a
 *                          sub-manifold conv nbr_in[k] == nbr_out[kvol-1-k])
 * plus, for API parity and for the weight-gradient kernel, the reference's
 *   indice_pairs (kvol, 2, n_in) int32, -1 padded; [k][0][s] = in, [k][1][s] = out
 *   indice_num   (kvol)
 * Pair slots are filled in ascending row order of the table they are read from
 * — output rows (= input rows) for a sub-manifold conv, INPUT rows for a strided
 * conv — deterministic; the reference GPU order is an atomicAdd race
 * (indice.cu.h:46-52).
 * Kernel offset index k = kz*ky_size*kx_size + ky*kx_size + kx with
 * k_axis = in - out*stride + pad  (geometry.h:62-71).
 * Output rows of a strided conv are sorted by ascending flat cell id
 * (b*vol + (z*Y + y)*X + x) — the reference GPU order (torch::_unique,
 * spconv_ops.h:130).  dilation is 1 (the only value VoxelBackBone8x uses).
 *
 * Strided convs are two-phase because n_out is data dependent:
 *   dm_rulebook_conv_count  -> n_out_dev (device int32); the occupancy bitmap of
 *                              the output grid with its popcount prefixes (or,
 *                              on the hash path, the candidate cell ids) kept in
 *                              the workspace
 *   (caller reads n_out, allocates exact-size outputs)
 *   dm_rulebook_conv_fill   -> out_ids, tables, pair lists
 * The same workspace must be passed, untouched, to both phases.
 *
 * Two implementations with identical results.  Bitmap (default whenever one bit
 * per output cell plus a 32-bit prefix per word fits the workspace, i.e. for any
 * real frame): mark -> popcount prefix -> rank lookups; 4 launches for the count
 * phase, 3 for the fill phase, no hash, no sort.  Hash + radix sort: inputs of a
 * few rows on a huge grid, where clearing and scanning the bitmap would cost
 * more than the candidates.  dm_rulebook_set_mode: 0 automatic, 1 hash + sort
 * always, 2 bitmap or DM_ERR_WORKSPACE (developer / test switch, process-wide).
 */
size_t dm_rulebook_workspace_bytes(int n_in, int kvol);
int dm_rulebook_set_mode(int mode);

int dm_rulebook_subm(const int32_t *indices /*(n,4) b,z,y,x*/, int n, int batch,
                     const int *spatial_shape_host /*[3] z,y,x*/,
                     const int *ksize_host /*[3]*/, int32_t *nbr_out /*(kvol,n)*/,
                     int32_t *indice_pairs /*(kvol,2,n) or NULL*/,
                     int32_t *indice_num /*(kvol)*/, void *workspace,
                     size_t workspace_bytes, dm_stream_t stream);

int dm_rulebook_conv_count(const int32_t *indices, int n, int batch,
                           const int *spatial_shape_host, const int *out_shape_host,
                           const int *ksize_host, const int *stride_host,
                           const int *padding_host, int32_t *n_out_dev,
                           void *workspace, size_t workspace_bytes,
                           dm_stream_t stream);

int dm_rulebook_conv_fill(const int32_t *indices, int n, int batch,
                          const int *spatial_shape_host, const int *out_shape_host,
                          const int *ksize_host, const int *stride_host,
                          const int *padding_host, int n_out,
                          int32_t *out_ids /*(n_out,4)*/, int32_t *nbr_out /*(kvol,n_out)*/,
                          int32_t *nbr_in /*(kvol,n)*/,
                          int32_t *indice_pairs /*(kvol,2,n) or NULL*/,
                          int32_t *indice_num /*(kvol)*/, void *workspace,
                          size_t workspace_bytes, dm_stream_t stream);

/* Capacity-sized builds — SURVEY 8(b) B2: "ops whose output size is data-dependent ... must offer a device-resident
 * count + capacity-bounded output so the caller can defer the read-back".  The reference reads numActOut back inside
 * every strided layer (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:58-141: `indiceNum.cpu()`, `outInds.slice(0, 0,
 * numAct)`), four times per backbone pass, and the voxel count before the first one.  Here the row count of the INPUT
 * is a device scalar too (`n_dev`: the voxelizer's device count, or the *n_out_dev of the previous layer) and `cap` /
 * `cap_in` its capacity: launches and table strides are sized by the capacity, the kernels stop at the count.  Layout:
 * nbr_out (kvol, cap_out), nbr_in (kvol, cap_in), indice_pairs (kvol, 2, cap_in), out_ids (cap_out, 4); rows beyond the
 * count are -1 in the tables, unspecified in out_ids.  *n_out_dev > cap_out means the tables are truncated (the caller
 * rebuilds that layer with dm_rulebook_conv_count / _fill).  The whole chain voxelize -> subm1 -> spconv2 -> ... of a
 * pass can be issued without a host value; ONE read-back at the end returns every count (spconv/ops.py:
 * build_rulebook_cap / finish_rulebook, pcdet/backbones_3d.py:build_rulebooks_deferred).  Results within the counts are
 * bit-identical to the two-phase entries.  DM_ERR_WORKSPACE: the occupancy bitmap does not fit the workspace
 * (dm_rulebook_workspace_bytes(cap_in, kvol)) — use the two-phase entries. */
int dm_rulebook_subm_cap(const int32_t *indices /*(cap,4)*/, const int32_t *n_dev, int cap, int batch,
                         const int *spatial_shape_host, const int *ksize_host, int32_t *nbr_out /*(kvol,cap)*/,
                         int32_t *indice_pairs /*(kvol,2,cap) or NULL*/, int32_t *indice_num /*(kvol)*/,
                         void *workspace, size_t workspace_bytes, dm_stream_t stream);
int dm_rulebook_conv_cap(const int32_t *indices /*(cap_in,4)*/, const int32_t *n_dev, int cap_in, int batch,
                         const int *spatial_shape_host, const int *out_shape_host, const int *ksize_host,
                         const int *stride_host, const int *padding_host, int cap_out, int32_t *n_out_dev,
                         int32_t *out_ids /*(cap_out,4)*/, int32_t *nbr_out /*(kvol,cap_out)*/,
                         int32_t *nbr_in /*(kvol,cap_in)*/, int32_t *indice_pairs /*(kvol,2,cap_in) or NULL*/,
                         int32_t *indice_num /*(kvol)*/, void *workspace, size_t workspace_bytes,
                         dm_stream_t stream);

/* Rebuild a gather table from reference-format pair lists (for callers that
 * hold a rulebook produced elsewhere).  side = 1: table[k][pairs[k][1][s]] =
 * pairs[k][0][s] (n_rows = n_out, forward); side = 0: the transpose. */
int dm_pairs_to_table(const int32_t *indice_pairs, const int32_t *indice_num,
                      int kvol, int pair_stride, int side, int32_t *table,
                      int n_rows, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* B. Sparse convolution: fused gather-GEMM-scatter                           */
/* ------------------------------------------------------------------------ */
/* Replaces sparse_conv_ext.indice_conv_fp32 / indice_conv_backward_fp32
 *   spconv/src/all.cc:32-33, spconv_ops.h:260-360 (fwd), :363-456 (bwd),
 *   kernels spconv/include/spconv/reordering.cu.h:22-160.
 *
 * out[o,:] = sum_k feat[nbr[k][o],:] @ W[k]          (output stationary, no
 * atomics, no intermediate buffers, one launch per layer; fp32 MFMA
 * v_mfma_f32_16x16x4_f32, exact fp32 fma chains).
 * filters (kvol, cin, cout) = the reference's (kz,ky,kx,cin,cout) viewed flat.
 *
 * transpose_w = 0: forward.      B_k = W[k]            (cin -> cout)
 * transpose_w = 1: input-gradient. feat := out_grad (n_rows_in x cout), the
 *                  table is nbr_in, B_k = W[k]^T (cout -> cin); for a
 *                  sub-manifold conv pass nbr_out and flip_k = 1 (B_k =
 *                  W[kvol-1-k]^T).
 * Supported channel counts: cin, cout in {4(in only),16,32,64,128}.
 */
size_t dm_spconv_workspace_bytes(int kvol, int cin, int cout);
int dm_spconv_gather_gemm(const float *feat, int n_rows_in, const float *filters,
                          const int32_t *nbr /*(kvol, n_rows_out)*/, int n_rows_out,
                          int kvol, int cin, int cout, int transpose_w, int flip_k,
                          float *out /*(n_rows_out, transpose_w ? cin : cout)*/,
                          const int32_t *tile_order /* from dm_spconv_tile_order, or NULL */,
                          const int32_t *row_perm /* from dm_spconv_pack_rows (then nbr is the packed
                                                     table and row p is written to out[row_perm[p]]), or NULL */,
                          void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* The same on the 16-bit matrix instructions (v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulation).
 * Replaces sparse_conv_ext.indice_conv_half / indice_conv_backward_half (spconv/src/all.cc:35-36; the
 * input-gradient half of the latter: transpose_w = 1) and serves the mixed-precision mode behind the
 * reference's fp16 configs (mmdet3d/apis/ssl_train.py:100-105).  `storage`:
 *   DM_SP16_F32ROWS  feat / filters / out are fp32 (same buffers as dm_spconv_gather_gemm); the multiplicands
 *                    are rounded to bf16 (round to nearest even) inside the kernel
 *   DM_SP16_F16      feat / filters / out are IEEE half (at::Half)
 *   DM_SP16_BF16     feat / filters / out are bfloat16
 *   DM_SP16_F32SPLIT feat / filters / out are fp32 and so is the arithmetic, to fp32 accuracy: each multiplicand is
 *                    split into three bf16 numbers (its 24 significand bits) inside the kernel and the six
 *                    significant cross products are accumulated in fp32 — an alternative to
 *                    dm_spconv_gather_gemm's v_mfma_f32_16x16x4_f32 at 3/8 of its matrix-pipe time
 * cin, cout in {16,32,64,128}; tables, tile order and row permutation exactly as above. */
enum { DM_SP16_F32ROWS = 0, DM_SP16_F16 = 1, DM_SP16_BF16 = 2, DM_SP16_F32SPLIT = 3 };
size_t dm_spconv16_workspace_bytes(int kvol, int cin, int cout);
int dm_spconv_gather_gemm16(const void *feat, int n_rows_in, const void *filters, int storage,
                            const int32_t *nbr, int n_rows_out, int kvol, int cin, int cout,
                            int transpose_w, int flip_k, void *out, const int32_t *tile_order,
                            const int32_t *row_perm, void *workspace, size_t workspace_bytes,
                            dm_stream_t stream);
/* Launch order of the 16-row output tiles of a gather table: tiles sorted by descending number of
 * active kernel offsets (a launch lasts as long as the compute unit that received the heaviest
 * tiles; heavy-first dispatch balances them).  A property of the table: build once per rulebook,
 * pass to every dm_spconv_gather_gemm on that table (same order for flip_k = 1).  Results do not
 * depend on it.  order: device int32[ceil(n_rows/16)], a permutation of the tile indices. */
size_t dm_spconv_tile_order_workspace_bytes(void);
/* Rows of a gather table grouped by equal neighbour mask (stable sort of the rows by the mask): a
 * 16-row tile pays a whole MFMA block per kernel offset any of its rows uses, so tiles of like rows do
 * 20-45 % less work.  perm (n_rows): packed position -> row; nbr_packed (kvol, n_rows) = nbr[:, perm].
 * A property of the table: build once per rulebook, then pass nbr_packed / perm (and the tile order
 * OF nbr_packed) to dm_spconv_gather_gemm.  Results equal the unpacked launch up to fp32 summation
 * order over the kernel offsets. */
size_t dm_spconv_pack_rows_workspace_bytes(int n_rows);
int dm_spconv_pack_rows(const int32_t *nbr /*(kvol, n_rows)*/, int n_rows, int kvol, int32_t *perm,
                        int32_t *nbr_packed, int32_t *tile_order_packed /* optional: dm_spconv_tile_order
                        of nbr_packed, int32[ceil(n_rows/16)] */, void *workspace, size_t workspace_bytes,
                        dm_stream_t stream);
int dm_spconv_tile_order(const int32_t *nbr /*(kvol, n_rows)*/, int n_rows, int kvol,
                         int32_t *order, void *workspace, size_t workspace_bytes,
                         dm_stream_t stream);

/* filt_grad[k] = sum_s feat[pairs[k][0][s],:]^T (x) out_grad[pairs[k][1][s],:]
 * (spconv_ops.h:436-441).  Deterministic two-stage reduction (partial slabs in
 * the workspace, then a fixed-order sum) — no float atomics. */
/* The weight gradients of SEVERAL layers (a whole backward pass of the sparse backbone: every layer's inputs exist
 * once the input-gradient chain has passed it) in one launch pair: one grid over all (layer, offset, chunk of
 * pairs) units + one reduce over all weight elements, instead of two launches per layer (spconv_ops.h:363-456 runs
 * per layer and per offset).  A job holds the arguments of dm_spconv_wgrad; accumulate != 0: filt_grad += (the
 * caller's gradient buffer already holds an earlier pass); layers outside the batched channel pairs run through
 * dm_spconv_wgrad inside the call (not with accumulate).  Launches: one rows kernel per register class of the
 * layers present (Cin, Cout <= 32 incl. the 4-channel input layer | 32 -> 64 and 64 -> 64 | 64 -> 128) + one reduce.
 * At most 16 batched jobs per call; results are bit-identical to dm_spconv_wgrad's (same chunks, same fixed-order
 * sums) except for the 4-channel layer, which rides on the 16-row tile here (equal to fp32 rounding). */
typedef struct dm_spconv_wgrad_job {
  const float *feat, *out_grad;
  const int32_t *indice_pairs, *indice_num;
  float *filt_grad;
  int pair_stride, kvol, cin, cout;
} dm_spconv_wgrad_job;
size_t dm_spconv_wgrad_batch_workspace_bytes(const dm_spconv_wgrad_job *jobs, int n_jobs);
int dm_spconv_wgrad_batch(const dm_spconv_wgrad_job *jobs, int n_jobs, int accumulate, void *workspace,
                          size_t workspace_bytes, dm_stream_t stream);
size_t dm_spconv_wgrad_workspace_bytes(int n_in, int kvol, int cin, int cout);
int dm_spconv_wgrad(const float *feat, const float *out_grad,
                    const int32_t *indice_pairs /*(kvol,2,pair_stride)*/,
                    const int32_t *indice_num /*(kvol) device*/, int pair_stride,
                    int kvol, int cin, int cout, float *filt_grad /*(kvol,cin,cout)*/,
                    void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* E. Rotated BEV overlap / IoU, rotated NMS                                  */
/* ------------------------------------------------------------------------ */
/* Replaces iou3d_nms_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu
 *   pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:50-88, iou3d_nms_kernel.cu:236-264.
 * boxes (N,7) [x,y,z,dx,dy,dz,heading]; ans (na, nb). */
size_t dm_iou3d_workspace_bytes(int na, int nb);
int dm_boxes_overlap_bev(const float *boxes_a, int na, const float *boxes_b, int nb,
                         float *ans_overlap, void *workspace, size_t workspace_bytes,
                         dm_stream_t stream);
/* Exact rotated BEV intersection area (no 1 cm corner-containment margin, convex clipping in double
 * precision) for threshold / zero comparisons: KITTI evaluation (reference: rotate_iou_gpu_eval,
 * mmdet3d/core/evaluation/kitti_utils/rotate_iou.py) and the GT-paste collision test
 * (box_np_ops.box_collision_test).  Same box format and output as dm_boxes_overlap_bev. */
int dm_boxes_overlap_bev_exact(const float *boxes_a, int na, const float *boxes_b, int nb,
                               float *ans_overlap, dm_stream_t stream);
int dm_boxes_iou_bev(const float *boxes_a, int na, const float *boxes_b, int nb, float *ans_iou,
                     void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* Replaces iou3d_nms_cuda.nms_gpu / nms_normal_gpu
 *   pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:91-185, iou3d_nms_kernel.cu:267-359.
 * boxes must be sorted by descending score.  keep (n) int64 DEVICE (the reference's is a
 * host tensor filled after a D2H copy of the whole mask); num_keep device int32.  The
 * greedy pass stops after max_keep survivors (<= 0: no limit) — callers slice
 * keep[:post_max_size] anyway (model_nms_utils.py:19-20). */
size_t dm_nms_workspace_bytes(int n);
int dm_nms(const float *boxes, int n, float thresh, int max_keep, long long *keep, int *num_keep,
           void *workspace, size_t workspace_bytes, dm_stream_t stream);
int dm_nms_normal(const float *boxes, int n, float thresh, int max_keep, long long *keep,
                  int *num_keep, void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* 2-D axis-aligned NMS on (n,4) xyxy boxes sorted by descending score: IoU =
 * inter / (area_a + area_b - inter), suppress when IoU > thresh.  Replaces mmcv.ops.nms
 * (mmcv-full 1.3.16, un-vendored) as used through batched_nms at
 * mmdet3d/models/ssl_modules/bbox_utils.py:97 and inside the 2D detector. */
int dm_nms_2d(const float *boxes_xyxy, int n, float thresh, int max_keep, long long *keep,
              int *num_keep, void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* `batch` independent problems of n boxes each in one launch chain (the per-sample loops of
 * class_agnostic_nms, model_nms_utils.py:5-22 called per sample from pvrcnn_head / roi_head_template.py:52-75, and
 * of mmdet's per-image RPN NMS): boxes (batch, n, 7 | 4) contiguous and score-sorted per problem, keep
 * (batch, keep_stride) with keep_stride >= n, num_keep (batch), workspace >= batch * dm_nms_workspace_bytes(n).
 * Results per problem are identical to the single-problem entries; the greedy pass is one wave per problem, so
 * problems issued one after the other leave the chip idle `batch` times as long.  normal != 0: dm_nms_normal. */
int dm_nms_batch(const float *boxes, int batch, int n, float thresh, int max_keep, int normal, long long *keep,
                 long long keep_stride, int *num_keep, void *workspace, size_t workspace_bytes,
                 dm_stream_t stream);
int dm_nms_2d_batch(const float *boxes_xyxy, int batch, int n, float thresh, int max_keep, long long *keep,
                    long long keep_stride, int *num_keep, void *workspace, size_t workspace_bytes,
                    dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* B/D. Fused training-mode BatchNorm (+ReLU) over row-major (N, C) activations */
/* ------------------------------------------------------------------------ */
/* Replaces the BatchNorm1d + ReLU of every sparse-conv block
 * (pcdet/models/backbones_3d/spconv_backbone.py:9-28, spconv/modules.py:125-137) and the
 * BatchNorm2d + ReLU of the shared set-abstraction MLPs
 * (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:31-40) — torch.nn.functional.batch_norm
 * semantics (biased variance for normalisation, unbiased for running_var, momentum update;
 * running_* may be NULL).  C % 4 == 0, C/4 divides 256.  save_mean / save_invstd (C) feed the
 * backward.  gamma / beta may be NULL (1 / 0). */
size_t dm_bn_rows_workspace_bytes(long long n, int c);
int dm_bn_rows_forward(const float *x, long long n, int c, const float *gamma, const float *beta,
                       float eps, float momentum, float *running_mean, float *running_var, int relu,
                       float *y, float *save_mean, float *save_invstd, void *workspace,
                       size_t workspace_bytes, dm_stream_t stream);
/* BatchNorm (training mode) + ReLU + max over the `ns` consecutive rows of each of `m` groups — the last layer
 * of a shared MLP followed by the max-pool over nsample (pointnet2_modules.py:79-90): x (m * ns, C) ->
 * pooled (m, C), argmax (m, C) bytes (row inside the group); the normalised (m * ns, C) tensor is never
 * written.  Backward: grad_pooled (m, C) -> dense grad_x (m * ns, C), grad_gamma, grad_beta. */
int dm_bn_rows_max_forward(const float *x, long long m, int ns, int c, const float *gamma, const float *beta,
                           float eps, float momentum, float *running_mean, float *running_var,
                           float *pooled, unsigned char *argmax, float *save_mean, float *save_invstd,
                           void *workspace, size_t workspace_bytes, dm_stream_t stream);
int dm_bn_rows_max_backward(const float *grad_pooled, const unsigned char *argmax, const float *x, long long m,
                            int ns, int c, const float *gamma, const float *beta, const float *save_mean,
                            const float *save_invstd, float *grad_x, float *grad_gamma, float *grad_beta,
                            void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* The same with grad_pooled rows `ldg` floats apart (ldg >= c, a multiple of 4, pointer 16-byte aligned): the pooled
 * gradient of one grouper is a column block of the gradient of the concatenated features (torch.cat backward). */
int dm_bn_rows_max_backward_ld(const float *grad_pooled, long long ldg, const unsigned char *argmax, const float *x,
                               long long m, int ns, int c, const float *gamma, const float *beta,
                               const float *save_mean, const float *save_invstd, float *grad_x, float *grad_gamma,
                               float *grad_beta, void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* dm_bn_rows_forward / dm_bn_rows_max_forward with the column statistics already reduced to `blocks`
 * partials (layout (2, c, blocks): mean, M2) over counts[b] rows each (dm_rowgemm_stats): no pass over x for
 * the statistics. */
int dm_bn_rows_forward_pre(const float *x, long long n, int c, const float *gamma, const float *beta, float eps,
                           float momentum, float *running_mean, float *running_var, int relu, float *y,
                           float *save_mean, float *save_invstd, const float *partial, const float *counts,
                           int blocks, dm_stream_t stream);
int dm_bn_rows_max_forward_pre(const float *x, long long m, int ns, int c, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *pooled, unsigned char *argmax, float *save_mean,
                               float *save_invstd, const float *partial, const float *counts, int blocks,
                               dm_stream_t stream);
/* The same in evaluation mode (running statistics, no gradient): one launch. */
int dm_bn_rows_eval_max(const float *x, long long m, int ns, int c, const float *gamma, const float *beta,
                        const float *running_mean, const float *running_var, float eps, float *pooled,
                        dm_stream_t stream);
/* Evaluation mode (running statistics; the EMA teacher's BatchNorm layers), optional ReLU, one launch. */
int dm_bn_rows_eval(const float *x, long long n, int c, const float *gamma, const float *beta,
                    const float *running_mean, const float *running_var, float eps, int relu, float *y,
                    dm_stream_t stream);
/* grad_x (N,C), grad_gamma (C), grad_beta (C); the ReLU mask is recomputed from x. */
int dm_bn_rows_backward(const float *grad_out, const float *x, long long n, int c, const float *gamma,
                        const float *beta, const float *save_mean, const float *save_invstd, int relu,
                        float *grad_x, float *grad_gamma, float *grad_beta, void *workspace,
                        size_t workspace_bytes, dm_stream_t stream);

/* Bboxes3DTo2D on one sample (mmdet3d/models/ssl_modules/processors_3d.py:81-155):
 * apply_3d_transformation_bboxes(reverse=True) (bbox_utils.py:110-200) composed on the host into
 * xf17_host = {A[9] (centre' = centre @ A, row-major), t[3], s, sigma, off, 0, 0}
 * (size' = s * size, yaw' = sigma * yaw + off), then bbox_3d_to_bbox_2d (bbox_utils.py:372-441) with the
 * row-major 4x4 lidar2img16_host and the ORIGINAL image size.  boxes3d (n, 7) -> boxes2d (n, 4) xyxy,
 * valid (n) bytes (>= 3 corners inside the image and mean clamped depth >= 0.5).  Backward: gradient of
 * boxes2d -> gradient of boxes3d (the forward is recomputed; clip / clamp / min / max as torch). */
int dm_box3d_project_forward(const float *boxes3d, int n, const float *xf17_host,
                             const float *lidar2img16_host, float img_w, float img_h, float *boxes2d,
                             unsigned char *valid, dm_stream_t stream);
int dm_box3d_project_backward(const float *boxes3d, int n, const float *xf17_host,
                              const float *lidar2img16_host, float img_w, float img_h,
                              const float *grad_boxes2d, float *grad_boxes3d, dm_stream_t stream);

/* HungarianConsistency on one sample's index-aligned matched pairs
 * (mmdet3d/models/ssl_modules/consumers_3d.py:11-117; DetMatch configuration: mmdet FocalLoss on
 * torch.logit(in_scores, logit_eps) vs arg-max of target_scores, L1Loss on boxes / (w, h, w, h),
 * GIoULoss; reduction 'mean', loss_weight 1).  losses3 = {cls, l1, iou}; the unit gradients
 * (d loss / d in_scores (n, n_cls), d l1 / d in_boxes (n, 4), d iou / d in_boxes (n, 4)) are kept for
 * the backward, which scales them by the upstream gradient of losses3. */
int dm_consistency_loss_forward(const float *in_boxes, const float *in_scores, const float *target_boxes,
                                const float *target_scores, int n, int n_cls, float img_w, float img_h,
                                float alpha, float gamma, float logit_eps, float iou_eps, float *losses3,
                                float *unit_grad_scores, float *unit_grad_l1, float *unit_grad_iou,
                                dm_stream_t stream);
int dm_consistency_loss_backward(const float *grad_losses3, const float *unit_grad_scores,
                                 const float *unit_grad_l1, const float *unit_grad_iou, int n, int n_cls,
                                 float *grad_scores, float *grad_boxes, dm_stream_t stream);

/* bbox_2d_transform (mmdet3d/models/fusion_layers/coord_transform.py:121-175) on (n, 4) xyxy boxes:
 * ori2new = scale -> crop offset -> h-flip, else the reverse; backward != 0 maps the gradient of the
 * transformed boxes to the gradient of the input boxes. */
int dm_bbox2d_transform(const float *boxes_or_grad, int n, float scale_x, float scale_y, float crop_x,
                        float crop_y, float img_w, int flip, int ori2new, int backward, float *out,
                        dm_stream_t stream);

/* y (rows, n) = x (rows, k) . w^T, w (n, k) row-major as an nn.Linear / 1x1-conv weight — the shared MLP
 * layers over grouped rows (pointnet2_modules.py:31-40) and their input gradients (pass w^T).
 * k, n multiples of 4, k <= 136, n <= 192; fp32 MFMA, weights resident in LDS. */
int dm_rowgemm_supported(int k, int n);
int dm_rowgemm(const float *x, const float *w, float *y, long long rows, int k, int n, dm_stream_t stream);
/* y = x . w^T plus the statistics of y's columns for the BatchNorm that follows: partial
 * (2, n, dm_rowgemm_parts(rows, k, n)) per-workgroup (mean, M2), counts (parts) rows per workgroup. */
int dm_rowgemm_parts(long long rows, int k, int n);
int dm_rowgemm_stats(const float *x, const float *w, float *y, long long rows, int k, int n, float *partial,
                     float *counts, dm_stream_t stream);
/* The same into rows of `ldy` floats starting at column `col0` (columns [0, col0) are written as zeros):
 * the input gradient of a first shared-MLP layer without the xyz / padding columns nobody differentiates. */
int dm_rowgemm_strided(const float *x, const float *w, float *y, long long rows, int k, int n, int ldy,
                       int col0, dm_stream_t stream);
/* The same with the weight given TRANSPOSED: y[:, col0:col0+n] = x (rows, k) . wt, wt (k, wt_ld) row-major, its n
 * columns starting at `wt` — the input gradient of a linear layer from its stored (out, in) weight (x = dY, k = out,
 * wt = weight + first live input column, wt_ld = in) without materialising weight^T per call (torch: `gy @ w`). */
int dm_rowgemm_wt(const float *x, const float *wt, int wt_ld, float *y, long long rows, int k, int n, int ldy,
                  int col0, dm_stream_t stream);

/* dw (n, k) [+]= dy (rows, n)^T . x (rows, k): the WEIGHT gradient of the same tall-skinny layers (torch: `gy.t() @ x`, in
 * the reference's autograd; here until round 4 a batched split-K BLAS call + sum).  n <= 64, k <= 160, multiples of 4;
 * fp32-class split arithmetic (dm_dconv_set_math mode 2's six bf16 products, fp32 accumulate), fixed row ranges per
 * workgroup and a fixed-order reduce: bitwise reproducible.  workspace: dm_tall_wgrad_workspace_bytes(rows, n, k). */
int dm_tall_wgrad_supported(int n, int k);
size_t dm_tall_wgrad_workspace_bytes(long long rows, int n, int k);
int dm_tall_wgrad(const float *dy, const float *x, float *dw, long long rows, int n, int k, int accumulate,
                  void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* Anchor-head box decoding: AnchorHeadTemplate.generate_predicted_boxes
 * (pcdet/models/dense_heads/anchor_head_template.py:225-272) = ResidualCoder.decode_torch
 * (pcdet/utils/box_coder_utils.py:43-76) + direction-bin correction (common_utils.limit_period).
 * box_encodings (n_total, 7), anchors (n_anchors, 7) reused every n_anchors rows, dir_logits
 * (n_total, n_dir_bins) or NULL -> boxes (n_total, 7).  Bit-identical to the fp32 tensor chain. */
int dm_anchor_decode(const float *box_encodings, const float *anchors, const float *dir_logits,
                     long long n_total, int n_anchors, int n_dir_bins, float dir_offset,
                     float dir_limit_offset, float period, float *boxes, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* G. 2D branch: RoIAlign over an FPN pyramid                                 */
/* ------------------------------------------------------------------------ */
/* Replaces mmcv.ops.RoIAlign (mmcv-full 1.3.16, un-vendored: "parity unpinned") as driven by
 * mmdet SingleRoIExtractor — configs/detmatch/001/detmatch/split_0.py:76-80,
 * mmdet3d/models/ssl_modules/processors/processors_2d.py:52-54.  avg pooling;
 * aligned != 0: pixel-centre shift -0.5, no minimum RoI size; sampling_ratio 0 = adaptive
 * ceil(roi / pooled).  feats_host[l]: device pointers (NCHW fp32) of the n_levels (<= 8)
 * maps; the three *_host arrays are HOST arrays of length n_levels; rois (R,5) device
 * [batch_idx, x1, y1, x2, y2]; roi_levels (R) device int32 (NULL = all level 0);
 * out (R, C, pooled_h, pooled_w).  max_grid: LDS sizing hint (samples per bin per axis,
 * 0 = default); RoIs needing more still produce exact results on a slower path. */
int dm_roi_align_forward(const float *const *feats_host, const int *heights_host,
                         const int *widths_host, const float *scales_host, int n_levels,
                         int channels, const float *rois, const int *roi_levels, int n_rois,
                         int pooled_h, int pooled_w, int sampling_ratio, int aligned, int max_grid,
                         float *out, dm_stream_t stream);
/* grads_host[l]: device pointers of the per-level input gradients, ACCUMULATED into (float
 * atomics; the caller zero-fills them). */
int dm_roi_align_backward(float *const *grads_host, const int *heights_host,
                          const int *widths_host, const float *scales_host, int n_levels,
                          int channels, const float *rois, const int *roi_levels, int n_rois,
                          int pooled_h, int pooled_w, int sampling_ratio, int aligned, int max_grid,
                          const float *grad_out, dm_stream_t stream);

/* Forward on NHWC maps (feats_nhwc_host[l]: (N, H_l, W_l, C) device pointers): separable taps,
 * thread = channel, every feature load a 256-byte-contiguous wave instruction.  Same results as
 * dm_roi_align_forward up to fp32 summation order.  pooled_h, pooled_w <= 16. */
int dm_roi_align_forward_nhwc(const float *const *feats_nhwc_host, const int *heights_host,
                              const int *widths_host, const float *scales_host, int n_levels,
                              int channels, const float *rois, const int *roi_levels, int n_rois,
                              int pooled_h, int pooled_w, int sampling_ratio, int aligned, float *out,
                              dm_stream_t stream);
/* Same gradient, separable form: grads_nhwc_host[l] are (N, H_l, W_l, C) buffers (caller-zeroed,
 * accumulated into with one 256-byte-contiguous float atomic per pixel and 64 channels; RoIs whose
 * pixel span exceeds the kernel's LDS tables fall back to per-sample atomics in place). */
int dm_roi_align_backward_nhwc(float *const *grads_nhwc_host, const int *heights_host,
                               const int *widths_host, const float *scales_host, int n_levels,
                               int channels, const float *rois, const int *roi_levels, int n_rois,
                               int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                               const float *grad_out, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* H. Teacher-student support: fused EMA, host LAP                            */
/* ------------------------------------------------------------------------ */
/* Replaces SSL._update_teacher (mmdet3d/models/detectors/ssl.py:146-163) over flat,
 * identically laid-out, 16-byte aligned arenas: t = s * f32(1-d) + t * f32(d). */
int dm_ema_update_f32(float *teacher, const float *student, size_t n, double decay,
                      dm_stream_t stream);
/* integer buffers: fp32 math, truncated back (Tensor.copy_ float -> long) */
int dm_ema_update_i64(long long *teacher, const long long *student, size_t n, double decay,
                      dm_stream_t stream);
/* I. Step driver: fused optimizer steps over flat, 16-byte aligned arenas (params, grads and
 * state index-aligned).  Replace the per-tensor torch.optim loops behind HybridOptimizer.step
 * (mmdet3d/core/optimizer/hybrid_optimizer.py:82-101) for the optimizers named at
 * configs/detmatch/001/detmatch/split_0.py:832-851.  grad_scale_dev: optional device scalar that
 * multiplies every gradient (the clip_grad_norm_ coefficient).  `step` is 1-based. */
int dm_adamw_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                      size_t n, double lr, double beta1, double beta2, double eps,
                      double weight_decay, long long step, const float *grad_scale_dev,
                      dm_stream_t stream);
int dm_sgd_step_f32(float *params, const float *grads, float *momentum_buf, size_t n, double lr,
                    double momentum, double dampening, double weight_decay, int first_step,
                    const float *grad_scale_dev, dm_stream_t stream);
/* The same with a liveness mask: block_live[i / 4] == 0 leaves elements 4i..4i+3 (parameter,
 * state) untouched — torch.optim skips parameters whose .grad is None, which is what a parameter
 * that never received a gradient has under mmcv's zero_grad + DDP(find_unused_parameters=True)
 * (mmdet3d/apis/ssl_train.py:65-69).  block_live may be NULL (= all live). */
int dm_adamw_step_masked_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                             size_t n, double lr, double beta1, double beta2, double eps,
                             double weight_decay, long long step, const float *grad_scale_dev,
                             const unsigned char *block_live, dm_stream_t stream);
int dm_sgd_step_masked_f32(float *params, const float *grads, float *momentum_buf, size_t n,
                           double lr, double momentum, double dampening, double weight_decay,
                           int first_step, const float *grad_scale_dev,
                           const unsigned char *block_live, dm_stream_t stream);
/* The same with a per-block FIRST step: block_first_step[i / 4] = the (1-based) optimizer step at which the
 * parameter holding elements 4i..4i+3 received its first gradient on any rank, 0 = never (skipped like
 * block_live == 0).  torch.optim keeps its state per parameter: AdamW's bias corrections run on the
 * parameter's own step count (step - first + 1), SGD creates the momentum buffer at the parameter's first
 * gradient (torch/optim/adamw.py, sgd.py, driven per parameter by hybrid_optimizer.py:82-101).  May be NULL
 * (every parameter live since step 1). */
int dm_adamw_step_blocks_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                             size_t n, double lr, double beta1, double beta2, double eps,
                             double weight_decay, long long step, const float *grad_scale_dev,
                             const unsigned char *block_live, const int *block_first_step,
                             dm_stream_t stream);
int dm_sgd_step_blocks_f32(float *params, const float *grads, float *momentum_buf, size_t n,
                           double lr, double momentum, double dampening, double weight_decay,
                           long long step, const float *grad_scale_dev, const unsigned char *block_live,
                           const int *block_first_step, dm_stream_t stream);
/* C. Anchor target assignment.  Replaces AxisAlignedTargetAssigner.assign_targets(_single)
 * (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-209) for all
 * samples x anchor classes of a batch (POS_FRACTION < 0, MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES
 * False).  anchors (C, A, 7) class-major, anchors_bev / gt_bev their axis-aligned BEV rectangles
 * [x1,y1,x2,y2] (box_utils.py:272-283, computed by the caller), gt_boxes (B, M, 8) with the class id
 * in the last column, class_ids / matched / unmatched (C).  Outputs in the reference's layout:
 * labels (B, L*C*R) int32 (-1 = ignore), reg_targets (B, L*C*R, 7), reg_weights (B, L*C*R), index
 * (l*C + c)*R + r with A = L*R.  Labels are bit-exact with the reference rules. */
size_t dm_anchor_assign_workspace_bytes(int B, int M, int C);
int dm_anchor_assign(const float *anchors, const float *anchors_bev, const float *gt_boxes,
                     const float *gt_bev, const int *class_ids, const float *matched,
                     const float *unmatched, int B, int M, int C, int A, int R, int num_class,
                     int *labels, float *reg_targets, float *reg_weights, void *workspace,
                     size_t workspace_bytes, dm_stream_t stream);
/* D. Key-point features from the BEV map (VoxelSetAbstraction.interpolate_from_bev_features +
 * bilinear_interpolate_torch, pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:9-40,113-117).
 * bev_nhwc (B, H, W, C) (C % 4 == 0, <= 1024); keypoints: B * n_keypoints rows of keypoint_stride floats
 * starting at (x, y); geom5 = {range_x0, range_y0, voxel_x, voxel_y, bev_stride}.  out (B, K, C); cells
 * (B, K, 4) int32 and weights (B, K, 4) are the four cells / weights of every key point (clamped as in
 * the reference), kept for the backward pass, which writes the whole gradient map (zeros + one
 * deterministic scatter, no atomics). */
int dm_bev_interpolate_forward(const float *bev_nhwc, int batch, int height, int width, int channels,
                               const float *keypoints, int keypoint_stride, int n_keypoints,
                               const float *geom5, float *out, int *cells, float *weights,
                               dm_stream_t stream);
int dm_bev_interpolate_backward(const float *grad_out, const int *cells, const float *weights, int batch,
                                int height, int width, int channels, int n_keypoints,
                                float *grad_bev_nhwc, dm_stream_t stream);
/* H. Cost matrix of the 2D <-> 3D Hungarian matching in one launch (FusionHungarianMatching.match,
 * mmdet3d/models/ssl_modules/processors_fusion.py:50-222, with ModHungarianAssigner's costs,
 * modified_hungarian_assigner.py:19-162: DoubleSidedFocalLossCost on torch.logit'ed foreground
 * probabilities, BBoxL1Cost on image-normalised xyxy, IoUCost giou).  Rows: n3 <= 512 LiDAR boxes
 * (x, y, z_bottom, dx, dy, dz, yaw) projected with lidar2img16_host (row-major 4x4; bbox_utils.py:372-441)
 * — or, with boxes3d == NULL, already projected xyxy boxes in boxes_proj.  Columns: n2 <= 512 image
 * boxes.  scores*: (n, n_cls) foreground probabilities, n_cls <= 8.  cost (n3, n2); proj_out (n3, 4)
 * optional.  The assignment itself is dm_lap_host on the copied-back matrix. */
int dm_fusion_match_cost(const float *boxes3d, const float *boxes_proj, const float *scores3, int n3,
                         const float *boxes2d, const float *scores2, int n2, int n_cls,
                         const float *lidar2img16_host, float img_w, float img_h, float w_cls,
                         float w_reg, float w_iou, float focal_alpha, float focal_eps, float logit_eps,
                         float *cost, float *proj_out, dm_stream_t stream);
/* D / E. Second-stage targets and losses of PV-RCNN, one or two launches per call for the whole batch.
 *
 * dm_roi_targets replaces ProposalTargetLayer.forward (sample_rois_for_rcnn, subsample_rois,
 * get_max_iou_with_same_class; pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:13-259)
 * plus the canonical transform of RoIHeadTemplate.assign_targets (roi_head_template.py:104-134).
 * rois (B, R, 7), roi_scores (B, R), roi_labels (B, R) int64, gt_boxes (B, G, gt_cols) with the class id
 * in the last column (rows after the last non-zero row are padding).  The reference draws its random
 * choices with numpy (:153,:185-199); here they are inputs: u_perm (B, R) orders the foreground RoIs
 * (ascending key), u_pick (B, S) picks with replacement (index = floor(u * count)), both uniform in
 * [0, 1).  Outputs (S = roi_per_image): rois (B, S, 7), gt_of_rois_src / gt_of_rois (canonical frame)
 * (B, S, gt_cols), iou / scores (B, S), labels / reg_valid (B, S) int64, rcnn_cls_labels (B, S) for
 * CLS_SCORE_TYPE roi_iou, the sampled RoI index (B, S) int64 and the per-sample "has fg or bg" flag. */
size_t dm_roi_targets_workspace_bytes(int batch, int n_rois);
int dm_roi_targets(const float *rois, const float *roi_scores, const long long *roi_labels,
                   const float *gt_boxes, int batch, int n_rois, int n_gt, int gt_cols,
                   const float *u_perm, const float *u_pick, int roi_per_image, int fg_per_image,
                   float reg_fg_thresh, float cls_fg_thresh, float cls_bg_thresh,
                   float cls_bg_thresh_lo, float hard_bg_ratio, float *out_rois, float *out_gt_src,
                   float *out_gt_canonical, float *out_iou, float *out_scores, long long *out_labels,
                   long long *out_reg_valid, float *out_cls_labels, long long *out_sampled,
                   float *out_ok, void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* RoIHeadTemplate.get_box_cls_layer_loss + get_box_reg_layer_loss (roi_head_template.py:136-218:
 * BinaryCrossEntropy on sigmoid(rcnn_cls), smooth-l1 on the residual code of the canonical GT, corner
 * regularisation loss_utils.py:209-233).  out3 = [cls, reg, corner] (already weighted by
 * loss_weights3); g_cls (n), g_sl1 / g_corner (n, 7) are the gradients of out3[0], out3[1], out3[2]
 * w.r.t. rcnn_cls / rcnn_reg, combined with the upstream gradient by dm_rcnn_loss_backward. */
int dm_rcnn_loss_forward(const float *rcnn_cls, const float *rcnn_reg, const float *rois,
                         const float *gt_canonical, const float *gt_src, const long long *reg_valid,
                         const float *cls_labels, int n, int gt_cols, const float *loss_weights3,
                         const float *code_weights7, float beta, int corner_loss, float *out3,
                         float *g_cls, float *g_sl1, float *g_corner, dm_stream_t stream);
int dm_rcnn_loss_backward(const float *upstream3, const float *g_cls, const float *g_sl1,
                          const float *g_corner, int n, float *d_rcnn_cls, float *d_rcnn_reg,
                          dm_stream_t stream);
/* PointHeadSimple.assign_targets (point_head_simple.py:20-48 -> assign_stack_targets,
 * point_head_template.py:49-129, set_ignore_flag branch): label of every key point = class of the
 * first GT box that holds it (1 when num_class == 1), -1 if only the box enlarged by extra_width3
 * holds it, else 0; the in-box rule is roiaware_pool3d's points_in_boxes.  points: rows of
 * point_stride floats starting at x (n_points per sample), labels (B * n_points) int64. */
int dm_point_targets(const float *points, int point_stride, const float *gt_boxes, int batch,
                     int n_points, int n_gt, int gt_cols, const float *extra_width3, int num_class,
                     long long *labels, dm_stream_t stream);
/* PointHeadTemplate.get_cls_layer_loss (point_head_template.py:131-154): sigmoid focal loss
 * (alpha, gamma 2) over (n, n_cls) with weights 1 / max(#positive, 1) on labels >= 0, times
 * loss_weight.  out2 = [loss, #positive]; grad (n, n_cls) = d loss / d preds. */
int dm_point_focal_loss(const float *preds, const long long *labels, int n, int n_cls, float alpha,
                        float loss_weight, float *out2, float *grad, dm_stream_t stream);
/* G. Target assignment, sampling and losses of the 2-D detector, 4-5 launches per call for the whole
 * batch (csrc/det2d_targets.hip).  mmdet 2.14 is an un-vendored dependency of the reference; the rules
 * are MaxIoUAssigner.assign_wrt_overlaps (gt_max_assign_all), RandomSampler (neg_pos_ub -1),
 * DeltaXYWHBBoxCoder.encode, AnchorHead.loss (sampling) and BBoxHead.get_targets / loss as configured at
 * configs/detmatch/001/detmatch/split_0.py:39-99,440-478.  Per-image inputs are host arrays of device
 * pointers (batch <= 8, <= 256 GT boxes per image).  `keys`: uniform [0,1) numbers, one per box; the
 * sampled positives / negatives are the boxes with the smallest keys (a uniform random subset).
 *
 * dm_rpn_loss_forward: RPNHead.loss.  level_outputs[l]: (B, H_l, W_l, channels) NHWC head output with
 * the A objectness logits first, then the 4A deltas; anchors (n_anchors, 4) in (level, h, w, a) order.
 * out2 = [loss_rpn_cls, loss_rpn_bbox]; the gradient w.r.t. the head outputs is returned as sparse
 * entries (5 per sampled slot: offset into one flat buffer laid out by grad_offsets[l], value) that
 * dm_rpn_loss_backward scatters, scaled by the upstream gradients, into a zeroed buffer.
 * assigned_out (B, n_anchors) int32, optional: assigned_gt_inds (-1 ignore, 0 negative, k+1). */
size_t dm_det2d_assign_workspace_bytes(int batch, int n_boxes_max);
int dm_rpn_loss_forward(const float *const *level_outputs, const int *level_hw, int n_levels,
                        int n_base_anchors, int channels, const long long *grad_offsets,
                        const float *anchors, int n_anchors, const float *const *gt_boxes,
                        const int *n_gt, int batch, const float *keys, float pos_iou_thr,
                        float neg_iou_thr, float min_pos_iou, int match_low_quality, int num,
                        int num_pos_max, const float *means4, const float *stds4,
                        float loss_cls_weight, float loss_bbox_weight, float *out2,
                        long long *entry_offsets, float *entry_values, int *assigned_out,
                        void *workspace, size_t workspace_bytes, dm_stream_t stream);
int dm_rpn_loss_backward(const long long *entry_offsets, const float *entry_values,
                         const float *upstream2, int n_entries, float *grad_flat, dm_stream_t stream);
/* RPNHead.get_bboxes up to the NMS for all images and levels in two launches: per (image, level) the
 * nms_pre highest objectness scores (ties: lower anchor index), DeltaXYWHBBoxCoder.decode with border
 * clipping, min_bbox_size flags, and the batched_nms inputs (boxes shifted by level * (max coordinate
 * of the image + 1), dropped boxes at -1e6 with score -1).  n_out = sum_l min(nms_pre, H_l W_l A);
 * img_hw = {h_0, w_0, h_1, w_1, ...}; wh_ratio_clip_log = |log(wh_ratio_clip)|.  Outputs (B, n_out, ...),
 * level-major, score-descending within a level. */
/* minimum; with batch * n_anchors * 4 + 512 more bytes the selection keys are computed once by a chip-wide
 * pre-pass (3x faster selection on the finest level) */
size_t dm_rpn_proposals_workspace_bytes(int batch, int n_out);
int dm_rpn_proposals_pre_nms(const float *const *level_outputs, const int *level_hw, int n_levels,
                             int n_base_anchors, int channels, const float *anchors, int n_anchors,
                             int batch, const float *img_hw, int nms_pre, const float *means4,
                             const float *stds4, float wh_ratio_clip_log, int clip_border,
                             float min_bbox_size, int n_out, float *boxes, float *scores,
                             unsigned char *live, float *nms_boxes, float *nms_scores, void *workspace,
                             size_t workspace_bytes, dm_stream_t stream);
/* StandardRoIHead.forward_train up to the RoI extractor: assign (proposals, optionally with the GT
 * boxes in front), sample `num` RoIs per image (at most num_pos_max positives), BBoxHead.get_targets.
 * proposals[b]: (n_proposals, proposal_stride) rows with xyxy first, proposal_ok[b]: (n_proposals) bool.
 * keys: (batch, keys_stride), keys_stride >= n_proposals + max n_gt.  Outputs: rois (B*num, 5)
 * [image, x1, y1, x2, y2], labels (B*num) int64 (n_classes = background), label_weights (B*num),
 * bbox_targets / bbox_weights (B*num, 4); rows beyond the sampled ones are zero with weight 0. */
int dm_roi2d_targets(const float *const *proposals, const unsigned char *const *proposal_ok,
                     int n_proposals, int proposal_stride, const float *const *gt_boxes,
                     const long long *const *gt_labels, const int *n_gt, int batch,
                     int add_gt_as_proposals, const float *keys, int keys_stride, float pos_iou_thr,
                     float neg_iou_thr, float min_pos_iou, int match_low_quality, int num,
                     int num_pos_max, int n_classes, const float *means4, const float *stds4,
                     float *rois, long long *labels, float *label_weights, float *bbox_targets,
                     float *bbox_weights, void *workspace, size_t workspace_bytes, dm_stream_t stream);
/* BBoxHead.loss with a sigmoid FocalLoss (gamma 2) and L1Loss: out3 = [loss_cls, loss_bbox, acc];
 * grad_cls (n_rows, n_cls_out), grad_bbox (n_rows, 4 * (reg_class_agnostic ? 1 : n_classes)) are the
 * gradients of out3[0] / out3[1] w.r.t. cls_score / bbox_pred. */
int dm_bbox_head_loss(const float *cls_score, const float *bbox_pred, const long long *labels,
                      const float *label_weights, const float *bbox_targets, const float *bbox_weights,
                      int n_rows, int n_cls_out, int n_classes, int reg_class_agnostic,
                      float focal_alpha, float loss_cls_weight, float loss_bbox_weight, float *out3,
                      float *grad_cls, float *grad_bbox, dm_stream_t stream);
/* ------------------------------------------------------------------------ */
/* C / G. Dense 2-D convolutions (BEV backbone, anchor-head convs, ResNet-50 + FPN + RPN)      */
/* ------------------------------------------------------------------------ */
/* Replace the cuDNN convolutions behind torch.nn.Conv2d / ConvTranspose2d at
 *   pcdet/models/backbones_2d/base_bev_backbone.py:38-69,94-112,
 *   pcdet/models/dense_heads/anchor_head_single.py:20-37,
 *   and mmdet's ResNet / FPN / RPNHead as configured at
 *   configs/detmatch/001/detmatch/split_0.py:39-99
 * (the reference has no binding of its own here: F.conv2d -> cudnn).  Activations NHWC fp32,
 * implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).
 *
 * dm_dconv_pack: weights -> the [S][N][K] layout the GEMM reads (K contiguous):
 *   dst[s][n][k] = src[n*sn + k*sk + s*st] * scale_n[n] * scale_k[k], zero for n >= Nsrc, k >= Ksrc
 *   (scales may be NULL; the frozen-BatchNorm fold of the 2D backbone is a scale per output
 *   channel).  Conv2d weight (Cout,Cin,KH,KW): forward N=Cout,K=Cin: sn=Cin*T, sk=T, st=1;
 *   input gradient N=Cin,K=Cout: sn=T, sk=Cin*T, st=1.
 *
 * dm_dconv_gemm: the lattice convolution
 *   out[b, oy0+i*oys, ox0+j*oxs, n] (+bias[n]) (relu) =
 *       sum_t sum_c in[b, i*iys+dy[t], j*ixs+dx[t], c] * w_packed[ws[t]][n][c]
 *   geom_host = 17 ints {B,Hin,Win,Cin, Hout,Wout,Cout, LH,LW, oy0,ox0,oys,oxs, iys,ixs, T, relu};
 *   taps_host = 3*T int16 {dy[T], dx[T], ws[T]} (T <= 64).  Cin % 4 == 0.  Forward convolution,
 *   input gradient (per residue class for stride > 1) and ConvTranspose2d are all instances.
 *   Small problems split the reduction over workgroups (partial sums in the workspace, summed in a
 *   fixed order: bitwise reproducible); large ones run full rounds of 128x128 / 64x64 tiles plus a
 *   spread-out tail launch.
 *
 * dm_dconv_wgrad: G[t][u][v] = sum_{b,i,j} U[(b,i,j)][u] * V[b, i*vys+dy[t], j*vxs+dx[t]][v],
 *   written as out[u*su + v*sv + t*st] = scale_u[u] * G (v < Cv_out; `accumulate` adds).
 *   geom_host = 10 ints {B, LH, LW, Cu, Cv, Hv, Wv, vys, vxs, T}.  Split over pixel chunks with a
 *   fixed-order reduce: bitwise reproducible. */
int dm_dconv_pack(const float *src, float *dst, const float *scale_n, const float *scale_k, int S,
                  int N, int K, int Nsrc, int Ksrc, long long sn, long long sk, long long st,
                  dm_stream_t stream);
/* Arithmetic of dm_dconv_gemm: 0 fp32 on the matrix pipe's own fp32 instruction (v_mfma_f32_32x32x2_f32);
 * 1 mixed precision — bf16 multiplicands (inputs and weights rounded to nearest-even on their way into
 * LDS), fp32 accumulation and storage (v_mfma_f32_32x32x16_bf16), the counterpart of the reference's fp16
 * (autocast) configs; layers with Cin % 64 != 0 stay fp32;
 * 2 fp32-class arithmetic on the bf16 instruction: both operands are split on their way into LDS into three
 * bf16 numbers (together their 24 significand bits) and the six cross products of weight >= 2^-16 are
 * accumulated in fp32 — more accurate against float64 than mode 0 (tools/probe_bf16_split.py) at 3/8 of its
 * matrix-pipe time; layers with Cin % 32 != 0 or Cout <= 32 use mode 0.
 * Range of mode 2 (tests/test_dense_conv_gpu.py: max-error and non-finite tests): full fp32-class precision for
 * finite operands with 2^-109 <= |x| <= 3.38e38 (below, the lowest — then the middle — bf16 plane underflows and
 * up to 16 of the 24 significand bits of THAT operand are dropped; above bf16's largest finite value 3.3895e38
 * the high plane rounds to infinity).  +-inf / NaN operands make the same output elements non-finite as in mode 0,
 * but an infinite operand yields NaN where mode 0 yields +-inf (inf splits into inf + NaN + NaN).
 * Process-wide; dm_dconv_wgrad follows the mode for layers with more than 64 channels on both sides; in mode 2 the
 * weight gradient of a 3 x 3 / stride-1 / same-size layer with >= 64 channels on both sides is computed by a
 * tap-fused kernel (nine taps per workgroup) when the layer gives a workgroup >= 24 steps of 16 pixels — same six
 * products per term, another summation order over the pixels than the per-tap kernel (agreement to the fp32
 * accumulation error; bitwise reproducible run to run).  Developer switches for A/B timing: 2 + 16 = mode 2 without
 * the patch kernels, 2 + 32 = mode 2 with the per-tap weight gradient everywhere (dm_dconv_get_math reports 2). */
int dm_dconv_set_math(int mode);
int dm_dconv_get_math(void);
/* The same for n_entries weights in one launch.  table_dev: device array of 80-byte rows
 * {const float *src; float *dst; const float *scale_n, *scale_k; int64 sn, sk, st;
 *  int32 S, N, K, Nsrc, Ksrc, pad} — every packed weight of a network is refreshed by one launch after
 * an optimizer / EMA step instead of one launch per convolution. */
int dm_dconv_pack_batch(const void *table_dev, int n_entries, int blocks_per_entry, dm_stream_t stream);
size_t dm_dconv_gemm_workspace_bytes(const int *geom_host);   /* 0 unless the reduction is split */
int dm_dconv_gemm(const float *x, const float *w_packed, const float *bias, float *y,
                  const int *geom_host, const short *taps_host, void *workspace,
                  size_t workspace_bytes, dm_stream_t stream);
/* The same with the shortcut branch of a residual block in the epilogue:
 * y = relu?(conv + bias + residual), residual in y's layout (mmdet ResNet Bottleneck.forward:
 * `out += identity; out = relu(out)`, mmdet/models/backbones/resnet.py:286-297 of mmdet 2.x —
 * un-vendored dependency of the reference's Faster R-CNN config). */
int dm_dconv_gemm_residual(const float *x, const float *w_packed, const float *bias,
                           const float *residual, float *y, const int *geom_host,
                           const short *taps_host, void *workspace, size_t workspace_bytes,
                           dm_stream_t stream);
/* The `planes` copy of a packed weight (math mode 2, the 3 x 3 / stride-1 layers that dconv_patch_* takes): the
 * three bf16 planes of the split weights in MFMA-fragment order, written ONCE per weight update instead of split
 * by every workgroup of every launch; the kernel moves its tiles into LDS with direct-to-LDS loads
 * (global_load_lds_dwordx4).  Layout: [s][k / 16][n / 32 (N padded to 32)][plane h, m, l][lane][8 bf16],
 * lane = 32 * ((k % 16) / 8) + n % 32; K % 16 == 0.  dm_dconv_planes_bytes: bytes of the copy (incl. one tile of
 * slack); dm_dconv_pack_planes: same arguments as dm_dconv_pack; a dm_dconv_pack_batch row with pad == 1 writes
 * the copy right behind its fp32 block (dst + S*N*K floats).  dm_dconv_gemm_planes = dm_dconv_gemm_residual with
 * the copy handed along (w_planes may be NULL: then it IS dm_dconv_gemm_residual); results are bit-identical
 * with and without it (same split, same product order).  Replaces the same cuDNN call as dm_dconv_gemm. */
size_t dm_dconv_planes_bytes(int S, int N, int K);
int dm_dconv_pack_planes(const float *src, void *planes, const float *scale_n, const float *scale_k,
                         int S, int N, int K, int Nsrc, int Ksrc, long long sn, long long sk, long long st,
                         dm_stream_t stream);
int dm_dconv_gemm_planes(const float *x, const float *w_packed, const void *w_planes, const float *bias,
                         const float *residual, float *y, const int *geom_host, const short *taps_host,
                         void *workspace, size_t workspace_bytes, dm_stream_t stream);
size_t dm_dconv_wgrad_workspace_bytes(const int *geom_host);
int dm_dconv_wgrad(const float *U, const float *V, float *out, const float *scale_u,
                   const int *geom_host, const short *taps_host, int Cv_out, long long su,
                   long long sv, long long st, int accumulate, void *workspace,
                   size_t workspace_bytes, dm_stream_t stream);
/* Replaces scipy.optimize.linear_sum_assignment at
 * mmdet3d/core/bbox/assigners/modified_hungarian_assigner.py:132.  HOST function: cost
 * (n_rows, n_cols) row-major host floats -> min(n_rows, n_cols) pairs sorted by row.
 * Returns the number of pairs, -1 on NaN / infeasible input. */
int dm_lap_host(const float *cost_host, int n_rows, int n_cols, int *row_ind_host,
                int *col_ind_host);

/* Fused 3D augmentation of a batch of point clouds (the point part of the
 * TS_SSL_Dataset pipelines).  Replaces, per view (= one sample as the student or the teacher pipeline
 * sees it), RandomFlip3D (mmdet3d/datasets/pipelines/transforms_3d.py:102-127 ->
 * core/points/lidar_points.py:28-33), GlobalRotScaleTrans (:566-690 -> base_points.py:139-205,263-269),
 * PointsRangeFilter (:783-797 -> base_points.py:207-229) and PointShuffle (:695-712 ->
 * base_points.py:129-137).
 *   points   device (*, n_feat) f32 rows [x, y, z, ...]; view v reads rows src_off[v] .. +src_len[v]
 *            (views may share a source segment); src_off / src_len / dst_off are HOST arrays
 *   params   device (n_views, DM_AUG_PARAMS) f32: [0] horizontal flip (y -> -y) != 0, [1] vertical flip,
 *            [2..10] M row-major with p' = p @ M (the recorded `pcd_rotation`), [11] scale,
 *            [12..14] translation, [15..20] range x/y/z min then max (strict), rest unused
 *   perm     device int32 or NULL: slot j of view v reads source row perm[dst_off[v] + j] (local index)
 *   out      device rows; view v's kept rows, in slot order, start at row dst_off[v]
 *   out_counts device (n_views) int32: kept rows per view (the caller decides when to read them) */
#define DM_AUG_MAX_VIEWS 128
#define DM_AUG_PARAMS 24
size_t dm_points_augment_workspace_bytes(int n_views, const int *src_len_host);
int dm_points_augment(const float *points, int n_feat, int n_views, const int *src_off_host,
                      const int *src_len_host, const int *dst_off_host, const float *params,
                      const int *perm, float *out, int *out_counts, void *workspace,
                      size_t workspace_bytes, dm_stream_t stream);

/* Fused anchor-head losses.  Replaces get_cls_layer_loss / get_box_reg_layer_loss of
 * pcdet/models/dense_heads/anchor_head_template.py:101-214 with the loss functions of
 * pcdet/utils/loss_utils.py:9-137,181-206 (sigmoid focal gamma 2, smooth-L1 with the sin-difference
 * heading term, direction-bin cross-entropy), ~160 element-wise launches each way in the reference.
 *   cls_preds (B,A,n_cls)  box_preds (B,A,7)  dir_preds (B,A,n_bins) or NULL
 *   labels (B,A) int32: -1 ignore, 0 background, c+1 class    reg_targets (B,A,7)   anchors (A,7)
 *   num_pos (B) float: positives per sample (the normaliser; clamped to >= 1 inside)
 *   weights3_host = {cls_weight, loc_weight, dir_weight}; code_weights7_host; all sums / B
 *   forward : losses3 = {cls, loc, dir} (deterministic two-level reduction)
 *   backward: grad_* = d(sum_i grad_losses3[i] * losses3[i]) / d(*_preds), dense (zeros off the positives) */
size_t dm_anchor_head_loss_workspace_bytes(int batch, int n_anchors);
int dm_anchor_head_loss_forward(const float *cls_preds, const float *box_preds, const float *dir_preds,
                                const int32_t *labels, const float *reg_targets, const float *anchors,
                                const float *num_pos, int batch, int n_anchors, int n_cls, int n_bins,
                                float alpha, float beta, float dir_offset, const float *weights3_host,
                                const float *code_weights7_host, float *losses3, void *workspace,
                                size_t workspace_bytes, dm_stream_t stream);
int dm_anchor_head_loss_backward(const float *cls_preds, const float *box_preds, const float *dir_preds,
                                 const int32_t *labels, const float *reg_targets, const float *anchors,
                                 const float *num_pos, int batch, int n_anchors, int n_cls, int n_bins,
                                 float alpha, float beta, float dir_offset, const float *weights3_host,
                                 const float *code_weights7_host, const float *grad_losses3,
                                 float *grad_cls, float *grad_box, float *grad_dir, dm_stream_t stream);

/* KITTI AP bookkeeping.  HOST functions replacing the numba-compiled loops
 * compute_statistics_jit / fused_compute_statistics of
 * mmdet3d/core/evaluation/kitti_utils/eval.py:161-279,291-338 for one (class, difficulty,
 * min_overlap) cell over all images.  Per image n: overlaps (dt_nums[n] x gt_nums[n]) row-major
 * doubles, gt_datas rows [x1,y1,x2,y2,alpha], dt_datas rows [x1,y1,x2,y2,alpha,score], dontcares
 * rows [x1,y1,x2,y2], ignore flags as clean_data produces them (eval.py:28-80); all arrays are the
 * per-image pieces concatenated.  metric 0 = 2D boxes (DontCare regions absorb false positives).
 *   dm_kitti_tp_scores_host  scores of the matched detections at threshold 0 -> count (< 0: bad input)
 *   dm_kitti_pr_host         pr (n_thresholds, 4) += [tp, fp, fn, orientation similarity] */
long long dm_kitti_tp_scores_host(const double *overlaps, const int64_t *gt_nums,
                                  const int64_t *dt_nums, const int64_t *dc_nums, int n_images,
                                  const double *gt_datas, const double *dt_datas,
                                  const double *dontcares, const int64_t *ignored_gts,
                                  const int64_t *ignored_dets, int metric, double min_overlap,
                                  double *tp_scores);
int dm_kitti_pr_host(const double *overlaps, const int64_t *gt_nums, const int64_t *dt_nums,
                     const int64_t *dc_nums, int n_images, const double *gt_datas,
                     const double *dt_datas, const double *dontcares, const int64_t *ignored_gts,
                     const int64_t *ignored_dets, int metric, double min_overlap,
                     const double *thresholds, int n_thresholds, int compute_aos, double *pr);

/* ------------------------------------------------------------------------ */
/* D. Stacked PointNet++ operators, points-in-boxes                           */
/* ------------------------------------------------------------------------ */
/* Replaces pointnet2_stack_cuda.ball_query_wrapper
 *   pcdet/ops/pointnet2/pointnet2_stack/src/ball_query.cpp, ball_query_gpu.cu:16-89.
 * max_m_per_sample: upper bound of new_xyz_batch_cnt[] known to the host (0: use m).
 * empty_mask NULL: reference output (idx[m][0] = -1 for an empty ball);
 * empty_mask != NULL: fused post-processing of pointnet2_utils.py:36-37 (mask written,
 * empty rows zeroed). */
int dm_ball_query_stack(int batch, int m, float radius, int nsample, const float *new_xyz,
                        const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt,
                        int max_m_per_sample, int *idx, unsigned char *empty_mask,
                        dm_stream_t stream);
/* Two radii around the same query centres in ONE scan of the points (the two groupers of a
 * StackSAModuleMSG source, pointnet2_modules.py:60-78): same results as two dm_ball_query_stack calls with
 * empty masks. */
int dm_ball_query_stack2(int batch, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                         const float *new_xyz, const int *new_xyz_batch_cnt, const float *xyz,
                         const int *xyz_batch_cnt, int *idx_a, int *idx_b, unsigned char *empty_a,
                         unsigned char *empty_b, dm_stream_t stream);
/* Replaces group_points_wrapper / group_points_grad_wrapper (group_points_gpu.cu:15-131).
 * out (m, c, nsample).  empty_mask (optional): rows of empty balls are written as zeros
 * (pointnet2_utils.py:145,150). grad_features (n, c) is zeroed by the callee. */
int dm_group_points_stack(int batch, int m, int c, int nsample, const float *features,
                          const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                          const unsigned char *empty_mask, float *out, dm_stream_t stream);
int dm_group_points_grad_stack(int batch, int m, int c, int n, int nsample, const float *grad_out,
                               const int *idx, const int *idx_batch_cnt,
                               const int *features_batch_cnt, float *grad_features,
                               dm_stream_t stream);
/* Fused QueryAndGroup gather in ROW layout (pointnet2_utils.py:119-156 = group xyz, subtract the
 * ball centre, group features, cat; then StackSAModuleMSG permutes to (1,C,M,ns) for 1x1 convs,
 * pointnet2_modules.py:72-76).  out (M, nsample, [4+]C): with use_xyz the row starts with a
 * 16-byte slot [xyz[src] - new_xyz[m], 0], then features[src] (so rows and feature blocks stay
 * 16-byte aligned for C % 4 == 0); rows of empty balls (empty_mask[m] != 0) are zero.  features
 * may be NULL with c == 0 (xyz only). */
int dm_query_group_rows(int batch, int m, int c, int nsample, int use_xyz, const float *xyz,
                        const float *new_xyz, const float *features, const int *xyz_batch_cnt,
                        const int *new_xyz_batch_cnt, const int *idx,
                        const unsigned char *empty_mask, float *out, dm_stream_t stream);
/* Its gradient w.r.t. features: grad_features[src,:] += grad_out[m,s,col_offset:col_offset+c]
 * (grad_out rows have row_width floats).  grad_features (n, c) is zeroed by the callee. */
int dm_group_rows_grad(int batch, int m, int c, int n, int nsample, int row_width, int col_offset,
                       const float *grad_out, const int *idx, const int *idx_batch_cnt,
                       const int *features_batch_cnt, const unsigned char *empty_mask,
                       float *grad_features, dm_stream_t stream);
/* Replaces furthest_point_sampling_wrapper (sampling_gpu.cu:25-189).  xyz (b,n,3),
 * temp (b,n) pre-filled with 1e10 by the caller, idxs (b,m).
 * RESIDENCY REQUIREMENT of the large-cloud path (> 24 576 points per sample: G = min(64, 128 / batch) workgroups of
 * 1 024 threads per sample exchange their round maxima through `temp` and SPIN on each other, csrc/pointnet2_stack.hip:
 * fps_kernel_multi): all G workgroups of a sample must become resident.  They do whenever the device's other work
 * retires — a spinning workgroup keeps its slot, the missing ones take the slots other kernels free — but NOT next to a
 * second kernel that spins the same way (another large-cloud FPS launch on another stream, or a launch confined by a
 * CU mask to fewer than G * batch workgroup slots): issue large-cloud FPS launches on ONE stream (the package does:
 * _lib.aux_stream; the passes of an iteration are even one launch, pcdet/pfe.py:FpsBatch).  Clouds up to 24 576 points
 * take one workgroup per sample and have no such requirement.  Exercised under the three stream lanes on the Waymo
 * shape by tests/test_waymo_shape_gpu.py. */
int dm_furthest_point_sampling(int batch, int n, int m, const float *xyz, float *temp, int *idxs,
                               dm_stream_t stream);
/* Same, ragged: sample b owns points [offsets_host[b], offsets_host[b+1]) of the stacked
 * xyz (sum N, 3) / temp (sum N); all samples run concurrently (the reference loops samples
 * in Python, voxel_set_abstraction.py:135-151). */
int dm_furthest_point_sampling_stack(int batch, const int *offsets_host, int m, const float *xyz,
                                     float *temp, int *idxs, dm_stream_t stream);
/* Voxel centres + rows per sample of a sparse level in one launch: xyz[i] = (coords[i][x,y,z] + 0.5) * v + r
 * (pcdet/utils/common_utils.py:65-82 get_voxel_centers, the same fp32 operations in the same order), counts[b] =
 * rows of sample b (voxel_set_abstraction.py:209-214; coords (n, 4) int32 [b, z, y, x], sample-major). */
int dm_voxel_centers(const int32_t *coords, int n, int batch, float vx, float vy, float vz, float rx, float ry,
                     float rz, float *xyz, int32_t *counts, dm_stream_t stream);
/* Test / tuning aid: 0 (default) clouds beyond one workgroup's registers (> 24576 points) are sampled by
 * several co-operating workgroups per sample (same indices); 1 forces one workgroup per sample.  Anything else:
 * DM_ERR_INVALID_ARG. */
int dm_fps_set_variant(int variant);
/* Replaces roiaware_pool3d_cuda.points_in_boxes_gpu (roiaware_pool3d.cpp:98-129,
 * roiaware_pool3d_kernel.cu:313-360).  box_idx (batch, pts_num): first containing box or -1
 * (the callee writes every element; no pre-fill needed). */
int dm_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts,
                       int *box_idx_of_points, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Measurement hook (bench.py roofline leg; not part of the reference ABI)    */
/* ------------------------------------------------------------------------ */
/* When enabled, every main sparse-conv kernel launch is bracketed by a pair of
 * HIP events recorded on the launch stream.  dm_profile_get synchronises on the
 * record's stop event and returns the elapsed milliseconds.
 * kind 0 = spconv_gg (a = B-operand rows "ci", b = B-operand cols "co",
 * c = column splits per row tile), kind 1 = wgrad partial+reduce (a = cin, b = cout). */
int dm_profile_enable(int on); /* clears all records */
/* diagnostic build aid: buf (device, 6 u64 per workgroup) or NULL to switch off */
int dm_spconv_debug_stamps(void *buf);
/* tuning aid: -1 auto, 0 LDS-staged-weights kernel, 1 register-weights kernel; anything else DM_ERR_INVALID_ARG */
int dm_spconv_set_variant(int v);
/* Developer switch: pairs per weight-gradient workgroup (0 = heuristic; a multiple of 64). */
int dm_spconv_set_wgrad_chunk(int pairs);
int dm_profile_count(void);
int dm_profile_get(int i, int *kind, int *a, int *b, int *c, int *rows, int *kvol,
                   unsigned long long *table, float *ms);

/* RoI-head box decoding — RoIHeadTemplate.generate_predicted_boxes (pcdet/models/roi_heads/roi_head_template.py:233-263):
 * ResidualCoder.decode_torch (pcdet/utils/box_coder_utils.py:43-76) of the (n, 7) refinements against each RoI taken as a
 * local anchor at the origin, rotate_points_along_z (pcdet/utils/common_utils.py:34-56) of the decoded centre by the
 * RoI's heading, translation by the RoI's centre: boxes (n, 7).  The backward gives the gradient w.r.t. the refinements
 * (the RoIs are detached, roi_head_template.py:96-99) — the 2D <-> 3D consistency losses differentiate these boxes. */
int dm_roi_decode_forward(const float *box_encodings, const float *rois, int n, float *boxes, dm_stream_t stream);
/* PVRCNNHead.get_global_grid_points_of_roi (pcdet/models/roi_heads/pvrcnn_head.py:127-149): grid^3 points per RoI
 * (rois (n, roi_dim >= 7) [x, y, z, dx, dy, dz, heading, ...]) -> points (n * grid^3, 3), point (i, j, k) of RoI r at row
 * r * grid^3 + (i * grid + j) * grid + k. */
int dm_roi_grid_points(const float *rois, int n_rois, int roi_dim, int grid, float *points, dm_stream_t stream);
int dm_roi_decode_backward(const float *grad_boxes, const float *box_encodings, const float *rois, int n,
                           float *grad_encodings, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Chain-level issue (host-side launch interpreter)                          */
/* ------------------------------------------------------------------------ */
/* Stable sort of short rows of float keys: the score orders of the proposal layers and NMS wrappers
 * (pcdet/models/model_utils/model_nms_utils.py:6-25, mmcv batched_nms / mmdet RPN `torch.sort(scores, descending=True)`;
 * the reference: torch.sort -> a multi-launch library merge sort per call).  One launch, one workgroup per row, (key, index)
 * pairs sorted in LDS: idx_out (rows, n) int64 = the permutation torch.sort(stable=True) returns (ties keep their original
 * order; -0.0 == +0.0; a positive NaN is the largest key), keys_out (rows, n) optional.  n <= dm_sort_rows_max() (16 384). */
int dm_sort_rows_max(void);
int dm_sort_rows_f32(const float *keys, int rows, int n, long long key_stride, int descending, long long *idx_out,
                     float *keys_out /*or NULL*/, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Fully connected layers (nn.Linear / Conv1d(kernel 1) of pcdet/models/roi_heads/pvrcnn_head.py:25-52,
 * voxel_set_abstraction.py:107-111, point_head_template.py:34-47; the reference: cuBLAS through torch.nn.functional.linear
 * and autograd's mm backward).  One exact-fp32 MFMA kernel, three operand forms, all row-major with leading dimensions:
 *   form 0  C[M][N] = A[M][K] . B[N][K]^T [+ bias[N]] [ReLU]     forward            y  = x w^T + b
 *   form 1  C[M][N] = A[M][K] . B[K][N]                          input gradient     gx = gy w
 *   form 2  C[M][N] = A[K][M]^T . B[K][N]                        weight gradient    gw = gy^T x
 * A long contraction over few output tiles is split along K; the partial products (workspace,
 * dm_fc_gemm_workspace_bytes(M, N, K)) are summed in split order, so the result depends on the shapes only. */
size_t dm_fc_gemm_workspace_bytes(int M, int N, int K);
int dm_fc_gemm(int form, const float *A, const float *B, const float *bias /*or NULL*/, float *C, int M, int N, int K,
               int lda, int ldb, int ldc, int relu, void *workspace, size_t workspace_bytes, dm_stream_t stream);

/* ------------------------------------------------------------------------ */
/* The reference issues its step one Python call -> one ATen / extension call -> one kernel at a time
 * (e.g. pcdet/models/backbones_2d/base_bev_backbone.py:38-69: 15 conv + 15 BatchNorm + 15 ReLU module calls,
 * each a pybind trip; mmdet ResNet-50 + FPN + RPNHead of configs/detmatch/001/detmatch/split_0.py:39-99:
 * ~70 module calls; pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:58-104).  A CHAIN replaces the
 * per-kernel trips of such a shape-static sub-graph by ONE call: `ops_host` is a table of launches of THIS
 * library's own entry points, argument k of op i = slots_host[slot[k]] + imm[k] (slot < 0: the immediate alone;
 * float / double arguments travel as their bit patterns in the low 32 / all 64 bits).  What varies between
 * calls (arena base addresses, the stream, data-dependent counts) sits in the slot table; the op table is built
 * once per shape signature by the host layer.  dm_chain_run calls the entries in order on the calling thread and
 * stops at the first non-zero return code (*failed_op_host = index of that op, -1 if none); an op whose `nargs`
 * differs from its entry's signature length is DM_ERR_INVALID_ARG before anything of it is called.  Same kernels, same
 * order, same arguments as the op-by-op path — results are bit-identical to it.
 *   dm_chain_fn_index      index of an entry point by name (-1: not launchable through a chain)
 *   dm_chain_fn_signature  its argument classes, one letter each: p pointer, i int, l long long, z size_t,
 *                          f float, d double (the host layer checks them against its own binding table) */
#define DM_CHAIN_MAX_ARGS 32
typedef struct dm_chain_op {
  int32_t fn;
  int32_t nargs;
  int32_t slot[DM_CHAIN_MAX_ARGS];
  int64_t imm[DM_CHAIN_MAX_ARGS];
} dm_chain_op;
int dm_chain_fn_count(void);
int dm_chain_fn_index(const char *name);
const char *dm_chain_fn_name(int fn);
const char *dm_chain_fn_signature(int fn);
int dm_chain_run(const dm_chain_op *ops_host, int n_ops, const long long *slots_host, int n_slots,
                 int *failed_op_host);

/* Element-wise / layout glue of the chained sub-graphs: what sat between this library's kernels as ATen launches
 * in the op-by-op path, as entry points so that a sub-graph is one op table.  NHWC fp32, C % 4 == 0.
 *   dm_relu_mask_f32   out = y > 0 ? grad : 0          (aten::threshold_backward behind F.relu, e.g.
 *                      mmdet ResNet Bottleneck.forward / mmcv ConvModule activations)
 *   dm_add_mask_f32    out = (a + b) [masked by y > 0]; b, y optional: the gradient of a residual block's input
 *                      (shortcut + main branch, then the previous block's ReLU) in one pass
 *   dm_colsum_f32      out[c] (+)= sum_r x[r][c]: the bias gradient of a convolution (aten::sum over N,H,W);
 *                      two stages, fixed order.  C <= 1024
 *   dm_resize_nearest_nhwc / _backward   F.interpolate(mode='nearest') to a given size and its gradient
 *                      (mmdet FPN.forward top-down path, fpn.py:167-176 of mmdet 2.14); backward may accumulate
 *   dm_maxpool_nhwc    F.max_pool2d(k, stride, pad): ResNet stem (3, 2, 1) and FPN's extra level (1, 2, 0)
 *   dm_subsample_nhwc_backward   gradient of max_pool2d(kernel 1, stride s): zeros except the sampled pixels
 *   dm_copy2d_f32      rows x cols floats between pitched matrices (torch.cat / channel slices of NHWC maps,
 *                      base_bev_backbone.py:108 `torch.cat(ups, dim=1)`; 16-byte accesses when cols, pitches and
 *                      addresses allow, scalar otherwise)
 *   dm_fill_bytes      hipMemsetAsync
 *   dm_bn_fold_batch   evaluation-mode BatchNorm as (scale, shift) for every layer of a table in one launch
 *                      (rows of 56 bytes {gamma, beta, mean, var, scale_out, shift_out: pointers; float eps; int C}):
 *                      scale = gamma * rsqrt(var + eps), shift = beta - mean * scale
 *   dm_multi_add_f32   dst_i += src_i (assign != 0: dst_i = src_i) for a table of (dst, src, n: 24-byte rows) in
 *                      one launch (gradient accumulation into the flat arena: torch._foreach_add_; the
 *                      concatenated bias of the anchor head's three 1x1 convolutions: torch.cat) */
/* HeightCompression.forward (pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-25: `.dense()` -> (B, C, D, H, W)
 * -> view (B, C * D, H, W)) written straight in NHWC: features (n, C) of the active voxels with indices (n, 4) int32
 * [b, z, y, x] -> out (B, H, W, C * D) with channel c * D + z, zero elsewhere (zeroed by the callee); the backward
 * gathers grad_features (n, C) from grad_out.  Replaces zeros + a 4-index index_put (and its autograd chain). */
int dm_height_compress_forward(const float *features, const int *indices, long long n, int C, int batch, int D, int H,
                               int W, float *out, dm_stream_t stream);
int dm_height_compress_backward(const float *grad_out, const int *indices, long long n, int C, int D, int H, int W,
                                float *grad_features, dm_stream_t stream);
int dm_relu_mask_f32(const float *grad, const float *y, float *out, long long n, dm_stream_t stream);
int dm_add_mask_f32(const float *a, const float *b, const float *y, float *out, long long n, dm_stream_t stream);
size_t dm_colsum_workspace_bytes(long long rows, int C);
int dm_colsum_f32(const float *x, long long rows, int C, float *out, int accumulate, void *workspace,
                  size_t workspace_bytes, dm_stream_t stream);
int dm_resize_nearest_nhwc(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, float *y,
                           dm_stream_t stream);
int dm_resize_nearest_nhwc_backward(const float *grad_y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                    float *grad_x, int accumulate, dm_stream_t stream);
int dm_maxpool_nhwc(const float *x, int B, int Hi, int Wi, int C, int k, int stride, int pad, float *y,
                    dm_stream_t stream);
int dm_subsample_nhwc_backward(const float *grad_y, int B, int Hi, int Wi, int C, int stride, float *grad_x,
                               dm_stream_t stream);
int dm_copy2d_f32(const float *src, long long src_pitch, float *dst, long long dst_pitch, long long rows, int cols,
                  dm_stream_t stream);
int dm_fill_bytes(void *dst, int byte_value, size_t nbytes, dm_stream_t stream);
int dm_bn_fold_batch(const void *table_dev, int n_rows, int max_channels, dm_stream_t stream);
int dm_multi_add_f32(const void *table_dev, int n_rows, long long max_n, int assign, dm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DETMATCH_HIP_H_ */
