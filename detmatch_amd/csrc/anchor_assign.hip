// Axis-aligned anchor target assignment of the anchor head as two device kernels.
//
// Replaces AxisAlignedTargetAssigner.assign_targets / assign_targets_single
// (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-209: a Python loop
// over samples x anchor classes with .cpu().numpy() argmax read-backs) for the settings the DetMatch
// configs use (POS_FRACTION < 0, MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES False), for ALL classes and
// samples of a batch at once, writing the reference's final layout directly:
//   box_cls_labels (B, L*C*R) int32, box_reg_targets (B, L*C*R, 7), reg_weights (B, L*C*R)
// with L locations, C anchor classes, R anchors per location and class (index (l*C + c)*R + r).
//
// The overlap is the nearest-BEV IoU of box_utils.py:286-298 on axis-aligned BEV rectangles; the
// rectangles themselves (limit_period / dims swap) are computed by the host layer with the same
// tensor ops as the reference and handed in, so the kernel's arithmetic is min / max / sub / mul /
// div in the reference's order (-ffp-contract=off): recomputing an overlap gives the same bits, which
// the "anchors that reach a ground truth's best overlap" rule (:151-156) relies on.
//   kernel 1: best overlap per ground truth (float bits as int: overlaps are >= 0) by atomicMax,
//             touched only by anchors that overlap it at all;
//   kernel 2: per anchor best ground truth (first maximum, like argmax), the label rules :156-188,
//             ResidualCoder.encode (box_coder_utils.py:19-40) for foreground anchors.
#include "dm_common.h"

namespace {

struct AssignArgs {
  int B, M, C, A, R;        // A anchors per class, R anchors per location and class
  int num_class;
};

__device__ __forceinline__ float bev_iou(const float4 a, const float area_a, const float4 g) {
  const float xl = fmaxf(fminf(a.z, g.z) - fmaxf(a.x, g.x), 0.0f);
  const float yl = fmaxf(fminf(a.w, g.w) - fmaxf(a.y, g.y), 0.0f);
  const float inter = xl * yl;
  const float area_b = (g.z - g.x) * (g.w - g.y);
  return inter / fmaxf(area_a + area_b - inter, 1e-6f);
}

// shared: per ground truth of sample b its BEV rectangle and whether anchor class c takes it
__device__ __forceinline__ void load_gt(const float *gt, const float *gbev, const int *cids,
                                        const AssignArgs s, int c, int b, float4 *g_rect, int *g_sel,
                                        int *any_sel) {
  __shared__ int last_nz;
  if (threadIdx.x == 0) last_nz = 0, *any_sel = 0;
  __syncthreads();
  // rows up to the last row whose 7 box values do not sum to zero (:53-58), at least row 0
  for (int m = threadIdx.x; m < s.M; m += blockDim.x) {
    const float *r = gt + ((size_t)b * s.M + m) * 8;
    float sum = r[0];
    for (int k = 1; k < 7; ++k) sum += r[k];
    if (sum != 0.0f) atomicMax(&last_nz, m);
  }
  __syncthreads();
  const int cid = cids[c];
  for (int m = threadIdx.x; m < s.M; m += blockDim.x) {
    const float *r = gt + ((size_t)b * s.M + m) * 8;
    int cls = (int)r[7];
    if (cls == 0) cls = s.num_class;       // class id 0 inside the kept range indexes class_names[-1]
    const int sel = (m <= last_nz) && (cls == cid);
    g_sel[m] = sel;
    g_rect[m] = *(const float4 *)(gbev + ((size_t)b * s.M + m) * 4);
    if (sel) *any_sel = 1;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void anchor_assign_gtmax_kernel(
    const float *__restrict__ abev, const float *__restrict__ gt, const float *__restrict__ gbev,
    const int *__restrict__ cids, const AssignArgs s, int *__restrict__ g2a) {
  extern __shared__ float4 smem[];
  float4 *g_rect = smem;
  int *g_sel = (int *)(smem + s.M);
  __shared__ int any_sel;
  const int c = blockIdx.y / s.B, b = blockIdx.y % s.B;
  load_gt(gt, gbev, cids, s, c, b, g_rect, g_sel, &any_sel);
  if (!any_sel) return;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= s.A) return;
  const float4 r = *(const float4 *)(abev + ((size_t)c * s.A + a) * 4);
  const float area_a = (r.z - r.x) * (r.w - r.y);
  int *dst = g2a + ((size_t)c * s.B + b) * s.M;
  for (int m = 0; m < s.M; ++m) {
    if (!g_sel[m]) continue;
    const float ov = bev_iou(r, area_a, g_rect[m]);
    if (ov > 0.0f) atomicMax(dst + m, __float_as_int(ov));
  }
}

__global__ __launch_bounds__(256) void anchor_assign_label_kernel(
    const float *__restrict__ anchors, const float *__restrict__ abev, const float *__restrict__ gt,
    const float *__restrict__ gbev, const int *__restrict__ cids, const float *__restrict__ matched,
    const float *__restrict__ unmatched, const AssignArgs s, const int *__restrict__ g2a,
    int *__restrict__ labels, float *__restrict__ targets, float *__restrict__ weights) {
  extern __shared__ float4 smem[];
  float4 *g_rect = smem;
  int *g_sel = (int *)(smem + s.M);
  float *g_best = (float *)(g_sel + s.M);
  __shared__ int any_sel;
  const int c = blockIdx.y / s.B, b = blockIdx.y % s.B;
  load_gt(gt, gbev, cids, s, c, b, g_rect, g_sel, &any_sel);
  for (int m = threadIdx.x; m < s.M; m += blockDim.x) {
    const float best = __int_as_float(g2a[((size_t)c * s.B + b) * s.M + m]);
    g_best[m] = (best == 0.0f || !g_sel[m]) ? -2.0f : best;          // :151-152
  }
  __syncthreads();
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= s.A) return;
  const size_t o = (size_t)b * ((size_t)s.A * s.C) + ((size_t)(a / s.R) * s.C + c) * s.R + (a % s.R);
  float *t = targets + o * 7;
  if (!any_sel) {                                   // no ground truth of this class: all background
    labels[o] = 0;
    weights[o] = 0.0f;
    for (int k = 0; k < 7; ++k) t[k] = 0.0f;
    return;
  }
  const float4 r = *(const float4 *)(abev + ((size_t)c * s.A + a) * 4);
  const float area_a = (r.z - r.x) * (r.w - r.y);
  float best = -1.0f;
  int arg = 0;
  bool force = false;
  for (int m = 0; m < s.M; ++m) {
    const float ov = g_sel[m] ? bev_iou(r, area_a, g_rect[m]) : -1.0f;
    if (ov > best) best = ov, arg = m;              // first maximum
    force |= (ov == g_best[m]);                     // :154
  }
  const float *g = gt + ((size_t)b * s.M + arg) * 8;
  const int cls_of_anchor = (int)g[7];
  int label = -1;
  if (force) label = cls_of_anchor;                 // :156
  if (best >= matched[c]) label = cls_of_anchor;    // :161
  if (best < unmatched[c]) label = 0;               // :187
  if (force) label = cls_of_anchor;                 // :188
  labels[o] = label;
  const bool fg = label > 0;
  weights[o] = fg ? 1.0f : 0.0f;
  if (!fg) {
    for (int k = 0; k < 7; ++k) t[k] = 0.0f;
    return;
  }
  const float *an = anchors + ((size_t)c * s.A + a) * 7;
  const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
  const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
  const float diag = sqrtf(dxa * dxa + dya * dya);
  t[0] = (g[0] - an[0]) / diag;
  t[1] = (g[1] - an[1]) / diag;
  t[2] = (g[2] - an[2]) / dza;
  t[3] = logf(dxg / dxa);
  t[4] = logf(dyg / dya);
  t[5] = logf(dzg / dza);
  t[6] = g[6] - an[6];
}

}  // namespace

extern "C" size_t dm_anchor_assign_workspace_bytes(int B, int M, int C) {
  return dm_align((size_t)B * M * C * sizeof(int));
}

extern "C" int dm_anchor_assign(const float *anchors, const float *anchors_bev, const float *gt_boxes,
                                const float *gt_bev, const int *class_ids, const float *matched,
                                const float *unmatched, int B, int M, int C, int A, int R,
                                int num_class, int *labels, float *reg_targets, float *reg_weights,
                                void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  if (B <= 0 || C <= 0 || A <= 0) return DM_OK;
  if (!anchors || !anchors_bev || !class_ids || !matched || !unmatched || !labels || !reg_targets ||
      !reg_weights || R <= 0 || A % R != 0 || (M > 0 && (!gt_boxes || !gt_bev)))
    return DM_ERR_INVALID_ARG;
  if (M > 2048) return DM_ERR_UNSUPPORTED;
  if (workspace_bytes < dm_anchor_assign_workspace_bytes(B, M, C) || !workspace) return DM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  AssignArgs s{B, M, C, A, R, num_class};
  int *g2a = (int *)workspace;
  DM_HIP(hipMemsetAsync(g2a, 0, (size_t)B * M * C * sizeof(int), st));
  const dim3 grid(dm_ceil_div(A, 256), C * B);
  const size_t smem = (size_t)(M > 0 ? M : 1) * (sizeof(float4) + sizeof(int) + sizeof(float));
  if (M > 0) {
    anchor_assign_gtmax_kernel<<<grid, 256, smem, st>>>(anchors_bev, gt_boxes, gt_bev, class_ids, s, g2a);
    DM_CHECK_LAUNCH();
  }
  anchor_assign_label_kernel<<<grid, 256, smem, st>>>(anchors, anchors_bev, gt_boxes, gt_bev, class_ids,
                                                      matched, unmatched, s, g2a, labels, reg_targets,
                                                      reg_weights);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
