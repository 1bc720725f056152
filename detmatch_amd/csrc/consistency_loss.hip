// The three losses of HungarianConsistency on one sample's matched pairs, values and gradients, one launch
// forward and one backward.
//
// Replaces the loop body of HungarianConsistency.forward (mmdet3d/models/ssl_modules/consumers_3d.py:11-117)
// for the configuration the DetMatch recipes use (configs/detmatch/001/split_0.py: FocalLoss on
// torch.logit(student probabilities, eps 1e-6) against the arg-max class of the matched 2D box,
// L1Loss on image-normalised boxes, GIoULoss; all reduction='mean'): ~64 element-wise launches forward and
// ~125 backward on a handful of pairs.  One workgroup: a thread evaluates one pair and its derivatives
// (mmdet sigmoid focal loss, mmdet bbox_overlaps giou aligned; clamp / min / max gradients as torch:
// inside-range only, ties of min / max split evenly), the three sums are folded in a fixed order.
// Forward stores the unit gradients (d loss_k / d input), backward scales them by the upstream scalars.
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

constexpr int kMaxCls = 8;

__device__ __forceinline__ float pick_max(float a, float g) { return a > g ? 1.f : (a == g ? 0.5f : 0.f); }
__device__ __forceinline__ float pick_min(float a, float g) { return a < g ? 1.f : (a == g ? 0.5f : 0.f); }

__global__ __launch_bounds__(256) void consistency_loss_kernel(
    const float *__restrict__ in_boxes, const float *__restrict__ in_scores, const float *__restrict__ tg_boxes,
    const float *__restrict__ tg_scores, int n, int C, float img_w, float img_h, float alpha, float gamma,
    float logit_eps, float iou_eps, float *__restrict__ losses /*[3] cls, l1, iou*/,
    float *__restrict__ g_scores /*(n, C) d cls / d score*/, float *__restrict__ g_l1 /*(n, 4)*/,
    float *__restrict__ g_iou /*(n, 4)*/) {
  __shared__ float red[3][256];
  const int tid = threadIdx.x;
  float s_cls = 0.f, s_l1 = 0.f, s_iou = 0.f;
  const float inv_cls = 1.0f / (float)(n * C), inv_l1 = 1.0f / (float)(n * 4), inv_iou = 1.0f / (float)n;
  const float fw4[4] = {img_w, img_h, img_w, img_h};
  for (int i = tid; i < n; i += 256) {
    // ---- class term ------------------------------------------------------------------------
    int lab = 0;
    float best = tg_scores[(size_t)i * C];
    for (int k = 1; k < C; ++k) {
      const float v = tg_scores[(size_t)i * C + k];
      if (v > best) best = v, lab = k;
    }
    for (int k = 0; k < C; ++k) {
      const float p = in_scores[(size_t)i * C + k];
      const float c = fminf(fmaxf(p, logit_eps), 1.f - logit_eps);
      const float x = logf(c / (1.f - c));
      const float t = k == lab ? 1.f : 0.f;
      const float sg = 1.f / (1.f + expf(-x));
      const float pt = (1.f - sg) * t + sg * (1.f - t);
      const float aw = alpha * t + (1.f - alpha) * (1.f - t);
      const float ptg = gamma == 2.f ? pt * pt : powf(pt, gamma);
      const float fw = aw * ptg;
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      s_cls += bce * fw;
      // d(bce * fw) / dx
      const float dpt = sg * (1.f - sg) * (1.f - 2.f * t);
      const float dptg = gamma == 2.f ? 2.f * pt : gamma * powf(pt, gamma - 1.f);
      const float dx = fw * (sg - t) + bce * aw * dptg * dpt;
      const float dxdp = (p < logit_eps || p > 1.f - logit_eps) ? 0.f : 1.f / (c * (1.f - c));
      g_scores[(size_t)i * C + k] = dx * dxdp * inv_cls;
    }
    // ---- L1 on image-normalised boxes ---------------------------------------------------------
    float a[4], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = in_boxes[(size_t)i * 4 + k], g[k] = tg_boxes[(size_t)i * 4 + k];
      const float d = a[k] / fw4[k] - g[k] / fw4[k];
      s_l1 += fabsf(d);
      g_l1[(size_t)i * 4 + k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / fw4[k] * inv_l1;
    }
    // ---- 1 - GIoU (aligned) -------------------------------------------------------------------
    const float area1 = (a[2] - a[0]) * (a[3] - a[1]), area2 = (g[2] - g[0]) * (g[3] - g[1]);
    const float ltx = fmaxf(a[0], g[0]), lty = fmaxf(a[1], g[1]), rbx = fminf(a[2], g[2]), rby = fminf(a[3], g[3]);
    const float w_raw = rbx - ltx, h_raw = rby - lty;
    const float w = fmaxf(w_raw, 0.f), h = fmaxf(h_raw, 0.f);
    const float overlap = w * h;
    const float u_raw = area1 + area2 - overlap;
    const float uni = fmaxf(u_raw, iou_eps);
    const float iou = overlap / uni;
    const float eltx = fminf(a[0], g[0]), elty = fminf(a[1], g[1]), erbx = fmaxf(a[2], g[2]), erby = fmaxf(a[3], g[3]);
    const float ew_raw = erbx - eltx, eh_raw = erby - elty;
    const float ew = fmaxf(ew_raw, 0.f), eh = fmaxf(eh_raw, 0.f);
    const float e_raw = ew * eh;
    const float enc = fmaxf(e_raw, iou_eps);
    const float giou = iou - (enc - uni) / enc;
    s_iou += 1.f - giou;
    // d(1 - giou)
    const float dg = -1.f;
    float d_overlap = dg * (1.f / uni);
    const float d_uni = dg * (-overlap / (uni * uni) + 1.f / enc);
    const float d_enc = dg * (-uni / (enc * enc));
    const float d_uraw = u_raw >= iou_eps ? d_uni : 0.f;
    const float d_area1 = d_uraw;
    d_overlap -= d_uraw;
    const float d_eraw = e_raw >= iou_eps ? d_enc : 0.f;
    const float d_ew = ew_raw >= 0.f ? d_eraw * eh : 0.f, d_eh = eh_raw >= 0.f ? d_eraw * ew : 0.f;
    const float d_w = w_raw >= 0.f ? d_overlap * h : 0.f, d_h = h_raw >= 0.f ? d_overlap * w : 0.f;
    float ga[4];
    ga[0] = -d_area1 * (a[3] - a[1]) - d_w * pick_max(a[0], g[0]) - d_ew * pick_min(a[0], g[0]);
    ga[1] = -d_area1 * (a[2] - a[0]) - d_h * pick_max(a[1], g[1]) - d_eh * pick_min(a[1], g[1]);
    ga[2] = d_area1 * (a[3] - a[1]) + d_w * pick_min(a[2], g[2]) + d_ew * pick_max(a[2], g[2]);
    ga[3] = d_area1 * (a[2] - a[0]) + d_h * pick_min(a[3], g[3]) + d_eh * pick_max(a[3], g[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) g_iou[(size_t)i * 4 + k] = ga[k] * inv_iou;
  }
  red[0][tid] = s_cls, red[1][tid] = s_l1, red[2][tid] = s_iou;
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int k = 0; k < 256; ++k) s += red[tid][k];      // fixed order
    losses[tid] = s * (tid == 0 ? inv_cls : (tid == 1 ? inv_l1 : inv_iou));
  }
}

__global__ __launch_bounds__(256) void consistency_loss_bwd_kernel(
    const float *__restrict__ up /*[3]*/, const float *__restrict__ g_scores, const float *__restrict__ g_l1,
    const float *__restrict__ g_iou, int n, int C, float *__restrict__ d_scores, float *__restrict__ d_boxes) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const float u0 = up[0], u1 = up[1], u2 = up[2];
  if (e < n * C) d_scores[e] = u0 * g_scores[e];
  if (e < n * 4) d_boxes[e] = u1 * g_l1[e] + u2 * g_iou[e];
}

}  // namespace

extern "C" int dm_consistency_loss_forward(const float *in_boxes, const float *in_scores,
                                           const float *target_boxes, const float *target_scores, int n,
                                           int n_cls, float img_w, float img_h, float alpha, float gamma,
                                           float logit_eps, float iou_eps, float *losses3,
                                           float *unit_grad_scores, float *unit_grad_l1,
                                           float *unit_grad_iou, dm_stream_t stream) {
  if (n <= 0 || n_cls < 1 || n_cls > kMaxCls || img_w <= 0.f || img_h <= 0.f) return DM_ERR_INVALID_ARG;
  if (!in_boxes || !in_scores || !target_boxes || !target_scores || !losses3 || !unit_grad_scores ||
      !unit_grad_l1 || !unit_grad_iou)
    return DM_ERR_INVALID_ARG;
  consistency_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(
      in_boxes, in_scores, target_boxes, target_scores, n, n_cls, img_w, img_h, alpha, gamma, logit_eps,
      iou_eps, losses3, unit_grad_scores, unit_grad_l1, unit_grad_iou);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_consistency_loss_backward(const float *grad_losses3, const float *unit_grad_scores,
                                            const float *unit_grad_l1, const float *unit_grad_iou, int n,
                                            int n_cls, float *grad_scores, float *grad_boxes,
                                            dm_stream_t stream) {
  if (n <= 0 || n_cls < 1) return DM_ERR_INVALID_ARG;
  if (!grad_losses3 || !unit_grad_scores || !unit_grad_l1 || !unit_grad_iou || !grad_scores || !grad_boxes)
    return DM_ERR_INVALID_ARG;
  const int m = n * (n_cls > 4 ? n_cls : 4);
  consistency_loss_bwd_kernel<<<dm_ceil_div(m, 256), 256, 0, (hipStream_t)stream>>>(
      grad_losses3, unit_grad_scores, unit_grad_l1, unit_grad_iou, n, n_cls, grad_scores, grad_boxes);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
