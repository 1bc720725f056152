// Cost matrix of the 2D <-> 3D Hungarian matching in one launch (gfx950).
//
// Replaces, per sample, the tensor chain of FusionHungarianMatching.match
// (mmdet3d/models/ssl_modules/processors_fusion.py:50-222): bbox_3d_to_bbox_2d (bbox_utils.py:372-441:
// corners -> lidar2img -> clamp / divide -> min / max -> clip), the normalised cxcywh form, torch.logit of
// both score sets and ModHungarianAssigner's three costs (modified_hungarian_assigner.py:19-162:
// DoubleSidedFocalLossCost, BBoxL1Cost on normalised xyxy, IoUCost giou) — ≈150 element-wise launches
// on at most a few hundred boxes.  One workgroup: per-row / per-column terms once (LDS), then the
// (N3, N2) cost matrix.  The one-to-one assignment itself stays a host LAP (dm_lap_host), as in the
// reference (scipy).
#include "dm_common.h"

namespace {

constexpr int kMaxBoxes = 512, kMaxCls = 8;

struct MatchCfg {
  int n3, n2, C, project;
  float m[16];                 // lidar2img, row-major
  float img_w, img_h;
  float w_cls, w_reg, w_iou, alpha, focal_eps, logit_eps;
};

__device__ __forceinline__ float sigmoid_of_logit(float s, float eps) {
  // torch.logit(s, eps) then .sigmoid()
  const float c = fminf(fmaxf(s, eps), 1.f - eps);
  const float z = logf(c / (1.f - c));
  return 1.f / (1.f + expf(-z));
}

__global__ __launch_bounds__(256) void fusion_match_cost_kernel(const float *__restrict__ boxes3d,
                                                                const float *__restrict__ proj_in,
                                                                const float *__restrict__ scores3,
                                                                const float *__restrict__ boxes2d,
                                                                const float *__restrict__ scores2, MatchCfg c,
                                                                float *__restrict__ cost,
                                                                float *__restrict__ proj_out) {
  __shared__ float r_box[kMaxBoxes][4];      // row: normalised predicted box as xyxy
  __shared__ float c_abs[kMaxBoxes][4];      // column: 2D box in image coordinates
  __shared__ float r_fl[kMaxBoxes][kMaxCls]; // pos - neg of the focal cost per class
  __shared__ float c_fl[kMaxBoxes][kMaxCls];
  __shared__ unsigned char r_lab[kMaxBoxes], c_lab[kMaxBoxes];
  const int tid = threadIdx.x;
  const float fw[4] = {c.img_w, c.img_h, c.img_w, c.img_h};
  for (int i = tid; i < c.n3; i += 256) {
    float x1, y1, x2, y2;
    if (c.project) {
      const float *b = boxes3d + (size_t)i * 7;
      const float cs = cosf(b[6]), sn = sinf(b[6]);
      float xmin = 3.0e38f, ymin = 3.0e38f, xmax = -3.0e38f, ymax = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float ox = (k & 4) ? 0.5f : -0.5f, oy = ((k & 2) ? 0.5f : -0.5f), oz = (k & 1) ? 1.f : 0.f;
        const float px = b[3] * ox, py = b[4] * oy, pz = b[5] * oz;
        const float X = (px * cs + py * sn) + b[0], Y = (-px * sn + py * cs) + b[1], Z = pz + b[2];
        const float u = ((X * c.m[0] + Y * c.m[1]) + Z * c.m[2]) + c.m[3];
        const float v = ((X * c.m[4] + Y * c.m[5]) + Z * c.m[6]) + c.m[7];
        const float d = fmaxf(((X * c.m[8] + Y * c.m[9]) + Z * c.m[10]) + c.m[11], 1e-5f);
        const float x = u / d, y = v / d;
        xmin = fminf(xmin, x), xmax = fmaxf(xmax, x), ymin = fminf(ymin, y), ymax = fmaxf(ymax, y);
      }
      x1 = fminf(fmaxf(xmin, 0.f), c.img_w), y1 = fminf(fmaxf(ymin, 0.f), c.img_h);
      x2 = fminf(fmaxf(xmax, 0.f), c.img_w), y2 = fminf(fmaxf(ymax, 0.f), c.img_h);
      if (proj_out) {
        float *o = proj_out + (size_t)i * 4;
        o[0] = x1, o[1] = y1, o[2] = x2, o[3] = y2;
      }
    } else {
      const float *b = proj_in + (size_t)i * 4;
      x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    }
    // xyxy -> cxcywh / factor -> (assigner) cxcywh -> xyxy, and * factor for the GIoU
    const float cx = ((x1 + x2) / 2) / c.img_w, cy = ((y1 + y2) / 2) / c.img_h;
    const float w = (x2 - x1) / c.img_w, h = (y2 - y1) / c.img_h;
    r_box[i][0] = cx - 0.5f * w, r_box[i][1] = cy - 0.5f * h, r_box[i][2] = cx + 0.5f * w, r_box[i][3] = cy + 0.5f * h;
    int lab = 0;
    float best = -1.f;
    for (int k = 0; k < c.C; ++k) {
      const float p = sigmoid_of_logit(scores3[(size_t)i * c.C + k], c.logit_eps);
      if (p > best) best = p, lab = k;
      const float neg = -logf(1.f - p + c.focal_eps) * (1.f - c.alpha) * (p * p);
      const float pos = -logf(p + c.focal_eps) * c.alpha * ((1.f - p) * (1.f - p));
      r_fl[i][k] = pos - neg;
    }
    r_lab[i] = (unsigned char)lab;
  }
  for (int j = tid; j < c.n2; j += 256) {
    const float *b = boxes2d + (size_t)j * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) c_abs[j][k] = b[k];
    int lab = 0;
    float best = -1.f;
    for (int k = 0; k < c.C; ++k) {
      const float p = sigmoid_of_logit(scores2[(size_t)j * c.C + k], c.logit_eps);
      if (p > best) best = p, lab = k;
      const float neg = -logf(1.f - p + c.focal_eps) * (1.f - c.alpha) * (p * p);
      const float pos = -logf(p + c.focal_eps) * c.alpha * ((1.f - p) * (1.f - p));
      c_fl[j][k] = pos - neg;
    }
    c_lab[j] = (unsigned char)lab;
  }
  __syncthreads();
  const int total = c.n3 * c.n2;
  for (int e = tid; e < total; e += 256) {
    const int i = e / c.n2, j = e - i * c.n2;
    // classification: (FL(p3, argmax p2) + FL(p2, argmax p3)) / 2, each already times the weight
    const float cls = (r_fl[i][c_lab[j]] * c.w_cls + c_fl[j][r_lab[i]] * c.w_cls) / 2;
    // L1 on normalised xyxy (torch.cdist p = 1)
    float l1 = 0.f, a[4];
    const float *g = c_abs[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      l1 += fabsf(r_box[i][k] - g[k] / fw[k]);
      a[k] = r_box[i][k] * fw[k];
    }
    const float reg = l1 * c.w_reg;
    // GIoU on image coordinates (bbox_overlaps mode 'giou', eps 1e-6)
    const float area1 = (a[2] - a[0]) * (a[3] - a[1]), area2 = (g[2] - g[0]) * (g[3] - g[1]);
    const float ww = fmaxf(fminf(a[2], g[2]) - fmaxf(a[0], g[0]), 0.f), hh = fmaxf(fminf(a[3], g[3]) - fmaxf(a[1], g[1]), 0.f);
    const float overlap = ww * hh;
    const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
    const float iou = overlap / uni;
    const float ew = fmaxf(fmaxf(a[2], g[2]) - fminf(a[0], g[0]), 0.f), eh = fmaxf(fmaxf(a[3], g[3]) - fminf(a[1], g[1]), 0.f);
    const float enclose = fmaxf(ew * eh, 1e-6f);
    const float giou = iou - (enclose - uni) / enclose;
    cost[e] = (cls + reg) + (-giou * c.w_iou);
  }
}

}  // namespace

extern "C" int dm_fusion_match_cost(const float *boxes3d, const float *boxes_proj, const float *scores3, int n3,
                                    const float *boxes2d, const float *scores2, int n2, int n_cls,
                                    const float *lidar2img16_host, float img_w, float img_h, float w_cls,
                                    float w_reg, float w_iou, float focal_alpha, float focal_eps, float logit_eps,
                                    float *cost, float *proj_out, dm_stream_t stream) {
  if (n3 <= 0 || n2 <= 0) return DM_OK;
  if (n3 > kMaxBoxes || n2 > kMaxBoxes || n_cls < 1 || n_cls > kMaxCls) return DM_ERR_UNSUPPORTED;
  if ((!boxes3d && !boxes_proj) || !scores3 || !boxes2d || !scores2 || !cost) return DM_ERR_INVALID_ARG;
  if (boxes3d && !lidar2img16_host) return DM_ERR_INVALID_ARG;
  MatchCfg c;
  c.n3 = n3, c.n2 = n2, c.C = n_cls, c.project = boxes3d != nullptr;
  for (int k = 0; k < 16; ++k) c.m[k] = lidar2img16_host ? lidar2img16_host[k] : 0.f;
  c.img_w = img_w, c.img_h = img_h;
  c.w_cls = w_cls, c.w_reg = w_reg, c.w_iou = w_iou;
  c.alpha = focal_alpha, c.focal_eps = focal_eps, c.logit_eps = logit_eps;
  fusion_match_cost_kernel<<<1, 256, 0, (hipStream_t)stream>>>(boxes3d, boxes_proj, scores3, boxes2d, scores2, c, cost,
                                                               proj_out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
