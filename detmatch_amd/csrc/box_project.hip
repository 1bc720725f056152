// 3D boxes of the augmented student frame -> 2D boxes of the original image, forward and backward, one
// launch each.
//
// Replaces Bboxes3DTo2D.forward (mmdet3d/models/ssl_modules/processors_3d.py:81-155) on one sample:
// apply_3d_transformation_bboxes(reverse=True) (bbox_utils.py:110-200 — translate / scale / rotate / flip
// records of the sample replayed backwards on centre, size and yaw) followed by bbox_3d_to_bbox_2d
// (bbox_utils.py:372-441: 8 corners -> lidar2img -> depth clamp -> divide -> min / max -> clip).  The
// tensor chain is ~75 element-wise launches per sample forward and twice that backward, on a few dozen
// boxes.  Here: the recorded transforms are composed on the host into
//     centre' = centre @ A + t,   size' = s * size,   yaw' = sigma * yaw + off
// (every recorded op is affine in these), one thread projects one box, and the backward kernel
// recomputes the forward (same code, same arg-min / arg-max corners) and applies the chain rule:
// clip and depth clamp pass gradient only inside their range (torch.clip / torch.clamp), min / max to
// the first extremal corner (torch.min / torch.max over a dimension).
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

struct ProjCfg {
  float A[9], t[3], s, sigma, off;
  float m[12];              // first three rows of lidar2img, row-major
  float img_w, img_h;
};

struct Corner {
  float px, py;             // size' * template offset (before the yaw rotation)
  float u, v, d_raw, d, x, y;
};

struct Proj {
  float c[3], dims[3], cs, sn;
  Corner k[8];
  int arg[4];               // corner giving xmin, ymin, xmax, ymax
  float raw[4];
  int n_inside;
  float depth_mean;
};

__device__ __forceinline__ void project_box(const float *b, const ProjCfg &c, Proj &p) {
  p.c[0] = ((b[0] * c.A[0] + b[1] * c.A[3]) + b[2] * c.A[6]) + c.t[0];
  p.c[1] = ((b[0] * c.A[1] + b[1] * c.A[4]) + b[2] * c.A[7]) + c.t[1];
  p.c[2] = ((b[0] * c.A[2] + b[1] * c.A[5]) + b[2] * c.A[8]) + c.t[2];
  p.dims[0] = b[3] * c.s, p.dims[1] = b[4] * c.s, p.dims[2] = b[5] * c.s;
  const float yaw = c.sigma * b[6] + c.off;
  p.cs = cosf(yaw), p.sn = sinf(yaw);
  p.raw[0] = p.raw[1] = 3.0e38f, p.raw[2] = p.raw[3] = -3.0e38f;
  p.arg[0] = p.arg[1] = p.arg[2] = p.arg[3] = 0;
  p.n_inside = 0;
  float dsum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // corner order of LiDARInstance3DBoxes.corners (lidar_box3d.py:48-87)
    const float ox = (k & 4) ? 0.5f : -0.5f, oy = (k & 2) ? 0.5f : -0.5f;
    const float oz = (((k & 3) == 1) || ((k & 3) == 2)) ? 1.f : 0.f;
    Corner &q = p.k[k];
    q.px = p.dims[0] * ox, q.py = p.dims[1] * oy;
    const float pz = p.dims[2] * oz;
    const float X = (q.px * p.cs + q.py * p.sn) + p.c[0], Y = (-q.px * p.sn + q.py * p.cs) + p.c[1], Z = pz + p.c[2];
    q.u = ((X * c.m[0] + Y * c.m[1]) + Z * c.m[2]) + c.m[3];
    q.v = ((X * c.m[4] + Y * c.m[5]) + Z * c.m[6]) + c.m[7];
    q.d_raw = ((X * c.m[8] + Y * c.m[9]) + Z * c.m[10]) + c.m[11];
    q.d = fmaxf(q.d_raw, 1e-5f);
    q.x = q.u / q.d, q.y = q.v / q.d;
    if (q.x >= 0.f && q.x < c.img_w && q.y >= 0.f && q.y < c.img_h && q.d > 0.f) ++p.n_inside;
    dsum += q.d;
    if (q.x < p.raw[0]) p.raw[0] = q.x, p.arg[0] = k;
    if (q.y < p.raw[1]) p.raw[1] = q.y, p.arg[1] = k;
    if (q.x > p.raw[2]) p.raw[2] = q.x, p.arg[2] = k;
    if (q.y > p.raw[3]) p.raw[3] = q.y, p.arg[3] = k;
  }
  p.depth_mean = dsum / 8.f;
}

__global__ __launch_bounds__(64) void box_project_fwd_kernel(const float *__restrict__ boxes, int n, ProjCfg c,
                                                             float *__restrict__ out,
                                                             unsigned char *__restrict__ valid) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Proj p;
  project_box(boxes + (size_t)i * 7, c, p);
  float *o = out + (size_t)i * 4;
  o[0] = fminf(fmaxf(p.raw[0], 0.f), c.img_w), o[1] = fminf(fmaxf(p.raw[1], 0.f), c.img_h);
  o[2] = fminf(fmaxf(p.raw[2], 0.f), c.img_w), o[3] = fminf(fmaxf(p.raw[3], 0.f), c.img_h);
  valid[i] = (p.n_inside >= 3 && p.depth_mean >= 0.5f) ? 1 : 0;
}

__global__ __launch_bounds__(64) void box_project_bwd_kernel(const float *__restrict__ boxes, int n, ProjCfg c,
                                                             const float *__restrict__ gout,
                                                             float *__restrict__ gboxes) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Proj p;
  project_box(boxes + (size_t)i * 7, c, p);
  float gc[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gyaw = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lim = (e & 1) ? c.img_h : c.img_w;
    const float g = gout[(size_t)i * 4 + e];
    if (!(p.raw[e] >= 0.f && p.raw[e] <= lim) || g == 0.f) continue;   // clipped: no gradient
    const int k = p.arg[e];
    const Corner &q = p.k[k];
    // x = u / d (or y = v / d), d = max(d_raw, 1e-5)
    const float num = (e & 1) ? q.v : q.u;
    const float gnum = g / q.d;
    const float gden = (q.d_raw >= 1e-5f) ? -g * num / (q.d * q.d) : 0.f;
    const int r = (e & 1) ? 4 : 0;
    const float gX = gnum * c.m[r + 0] + gden * c.m[8], gY = gnum * c.m[r + 1] + gden * c.m[9],
                gZ = gnum * c.m[r + 2] + gden * c.m[10];
    gc[0] += gX, gc[1] += gY, gc[2] += gZ;
    // X = px cs + py sn + cx, Y = -px sn + py cs + cy, Z = pz + cz
    const float gpx = gX * p.cs - gY * p.sn, gpy = gX * p.sn + gY * p.cs;
    gyaw += gX * (-q.px * p.sn + q.py * p.cs) + gY * (-q.px * p.cs - q.py * p.sn);
    const float ox = (k & 4) ? 0.5f : -0.5f, oy = (k & 2) ? 0.5f : -0.5f;
    const float oz = (((k & 3) == 1) || ((k & 3) == 2)) ? 1.f : 0.f;
    gd[0] += gpx * ox, gd[1] += gpy * oy, gd[2] += gZ * oz;
  }
  float *o = gboxes + (size_t)i * 7;
  o[0] = (gc[0] * c.A[0] + gc[1] * c.A[1]) + gc[2] * c.A[2];
  o[1] = (gc[0] * c.A[3] + gc[1] * c.A[4]) + gc[2] * c.A[5];
  o[2] = (gc[0] * c.A[6] + gc[1] * c.A[7]) + gc[2] * c.A[8];
  o[3] = gd[0] * c.s, o[4] = gd[1] * c.s, o[5] = gd[2] * c.s;
  o[6] = gyaw * c.sigma;
}

// bbox_2d_transform (mmdet3d/models/fusion_layers/coord_transform.py:121-175): scale -> crop offset ->
// h-flip of xyxy boxes (ori2new), or the reverse order with the inverse operations; the same fp32
// operations in the same order as the tensor chain (13 launches), and its transpose for the gradient.
struct Box2DXf {
  float sx, sy, cx, cy, img_w;
  int flip, ori2new;
};

__global__ __launch_bounds__(256) void bbox2d_transform_kernel(const float *__restrict__ in, int n, Box2DXf c,
                                                               int backward, float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *b = in + (size_t)i * 4;
  float *o = out + (size_t)i * 4;
  if (!backward) {
    float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    if (c.ori2new) {
      x1 = x1 * c.sx + c.cx, x2 = x2 * c.sx + c.cx, y1 = y1 * c.sy + c.cy, y2 = y2 * c.sy + c.cy;
      if (c.flip) {
        const float t = x1;
        x1 = c.img_w - x2, x2 = c.img_w - t;
      }
    } else {
      if (c.flip) {
        const float t = x1;
        x1 = c.img_w - x2, x2 = c.img_w - t;
      }
      x1 = (x1 - c.cx) / c.sx, x2 = (x2 - c.cx) / c.sx, y1 = (y1 - c.cy) / c.sy, y2 = (y2 - c.cy) / c.sy;
    }
    o[0] = x1, o[1] = y1, o[2] = x2, o[3] = y2;
  } else {            // in = gradient of the transformed boxes
    float g1 = b[0], h1 = b[1], g2 = b[2], h2 = b[3];
    if (c.ori2new) {
      if (c.flip) {
        const float t = g1;
        g1 = -g2, g2 = -t;
      }
      g1 *= c.sx, g2 *= c.sx, h1 *= c.sy, h2 *= c.sy;
    } else {
      g1 /= c.sx, g2 /= c.sx, h1 /= c.sy, h2 /= c.sy;
      if (c.flip) {
        const float t = g1;
        g1 = -g2, g2 = -t;
      }
    }
    o[0] = g1, o[1] = h1, o[2] = g2, o[3] = h2;
  }
}

bool fill_cfg(const float *xf17_host, const float *lidar2img16_host, float img_w, float img_h, ProjCfg &c) {
  if (!xf17_host || !lidar2img16_host) return false;
  for (int k = 0; k < 9; ++k) c.A[k] = xf17_host[k];
  for (int k = 0; k < 3; ++k) c.t[k] = xf17_host[9 + k];
  c.s = xf17_host[12], c.sigma = xf17_host[13], c.off = xf17_host[14];
  for (int k = 0; k < 12; ++k) c.m[k] = lidar2img16_host[k];
  c.img_w = img_w, c.img_h = img_h;
  return true;
}

}  // namespace

extern "C" int dm_box3d_project_forward(const float *boxes3d, int n, const float *xf17_host,
                                        const float *lidar2img16_host, float img_w, float img_h,
                                        float *boxes2d, unsigned char *valid, dm_stream_t stream) {
  if (n < 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  ProjCfg c;
  if (!boxes3d || !boxes2d || !valid || !fill_cfg(xf17_host, lidar2img16_host, img_w, img_h, c))
    return DM_ERR_INVALID_ARG;
  box_project_fwd_kernel<<<dm_ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(boxes3d, n, c, boxes2d, valid);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_box3d_project_backward(const float *boxes3d, int n, const float *xf17_host,
                                         const float *lidar2img16_host, float img_w, float img_h,
                                         const float *grad_boxes2d, float *grad_boxes3d,
                                         dm_stream_t stream) {
  if (n < 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  ProjCfg c;
  if (!boxes3d || !grad_boxes2d || !grad_boxes3d || !fill_cfg(xf17_host, lidar2img16_host, img_w, img_h, c))
    return DM_ERR_INVALID_ARG;
  box_project_bwd_kernel<<<dm_ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(boxes3d, n, c, grad_boxes2d,
                                                                              grad_boxes3d);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bbox2d_transform(const float *boxes_or_grad, int n, float scale_x, float scale_y,
                                   float crop_x, float crop_y, float img_w, int flip, int ori2new,
                                   int backward, float *out, dm_stream_t stream) {
  if (n < 0 || scale_x == 0.f || scale_y == 0.f) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  if (!boxes_or_grad || !out) return DM_ERR_INVALID_ARG;
  Box2DXf c;
  c.sx = scale_x, c.sy = scale_y, c.cx = crop_x, c.cy = crop_y, c.img_w = img_w, c.flip = flip, c.ori2new = ori2new;
  bbox2d_transform_kernel<<<dm_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(boxes_or_grad, n, c, backward, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
