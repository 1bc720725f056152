// Anchor-head box decoding in one launch.
//
// AnchorHeadTemplate.generate_predicted_boxes (pcdet/models/dense_heads/anchor_head_template.py:225-272)
// = ResidualCoder.decode_torch (pcdet/utils/box_coder_utils.py:43-76) + the direction-classifier
// correction (limit_period, common_utils.py:20-23): 34 element-wise torch launches over the
// (B, 211 200, 7) anchor grid per pass.  Here one thread decodes one anchor with the SAME fp32
// operation sequence (separate multiply and add — the file is compiled with -ffp-contract=off; division
// by the Python scalar `period` is torch's multiplication by its float reciprocal), so the result is the
// tensor chain's bit for bit.  No gradient: PV-RCNN's proposal layer detaches these boxes
// (roi_head_template.py:96-99).
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

__global__ __launch_bounds__(256) void anchor_decode_kernel(
    const float *__restrict__ enc, const float *__restrict__ anchors, const float *__restrict__ dir_logits,
    long long total, int n_anchors, int n_bins, float dir_offset, float dir_limit_offset, float period,
    float inv_period, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float *a = anchors + (size_t)(i % n_anchors) * 7;
  const float *t = enc + (size_t)i * 7;
  const float xa = a[0], ya = a[1], za = a[2], dxa = a[3], dya = a[4], dza = a[5], ra = a[6];
  const float diagonal = sqrtf(dxa * dxa + dya * dya);
  float *o = out + (size_t)i * 7;
  o[0] = t[0] * diagonal + xa;
  o[1] = t[1] * diagonal + ya;
  o[2] = t[2] * dza + za;
  o[3] = expf(t[3]) * dxa;
  o[4] = expf(t[4]) * dya;
  o[5] = expf(t[5]) * dza;
  float r = t[6] + ra;
  if (dir_logits != nullptr) {
    const float *d = dir_logits + (size_t)i * n_bins;
    int label = 0;
    float best = d[0];
    for (int k = 1; k < n_bins; ++k)
      if (d[k] > best) best = d[k], label = k;       // first maximum, as torch.max
    const float val = r - dir_offset;
    const float dir_rot = val - floorf(val * inv_period + dir_limit_offset) * period;
    r = dir_rot + dir_offset + period * (float)label;
  }
  o[6] = r;
}

// RoI-head box decoding (roi_head_template.py:233-263): ResidualCoder.decode_torch against the RoI as a local anchor at
// the origin, rotation of the decoded centre by the RoI's heading (common_utils.rotate_points_along_z), translation by
// the RoI's centre — 20 element-wise / matmul / cat launches per pass in the tensor formulation, one here; and its
// gradient w.r.t. the encodings (the RoIs are detached), which the consistency losses need.
__global__ __launch_bounds__(256) void roi_decode_kernel(const float *__restrict__ enc, const float *__restrict__ rois,
                                                          int n, float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *r = rois + (size_t)i * 7, *t = enc + (size_t)i * 7;
  const float dx = r[3], dy = r[4], dz = r[5], ry = r[6];
  const float diagonal = sqrtf(dx * dx + dy * dy);
  const float xg = t[0] * diagonal, yg = t[1] * diagonal, zg = t[2] * dz;
  const float c = cosf(ry), s = sinf(ry);
  float *o = out + (size_t)i * 7;
  o[0] = (xg * c - yg * s) + r[0];
  o[1] = (xg * s + yg * c) + r[1];
  o[2] = zg + r[2];
  o[3] = expf(t[3]) * dx;
  o[4] = expf(t[4]) * dy;
  o[5] = expf(t[5]) * dz;
  o[6] = t[6] + ry;
}

__global__ __launch_bounds__(256) void roi_decode_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ enc,
                                                              const float *__restrict__ rois, int n,
                                                              float *__restrict__ grad_enc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *r = rois + (size_t)i * 7, *t = enc + (size_t)i * 7, *g = grad + (size_t)i * 7;
  const float dx = r[3], dy = r[4], dz = r[5], ry = r[6];
  const float diagonal = sqrtf(dx * dx + dy * dy);
  const float c = cosf(ry), s = sinf(ry);
  float *o = grad_enc + (size_t)i * 7;
  o[0] = (g[0] * c + g[1] * s) * diagonal;
  o[1] = (g[1] * c - g[0] * s) * diagonal;
  o[2] = g[2] * dz;
  o[3] = g[3] * (expf(t[3]) * dx);
  o[4] = g[4] * (expf(t[4]) * dy);
  o[5] = g[5] * (expf(t[5]) * dz);
  o[6] = g[6];
}

// RoI grid points (pvrcnn_head.py:127-149): grid^3 points per RoI, ((i, j, k) + 0.5) / grid * size - size / 2 in the box
// frame (k fastest), rotated by the heading, shifted to the centre.
__global__ __launch_bounds__(256) void roi_grid_points_kernel(const float *__restrict__ rois, int n_rois, int roi_dim,
                                                               int grid, float *__restrict__ out) {
  const int g3 = grid * grid * grid;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_rois * g3) return;
  const int r = (int)(i / g3), c = (int)(i % g3);
  const float *b = rois + (size_t)r * roi_dim;
  const int gi = c / (grid * grid), gj = (c / grid) % grid, gk = c % grid;
  const float fg = (float)grid;
  const float lx = ((float)gi + 0.5f) / fg * b[3] - b[3] / 2.0f;
  const float ly = ((float)gj + 0.5f) / fg * b[4] - b[4] / 2.0f;
  const float lz = ((float)gk + 0.5f) / fg * b[5] - b[5] / 2.0f;
  const float cs = cosf(b[6]), sn = sinf(b[6]);
  float *o = out + (size_t)i * 3;
  o[0] = (lx * cs - ly * sn) + b[0];
  o[1] = (lx * sn + ly * cs) + b[1];
  o[2] = lz + b[2];
}

}  // namespace

extern "C" int dm_roi_grid_points(const float *rois, int n_rois, int roi_dim, int grid, float *points,
                                  dm_stream_t stream) {
  if (n_rois < 0 || roi_dim < 7 || grid < 1 || grid > 32) return DM_ERR_INVALID_ARG;
  if (n_rois == 0) return DM_OK;
  if (!rois || !points) return DM_ERR_INVALID_ARG;
  const long long total = (long long)n_rois * grid * grid * grid;
  roi_grid_points_kernel<<<dm_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(rois, n_rois, roi_dim, grid, points);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_roi_decode_forward(const float *box_encodings, const float *rois, int n, float *boxes,
                                     dm_stream_t stream) {
  if (n < 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  if (!box_encodings || !rois || !boxes) return DM_ERR_INVALID_ARG;
  roi_decode_kernel<<<dm_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(box_encodings, rois, n, boxes);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_roi_decode_backward(const float *grad_boxes, const float *box_encodings, const float *rois, int n,
                                      float *grad_encodings, dm_stream_t stream) {
  if (n < 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  if (!grad_boxes || !box_encodings || !rois || !grad_encodings) return DM_ERR_INVALID_ARG;
  roi_decode_bwd_kernel<<<dm_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(grad_boxes, box_encodings, rois, n,
                                                                             grad_encodings);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_anchor_decode(const float *box_encodings, const float *anchors, const float *dir_logits,
                                long long n_total, int n_anchors, int n_dir_bins, float dir_offset,
                                float dir_limit_offset, float period, float *boxes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_total < 0 || n_anchors <= 0 || (dir_logits && n_dir_bins < 1) || period == 0.0f)
    return DM_ERR_INVALID_ARG;
  if (n_total == 0) return DM_OK;
  if (!box_encodings || !anchors || !boxes) return DM_ERR_INVALID_ARG;
  const float inv_period = 1.0f / period;   // torch: x / scalar == x * (1.0f / (float)scalar)
  anchor_decode_kernel<<<dm_ceil_div(n_total, 256), 256, 0, st>>>(
      box_encodings, anchors, dir_logits, n_total, n_anchors, n_dir_bins, dir_offset, dir_limit_offset,
      period, inv_period, boxes);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
