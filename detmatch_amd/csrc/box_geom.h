// Rotated-rectangle geometry shared by the IoU / NMS kernels and the RoI target kernels (gfx950).
// Follows pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:22-234; see iou3d_nms.hip for the rules
// (cos/sin evaluated once per box, "double libm rounded to float", fp32 op order as written).
#pragma once
#include "dm_common.h"

namespace {


struct Pt {
  float x, y;
};

constexpr float kEps = 1e-8f;

__device__ __forceinline__ float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__device__ __forceinline__ float fmin2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax2(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ bool check_rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
         fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

// iou3d_nms_kernel.cu:51-62 with cos(-h) = cs.x, sin(-h) = -cs.y precomputed
__device__ __forceinline__ bool check_in_box2d(const float *box, float2 cs, Pt p) {
  const float MARGIN = 1e-2f;
  float center_x = box[0], center_y = box[1];
  float angle_cos = cs.x, angle_sin = -cs.y;
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * (-angle_sin);
  float rot_y = (p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos;
  return fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN;
}

__device__ __forceinline__ bool intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt *ans) {
  if (!check_rect_cross(p0, p1, q0, q1)) return false;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ __forceinline__ Pt rotate_around_center(Pt c, float acos_, float asin_, Pt p) {
  Pt r;
  r.x = (p.x - c.x) * acos_ + (p.y - c.y) * (-asin_) + c.x;
  r.y = (p.x - c.x) * asin_ + (p.y - c.y) * acos_ + c.y;
  return r;
}

__device__ __forceinline__ float atan2f_d(float y, float x) {
  return (float)atan2((double)y, (double)x);
}

// iou3d_nms_kernel.cu:104-224; csa / csb = (cos, sin) of the two headings
__device__ float box_overlap(const float *box_a, float2 csa, const float *box_b, float2 csb) {
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2;
  float a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  Pt center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
  Pt ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  Pt cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    ca[k] = rotate_around_center(center_a, csa.x, csa.y, ca[k]);
    cb[k] = rotate_around_center(center_b, csb.x, csb.y, cb[k]);
  }
  ca[4] = ca[0];
  cb[4] = cb[0];
  Pt cross_points[16];
  Pt poly_center = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      Pt ans;
      if (intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &ans)) {
        poly_center.x = poly_center.x + ans.x;
        poly_center.y = poly_center.y + ans.y;
        cross_points[cnt++] = ans;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d(box_a, csa, cb[k])) {
      poly_center.x = poly_center.x + cb[k].x;
      poly_center.y = poly_center.y + cb[k].y;
      cross_points[cnt++] = cb[k];
    }
    if (check_in_box2d(box_b, csb, ca[k])) {
      poly_center.x = poly_center.x + ca[k].x;
      poly_center.y = poly_center.y + ca[k].y;
      cross_points[cnt++] = ca[k];
    }
  }
  if (cnt == 0) return 0.f;
  poly_center.x /= cnt;
  poly_center.y /= cnt;
  // bubble sort by angle about the centroid (:200-209); angles evaluated once per point
  float ang[16];
  for (int i = 0; i < cnt; ++i)
    ang[i] = atan2f_d(cross_points[i].y - poly_center.y, cross_points[i].x - poly_center.x);
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (ang[i] > ang[i + 1]) {
        Pt t = cross_points[i];
        cross_points[i] = cross_points[i + 1];
        cross_points[i + 1] = t;
        float ta = ang[i];
        ang[i] = ang[i + 1];
        ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    Pt u = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
    Pt v = {cross_points[k + 1].x - cross_points[0].x, cross_points[k + 1].y - cross_points[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float *a, float2 csa, const float *b, float2 csb) {
  float sa = a[3] * a[4];
  float sb = b[3] * b[4];
  float s_overlap = box_overlap(a, csa, b, csb);
  return s_overlap / fmaxf(sa + sb - s_overlap, kEps);
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4];
  float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, kEps);
}

}  // namespace
