/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see dm_oracle.h).  Plain C, scalar, one core.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * Floating-point contraction is OFF; where the reference's device code is an
 * FMA chain by construction (nvcc default contraction of a*a+b*b+c*c) the chain
 * is written out with fmaf() so CPU and GPU agree bit for bit.
 */
#include "dm_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* A. hard voxelization                                                      */
/* ------------------------------------------------------------------------ */

/* voxelization_cpu.cpp:21-30: c = floor((p - min) / size) in fp32, per axis,
 * out of range if c < 0 || c >= grid; stored reversed (z,y,x). */
static int point_to_coor(const float *p, const float *voxel_size,
                         const float *coors_range, const int *grid, int *coor) {
  for (int j = 0; j < 3; ++j) {
    float q = (p[j] - coors_range[j]) / voxel_size[j];
    int c = (int)floorf(q);
    if (c < 0 || c >= grid[j]) return 0;
    coor[2 - j] = c;
  }
  return 1;
}

int orc_hard_voxelize(const float *points, int n, int c, const float *voxel_size,
                      const float *coors_range, int max_points, int max_voxels,
                      float *voxels, int32_t *coors, int32_t *num_points) {
  int grid[3];
  for (int i = 0; i < 3; ++i) /* voxelization_cpu.cpp:120-123 */
    grid[i] = (int)roundf((coors_range[3 + i] - coors_range[i]) / voxel_size[i]);
  size_t vol = (size_t)grid[0] * grid[1] * grid[2];
  /* voxelization_cpu.cpp:127-128: dense coor_to_voxelidx[z][y][x] = -1 */
  int32_t *coor_to_voxelidx = (int32_t *)malloc(vol * sizeof(int32_t));
  memset(coor_to_voxelidx, 0xff, vol * sizeof(int32_t));
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) { /* voxelization_cpu.cpp:68-99 */
    int coor[3];
    if (!point_to_coor(points + (size_t)i * c, voxel_size, coors_range, grid, coor))
      continue;
    size_t cell = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int voxelidx = coor_to_voxelidx[cell];
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) continue;
      voxel_num += 1;
      coor_to_voxelidx[cell] = voxelidx;
      for (int k = 0; k < 3; ++k) coors[voxelidx * 3 + k] = coor[k];
    }
    int num = num_points[voxelidx];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)voxelidx * max_points + num) * c,
             points + (size_t)i * c, sizeof(float) * c);
      num_points[voxelidx] += 1;
    }
  }
  free(coor_to_voxelidx);
  return voxel_num;
}

/* ------------------------------------------------------------------------ */
/* B. spconv rulebook                                                        */
/* ------------------------------------------------------------------------ */

/* geometry.h:25-86 getValidOutPos<Index,3>: enumerate the output positions an
 * input voxel contributes to, from the UPPER corner downwards, with the kernel
 * offset index accumulated from the last (x) axis.  out holds up to kvol rows
 * of (z,y,x,offset). */
static int valid_out_pos(const int *input_pos, const int *ksize, const int *stride,
                         const int *padding, const int *dilation,
                         const int *out_shape, int *out) {
  int lowers[3], uppers[3], counter[3], counter_size[3];
  int point_counter = 0, num_points = 1;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = (input_pos[i] - (ksize[i] - 1) * dilation[i] - 1 + stride[i] +
                 padding[i]) / stride[i];
    uppers[i] = (input_pos[i] + padding[i]) / stride[i];
  }
  for (int i = 0; i < 3; ++i) {
    counter_size[i] = (uppers[i] - lowers[i]) / dilation[i] + 1;
    num_points *= counter_size[i];
    counter[i] = 0;
  }
  for (int i = 0; i < num_points; ++i) {
    int valid = 1, m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dilation[j];
      out[point_counter * 4 + j] = val;
      if (val < 0 || val > out_shape[j] - 1) valid = 0;
      offset += m * (input_pos[j] - val * stride[j] + padding[j]) / dilation[j];
      m *= ksize[j];
    }
    out[point_counter * 4 + 3] = offset;
    if (valid) ++point_counter;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c) {
      if (counter[c] == counter_size[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return point_counter;
}

typedef struct {
  int32_t key;
  int32_t rank;
} key_rank_t;

static int cmp_key(const void *a, const void *b) {
  int32_t x = ((const key_rank_t *)a)->key, y = ((const key_rank_t *)b)->key;
  return (x > y) - (x < y);
}

int orc_get_indice_pairs(const int32_t *indices, int n, int batch_size,
                         const int *out_shape, const int *spatial_shape,
                         const int *ksize, const int *stride_in,
                         const int *padding_in, const int *dilation, int subm,
                         int sort_out, int32_t *out_ids, int32_t *indice_pairs,
                         int32_t *indice_num) {
  (void)spatial_shape;
  int kvol = ksize[0] * ksize[1] * ksize[2];
  int stride[3], padding[3];
  for (int i = 0; i < 3; ++i) { /* spconv_ops.h:73-85: SubM forces stride 1, pad k/2 */
    stride[i] = subm ? 1 : stride_in[i];
    padding[i] = subm ? ksize[i] / 2 : padding_in[i];
  }
  int64_t vol = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
  if (vol * batch_size > INT_MAX) return -1;
  int32_t *grid = (int32_t *)malloc((size_t)vol * batch_size * sizeof(int32_t));
  memset(grid, 0xff, (size_t)vol * batch_size * sizeof(int32_t)); /* spconv_ops.h:59-61 */
  memset(indice_pairs, 0xff, (size_t)kvol * 2 * n * sizeof(int32_t)); /* :55-57 */
  memset(indice_num, 0, kvol * sizeof(int32_t));
  int *valid = (int *)malloc((size_t)kvol * 4 * sizeof(int));
  int num_act = 0;
  if (subm) { /* geometry.h:248-296 getIndicePairsSubM */
    for (int j = 0; j < n; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      int64_t index = ((int64_t)p[1] * out_shape[1] + p[2]) * out_shape[2] + p[3] +
                      vol * p[0];
      grid[index] = j;
    }
    for (int j = 0; j < n; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      int pos[3] = {p[1], p[2], p[3]};
      int nv = valid_out_pos(pos, ksize, stride, padding, dilation, out_shape, valid);
      for (int i = 0; i < nv; ++i) {
        const int *pp = valid + i * 4;
        int offset = pp[3];
        int64_t index = ((int64_t)pp[0] * out_shape[1] + pp[1]) * out_shape[2] +
                        pp[2] + vol * p[0];
        if (grid[index] > -1) {
          indice_pairs[((size_t)offset * 2 + 0) * n + indice_num[offset]] = j;
          indice_pairs[((size_t)offset * 2 + 1) * n + indice_num[offset]++] =
              grid[index];
        }
      }
    }
    memcpy(out_ids, indices, (size_t)n * 4 * sizeof(int32_t));
    num_act = n;
  } else { /* geometry.h:145-192 getIndicePairsConv (first-touch output order) */
    for (int j = 0; j < n; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      int pos[3] = {p[1], p[2], p[3]};
      int nv = valid_out_pos(pos, ksize, stride, padding, dilation, out_shape, valid);
      for (int i = 0; i < nv; ++i) {
        const int *pp = valid + i * 4;
        int offset = pp[3];
        int64_t index = ((int64_t)pp[0] * out_shape[1] + pp[1]) * out_shape[2] +
                        pp[2] + vol * p[0];
        if (grid[index] == -1) {
          out_ids[(size_t)num_act * 4 + 0] = p[0];
          for (int k = 0; k < 3; ++k) out_ids[(size_t)num_act * 4 + 1 + k] = pp[k];
          grid[index] = num_act++;
        }
        indice_pairs[((size_t)offset * 2 + 0) * n + indice_num[offset]] = j;
        indice_pairs[((size_t)offset * 2 + 1) * n + indice_num[offset]++] =
            grid[index];
      }
    }
    if (sort_out && num_act > 0) {
      /* GPU order: ascending flat cell id (indice.cu.h:59-60 + torch::_unique,
       * spconv_ops.h:130; assignGridAndIndiceOutKernel indice.cu.h:113-128). */
      key_rank_t *kr = (key_rank_t *)malloc((size_t)num_act * sizeof(key_rank_t));
      for (int r = 0; r < num_act; ++r) {
        const int32_t *o = out_ids + (size_t)r * 4;
        kr[r].key = (int32_t)(((int64_t)o[1] * out_shape[1] + o[2]) * out_shape[2] +
                              o[3] + vol * o[0]);
        kr[r].rank = r;
      }
      qsort(kr, num_act, sizeof(key_rank_t), cmp_key);
      int32_t *remap = (int32_t *)malloc((size_t)num_act * sizeof(int32_t));
      int32_t *tmp = (int32_t *)malloc((size_t)num_act * 4 * sizeof(int32_t));
      for (int r = 0; r < num_act; ++r) {
        remap[kr[r].rank] = r;
        memcpy(tmp + (size_t)r * 4, out_ids + (size_t)kr[r].rank * 4,
               4 * sizeof(int32_t));
      }
      memcpy(out_ids, tmp, (size_t)num_act * 4 * sizeof(int32_t));
      for (int k = 0; k < kvol; ++k)
        for (int s = 0; s < indice_num[k]; ++s) {
          int32_t *o = &indice_pairs[((size_t)k * 2 + 1) * n + s];
          *o = remap[*o];
        }
      free(kr);
      free(remap);
      free(tmp);
    }
  }
  free(valid);
  free(grid);
  return num_act;
}

/* ------------------------------------------------------------------------ */
/* B. gather - GEMM - scatter                                                */
/* ------------------------------------------------------------------------ */

/* C (m,n) = A (m,k) @ B (k,n), row-major; stands in for torch::mm_out */
static void mm(const float *a, const float *b, float *c, int m, int k, int n) {
  for (int i = 0; i < m; ++i) {
    float *ci = c + (size_t)i * n;
    for (int j = 0; j < n; ++j) ci[j] = 0.f;
    for (int p = 0; p < k; ++p) {
      float av = a[(size_t)i * k + p];
      const float *bp = b + (size_t)p * n;
      for (int j = 0; j < n; ++j) ci[j] += av * bp[j];
    }
  }
}

/* C (k,n) = A(m,k)^T @ B (m,n) */
static void mm_tn(const float *a, const float *b, float *c, int m, int k, int n) {
  memset(c, 0, (size_t)k * n * sizeof(float));
  for (int i = 0; i < m; ++i) {
    const float *bi = b + (size_t)i * n;
    for (int p = 0; p < k; ++p) {
      float av = a[(size_t)i * k + p];
      float *cp = c + (size_t)p * n;
      for (int j = 0; j < n; ++j) cp[j] += av * bi[j];
    }
  }
}

/* C (m,k) = A(m,n) @ B(k,n)^T.  Every c[i][p] is the sum over j = 0 .. n-1 in that order (what the plain
 * dot-product loop computes); B is transposed once so that the inner loop runs over p with unit stride and the
 * compiler can vectorise it WITHOUT reassociating any sum (the scalar port was 2-5x slower than the
 * reference's MKL-backed path, which made the CPU baseline look worse than the CPU is). */
static void mm_nt(const float *a, const float *b, float *c, int m, int n, int k) {
  float *bt = (float *)malloc((size_t)n * k * sizeof(float));
  for (int p = 0; p < k; ++p)
    for (int j = 0; j < n; ++j) bt[(size_t)j * k + p] = b[(size_t)p * n + j];
  for (int i = 0; i < m; ++i) {
    const float *ai = a + (size_t)i * n;
    float *ci = c + (size_t)i * k;
    for (int p = 0; p < k; ++p) ci[p] = 0.f;
    for (int j = 0; j < n; ++j) {
      const float av = ai[j];
      const float *bj = bt + (size_t)j * k;
      for (int p = 0; p < k; ++p) ci[p] += av * bj[p];
    }
  }
  free(bt);
}

static int argmax_first(const int32_t *v, int n) { /* std::max_element, spconv_ops.h:272 */
  int best = 0;
  for (int i = 1; i < n; ++i)
    if (v[i] > v[best]) best = i;
  return best;
}

void orc_indice_conv(const float *features, int n_in, const float *filters,
                     const int32_t *indice_pairs, const int32_t *indice_num,
                     int pair_stride, int kvol, int cin, int cout, int n_out,
                     int subm, float *out) {
  int max_off = argmax_first(indice_num, kvol);
  int max_size = indice_num[max_off];
  memset(out, 0, (size_t)n_out * cout * sizeof(float));
  float *ibuf = (float *)malloc((size_t)(max_size > 0 ? max_size : 1) * cin * sizeof(float));
  float *obuf = (float *)malloc((size_t)(max_size > 0 ? max_size : 1) * cout * sizeof(float));
  if (subm) /* spconv_ops.h:300-303: centre offset as one dense mm */
    mm(features, filters + (size_t)max_off * cin * cout, out, n_in, cin, cout);
  for (int k = 0; k < kvol; ++k) { /* spconv_ops.h:307-355 */
    int nhot = indice_num[k];
    if (nhot <= 0 || (subm && k == max_off)) continue;
    const int32_t *in_idx = indice_pairs + ((size_t)k * 2 + 0) * pair_stride;
    const int32_t *out_idx = indice_pairs + ((size_t)k * 2 + 1) * pair_stride;
    for (int i = 0; i < nhot; ++i) /* reordering.cc:21-32 gather */
      memcpy(ibuf + (size_t)i * cin, features + (size_t)in_idx[i] * cin,
             cin * sizeof(float));
    mm(ibuf, filters + (size_t)k * cin * cout, obuf, nhot, cin, cout);
    for (int i = 0; i < nhot; ++i) { /* reordering.cc:34-50 scatter-add */
      float *o = out + (size_t)out_idx[i] * cout;
      const float *b = obuf + (size_t)i * cout;
      for (int j = 0; j < cout; ++j) o[j] += b[j];
    }
  }
  free(ibuf);
  free(obuf);
}

void orc_indice_conv_backward(const float *features, int n_in, const float *filters,
                              const float *out_grad, int n_out,
                              const int32_t *indice_pairs, const int32_t *indice_num,
                              int pair_stride, int kvol, int cin, int cout, int subm,
                              float *in_grad, float *filt_grad) {
  int max_off = argmax_first(indice_num, kvol);
  int max_size = indice_num[max_off];
  memset(in_grad, 0, (size_t)n_in * cin * sizeof(float));
  memset(filt_grad, 0, (size_t)kvol * cin * cout * sizeof(float));
  float *ibuf = (float *)malloc((size_t)(max_size > 0 ? max_size : 1) * cin * sizeof(float));
  float *obuf = (float *)malloc((size_t)(max_size > 0 ? max_size : 1) * cout * sizeof(float));
  if (subm) { /* spconv_ops.h:393-397 */
    mm_tn(features, out_grad, filt_grad + (size_t)max_off * cin * cout, n_in, cin, cout);
    mm_nt(out_grad, filters + (size_t)max_off * cin * cout, in_grad, n_out, cout, cin);
  }
  for (int k = 0; k < kvol; ++k) { /* spconv_ops.h:398-453 */
    int nhot = indice_num[k];
    if (nhot <= 0 || (subm && k == max_off)) continue;
    const int32_t *in_idx = indice_pairs + ((size_t)k * 2 + 0) * pair_stride;
    const int32_t *out_idx = indice_pairs + ((size_t)k * 2 + 1) * pair_stride;
    for (int i = 0; i < nhot; ++i) {
      memcpy(ibuf + (size_t)i * cin, features + (size_t)in_idx[i] * cin,
             cin * sizeof(float));
      memcpy(obuf + (size_t)i * cout, out_grad + (size_t)out_idx[i] * cout,
             cout * sizeof(float));
    }
    mm_tn(ibuf, obuf, filt_grad + (size_t)k * cin * cout, nhot, cin, cout);
    mm_nt(obuf, filters + (size_t)k * cin * cout, ibuf, nhot, cout, cin);
    for (int i = 0; i < nhot; ++i) {
      float *o = in_grad + (size_t)in_idx[i] * cin;
      const float *b = ibuf + (size_t)i * cin;
      for (int j = 0; j < cin; ++j) o[j] += b[j];
    }
  }
  free(ibuf);
  free(obuf);
}

/* ------------------------------------------------------------------------ */
/* E. rotated BEV overlap / IoU / NMS                                        */
/* ------------------------------------------------------------------------ */
/* Transcendentals: the device code calls cos/sin/atan2 on floats; the CPU twin
 * (iou3d_cpu.cpp, C <math.h>) evaluates them in double and rounds to float.
 * Oracle and HIP kernel both use "double libm, round to float" so that results
 * agree wherever the double result is not within 2^-29 of a float tie. */
/* Round 2: compiling the reference's iou3d_cpu.cpp here (oracle/_ref) showed that g++ resolves
 * its cos(float) / sin(float) / atan2(float, float) to the FLOAT libm overloads (cosf ...), which
 * are not correctly rounded everywhere: 1 box in ~500 gets a 1-ulp different cosine and its IoUs
 * move by up to 2e-6.  orc_set_trig_mode(1) selects the float libm calls, with which this file
 * reproduces the compiled reference bit for bit (tests/test_oracle_ops.py); the default mode 0
 * (correctly rounded) is what the HIP pre-pass computes. */
static int g_trig_mode = 0;
void orc_set_trig_mode(int m) { g_trig_mode = m; }
static float cosf_d(float x) { return g_trig_mode ? cosf(x) : (float)cos((double)x); }
static float sinf_d(float x) { return g_trig_mode ? sinf(x) : (float)sin((double)x); }
static float atan2f_d(float y, float x) {
  return g_trig_mode ? atan2f(y, x) : (float)atan2((double)y, (double)x);
}

typedef struct {
  float x, y;
} pt_t;

static const float IOU_EPS = 1e-8f;

static float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; } /* :35-37 */
static float cross3(pt_t p1, pt_t p2, pt_t p0) {                      /* :39-41 */
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float fmin2(float a, float b) { return a < b ? a : b; }
static float fmax2(float a, float b) { return a > b ? a : b; }

static int check_rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) { /* :43-49 */
  return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
         fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

static int check_in_box2d(const float *box, pt_t p) { /* :51-62, MARGIN 1e-2 */
  const float MARGIN = 1e-2f;
  float center_x = box[0], center_y = box[1];
  float angle_cos = cosf_d(-box[6]), angle_sin = sinf_d(-box[6]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * (-angle_sin);
  float rot_y = (p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos;
  return fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN;
}

static int intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t *ans) { /* :64-93 */
  if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static pt_t rotate_around_center(pt_t center, float c, float s, pt_t p) { /* :95-99 */
  pt_t r;
  r.x = (p.x - center.x) * c + (p.y - center.y) * (-s) + center.x;
  r.y = (p.x - center.x) * s + (p.y - center.y) * c + center.y;
  return r;
}

static int point_cmp(pt_t a, pt_t b, pt_t center) { /* :101-103 */
  return atan2f_d(a.y - center.y, a.x - center.x) > atan2f_d(b.y - center.y, b.x - center.x);
}

float orc_box_overlap(const float *box_a, const float *box_b) { /* :104-224 */
  float a_angle = box_a[6], b_angle = box_b[6];
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2;
  float a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  pt_t center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
  pt_t ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt_t cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf_d(a_angle), a_sin = sinf_d(a_angle);
  float b_cos = cosf_d(b_angle), b_sin = sinf_d(b_angle);
  for (int k = 0; k < 4; k++) {
    ca[k] = rotate_around_center(center_a, a_cos, a_sin, ca[k]);
    cb[k] = rotate_around_center(center_b, b_cos, b_sin, cb[k]);
  }
  ca[4] = ca[0];
  cb[4] = cb[0];
  pt_t cross_points[16];
  pt_t poly_center = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      int flag = intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &cross_points[cnt]);
      if (flag) {
        poly_center.x = poly_center.x + cross_points[cnt].x;
        poly_center.y = poly_center.y + cross_points[cnt].y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d(box_a, cb[k])) {
      poly_center.x = poly_center.x + cb[k].x;
      poly_center.y = poly_center.y + cb[k].y;
      cross_points[cnt++] = cb[k];
    }
    if (check_in_box2d(box_b, ca[k])) {
      poly_center.x = poly_center.x + ca[k].x;
      poly_center.y = poly_center.y + ca[k].y;
      cross_points[cnt++] = ca[k];
    }
  }
  if (cnt == 0) return 0.f; /* :196-197 divide gives NaN that is never read */
  poly_center.x /= cnt;
  poly_center.y /= cnt;
  for (int j = 0; j < cnt - 1; j++) /* :200-209 bubble sort by atan2 */
    for (int i = 0; i < cnt - j - 1; i++)
      if (point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
        pt_t t = cross_points[i];
        cross_points[i] = cross_points[i + 1];
        cross_points[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    pt_t u = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
    pt_t v = {cross_points[k + 1].x - cross_points[0].x,
              cross_points[k + 1].y - cross_points[0].y};
    area += cross2(u, v);
  }
  return (float)(fabs((double)area) / 2.0); /* :224 fabs(area) / 2.0 in double */
}

float orc_iou_bev(const float *a, const float *b) { /* :226-233 */
  float sa = a[3] * a[4];
  float sb = b[3] * b[4];
  float s_overlap = orc_box_overlap(a, b);
  return s_overlap / fmaxf(sa + sb - s_overlap, IOU_EPS);
}

void orc_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_box_overlap(a + i * 7, b + j * 7);
}

void orc_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_iou_bev(a + i * 7, b + j * 7);
}

static float iou_normal(const float *a, const float *b) { /* :314-326 */
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
  float right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
  float bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4];
  float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, IOU_EPS);
}

/* mask[r][c] bit i = iou(box_r, box_{64c+i}) > thresh, diagonal tile from
 * i = r%64+1 (kernel :299-307); greedy over remv words (iou3d_nms.cpp:117-133). */
static int nms_generic(const float *boxes, int n, float thresh, int64_t *keep, int normal) {
  int col_blocks = (n + 63) / 64;
  if (n == 0) return 0;
  uint64_t *mask = (uint64_t *)calloc((size_t)n * col_blocks, sizeof(uint64_t));
  for (int r = 0; r < n; ++r)
    for (int cb = r / 64; cb < col_blocks; ++cb) {
      uint64_t t = 0;
      int col_size = n - cb * 64 < 64 ? n - cb * 64 : 64;
      int start = (r / 64 == cb) ? (r % 64) + 1 : 0;
      for (int i = start; i < col_size; ++i) {
        const float *bj = boxes + (size_t)(cb * 64 + i) * 7;
        float v = normal ? iou_normal(boxes + (size_t)r * 7, bj)
                         : orc_iou_bev(boxes + (size_t)r * 7, bj);
        if (v > thresh) t |= 1ULL << i;
      }
      mask[(size_t)r * col_blocks + cb] = t;
    }
  uint64_t *remv = (uint64_t *)calloc(col_blocks, sizeof(uint64_t));
  int num_to_keep = 0;
  for (int i = 0; i < n; i++) {
    int nblock = i / 64, inblock = i % 64;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num_to_keep++] = i;
      const uint64_t *p = mask + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
    }
  }
  free(mask);
  free(remv);
  return num_to_keep;
}

int orc_nms(const float *boxes, int n, float thresh, int64_t *keep) {
  return nms_generic(boxes, n, thresh, keep, 0);
}
int orc_nms_normal(const float *boxes, int n, float thresh, int64_t *keep) {
  return nms_generic(boxes, n, thresh, keep, 1);
}

/* ------------------------------------------------------------------------ */
/* D. stacked PointNet++ ops                                                 */
/* ------------------------------------------------------------------------ */

/* a*a + b*b + c*c as the device compiler contracts it: fma(c,c, fma(b,b, a*a)) */
static float dist2_fma(float dx, float dy, float dz) {
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

void orc_ball_query_stack(int b, int m, float radius, int nsample, const float *new_xyz,
                          const int32_t *new_xyz_batch_cnt, const float *xyz,
                          const int32_t *xyz_batch_cnt, int32_t *idx) {
  float radius2 = radius * radius;
  for (int pt = 0; pt < m; ++pt) {
    int bs_idx = 0, pt_cnt = new_xyz_batch_cnt[0];
    for (int k = 1; k < b; k++) { /* ball_query_gpu.cu:27-32 */
      if (pt < pt_cnt) break;
      pt_cnt += new_xyz_batch_cnt[k];
      bs_idx = k;
    }
    int start = 0;
    for (int k = 0; k < bs_idx; k++) start += xyz_batch_cnt[k];
    const float *q = new_xyz + (size_t)pt * 3;
    const float *base = xyz + (size_t)start * 3;
    int32_t *out = idx + (size_t)pt * nsample;
    int n = xyz_batch_cnt[bs_idx], cnt = 0;
    for (int k = 0; k < n; ++k) { /* :48-63 */
      float d2 = dist2_fma(q[0] - base[k * 3 + 0], q[1] - base[k * 3 + 1],
                           q[2] - base[k * 3 + 2]);
      if (d2 < radius2) {
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k;
        out[cnt] = k;
        ++cnt;
        if (cnt >= nsample) break;
      }
    }
    if (cnt == 0) out[0] = -1; /* :65; the rest of the row keeps its prior contents */
  }
}

static int batch_of(const int32_t *cnt, int b, int pt) {
  int bs_idx = 0, pt_cnt = cnt[0];
  for (int k = 1; k < b; k++) {
    if (pt < pt_cnt) break;
    pt_cnt += cnt[k];
    bs_idx = k;
  }
  return bs_idx;
}

void orc_group_points_stack(int b, int m, int c, int nsample, const float *features,
                            const int32_t *features_batch_cnt, const int32_t *idx,
                            const int32_t *idx_batch_cnt, float *out) {
  for (int pt = 0; pt < m; ++pt) {
    int bs_idx = batch_of(idx_batch_cnt, b, pt);
    int start = 0;
    for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
    for (int ci = 0; ci < c; ++ci)
      for (int s = 0; s < nsample; ++s)
        out[((size_t)pt * c + ci) * nsample + s] =
            features[((size_t)start + idx[(size_t)pt * nsample + s]) * c + ci];
  }
}

void orc_group_points_grad_stack(int b, int m, int c, int n, int nsample,
                                 const float *grad_out, const int32_t *idx,
                                 const int32_t *idx_batch_cnt,
                                 const int32_t *features_batch_cnt, float *grad_features) {
  memset(grad_features, 0, (size_t)n * c * sizeof(float));
  for (int pt = 0; pt < m; ++pt) {
    int bs_idx = batch_of(idx_batch_cnt, b, pt);
    int start = 0;
    for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
    for (int ci = 0; ci < c; ++ci)
      for (int s = 0; s < nsample; ++s)
        grad_features[((size_t)start + idx[(size_t)pt * nsample + s]) * c + ci] +=
            grad_out[((size_t)pt * c + ci) * nsample + s];
  }
}

void orc_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                 int32_t *idxs) {
  if (m <= 0) return;
  /* sampling_gpu.cu:9-13 opt_n_threads: min(2^floor(log2 n), 1024), at least 1 */
  int pow2 = (int)(log((double)n) / log(2.0));
  int bs = 1 << pow2;
  if (bs > 1024) bs = 1024;
  if (bs < 1) bs = 1;
  float *dists = (float *)malloc(bs * sizeof(float));
  int *dists_i = (int *)malloc(bs * sizeof(int));
  for (int bi = 0; bi < b; ++bi) {
    const float *data = xyz + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int32_t *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; j++) {
      float x1 = data[old * 3 + 0], y1 = data[old * 3 + 1], z1 = data[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) { /* :55-70 per-thread strided scan */
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += bs) {
          float d = dist2_fma(data[k * 3 + 0] - x1, data[k * 3 + 1] - y1,
                              data[k * 3 + 2] - z1);
          float d2 = d < tmp[k] ? d : tmp[k];
          tmp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) /* :75-134 tree, __update :16-21 */
        for (int tid = 0; tid < s; ++tid) {
          float v1 = dists[tid], v2 = dists[tid + s];
          int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* ------------------------------------------------------------------------ */
/* D. points in boxes (GPU semantics, MARGIN 1e-5)                           */
/* ------------------------------------------------------------------------ */
void orc_points_in_boxes(int batch, int nboxes, int npts, const float *boxes,
                         const float *pts, int32_t *box_idx) {
  const float MARGIN = 1e-5f;
  for (int bi = 0; bi < batch; ++bi)
    for (int p = 0; p < npts; ++p) {
      const float *pt = pts + ((size_t)bi * npts + p) * 3;
      int32_t res = -1;
      for (int k = 0; k < nboxes; ++k) {
        const float *bx = boxes + ((size_t)bi * nboxes + k) * 7;
        float x = pt[0], y = pt[1], z = pt[2];
        float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5], rz = bx[6];
        /* roiaware_pool3d_kernel.cu:32: fabsf(z - cz) > dz / 2.0 (double compare) */
        if ((double)fabsf(z - cz) > (double)dz / 2.0) continue;
        float cosa = cosf_d(-rz), sina = sinf_d(-rz); /* :17-21 */
        float sx = x - cx, sy = y - cy;
        float local_x = sx * cosa + sy * (-sina);
        float local_y = sx * sina + sy * cosa;
        /* :34: fabs(local) < d / 2.0 + MARGIN evaluated in double */
        int in_flag = ((double)fabsf(local_x) < (double)dx / 2.0 + (double)MARGIN) &
                      ((double)fabsf(local_y) < (double)dy / 2.0 + (double)MARGIN);
        if (in_flag) {
          res = k;
          break;
        }
      }
      box_idx[(size_t)bi * npts + p] = res;
    }
}

/* ------------------------------------------------------------------------------------------
 * RoIAlign (avg) — mmcv.ops.RoIAlign as configured at configs/detmatch/001/detmatch/split_0.py:78
 * (output_size 7, sampling_ratio 0, aligned default True).  The algorithm lives in mmcv-full
 * 1.3.16 (mmcv/ops/csrc/roi_align_cuda_kernel.cuh), absent from /root/reference: PARITY UNPINNED.
 * ------------------------------------------------------------------------------------------ */
static int orc_ra_tap(float p, int size, int *lo, int *hi, float *wl, float *wh) {
  if (p < -1.0f || p > (float)size) return 0;
  if (p <= 0.0f) p = 0.0f;
  *lo = (int)p;
  if (*lo >= size - 1) {
    *hi = *lo = size - 1;
    p = (float)*lo;
  } else {
    *hi = *lo + 1;
  }
  float l = p - (float)*lo;
  *wl = 1.0f - l;
  *wh = l;
  return 1;
}

static void orc_roi_align_impl(const float *feat, const float *gout, double *gfeat, int c, int h,
                               int w, const float *rois, int r, float scale, int ph, int pw,
                               int sampling_ratio, int aligned, float *out) {
  for (int n = 0; n < r; ++n) {
    const float *roi = rois + (size_t)n * 5;
    int batch = (int)roi[0];
    float off = aligned ? 0.5f : 0.0f;
    float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
    float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
      rw = fmaxf(rw, 1.0f);
      rh = fmaxf(rh, 1.0f);
    }
    float bh = rh / (float)ph, bw = rw / (float)pw;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int ch = 0; ch < c; ++ch) {
      size_t plane = ((size_t)batch * c + ch) * h * w;
      for (int by = 0; by < ph; ++by)
        for (int bx = 0; bx < pw; ++bx) {
          size_t oidx = (((size_t)n * c + ch) * ph + by) * pw + bx;
          double acc = 0.0;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)by * bh + ((float)iy + 0.5f) * bh / (float)gh;
            int yl, yh;
            float wyl, wyh;
            if (!orc_ra_tap(y, h, &yl, &yh, &wyl, &wyh)) continue;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)bx * bw + ((float)ix + 0.5f) * bw / (float)gw;
              int xl, xh;
              float wxl, wxh;
              if (!orc_ra_tap(x, w, &xl, &xh, &wxl, &wxh)) continue;
              float w1 = wyl * wxl, w2 = wyl * wxh, w3 = wyh * wxl, w4 = wyh * wxh;
              if (feat) {
                acc += (double)w1 * feat[plane + (size_t)yl * w + xl] +
                       (double)w2 * feat[plane + (size_t)yl * w + xh] +
                       (double)w3 * feat[plane + (size_t)yh * w + xl] +
                       (double)w4 * feat[plane + (size_t)yh * w + xh];
              } else {
                double g = (double)gout[oidx] / count;
                gfeat[plane + (size_t)yl * w + xl] += g * w1;
                gfeat[plane + (size_t)yl * w + xh] += g * w2;
                gfeat[plane + (size_t)yh * w + xl] += g * w3;
                gfeat[plane + (size_t)yh * w + xh] += g * w4;
              }
            }
          }
          if (feat) out[oidx] = (float)(acc / count);
        }
    }
  }
}

void orc_roi_align_forward(const float *feat, int c, int h, int w, const float *rois, int r,
                           float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                           float *out) {
  orc_roi_align_impl(feat, NULL, NULL, c, h, w, rois, r, spatial_scale, ph, pw, sampling_ratio,
                     aligned, out);
}

void orc_roi_align_backward(const float *grad_out, int c, int h, int w, const float *rois, int r,
                            float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                            double *grad_feat) {
  orc_roi_align_impl(NULL, grad_out, grad_feat, c, h, w, rois, r, spatial_scale, ph, pw,
                     sampling_ratio, aligned, NULL);
}


/* ---- 3D augmentation of one view (transforms_3d.py RandomFlip3D / GlobalRotScaleTrans /
 * PointsRangeFilter on LiDARPoints: lidar_points.py:28-33, base_points.py:139-229,263-269) ---- */
int orc_points_augment(const float *points, int n, int n_feat, const float *P,
                       const int32_t *perm, float *out) {
  int kept = 0;
  for (int j = 0; j < n; ++j) {
    const float *row = points + (size_t)(perm ? perm[j] : j) * n_feat;
    float x = row[0], y = row[1], z = row[2];
    if (P[0] != 0.f) y = -y;
    if (P[1] != 0.f) x = -x;
    float rx = x * P[2];
    rx = fmaf(y, P[5], rx);
    rx = fmaf(z, P[8], rx);
    float ry = x * P[3];
    ry = fmaf(y, P[6], ry);
    ry = fmaf(z, P[9], ry);
    float rz = x * P[4];
    rz = fmaf(y, P[7], rz);
    rz = fmaf(z, P[10], rz);
    x = rx * P[11] + P[12];      /* -ffp-contract=off: product and sum round separately */
    y = ry * P[11] + P[13];
    z = rz * P[11] + P[14];
    if (x > P[15] && y > P[16] && z > P[17] && x < P[18] && y < P[19] && z < P[20]) {
      float *dst = out + (size_t)kept * n_feat;
      dst[0] = x, dst[1] = y, dst[2] = z;
      for (int c = 3; c < n_feat; ++c) dst[c] = row[c];
      ++kept;
    }
  }
  return kept;
}
