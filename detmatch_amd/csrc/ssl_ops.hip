// Teacher-student support ops for gfx950: fused EMA over flat parameter arenas and the
// small host-side linear-assignment solve.
//
// EMA replaces mmdet3d/models/detectors/ssl.py:146-163 (_update_teacher): the reference
// rebuilds a ~600-entry state dict with three elementwise launches per entry every
// iteration; here teacher and student live in two identically laid-out arenas and one
// launch updates all floats: t = s * f32(1 - d) + t * f32(d)   (two roundings of the
// products, one of the sum — the reference's fp32 expression order; HBM-bound, 12 B/elem).
// Integer buffers (BatchNorm num_batches_tracked, pcdet global_step) follow the reference's
// promotion rule: the product/sum is evaluated in fp32 and truncated back to int64
// (load_state_dict copy_, ssl.py:163).
//
// LAP replaces scipy.optimize.linear_sum_assignment at
// mmdet3d/core/bbox/assigners/modified_hungarian_assigner.py:132 (n, m <= ~100):
// shortest augmenting path (Jonker-Volgenant style) on the host, in double.
#include <cmath>
#include <limits>
#include <vector>

#include "dm_common.h"

namespace {

__global__ __launch_bounds__(256) void ema_f32_kernel(float *__restrict__ t,
                                                      const float *__restrict__ s, size_t n,
                                                      float one_minus_d, float d) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 tv = *(float4 *)(t + i);
    float4 sv = *(const float4 *)(s + i);
    tv.x = __fadd_rn(__fmul_rn(sv.x, one_minus_d), __fmul_rn(tv.x, d));
    tv.y = __fadd_rn(__fmul_rn(sv.y, one_minus_d), __fmul_rn(tv.y, d));
    tv.z = __fadd_rn(__fmul_rn(sv.z, one_minus_d), __fmul_rn(tv.z, d));
    tv.w = __fadd_rn(__fmul_rn(sv.w, one_minus_d), __fmul_rn(tv.w, d));
    *(float4 *)(t + i) = tv;
  } else {
    for (; i < n; ++i) t[i] = __fadd_rn(__fmul_rn(s[i], one_minus_d), __fmul_rn(t[i], d));
  }
}

__global__ __launch_bounds__(256) void ema_i64_kernel(long long *__restrict__ t,
                                                      const long long *__restrict__ s, size_t n,
                                                      float one_minus_d, float d) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = __fadd_rn(__fmul_rn((float)s[i], one_minus_d), __fmul_rn((float)t[i], d));
  t[i] = (long long)v;  // truncation toward zero, as Tensor.copy_(float -> long)
}

}  // namespace

extern "C" int dm_ema_update_f32(float *teacher, const float *student, size_t n, double decay,
                                 dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return DM_OK;
  if (!teacher || !student) return DM_ERR_INVALID_ARG;
  if (((uintptr_t)teacher & 15) || ((uintptr_t)student & 15)) return DM_ERR_INVALID_ARG;
  // a Python float scalar becomes an fp32 scalar when it multiplies an fp32 tensor
  float omd = (float)(1.0 - decay), d = (float)decay;
  size_t threads = (n + 3) / 4;
  ema_f32_kernel<<<dm_ceil_div((long long)threads, 256), 256, 0, st>>>(teacher, student, n, omd, d);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_ema_update_i64(long long *teacher, const long long *student, size_t n,
                                 double decay, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return DM_OK;
  if (!teacher || !student) return DM_ERR_INVALID_ARG;
  float omd = (float)(1.0 - decay), d = (float)decay;
  ema_i64_kernel<<<dm_ceil_div((long long)n, 256), 256, 0, st>>>(teacher, student, n, omd, d);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// Rectangular linear sum assignment, minimisation.  cost (n_rows, n_cols) row-major HOST
// floats.  Writes min(n_rows, n_cols) matched pairs sorted by row (the scipy contract).
// Returns the number of pairs or -1 on invalid input (NaN / all-infinite rows).
extern "C" int dm_lap_host(const float *cost_host, int n_rows, int n_cols, int *row_ind_host,
                           int *col_ind_host) {
  if (n_rows < 0 || n_cols < 0) return -1;
  if (n_rows == 0 || n_cols == 0) return 0;
  if (!cost_host || !row_ind_host || !col_ind_host) return -1;
  // work on the orientation with rows <= cols
  const bool transpose = n_rows > n_cols;
  const int nr = transpose ? n_cols : n_rows, nc = transpose ? n_rows : n_cols;
  auto C = [&](int i, int j) -> double {
    return transpose ? (double)cost_host[(size_t)j * n_cols + i]
                     : (double)cost_host[(size_t)i * n_cols + j];
  };
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j)
      if (std::isnan(C(i, j))) return -1;
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), shortest(nc);
  std::vector<int> col4row(nr, -1), row4col(nc, -1), path(nc, -1);
  std::vector<char> SR(nr), SC(nc);
  for (int cur = 0; cur < nr; ++cur) {
    std::fill(shortest.begin(), shortest.end(), INF);
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    double min_val = 0.0;
    int i = cur, sink = -1;
    while (sink == -1) {
      int index = -1;
      double lowest = INF;
      SR[i] = 1;
      for (int j = 0; j < nc; ++j) {
        if (SC[j]) continue;
        double r = min_val + C(i, j) - u[i] - v[j];
        if (r < shortest[j]) {
          path[j] = i;
          shortest[j] = r;
        }
        // ties prefer a still-unassigned column (keeps augmenting paths short)
        if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
          lowest = shortest[j];
          index = j;
        }
      }
      min_val = lowest;
      if (index < 0 || min_val == INF) return -1;  // infeasible
      int j = index;
      if (row4col[j] == -1) sink = j;
      else i = row4col[j];
      SC[j] = 1;
    }
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - shortest[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - shortest[j];
    int j = sink;
    while (true) {
      int r = path[j];
      row4col[j] = r;
      int prev = col4row[r];
      col4row[r] = j;
      j = prev;
      if (r == cur) break;
    }
  }
  if (!transpose) {
    for (int i = 0; i < nr; ++i) {
      row_ind_host[i] = i;
      col_ind_host[i] = col4row[i];
    }
  } else {  // pairs sorted by (original) row = our column index
    int k = 0;
    for (int j = 0; j < nc; ++j)
      if (row4col[j] != -1) {
        row_ind_host[k] = j;
        col_ind_host[k] = row4col[j];
        ++k;
      }
  }
  return nr;
}
