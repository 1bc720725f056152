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

// ---- fused optimizer steps over flat arenas (HybridOptimizer: AdamW on the 3D student, SGD with
// momentum on the 2D student — configs/detmatch/001/detmatch/split_0.py:829-852).  The reference
// runs torch.optim per parameter tensor (~600 tensors x ~6 element-wise launches per step); with
// parameters, gradients and optimizer state laid out in aligned flat arenas a step is one launch
// per optimizer.  `grad_scale` (device scalar, optional) carries the clip coefficient of
// clip_grad_norm_ so clipping needs no pass of its own.  Formulas = torch.optim.AdamW / SGD.
struct AdamWArgs {
  float lr_wd;          // lr * weight_decay
  float one_minus_b1, b2, one_minus_b2;
  float step_size;      // lr / (1 - b1^t)
  float inv_bc2_sqrt;   // 1 / sqrt(1 - b2^t)
  float eps;
};

__global__ __launch_bounds__(256) void adamw_f32_kernel(float *__restrict__ p,
                                                        const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v,
                                                        size_t n, AdamWArgs a,
                                                        const float *__restrict__ grad_scale,
                                                        const unsigned char *__restrict__ live,
                                                        const int *__restrict__ first_step, long long step,
                                                        double lr, double beta1, double beta2) {
  const float gs = grad_scale ? *grad_scale : 1.0f;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (live && !live[i >> 2]) return;   // parameter never received a gradient: torch skips it
  if (first_step) {
    // torch.optim.AdamW keeps a per-parameter step that starts at 1 with the parameter's FIRST gradient:
    // a parameter that became live at step f > 1 takes its bias corrections from step - f + 1
    const int f = first_step[i >> 2];
    if (f <= 0) return;
    if (f > 1) {
      const double t = (double)(step - f + 1);
      a.step_size = (float)(lr / (1.0 - pow(beta1, t)));
      a.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
    }
  }
  const int cnt = (int)(n - i < 4 ? n - i : 4);
  float pv[4], gv[4], mv[4], vv[4];
  if (cnt == 4) {
    *(float4 *)pv = *(const float4 *)(p + i);
    *(float4 *)gv = *(const float4 *)(g + i);
    *(float4 *)mv = *(const float4 *)(m + i);
    *(float4 *)vv = *(const float4 *)(v + i);
  } else {
    for (int k = 0; k < cnt; ++k) pv[k] = p[i + k], gv[k] = g[i + k], mv[k] = m[i + k], vv[k] = v[i + k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gk = gv[k] * gs;
    float pk = pv[k] - pv[k] * a.lr_wd;                    // param.mul_(1 - lr * wd)
    const float mk = mv[k] + (gk - mv[k]) * a.one_minus_b1;   // exp_avg.lerp_(grad, 1 - beta1)
    const float vk = vv[k] * a.b2 + gk * gk * a.one_minus_b2;
    const float denom = sqrtf(vk) * a.inv_bc2_sqrt + a.eps;
    pk -= a.step_size * (mk / denom);
    pv[k] = pk, mv[k] = mk, vv[k] = vk;
  }
  if (cnt == 4) {
    *(float4 *)(p + i) = *(float4 *)pv;
    *(float4 *)(m + i) = *(float4 *)mv;
    *(float4 *)(v + i) = *(float4 *)vv;
  } else {
    for (int k = 0; k < cnt; ++k) p[i + k] = pv[k], m[i + k] = mv[k], v[i + k] = vv[k];
  }
}

__global__ __launch_bounds__(256) void sgd_f32_kernel(float *__restrict__ p,
                                                      const float *__restrict__ g,
                                                      float *__restrict__ buf, size_t n, float lr,
                                                      float momentum, float dampening, float wd,
                                                      int first, const float *__restrict__ grad_scale,
                                                      const unsigned char *__restrict__ live,
                                                      const int *__restrict__ first_step, long long step) {
  const float gs = grad_scale ? *grad_scale : 1.0f;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (live && !live[i >> 2]) return;
  if (first_step) {          // torch.optim.SGD creates a parameter's momentum buffer at ITS first gradient
    const int f = first_step[i >> 2];
    if (f <= 0) return;
    first = (long long)f == step;
  }
  const int cnt = (int)(n - i < 4 ? n - i : 4);
  const bool mom = momentum != 0.0f;
  // whole 16-byte groups as float4 when the three arrays are 16-byte aligned (the arenas are): the scalar form of this
  // kernel ran at half the bandwidth of adamw_f32_kernel (317 us for the 41 M parameters of the 2D student: 2.6 TB/s)
  const bool vec = cnt == 4 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)(mom ? buf : p)) & 15) == 0);
  float pv[4], gv[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    *(float4 *)pv = *(const float4 *)(p + i);
    *(float4 *)gv = *(const float4 *)(g + i);
    if (mom && !first) *(float4 *)bv = *(const float4 *)(buf + i);
  } else {
    for (int k = 0; k < cnt; ++k) {
      pv[k] = p[i + k], gv[k] = g[i + k];
      if (mom && !first) bv[k] = buf[i + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float pk = pv[k];
    float d = gv[k] * gs + wd * pk;                          // grad.add(param, alpha=wd)
    if (mom) {
      const float b = first ? d : bv[k] * momentum + d * (1.0f - dampening);
      bv[k] = b;
      d = b;
    }
    pv[k] = pk - lr * d;
  }
  if (vec) {
    *(float4 *)(p + i) = *(float4 *)pv;
    if (mom) *(float4 *)(buf + i) = *(float4 *)bv;
  } else {
    for (int k = 0; k < cnt; ++k) {
      p[i + k] = pv[k];
      if (mom) buf[i + k] = bv[k];
    }
  }
}

}  // namespace

extern "C" int dm_adamw_step_blocks_f32(float *params, const float *grads, float *exp_avg,
                                        float *exp_avg_sq, size_t n, double lr, double beta1,
                                        double beta2, double eps, double weight_decay,
                                        long long step, const float *grad_scale_dev,
                                        const unsigned char *block_live, const int *block_first_step,
                                        dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return DM_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return DM_ERR_INVALID_ARG;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
    return DM_ERR_INVALID_ARG;
  AdamWArgs a;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.lr_wd = (float)(lr * weight_decay);
  a.one_minus_b1 = (float)(1.0 - beta1);
  a.b2 = (float)beta2;
  a.one_minus_b2 = (float)(1.0 - beta2);
  a.step_size = (float)(lr / bc1);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.eps = (float)eps;
  adamw_f32_kernel<<<dm_ceil_div((long long)((n + 3) / 4), 256), 256, 0, st>>>(
      params, grads, exp_avg, exp_avg_sq, n, a, grad_scale_dev, block_live, block_first_step, step, lr, beta1,
      beta2);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_adamw_step_masked_f32(float *params, const float *grads, float *exp_avg,
                                        float *exp_avg_sq, size_t n, double lr, double beta1,
                                        double beta2, double eps, double weight_decay,
                                        long long step, const float *grad_scale_dev,
                                        const unsigned char *block_live, dm_stream_t stream) {
  return dm_adamw_step_blocks_f32(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                                  step, grad_scale_dev, block_live, nullptr, stream);
}

extern "C" int dm_adamw_step_f32(float *params, const float *grads, float *exp_avg,
                                 float *exp_avg_sq, size_t n, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, long long step,
                                 const float *grad_scale_dev, dm_stream_t stream) {
  return dm_adamw_step_masked_f32(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                  weight_decay, step, grad_scale_dev, nullptr, stream);
}

extern "C" int dm_sgd_step_blocks_f32(float *params, const float *grads, float *momentum_buf,
                                      size_t n, double lr, double momentum, double dampening,
                                      double weight_decay, long long step,
                                      const float *grad_scale_dev, const unsigned char *block_live,
                                      const int *block_first_step, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return DM_OK;
  if (!params || !grads || (momentum != 0.0 && !momentum_buf) || step < 1) return DM_ERR_INVALID_ARG;
  sgd_f32_kernel<<<dm_ceil_div((long long)((n + 3) / 4), 256), 256, 0, st>>>(
      params, grads, momentum_buf, n, (float)lr, (float)momentum, (float)dampening,
      (float)weight_decay, step == 1 ? 1 : 0, grad_scale_dev, block_live, block_first_step, step);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_sgd_step_masked_f32(float *params, const float *grads, float *momentum_buf,
                                      size_t n, double lr, double momentum, double dampening,
                                      double weight_decay, int first_step,
                                      const float *grad_scale_dev, const unsigned char *block_live,
                                      dm_stream_t stream) {
  return dm_sgd_step_blocks_f32(params, grads, momentum_buf, n, lr, momentum, dampening, weight_decay,
                                first_step ? 1 : 2, grad_scale_dev, block_live, nullptr, stream);
}

extern "C" int dm_sgd_step_f32(float *params, const float *grads, float *momentum_buf, size_t n,
                               double lr, double momentum, double dampening, double weight_decay,
                               int first_step, const float *grad_scale_dev, dm_stream_t stream) {
  return dm_sgd_step_masked_f32(params, grads, momentum_buf, n, lr, momentum, dampening,
                                weight_decay, first_step, grad_scale_dev, nullptr, stream);
}

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
