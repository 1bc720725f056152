// Fused losses of the anchor head for gfx950.
//
// Replaces AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss
// (pcdet/models/dense_heads/anchor_head_template.py:101-214) with the loss functions of
// pcdet/utils/loss_utils.py (SigmoidFocalClassificationLoss :9-73, WeightedSmoothL1Loss :75-137,
// WeightedCrossEntropyLoss :181-206): in the reference ~160 element-wise torch launches forward and as
// many backward over (B, 211 200, k) tensors per student pass, almost all of them on anchors that
// carry no regression target.  Here one launch computes the three weighted sums (deterministic: per
// block partials, fixed-order final reduce) and one launch writes the three gradients; regression and
// direction terms only touch the positive anchors.
//
//   cls   sum_c focal(x_c, [label == c+1]) * [label >= 0] / max(npos_b, 1)          * cls_w / B
//   loc   sum_k smoothl1((p_k - t_k) * cw_k ; k=6: sin(p)cos(t) - cos(p)sin(t))
//                                         * [label > 0] / max(npos_b, 1)            * loc_w / B
//   dir   CE(dir logits, bin(t_6 + anchor_6)) * [label > 0] / max(npos_b, 1)        * dir_w / B
// HBM-bound: reads 12 + 7 + 1 floats per anchor forward, the same plus 12 written backward.
#include <cmath>

#include "dm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxCls = 8;
constexpr int kMaxBins = 8;

struct AnchorLossCfg {
  int batch, n_anchors, n_cls, n_bins;
  float alpha, beta, dir_offset;
  float scale[3];            // cls_w / B, loc_w / B, dir_w / B
  float cw[7];
};

__device__ __forceinline__ int dir_bin(float t6, float a6, float dir_offset, int n_bins) {
  const float two_pi = 6.283185307179586f;
  float v = (t6 + a6) - dir_offset;
  if (!(v == v)) return 0;                                // NaN target (never on a positive anchor)
  v = v - floorf(v / two_pi + 0.f) * two_pi;              // limit_period(v, 0, 2 pi)
  int b = (int)floorf(v / (two_pi / (float)n_bins));
  return b < 0 ? 0 : (b > n_bins - 1 ? n_bins - 1 : b);
}

template <bool GRAD>
__global__ __launch_bounds__(kBlock) void anchor_loss_kernel(
    const float *__restrict__ cls, const float *__restrict__ box, const float *__restrict__ dir,
    const int32_t *__restrict__ labels, const float *__restrict__ tgt, const float *__restrict__ anchors,
    const float *__restrict__ npos, AnchorLossCfg c, const float *__restrict__ upstream,
    float *__restrict__ gcls, float *__restrict__ gbox, float *__restrict__ gdir,
    float *__restrict__ partial) {
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long total = (long long)c.batch * c.n_anchors;
  float s_cls = 0.f, s_loc = 0.f, s_dir = 0.f;
  if (idx < total) {
    const int b = (int)(idx / c.n_anchors), a = (int)(idx % c.n_anchors);
    const int label = labels[idx];
    const float inv = 1.f / fmaxf(npos[b], 1.f);
    const float wc = label >= 0 ? inv : 0.f;
    const float up0 = GRAD ? upstream[0] * c.scale[0] : 0.f;
    for (int k = 0; k < c.n_cls; ++k) {
      const float x = cls[idx * c.n_cls + k];
      const float t = (label == k + 1) ? 1.f : 0.f;
      const float p = 1.f / (1.f + expf(-x));
      const float aw = t * c.alpha + (1.f - t) * (1.f - c.alpha);
      const float pt = t * (1.f - p) + (1.f - t) * p;
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      if (!GRAD) {
        s_cls += aw * pt * pt * bce * wc;
      } else {
        const float dpt = (1.f - 2.f * t) * p * (1.f - p);
        gcls[idx * c.n_cls + k] = aw * (2.f * pt * dpt * bce + pt * pt * (p - t)) * wc * up0;
      }
    }
    const bool pos = label > 0;
    const float wr = pos ? inv : 0.f;
    float t6 = 0.f;
    {
      const float up1 = GRAD ? upstream[1] * c.scale[1] : 0.f;
      for (int k = 0; k < 7; ++k) {
        float g = 0.f;
        if (pos) {
          const float p = box[idx * 7 + k];
          float t = tgt[idx * 7 + k];
          if (k == 6) t6 = t;
          float diff, dd;
          if (t != t) {                                   // NaN target: neutralised (loss_utils.py:117)
            diff = 0.f, dd = 0.f;
          } else if (k == 6) {
            float sp, cp, st, ct;
            sincosf(p, &sp, &cp);
            sincosf(t, &st, &ct);
            diff = (sp * ct - cp * st) * c.cw[6];
            dd = (cp * ct + sp * st) * c.cw[6];
          } else {
            diff = (p - t) * c.cw[k];
            dd = c.cw[k];
          }
          const float n = fabsf(diff);
          if (!GRAD) {
            s_loc += (n < c.beta ? 0.5f * n * n / c.beta : n - 0.5f * c.beta) * wr;
          } else {
            const float dl = n < c.beta ? diff / c.beta : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
            g = dl * dd * wr * up1;
          }
        }
        if (GRAD) gbox[idx * 7 + k] = g;
      }
    }
    if (dir != nullptr) {
      const float up2 = GRAD ? upstream[2] * c.scale[2] : 0.f;
      if (pos) {
        const int bin = dir_bin(t6, anchors[a * 7 + 6], c.dir_offset, c.n_bins);
        float z[kMaxBins], m = -INFINITY;
        for (int j = 0; j < c.n_bins; ++j) {
          z[j] = dir[idx * c.n_bins + j];
          m = fmaxf(m, z[j]);
        }
        float se = 0.f;
        for (int j = 0; j < c.n_bins; ++j) se += expf(z[j] - m);
        if (!GRAD) {
          s_dir += (logf(se) + m - z[bin]) * wr;
        } else {
          for (int j = 0; j < c.n_bins; ++j)
            gdir[idx * c.n_bins + j] = (expf(z[j] - m) / se - (j == bin ? 1.f : 0.f)) * wr * up2;
        }
      } else if (GRAD) {
        for (int j = 0; j < c.n_bins; ++j) gdir[idx * c.n_bins + j] = 0.f;
      }
    }
  }
  if (GRAD) return;
  // block sums in a fixed order: wave butterfly, then the 4 wave sums by thread 0
  __shared__ float red[3][kBlock / DM_WAVE];
  float v[3] = {s_cls, s_loc, s_dir};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    for (int o = DM_WAVE / 2; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o, DM_WAVE);
    if ((threadIdx.x & (DM_WAVE - 1)) == 0) red[i][threadIdx.x / DM_WAVE] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < kBlock / DM_WAVE; ++w) s += red[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void anchor_loss_reduce(const float *__restrict__ partial, int n_blocks,
                                                             AnchorLossCfg c, float *__restrict__ out) {
  __shared__ double red[3][kBlock];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_blocks; i += kBlock)
    for (int k = 0; k < 3; ++k) s[k] += (double)partial[(size_t)i * 3 + k];
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) out[threadIdx.x] = (float)(red[threadIdx.x][0] * (double)c.scale[threadIdx.x]);
}

int fill_cfg(AnchorLossCfg &c, int batch, int n_anchors, int n_cls, int n_bins, float alpha, float beta,
             float dir_offset, const float *weights3, const float *code_weights7) {
  if (batch <= 0 || n_anchors <= 0 || n_cls <= 0 || n_cls > kMaxCls || n_bins < 0 || n_bins > kMaxBins ||
      !weights3 || !code_weights7 || !(beta > 0.f))
    return DM_ERR_INVALID_ARG;
  c.batch = batch, c.n_anchors = n_anchors, c.n_cls = n_cls, c.n_bins = n_bins;
  c.alpha = alpha, c.beta = beta, c.dir_offset = dir_offset;
  for (int i = 0; i < 3; ++i) c.scale[i] = weights3[i] / (float)batch;
  for (int i = 0; i < 7; ++i) c.cw[i] = code_weights7[i];
  return DM_OK;
}

}  // namespace

extern "C" size_t dm_anchor_head_loss_workspace_bytes(int batch, int n_anchors) {
  long long blocks = ((long long)batch * n_anchors + kBlock - 1) / kBlock;
  return dm_align((size_t)blocks * 3 * sizeof(float));
}

extern "C" int dm_anchor_head_loss_forward(const float *cls_preds, const float *box_preds,
                                           const float *dir_preds, const int32_t *labels,
                                           const float *reg_targets, const float *anchors,
                                           const float *num_pos, int batch, int n_anchors, int n_cls,
                                           int n_bins, float alpha, float beta, float dir_offset,
                                           const float *weights3_host, const float *code_weights7_host,
                                           float *losses3, void *workspace, size_t workspace_bytes,
                                           dm_stream_t stream) {
  AnchorLossCfg c;
  int rc = fill_cfg(c, batch, n_anchors, n_cls, dir_preds ? n_bins : 0, alpha, beta, dir_offset, weights3_host,
                    code_weights7_host);
  if (rc != DM_OK) return rc;
  if (!cls_preds || !box_preds || !labels || !reg_targets || !anchors || !num_pos || !losses3 || !workspace)
    return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_anchor_head_loss_workspace_bytes(batch, n_anchors)) return DM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(((long long)batch * n_anchors + kBlock - 1) / kBlock);
  float *partial = (float *)workspace;
  anchor_loss_kernel<false><<<blocks, kBlock, 0, st>>>(cls_preds, box_preds, dir_preds, labels, reg_targets,
                                                       anchors, num_pos, c, nullptr, nullptr, nullptr, nullptr,
                                                       partial);
  DM_CHECK_LAUNCH();
  anchor_loss_reduce<<<1, kBlock, 0, st>>>(partial, blocks, c, losses3);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_anchor_head_loss_backward(const float *cls_preds, const float *box_preds,
                                            const float *dir_preds, const int32_t *labels,
                                            const float *reg_targets, const float *anchors,
                                            const float *num_pos, int batch, int n_anchors, int n_cls,
                                            int n_bins, float alpha, float beta, float dir_offset,
                                            const float *weights3_host, const float *code_weights7_host,
                                            const float *grad_losses3, float *grad_cls, float *grad_box,
                                            float *grad_dir, dm_stream_t stream) {
  AnchorLossCfg c;
  int rc = fill_cfg(c, batch, n_anchors, n_cls, dir_preds ? n_bins : 0, alpha, beta, dir_offset, weights3_host,
                    code_weights7_host);
  if (rc != DM_OK) return rc;
  if (!cls_preds || !box_preds || !labels || !reg_targets || !anchors || !num_pos || !grad_losses3 ||
      !grad_cls || !grad_box || (dir_preds && !grad_dir))
    return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(((long long)batch * n_anchors + kBlock - 1) / kBlock);
  anchor_loss_kernel<true><<<blocks, kBlock, 0, st>>>(cls_preds, box_preds, dir_preds, labels, reg_targets,
                                                      anchors, num_pos, c, grad_losses3, grad_cls, grad_box,
                                                      grad_dir, nullptr);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
