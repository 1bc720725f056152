// Target assignment, random sampling and losses of the 2-D detector (Faster R-CNN) as fused kernels
// (gfx950).  mmdet 2.14 is an un-vendored third-party dependency of the reference; the rules restated
// here are MaxIoUAssigner.assign_wrt_overlaps (gt_max_assign_all), RandomSampler (neg_pos_ub -1),
// DeltaXYWHBBoxCoder.encode, AnchorHead.loss with sampling (RPNHead), BBoxHead.get_targets / loss with
// a sigmoid FocalLoss, as configured at configs/detmatch/001/detmatch/split_0.py:39-99,440-478.
//
//   assign_iou_kernel     per box: max / argmax IoU over the image's GT; per GT: max IoU over the boxes
//   assign_label_kernel   assigned_gt_inds: -1 ignore, 0 negative, k+1 = GT k (incl. the low-quality rule)
//   select_kernel         the `count` smallest random keys among the positives / negatives of an image
//                         (three-pass radix select in LDS + rank sort): a uniform random subset
//   rpn_loss_kernel       BCE over the sampled anchors + L1 over the encoded deltas of the sampled
//                         positives, read from / differentiated into the NHWC head outputs directly
//   roi2d_target_kernel   the sampled RoIs of every image with labels, weights and delta targets
//   bbox_head_loss_kernel sigmoid focal loss + accuracy + class-specific L1 with their gradients
//
// In the reference these are per-image chains of dense tensor operations (RPN: ~140 launches per
// image); here one call is 4-5 launches for the whole batch.  Random keys are inputs (drawn by the
// caller), every reduction has a fixed order.
#include "dm_common.h"

#define DM2D_MAX_IMGS 8
#define DM2D_MAX_GT 256
#define DM2D_MAX_LEVELS 8

namespace {

struct Det2DBatch {
  const float *gt[DM2D_MAX_IMGS];              // (K_i, 4) xyxy
  int K[DM2D_MAX_IMGS];
  const float *boxes[DM2D_MAX_IMGS];           // (P, box_stride) rows, xyxy first
  const unsigned char *valid[DM2D_MAX_IMGS];   // (P) bool or null
  int P, box_stride, prefix_gt;                // prefix_gt: the image's GT boxes come first
};

__device__ __forceinline__ float4 ld4(const float *p) { return make_float4(p[0], p[1], p[2], p[3]); }

__device__ __forceinline__ int n_boxes(const Det2DBatch &d, int b) { return d.P + (d.prefix_gt ? d.K[b] : 0); }

__device__ __forceinline__ float4 load_box(const Det2DBatch &d, int b, int n) {
  const float *p;
  if (d.prefix_gt) {
    p = n < d.K[b] ? d.gt[b] + (size_t)n * 4 : d.boxes[b] + (size_t)(n - d.K[b]) * d.box_stride;
  } else {
    p = d.boxes[b] + (size_t)n * d.box_stride;
  }
  return ld4(p);
}

__device__ __forceinline__ bool box_valid(const Det2DBatch &d, int b, int n) {
  if (d.prefix_gt) {
    if (n < d.K[b]) return true;
    n -= d.K[b];
  }
  return d.valid[b] ? d.valid[b][n] != 0 : true;
}

// mmdet bbox_overlaps(gt, box), mode iou, eps 1e-6
__device__ __forceinline__ float iou_xyxy(float4 g, float4 q) {
  const float a1 = (g.z - g.x) * (g.w - g.y), a2 = (q.z - q.x) * (q.w - q.y);
  const float w = fmaxf(fminf(g.z, q.z) - fmaxf(g.x, q.x), 0.f), h = fmaxf(fminf(g.w, q.w) - fmaxf(g.y, q.y), 0.f);
  const float ov = w * h;
  return ov / fmaxf(a1 + a2 - ov, 1e-6f);
}

__global__ __launch_bounds__(256) void assign_iou_kernel(Det2DBatch d, int n_max, float *__restrict__ max_ov,
                                                         int *__restrict__ arg_ov,
                                                         unsigned *__restrict__ gt_max) {
  __shared__ float4 sg[DM2D_MAX_GT];
  const int b = blockIdx.y, K = d.K[b], N = n_boxes(d, b);
  const int n = blockIdx.x * 256 + threadIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) sg[k] = ld4(d.gt[b] + (size_t)k * 4);
  __syncthreads();
  const bool in = n < N;
  const float4 q = in ? load_box(d, b, n) : make_float4(0, 0, 0, 0);
  float best = -1.f;
  int bi = 0;
  for (int k = 0; k < K; ++k) {
    const float v = in ? iou_xyxy(sg[k], q) : 0.f;
    if (v > best) {
      best = v;
      bi = k;
    }
    float m = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&gt_max[b * DM2D_MAX_GT + k], __float_as_uint(m));
  }
  if (in) {
    max_ov[(size_t)b * n_max + n] = best;
    arg_ov[(size_t)b * n_max + n] = bi;
  }
}

__global__ __launch_bounds__(256) void assign_label_kernel(Det2DBatch d, int n_max, const float *__restrict__ max_ov,
                                                           const int *__restrict__ arg_ov,
                                                           const unsigned *__restrict__ gt_max, float pos_thr,
                                                           float neg_thr, float min_pos, int low_quality,
                                                           int *__restrict__ assigned) {
  __shared__ float4 sg[DM2D_MAX_GT];
  __shared__ float sm[DM2D_MAX_GT];
  const int b = blockIdx.y, K = d.K[b], N = n_boxes(d, b);
  const int n = blockIdx.x * 256 + threadIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) {
    sg[k] = ld4(d.gt[b] + (size_t)k * 4);
    sm[k] = __uint_as_float(gt_max[b * DM2D_MAX_GT + k]);
  }
  __syncthreads();
  if (n >= N) return;
  int a = 0;
  if (K > 0) {
    const float mo = max_ov[(size_t)b * n_max + n];
    a = -1;
    if (mo >= 0.f && mo < neg_thr) a = 0;
    if (mo >= pos_thr) a = arg_ov[(size_t)b * n_max + n] + 1;
    if (low_quality) {
      const float4 q = load_box(d, b, n);
      for (int k = 0; k < K; ++k)
        if (sm[k] >= min_pos && iou_xyxy(sg[k], q) == sm[k]) a = k + 1;  // later GT overwrite
    }
  }
  if (!box_valid(d, b, n)) a = -1;
  assigned[(size_t)b * n_max + n] = a;
}

// ---------------------------------------------------------------------------------------------
// the `want` smallest keys among the candidates of image blockIdx.y (blockIdx.x: 0 = positives
// (assigned > 0), 1 = negatives (assigned == 0)), ascending by (key, index).  Keys are uniform in
// [0, 1): their IEEE bit patterns order like the values.
constexpr int kSelThreads = 1024;
constexpr int kSelMax = 512;

__global__ __launch_bounds__(kSelThreads) void select_kernel(Det2DBatch d, int n_max, const int *__restrict__ assigned,
                                                             const float *__restrict__ keys, int want_pos,
                                                             int want_neg, int *__restrict__ sel_idx,
                                                             int *__restrict__ sel_cnt) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned s_prefix, s_rem, s_done;
  __shared__ unsigned l_key[kSelMax], t_idx[kSelMax];
  __shared__ int l_idx[kSelMax];
  __shared__ unsigned s_nl, s_nt;
  const int b = blockIdx.y, which = blockIdx.x, tid = threadIdx.x;
  const int N = n_boxes(d, b);
  const int want = min(which == 0 ? want_pos : want_neg, N);
  const int *as = assigned + (size_t)b * n_max;
  const float *ky = keys + (size_t)b * n_max;
  int *out = sel_idx + ((size_t)b * 2 + which) * kSelMax;
  auto cand = [&](int n) { return which == 0 ? as[n] > 0 : as[n] == 0; };
  if (tid == 0) {
    s_prefix = 0;
    s_rem = want;
    s_done = 0;
    s_nl = 0;
    s_nt = 0;
  }
  // radix select of the want-th smallest key: 12 + 12 + 8 bits
  const int shifts[3] = {20, 8, 0}, bits[3] = {12, 12, 8};
  unsigned mask_hi = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = shifts[pass], nb = 1 << bits[pass];
    for (int i = tid; i < nb; i += kSelThreads) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    if (!s_done && want > 0)
      for (int n = tid; n < N; n += kSelThreads)
        if (cand(n)) {
          const unsigned kb = __float_as_uint(ky[n]);
          if ((kb & mask_hi) == prefix) atomicAdd(&hist[(kb >> sh) & (nb - 1)], 1u);
        }
    __syncthreads();
    if (tid < 64 && !s_done && want > 0) {  // wave 0: locate the bin that holds the s_rem-th element
      const int per = nb / 64;
      unsigned mine = 0;
      for (int i = 0; i < per; ++i) mine += hist[tid * per + i];
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        unsigned o = __shfl_up(incl, off);
        if (tid >= off) incl += o;
      }
      const unsigned total = __shfl(incl, 63);
      const unsigned rem = s_rem;
      if (pass == 0 && total <= rem) {
        if (tid == 0) s_done = 1;  // fewer candidates than wanted: take them all
      } else {
        const unsigned excl = incl - mine;
        if (excl < rem && rem <= incl) {
          unsigned acc = excl;
          for (int i = 0; i < per; ++i) {
            const unsigned h = hist[tid * per + i];
            if (acc + h >= rem) {
              s_prefix = prefix | ((unsigned)(tid * per + i) << sh);
              s_rem = rem - acc;
              break;
            }
            acc += h;
          }
        }
      }
    }
    mask_hi |= (unsigned)((1 << bits[pass]) - 1) << sh;
    __syncthreads();
  }
  const bool all = s_done != 0;
  const unsigned T = s_prefix;  // the want-th smallest key; s_rem of the keys equal to T are taken
  if (want > 0)
    for (int n = tid; n < N; n += kSelThreads)
      if (cand(n)) {
        const unsigned kb = __float_as_uint(ky[n]);
        if (all || kb < T) {
          const unsigned p = atomicAdd(&s_nl, 1u);
          if (p < kSelMax) {
            l_key[p] = kb;
            l_idx[p] = n;
          }
        } else if (kb == T) {
          const unsigned p = atomicAdd(&s_nt, 1u);
          if (p < kSelMax) t_idx[p] = n;
        }
      }
  __syncthreads();
  unsigned nl = min(s_nl, (unsigned)kSelMax);
  if (!all && want > 0) {  // ties at the threshold: lowest indices first
    const unsigned nt = min(s_nt, (unsigned)kSelMax), take = s_rem;
    for (unsigned i = tid; i < nt; i += kSelThreads) {
      unsigned r = 0;
      for (unsigned j = 0; j < nt; ++j) r += t_idx[j] < t_idx[i];
      if (r < take && nl + r < kSelMax) {
        l_key[nl + r] = T;
        l_idx[nl + r] = (int)t_idx[i];
      }
    }
    nl = min(nl + min(take, nt), (unsigned)kSelMax);
  }
  __syncthreads();
  for (unsigned i = tid; i < kSelMax; i += kSelThreads) {
    if (i < nl) {
      unsigned r = 0;
      for (unsigned j = 0; j < nl; ++j)
        r += (l_key[j] < l_key[i]) | ((l_key[j] == l_key[i]) & (l_idx[j] < l_idx[i]));
      out[r] = l_idx[i];
    }
  }
  __syncthreads();
  for (unsigned i = nl + tid; i < kSelMax; i += kSelThreads) out[i] = 0;
  if (tid == 0) sel_cnt[b * 2 + which] = (int)nl;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_delta(float4 p, float4 g, const float *mean, const float *stdv, float *t) {
  const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f, pw = p.z - p.x, ph = p.w - p.y;
  const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
  t[0] = ((gx - px) / pw - mean[0]) / stdv[0];
  t[1] = ((gy - py) / ph - mean[1]) / stdv[1];
  t[2] = (logf(gw / pw) - mean[2]) / stdv[2];
  t[3] = (logf(gh / ph) - mean[3]) / stdv[3];
}

template <typename T>
__device__ __forceinline__ T block_sum_1024(T v, T *red) {
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  return red[0];
}

struct GtLabels {
  const long long *p[DM2D_MAX_IMGS];
};

struct RpnLevels {
  const float *y[DM2D_MAX_LEVELS];   // (B, H, W, C) head output: A objectness logits, then 4A deltas
  long long grad_off[DM2D_MAX_LEVELS];  // element offset of the level in the flat gradient buffer
  int hw[DM2D_MAX_LEVELS];
  int first[DM2D_MAX_LEVELS + 1];    // first anchor of the level
  int n_levels, A, C;
};

// one block; out = [loss_cls, loss_bbox]; entries: (offset into the flat gradient, kind, value)
__global__ __launch_bounds__(1024) void rpn_loss_kernel(Det2DBatch d, RpnLevels lv, int B, int n_max,
                                                        const int *__restrict__ assigned,
                                                        const int *__restrict__ sel_idx, const int *__restrict__ sel_cnt,
                                                        int want_pos, int want_neg, int num, float w_cls, float w_box,
                                                        float4 mean, float4 stdv, float *__restrict__ out,
                                                        long long *__restrict__ ent_off, float *__restrict__ ent_val) {
  __shared__ double red[1024];
  const int tid = threadIdx.x, per = want_pos + want_neg, total = B * per;
  const float mean_a[4] = {mean.x, mean.y, mean.z, mean.w}, std_a[4] = {stdv.x, stdv.y, stdv.z, stdv.w};
  auto entry_ok = [&](int b, int j, bool &pos, int &n) {
    const int pc = sel_cnt[b * 2], nc = sel_cnt[b * 2 + 1];
    pos = j < want_pos;
    const int jj = pos ? j : j - want_pos;
    n = sel_idx[((size_t)b * 2 + (pos ? 0 : 1)) * kSelMax + min(jj, kSelMax - 1)];
    return pos ? jj < pc : (jj < nc && jj < num - pc);
  };
  // mmdet anchor_head.py get_targets: num_total_samples = sum_i max(#pos_i, 1) + sum_i max(#neg_i, 1)
  int cnt = 0;
  for (int b = 0; b < B; ++b) {
    const int pc = min(sel_cnt[b * 2], want_pos), nc = max(min(min(sel_cnt[b * 2 + 1], want_neg), num - pc), 0);
    cnt += max(pc, 1) + max(nc, 1);
  }
  const float inv = 1.f / fmaxf((float)cnt, 1.f);
  double s_cls = 0.0, s_box = 0.0;
  for (int e = tid; e < total; e += 1024) {
    const int b = e / per, j = e % per;
    bool pos;
    int n;
    const bool ok = entry_ok(b, j, pos, n);
    long long off[5] = {-1, -1, -1, -1, -1};
    float val[5] = {0, 0, 0, 0, 0};
    if (ok) {
      int l = 0;
      while (l + 1 < lv.n_levels && n >= lv.first[l + 1]) ++l;
      const int loc = n - lv.first[l], a = loc % lv.A, cell = loc / lv.A;
      const long long row = ((long long)b * lv.hw[l] + cell) * lv.C;
      const float *y = lv.y[l] + row;
      const float x = y[a], t = pos ? 1.f : 0.f;
      s_cls += (double)(fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
      off[0] = lv.grad_off[l] + row + a;
      val[0] = (1.f / (1.f + expf(-x)) - t) * inv * w_cls;
      if (pos) {
        const int g = assigned[(size_t)b * n_max + n] - 1;
        float tg[4];
        encode_delta(load_box(d, b, n), ld4(d.gt[b] + (size_t)max(g, 0) * 4), mean_a, std_a, tg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float df = y[lv.A + 4 * a + k] - tg[k];
          s_box += (double)fabsf(df);
          off[1 + k] = lv.grad_off[l] + row + lv.A + 4 * a + k;
          val[1 + k] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv * w_box;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      ent_off[(size_t)e * 5 + k] = off[k];
      ent_val[(size_t)e * 5 + k] = val[k];
    }
  }
  s_cls = block_sum_1024(s_cls, red);
  s_box = block_sum_1024(s_box, red);
  if (tid == 0) {
    out[0] = (float)s_cls * inv * w_cls;
    out[1] = (float)s_box * inv * w_box;
  }
}

// grad[off] = val * upstream[kind]  (kind: column 0 of an entry = objectness, 1..4 = deltas)
__global__ __launch_bounds__(256) void rpn_loss_scatter_kernel(const long long *__restrict__ ent_off,
                                                               const float *__restrict__ ent_val,
                                                               const float *__restrict__ upstream, int n,
                                                               float *__restrict__ grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long o = ent_off[i];
  if (o >= 0) grad[o] = ent_val[i] * upstream[(i % 5) == 0 ? 0 : 1];
}

// ---------------------------------------------------------------------------------------------
// BBoxHead.get_targets on the sampled set: `num` rows per image, positives, then negatives, then
// zero rows (weight 0)
__global__ __launch_bounds__(256) void roi2d_target_kernel(Det2DBatch d, int n_max, const int *__restrict__ assigned,
                                                           const int *__restrict__ sel_idx,
                                                           const int *__restrict__ sel_cnt, int num, int n_classes,
                                                           GtLabels gt_labels,
                                                           float4 mean, float4 stdv, float *__restrict__ rois,
                                                           long long *__restrict__ labels, float *__restrict__ label_w,
                                                           float *__restrict__ tgt, float *__restrict__ box_w) {
  const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= num) return;
  const float mean_a[4] = {mean.x, mean.y, mean.z, mean.w}, std_a[4] = {stdv.x, stdv.y, stdv.z, stdv.w};
  const int pc = sel_cnt[b * 2], nc = min(sel_cnt[b * 2 + 1], num - pc);
  const bool pos = r < pc, ok = r < pc + nc;
  const size_t o = (size_t)b * num + r;
  float roi[5] = {0, 0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
  long long lab = n_classes;
  if (ok) {
    const int n = sel_idx[((size_t)b * 2 + (pos ? 0 : 1)) * kSelMax + (pos ? r : r - pc)];
    const float4 q = load_box(d, b, n);
    roi[0] = (float)b;
    roi[1] = q.x;
    roi[2] = q.y;
    roi[3] = q.z;
    roi[4] = q.w;
    if (pos) {
      const int g = max(assigned[(size_t)b * n_max + n] - 1, 0);
      lab = gt_labels.p[b][g];
      encode_delta(q, ld4(d.gt[b] + (size_t)g * 4), mean_a, std_a, t);
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) rois[o * 5 + k] = roi[k];
  labels[o] = lab;
  label_w[o] = ok ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    tgt[o * 4 + k] = t[k];
    box_w[o * 4 + k] = pos ? 1.f : 0.f;
  }
}

// BBoxHead.loss: sigmoid focal loss (weights label_w, / n_valid), accuracy, class-specific L1 over the
// positives / n_valid.  out = [loss_cls, loss_bbox, acc]
__global__ __launch_bounds__(1024) void bbox_head_loss_kernel(const float *__restrict__ cls, const float *__restrict__ box,
                                                              const long long *__restrict__ labels,
                                                              const float *__restrict__ label_w,
                                                              const float *__restrict__ tgt, const float *__restrict__ box_w,
                                                              int M, int C, int n_classes, int agnostic, float alpha,
                                                              float w_cls, float w_box, float *__restrict__ out,
                                                              float *__restrict__ g_cls, float *__restrict__ g_box) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double nv = 0.0;
  for (int m = tid; m < M; m += 1024) nv += label_w[m] > 0.f ? 1.0 : 0.0;
  nv = block_sum_1024(nv, red);
  const float inv = 1.f / fmaxf((float)nv, 1.f);
  double s_cls = 0.0, s_box = 0.0, s_acc = 0.0;
  const int box_c = agnostic ? 1 : n_classes;
  for (int m = tid; m < M; m += 1024) {
    const long long l = labels[m];
    const float lw = label_w[m];
    int arg = 0;
    float best = cls[(size_t)m * C];
    for (int c = 0; c < C; ++c) {
      const float x = cls[(size_t)m * C + c];
      if (x > best) {
        best = x;
        arg = c;
      }
      const float t = l == c ? 1.f : 0.f;
      const float p = 1.f / (1.f + expf(-x));
      const float pt = (1.f - p) * t + p * (1.f - t);
      const float fw = (alpha * t + (1.f - alpha) * (1.f - t)) * pt * pt;
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      s_cls += (double)(bce * fw * lw);
      const float dpt = (t > 0.5f ? -1.f : 1.f) * p * (1.f - p);
      g_cls[(size_t)m * C + c] =
          (alpha * t + (1.f - alpha) * (1.f - t)) * (2.f * pt * dpt * bce + pt * pt * (p - t)) * lw * inv * w_cls;
    }
    if (lw > 0.f && arg == l) s_acc += 1.0;
    const bool pos = l >= 0 && l < n_classes && lw > 0.f;
    const int sel = agnostic ? 0 : (int)(l < n_classes - 1 ? (l < 0 ? 0 : l) : n_classes - 1);
    for (int c = 0; c < box_c; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float g = 0.f;
        if (c == sel) {
          const float bw = box_w[(size_t)m * 4 + k] * (pos ? 1.f : 0.f);
          const float df = box[((size_t)m * box_c + c) * 4 + k] - tgt[(size_t)m * 4 + k];
          s_box += (double)(fabsf(df) * bw);
          g = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * bw * inv * w_box;
        }
        g_box[((size_t)m * box_c + c) * 4 + k] = g;
      }
  }
  s_cls = block_sum_1024(s_cls, red);
  s_box = block_sum_1024(s_box, red);
  s_acc = block_sum_1024(s_acc, red);
  if (tid == 0) {
    out[0] = (float)s_cls * inv * w_cls;
    out[1] = (float)s_box * inv * w_box;
    out[2] = (float)s_acc * 100.f * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// RPNHead._get_bboxes_single up to the NMS, for all images and levels at once: per (image, level) the
// nms_pre highest objectness scores (all anchors of the level if it has no more than that; ties go to
// the lower anchor index), decoded and clipped boxes, "large enough" flags, and the largest coordinate
// of the image (batched_nms separates the levels by multiples of max + 1).
constexpr int kTopMax = 2048;

struct RpnProposalCfg {
  int nms_pre, T;                       // T = sum over levels of min(nms_pre, level size)
  int base[DM2D_MAX_LEVELS];            // first output slot of the level
  float img_h[DM2D_MAX_IMGS], img_w[DM2D_MAX_IMGS];
  float mean[4], stdv[4], max_ratio, min_size;
  int clip;
};

__device__ __forceinline__ int ordered_int(float v) {
  const int b = __float_as_int(v);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}

// Selection keys of every anchor of every image, once, on the whole chip: the top-k kernel below walks the
// anchors of a level four times on ONE CU (three radix passes + the compaction), and the sigmoid + the
// (location, anchor) index split per visit made that walk instruction bound (265 us for the 92 k anchors of
// the finest level).
__global__ __launch_bounds__(256) void rpn_keys_kernel(RpnLevels lv, int n_anchors, unsigned *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n_anchors) return;
  int l = 0;
  while (l + 1 < lv.n_levels && i >= lv.first[l + 1]) ++l;
  const int loc = i - lv.first[l];
  const float x = lv.y[l][((size_t)b * lv.hw[l] + loc / lv.A) * lv.C + (loc % lv.A)];
  keys[(size_t)b * n_anchors + i] = 0x7FFFFFFFu - __float_as_uint(1.f / (1.f + expf(-x)));
}

__global__ __launch_bounds__(1024) void rpn_topk_decode_kernel(RpnLevels lv, RpnProposalCfg c,
                                                               const unsigned *__restrict__ keys, int n_anchors,
                                                               const float *__restrict__ anchors,
                                                               float *__restrict__ boxes, float *__restrict__ scores,
                                                               int *__restrict__ level_of, unsigned char *__restrict__ live,
                                                               int *__restrict__ coord_max) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned s_prefix, s_rem, s_nl, s_nt;
  __shared__ unsigned l_key[kTopMax], t_idx[kTopMax];
  __shared__ int l_idx[kTopMax];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int A = lv.A, C = lv.C, N = lv.hw[l] * A;
  const float *y = lv.y[l] + (size_t)b * lv.hw[l] * C;
  auto score_of = [&](int loc) {
    const float x = y[(size_t)(loc / A) * C + (loc % A)];
    return 1.f / (1.f + expf(-x));
  };
  // descending order of positive floats = ascending order of the complemented bit pattern
  const unsigned *kl = keys ? keys + (size_t)b * n_anchors + lv.first[l] : nullptr;
  auto key_of = [&](int loc) { return kl ? kl[loc] : 0x7FFFFFFFu - __float_as_uint(score_of(loc)); };
  const int want = min(c.nms_pre > 0 ? c.nms_pre : N, N);
  const bool need_select = want < N;
  if (tid == 0) {
    s_prefix = 0;
    s_rem = want;
    s_nl = 0;
    s_nt = 0;
  }
  __syncthreads();
  unsigned nl = 0;
  if (need_select) {
    const int shifts[3] = {20, 8, 0}, bits[3] = {12, 12, 8};
    unsigned mask_hi = 0;
    for (int pass = 0; pass < 3; ++pass) {
      const int sh = shifts[pass], nb = 1 << bits[pass];
      for (int i = tid; i < nb; i += 1024) hist[i] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix;
      for (int n = tid; n < N; n += 1024) {
        const unsigned kb = key_of(n);
        if ((kb & mask_hi) == prefix) atomicAdd(&hist[(kb >> sh) & (nb - 1)], 1u);
      }
      __syncthreads();
      if (tid < 64) {
        const int per = nb / 64;
        unsigned mine = 0;
        for (int i = 0; i < per; ++i) mine += hist[tid * per + i];
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          unsigned o = __shfl_up(incl, off);
          if (tid >= off) incl += o;
        }
        const unsigned rem = s_rem, excl = incl - mine;
        if (excl < rem && rem <= incl) {
          unsigned acc = excl;
          for (int i = 0; i < per; ++i) {
            const unsigned h = hist[tid * per + i];
            if (acc + h >= rem) {
              s_prefix = prefix | ((unsigned)(tid * per + i) << sh);
              s_rem = rem - acc;
              break;
            }
            acc += h;
          }
        }
      }
      mask_hi |= (unsigned)((1 << bits[pass]) - 1) << sh;
      __syncthreads();
    }
    const unsigned T = s_prefix;
    for (int n = tid; n < N; n += 1024) {
      const unsigned kb = key_of(n);
      if (kb < T) {
        const unsigned p = atomicAdd(&s_nl, 1u);
        if (p < kTopMax) {
          l_key[p] = kb;
          l_idx[p] = n;
        }
      } else if (kb == T) {
        const unsigned p = atomicAdd(&s_nt, 1u);
        if (p < kTopMax) t_idx[p] = n;
      }
    }
    __syncthreads();
    nl = min(s_nl, (unsigned)kTopMax);
    const unsigned nt = min(s_nt, (unsigned)kTopMax), take = s_rem;
    for (unsigned i = tid; i < nt; i += 1024) {
      unsigned r = 0;
      for (unsigned j = 0; j < nt; ++j) r += t_idx[j] < t_idx[i];
      if (r < take && nl + r < kTopMax) {
        l_key[nl + r] = T;
        l_idx[nl + r] = (int)t_idx[i];
      }
    }
    nl = min(nl + min(take, nt), (unsigned)kTopMax);
    __syncthreads();
  }
  int cmax = ordered_int(-3.0e38f);
  for (int i = tid; i < want; i += 1024) {
    int loc, r;
    if (need_select) {
      if ((unsigned)i >= nl) continue;
      r = 0;
      for (unsigned j = 0; j < nl; ++j) r += (l_key[j] < l_key[i]) | ((l_key[j] == l_key[i]) & (l_idx[j] < l_idx[i]));
      loc = l_idx[i];
    } else {
      loc = r = i;
    }
    const int a = loc % A, cell = loc / A;
    const float *row = y + (size_t)cell * C;
    const float sc = 1.f / (1.f + expf(-row[a]));
    const float *an = anchors + (size_t)(lv.first[l] + loc) * 4;
    const float dx = row[A + 4 * a] * c.stdv[0] + c.mean[0], dy = row[A + 4 * a + 1] * c.stdv[1] + c.mean[1];
    float dw = row[A + 4 * a + 2] * c.stdv[2] + c.mean[2], dh = row[A + 4 * a + 3] * c.stdv[3] + c.mean[3];
    dw = fminf(fmaxf(dw, -c.max_ratio), c.max_ratio);
    dh = fminf(fmaxf(dh, -c.max_ratio), c.max_ratio);
    const float px = (an[0] + an[2]) * 0.5f, py = (an[1] + an[3]) * 0.5f, pw = an[2] - an[0], ph = an[3] - an[1];
    const float gw = pw * expf(dw), gh = ph * expf(dh), gx = px + pw * dx, gy = py + ph * dy;
    float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
    if (c.clip) {
      x1 = fminf(fmaxf(x1, 0.f), c.img_w[b]);
      x2 = fminf(fmaxf(x2, 0.f), c.img_w[b]);
      y1 = fminf(fmaxf(y1, 0.f), c.img_h[b]);
      y2 = fminf(fmaxf(y2, 0.f), c.img_h[b]);
    }
    const size_t o = (size_t)b * c.T + c.base[l] + r;
    boxes[o * 4 + 0] = x1;
    boxes[o * 4 + 1] = y1;
    boxes[o * 4 + 2] = x2;
    boxes[o * 4 + 3] = y2;
    scores[o] = sc;
    level_of[o] = l;
    live[o] = c.min_size >= 0.f ? ((x2 - x1) > c.min_size && (y2 - y1) > c.min_size) : 1;
    cmax = max(cmax, max(max(ordered_int(x1), ordered_int(y1)), max(ordered_int(x2), ordered_int(y2))));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
  if ((tid & 63) == 0) atomicMax(&coord_max[b], cmax);
}

// batched_nms inputs: boxes shifted by level * (max coordinate + 1); dropped boxes far away, score -1
__global__ __launch_bounds__(256) void rpn_nms_prep_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                           const int *__restrict__ level_of,
                                                           const unsigned char *__restrict__ live,
                                                           const int *__restrict__ coord_max, int T,
                                                           float *__restrict__ b_nms, float *__restrict__ s_nms) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T) return;
  const int om = coord_max[b];
  const float mx = __int_as_float(om ^ ((om >> 31) & 0x7FFFFFFF));
  const size_t o = (size_t)b * T + i;
  const float off = (float)level_of[o] * (mx + 1.f);
  const bool lv = live[o] != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) b_nms[o * 4 + k] = lv ? boxes[o * 4 + k] + off : -1e6f;
  s_nms[o] = lv ? scores[o] : -1.f;
}

int fill_batch(Det2DBatch &d, int batch, const float *const *gt, const int *n_gt, const float *const *boxes,
               const unsigned char *const *valid, int shared_boxes, int P, int box_stride, int prefix_gt) {
  if (batch < 1 || batch > DM2D_MAX_IMGS || P < 0 || box_stride < 4 || !gt || !n_gt || !boxes) return DM_ERR_INVALID_ARG;
  for (int b = 0; b < batch; ++b) {
    if (n_gt[b] < 0 || n_gt[b] > DM2D_MAX_GT || (n_gt[b] > 0 && !gt[b])) return DM_ERR_UNSUPPORTED;
    d.gt[b] = gt[b];
    d.K[b] = n_gt[b];
    d.boxes[b] = boxes[shared_boxes ? 0 : b];
    d.valid[b] = valid ? valid[b] : nullptr;
    if (!d.boxes[b] && P > 0) return DM_ERR_INVALID_ARG;
  }
  d.P = P;
  d.box_stride = box_stride;
  d.prefix_gt = prefix_gt;
  return DM_OK;
}

struct AssignWs {
  float *max_ov;
  int *arg_ov, *assigned, *sel_idx, *sel_cnt;
  unsigned *gt_max;
};

size_t assign_ws_bytes(int batch, int n_max) {
  return 3 * dm_align((size_t)batch * n_max * 4) + dm_align((size_t)batch * 2 * kSelMax * 4) +
         dm_align((size_t)batch * 2 * 4) + dm_align((size_t)batch * DM2D_MAX_GT * 4);
}

// the three stages shared by the RPN and the RoI head
int assign_and_sample(const Det2DBatch &d, int batch, int n_max, const float *keys, float pos_thr, float neg_thr,
                      float min_pos, int low_quality, int want_pos, int want_neg, DmArena &arena, AssignWs &w,
                      hipStream_t st) {
  w.max_ov = arena.take<float>((size_t)batch * n_max);
  w.arg_ov = arena.take<int>((size_t)batch * n_max);
  w.assigned = arena.take<int>((size_t)batch * n_max);
  w.sel_idx = arena.take<int>((size_t)batch * 2 * kSelMax);
  w.sel_cnt = arena.take<int>((size_t)batch * 2);
  w.gt_max = arena.take<unsigned>((size_t)batch * DM2D_MAX_GT);
  if (!arena.ok()) return DM_ERR_WORKSPACE;
  if (want_pos > kSelMax || want_neg > kSelMax) return DM_ERR_UNSUPPORTED;
  DM_HIP(hipMemsetAsync(w.gt_max, 0, (size_t)batch * DM2D_MAX_GT * 4, st));
  dim3 grid(dm_ceil_div(n_max, 256), batch);
  assign_iou_kernel<<<grid, 256, 0, st>>>(d, n_max, w.max_ov, w.arg_ov, w.gt_max);
  DM_CHECK_LAUNCH();
  assign_label_kernel<<<grid, 256, 0, st>>>(d, n_max, w.max_ov, w.arg_ov, w.gt_max, pos_thr, neg_thr, min_pos,
                                            low_quality, w.assigned);
  DM_CHECK_LAUNCH();
  select_kernel<<<dim3(2, batch), kSelThreads, 0, st>>>(d, n_max, w.assigned, keys, want_pos, want_neg, w.sel_idx,
                                                        w.sel_cnt);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

}  // namespace

extern "C" size_t dm_det2d_assign_workspace_bytes(int batch, int n_boxes_max) {
  if (batch <= 0 || n_boxes_max <= 0) return 0;
  return assign_ws_bytes(batch, n_boxes_max);
}

extern "C" int dm_rpn_loss_forward(const float *const *level_outputs, const int *level_hw, int n_levels,
                                   int n_base_anchors, int channels, const long long *grad_offsets,
                                   const float *anchors, int n_anchors, const float *const *gt_boxes,
                                   const int *n_gt, int batch, const float *keys, float pos_iou_thr,
                                   float neg_iou_thr, float min_pos_iou, int match_low_quality, int num,
                                   int num_pos_max, const float *means4, const float *stds4,
                                   float loss_cls_weight, float loss_bbox_weight, float *out2,
                                   long long *entry_offsets, float *entry_values, int *assigned_out,
                                   void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!level_outputs || !level_hw || !grad_offsets || !anchors || !keys || !means4 || !stds4 || !out2 ||
      !entry_offsets || !entry_values || !workspace)
    return DM_ERR_INVALID_ARG;
  if (n_levels < 1 || n_levels > DM2D_MAX_LEVELS || n_anchors <= 0 || num <= 0) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_det2d_assign_workspace_bytes(batch, n_anchors)) return DM_ERR_WORKSPACE;
  Det2DBatch d;
  const float *boxes[1] = {anchors};
  int rc = fill_batch(d, batch, gt_boxes, n_gt, boxes, nullptr, 1, n_anchors, 4, 0);
  if (rc != DM_OK) return rc;
  RpnLevels lv;
  lv.n_levels = n_levels;
  lv.A = n_base_anchors;
  lv.C = channels;
  int first = 0;
  for (int l = 0; l < n_levels; ++l) {
    lv.y[l] = level_outputs[l];
    lv.hw[l] = level_hw[l];
    lv.grad_off[l] = grad_offsets[l];
    lv.first[l] = first;
    first += level_hw[l] * n_base_anchors;
  }
  lv.first[n_levels] = first;
  if (first != n_anchors || channels < 5 * n_base_anchors) return DM_ERR_INVALID_ARG;
  const int want_pos = num_pos_max, want_neg = num;
  if (want_pos < 0 || want_pos > num) return DM_ERR_INVALID_ARG;
  DmArena arena(workspace, workspace_bytes);
  AssignWs w;
  rc = assign_and_sample(d, batch, n_anchors, keys, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality,
                         want_pos, want_neg, arena, w, st);
  if (rc != DM_OK) return rc;
  rpn_loss_kernel<<<1, 1024, 0, st>>>(d, lv, batch, n_anchors, w.assigned, w.sel_idx, w.sel_cnt, want_pos, want_neg,
                                      num, loss_cls_weight, loss_bbox_weight,
                                      make_float4(means4[0], means4[1], means4[2], means4[3]),
                                      make_float4(stds4[0], stds4[1], stds4[2], stds4[3]), out2, entry_offsets,
                                      entry_values);
  DM_CHECK_LAUNCH();
  if (assigned_out)
    DM_HIP(hipMemcpyAsync(assigned_out, w.assigned, (size_t)batch * n_anchors * 4, hipMemcpyDeviceToDevice, st));
  return DM_OK;
}

extern "C" int dm_rpn_loss_backward(const long long *entry_offsets, const float *entry_values,
                                    const float *upstream2, int n_entries, float *grad_flat,
                                    dm_stream_t stream) {
  if (n_entries <= 0) return DM_OK;
  if (!entry_offsets || !entry_values || !upstream2 || !grad_flat) return DM_ERR_INVALID_ARG;
  rpn_loss_scatter_kernel<<<dm_ceil_div(n_entries, 256), 256, 0, (hipStream_t)stream>>>(
      entry_offsets, entry_values, upstream2, n_entries, grad_flat);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_roi2d_targets(const float *const *proposals, const unsigned char *const *proposal_ok,
                                int n_proposals, int proposal_stride, const float *const *gt_boxes,
                                const long long *const *gt_labels, const int *n_gt, int batch,
                                int add_gt_as_proposals, const float *keys, int keys_stride, float pos_iou_thr,
                                float neg_iou_thr, float min_pos_iou, int match_low_quality, int num,
                                int num_pos_max, int n_classes, const float *means4, const float *stds4,
                                float *rois, long long *labels, float *label_weights, float *bbox_targets,
                                float *bbox_weights, void *workspace, size_t workspace_bytes,
                                dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!proposals || !gt_boxes || !gt_labels || !n_gt || !keys || !means4 || !stds4 || !rois || !labels ||
      !label_weights || !bbox_targets || !bbox_weights || !workspace || num <= 0)
    return DM_ERR_INVALID_ARG;
  Det2DBatch d;
  int rc = fill_batch(d, batch, gt_boxes, n_gt, proposals, proposal_ok, 0, n_proposals, proposal_stride,
                      add_gt_as_proposals);
  if (rc != DM_OK) return rc;
  const int n_max = keys_stride;
  for (int b = 0; b < batch; ++b)
    if (n_proposals + (add_gt_as_proposals ? n_gt[b] : 0) > n_max) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_det2d_assign_workspace_bytes(batch, n_max)) return DM_ERR_WORKSPACE;
  const int want_pos = num_pos_max, want_neg = num;  // clamped to the image's box count in the kernel
  if (want_pos < 0 || want_pos > num) return DM_ERR_INVALID_ARG;
  DmArena arena(workspace, workspace_bytes);
  AssignWs w;
  rc = assign_and_sample(d, batch, n_max, keys, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality, want_pos,
                         want_neg, arena, w, st);
  if (rc != DM_OK) return rc;
  GtLabels lab_dev;
  for (int b = 0; b < batch; ++b) {
    if (n_gt[b] > 0 && !gt_labels[b]) return DM_ERR_INVALID_ARG;
    lab_dev.p[b] = gt_labels[b];
  }
  roi2d_target_kernel<<<dim3(dm_ceil_div(num, 256), batch), 256, 0, st>>>(
      d, n_max, w.assigned, w.sel_idx, w.sel_cnt, num, n_classes, lab_dev,
      make_float4(means4[0], means4[1], means4[2], means4[3]), make_float4(stds4[0], stds4[1], stds4[2], stds4[3]),
      rois, labels, label_weights, bbox_targets, bbox_weights);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bbox_head_loss(const float *cls_score, const float *bbox_pred, const long long *labels,
                                 const float *label_weights, const float *bbox_targets,
                                 const float *bbox_weights, int n_rows, int n_cls_out, int n_classes,
                                 int reg_class_agnostic, float focal_alpha, float loss_cls_weight,
                                 float loss_bbox_weight, float *out3, float *grad_cls, float *grad_bbox,
                                 dm_stream_t stream) {
  if (n_rows <= 0 || n_cls_out <= 0 || n_classes <= 0) return DM_ERR_INVALID_ARG;
  if (!cls_score || !bbox_pred || !labels || !label_weights || !bbox_targets || !bbox_weights || !out3 ||
      !grad_cls || !grad_bbox)
    return DM_ERR_INVALID_ARG;
  bbox_head_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(cls_score, bbox_pred, labels, label_weights,
                                                             bbox_targets, bbox_weights, n_rows, n_cls_out, n_classes,
                                                             reg_class_agnostic, focal_alpha, loss_cls_weight,
                                                             loss_bbox_weight, out3, grad_cls, grad_bbox);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_rpn_proposals_pre_nms(const float *const *level_outputs, const int *level_hw, int n_levels,
                                        int n_base_anchors, int channels, const float *anchors, int n_anchors,
                                        int batch, const float *img_hw, int nms_pre, const float *means4,
                                        const float *stds4, float wh_ratio_clip_log, int clip_border,
                                        float min_bbox_size, int n_out, float *boxes, float *scores,
                                        unsigned char *live, float *nms_boxes, float *nms_scores,
                                        void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!level_outputs || !level_hw || !anchors || !img_hw || !means4 || !stds4 || !boxes || !scores || !live ||
      !nms_boxes || !nms_scores || !workspace)
    return DM_ERR_INVALID_ARG;
  if (n_levels < 1 || n_levels > DM2D_MAX_LEVELS || batch < 1 || batch > DM2D_MAX_IMGS) return DM_ERR_INVALID_ARG;
  if (nms_pre > kTopMax) return DM_ERR_UNSUPPORTED;
  RpnLevels lv;
  RpnProposalCfg c;
  lv.n_levels = n_levels;
  lv.A = n_base_anchors;
  lv.C = channels;
  int first = 0, T = 0;
  for (int l = 0; l < n_levels; ++l) {
    lv.y[l] = level_outputs[l];
    lv.hw[l] = level_hw[l];
    lv.grad_off[l] = 0;
    lv.first[l] = first;
    const int n = level_hw[l] * n_base_anchors;
    first += n;
    c.base[l] = T;
    T += (nms_pre > 0 && n > nms_pre) ? nms_pre : n;
  }
  lv.first[n_levels] = first;
  if (first != n_anchors || T != n_out || channels < 5 * n_base_anchors) return DM_ERR_INVALID_ARG;
  c.nms_pre = nms_pre;
  c.T = T;
  for (int b = 0; b < batch; ++b) {
    c.img_h[b] = img_hw[2 * b];
    c.img_w[b] = img_hw[2 * b + 1];
  }
  for (int k = 0; k < 4; ++k) {
    c.mean[k] = means4[k];
    c.stdv[k] = stds4[k];
  }
  c.max_ratio = wh_ratio_clip_log;
  c.min_size = min_bbox_size;
  c.clip = clip_border;
  DmArena arena(workspace, workspace_bytes);
  int *level_of = arena.take<int>((size_t)batch * T);
  int *coord_max = arena.take<int>(DM2D_MAX_IMGS);
  if (!arena.ok()) return DM_ERR_WORKSPACE;
  DM_HIP(hipMemsetAsync(coord_max, 0x80, DM2D_MAX_IMGS * sizeof(int), st));   // very negative ordered ints
  // optional: room for one key per anchor and image behind the mandatory part of the workspace
  unsigned *keys = arena.take<unsigned>((size_t)batch * n_anchors);
  if (!arena.ok()) keys = nullptr;
  if (keys) {
    rpn_keys_kernel<<<dim3(dm_ceil_div(n_anchors, 256), batch), 256, 0, st>>>(lv, n_anchors, keys);
    DM_CHECK_LAUNCH();
  }
  rpn_topk_decode_kernel<<<dim3(n_levels, batch), 1024, 0, st>>>(lv, c, keys, n_anchors, anchors, boxes, scores,
                                                                 level_of, live, coord_max);
  DM_CHECK_LAUNCH();
  rpn_nms_prep_kernel<<<dim3(dm_ceil_div(T, 256), batch), 256, 0, st>>>(boxes, scores, level_of, live, coord_max, T,
                                                                        nms_boxes, nms_scores);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" size_t dm_rpn_proposals_workspace_bytes(int batch, int n_out) {
  if (batch <= 0 || n_out <= 0) return 0;
  return dm_align((size_t)batch * n_out * sizeof(int)) + dm_align(DM2D_MAX_IMGS * sizeof(int));
}
