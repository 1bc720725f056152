// Second-stage target assignment and losses of PV-RCNN as fused kernels (gfx950).
//
// Replaces, for a whole batch per launch,
//   * ProposalTargetLayer.forward / sample_rois_for_rcnn / get_max_iou_with_same_class / subsample_rois
//     (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:13-259) together with the
//     canonical transform of RoIHeadTemplate.assign_targets (roi_head_template.py:104-134):
//     roi_gt_match_kernel + roi_sample_kernel;
//   * get_box_cls_layer_loss + get_box_reg_layer_loss incl. the corner regularisation
//     (roi_head_template.py:136-218, loss_utils.py:209-233, box_coder_utils.py:16-77), forward and the
//     gradients w.r.t. rcnn_cls / rcnn_reg: rcnn_loss_kernel;
//   * PointHeadSimple.assign_targets / assign_stack_targets (point_head_simple.py:20-48,
//     point_head_template.py:49-129) and the focal loss of get_cls_layer_loss (:131-154):
//     point_targets_kernel + point_focal_kernel.
// Each of these is a chain of 35-230 element-wise torch launches on a few hundred rows in the
// reference; here every chain is one or two launches, with fixed-order reductions (results do not
// depend on the launch configuration).  The random numbers of the sampler are inputs (drawn by the
// caller with the framework's generator), so the kernels are deterministic functions.
#include "box_geom.h"

namespace {

constexpr int kThreads = 256;
constexpr int kGtChunk = 128;

struct RoiTargetCfg {
  int roi_per_image, fg_per_image;
  float reg_fg, cls_fg, cls_bg, cls_bg_lo, hard_bg_ratio, fg_thresh, cls_span;
};

// python's float % for a positive period (torch.remainder)
__device__ __forceinline__ float py_mod(float a, float period) {
  float m = fmodf(a, period);
  if (m != 0.f && m < 0.f) m += period;
  return m;
}

// ---------------------------------------------------------------------------------------------
// max IoU3D of every RoI with the ground-truth boxes of its own class (:217-259): 16 RoIs x 16 lanes
// per block, the lanes of a RoI stride over the ground truth; ties go to the lower index.
__global__ __launch_bounds__(kThreads) void roi_gt_match_kernel(
    const float *__restrict__ rois, const long long *__restrict__ roi_labels, const float *__restrict__ gt,
    int R, int G, int gtc, float *__restrict__ max_iou, int *__restrict__ gt_idx) {
  __shared__ float sg[kGtChunk * 8];
  __shared__ float2 scs[kGtChunk];
  __shared__ int s_last;
  const int b = blockIdx.y, tid = threadIdx.x;
  const float *gb = gt + (size_t)b * G * gtc;
  if (tid == 0) s_last = 0;
  __syncthreads();
  {  // rows after the last non-zero row are padding (:101-104); row 0 always counts
    int last = 0;
    for (int g = tid; g < G; g += kThreads) {
      float s = 0.f;
      for (int c = 0; c < gtc; ++c) s += gb[(size_t)g * gtc + c];
      if (s != 0.f) last = g;
    }
    if (last > 0) atomicMax(&s_last, last);
  }
  __syncthreads();
  const int last = s_last;
  const int ri = blockIdx.x * 16 + (tid >> 4), lane = tid & 15;
  const int rc = min(ri, R - 1);
  float a[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    float v = rois[((size_t)b * R + rc) * 7 + c];
    a[c] = (v != v) ? 0.f : v;  // :109-112
  }
  const long long label = roi_labels[(size_t)b * R + rc];
  const float2 csa = make_float2((float)cos((double)a[6]), (float)sin((double)a[6]));
  const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2, vol_a = a[3] * a[4] * a[5];
  float best = -1.f;
  int bidx = 0;
  for (int g0 = 0; g0 < G; g0 += kGtChunk) {
    const int ng = min(kGtChunk, G - g0);
    __syncthreads();
    for (int e = tid; e < ng * 8; e += kThreads) {
      int g = e >> 3, c = e & 7;
      sg[e] = c < gtc ? gb[(size_t)(g0 + g) * gtc + (c < 7 ? c : gtc - 1)] : 0.f;
    }
    for (int g = tid; g < ng; g += kThreads) {
      double h = (double)gb[(size_t)(g0 + g) * gtc + 6];
      scs[g] = make_float2((float)cos(h), (float)sin(h));
    }
    __syncthreads();
    for (int g = lane; g < ng; g += 16) {
      if (g0 + g > last) continue;
      const float *q = sg + g * 8;
      if ((long long)q[7] != label) continue;
      float ov = box_overlap(a, csa, q, scs[g]);
      float b_max = q[2] + q[5] / 2, b_min = q[2] - q[5] / 2;
      float oh = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
      float o3 = ov * oh;
      float vol_b = q[3] * q[4] * q[5];
      float iou = o3 / fmaxf(vol_a + vol_b - o3, 1e-6f);
      if (iou > best) {
        best = iou;
        bidx = g0 + g;
      }
    }
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    float ob = __shfl_xor(best, off, 16);
    int oi = __shfl_xor(bidx, off, 16);
    if (ob > best || (ob == best && oi < bidx)) {
      best = ob;
      bidx = oi;
    }
  }
  if (lane == 0 && ri < R) {
    bool has = best >= 0.f;
    max_iou[(size_t)b * R + ri] = has ? best : 0.f;
    gt_idx[(size_t)b * R + ri] = has ? bidx : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// fg / bg sampling (:136-215), gathers, classification / regression flags (:13-67) and the canonical
// transform (roi_head_template.py:104-134); one block per sample.
__global__ __launch_bounds__(kThreads) void roi_sample_kernel(
    const float *__restrict__ rois, const float *__restrict__ roi_scores,
    const long long *__restrict__ roi_labels, const float *__restrict__ gt, const float *__restrict__ max_iou,
    const int *__restrict__ gt_idx, const float *__restrict__ u_perm, const float *__restrict__ u_pick,
    RoiTargetCfg cfg, int R, int G, int gtc, float *__restrict__ o_rois, float *__restrict__ o_gt_src,
    float *__restrict__ o_gt_ct, float *__restrict__ o_iou, float *__restrict__ o_score,
    long long *__restrict__ o_label, long long *__restrict__ o_reg_valid, float *__restrict__ o_cls_label,
    long long *__restrict__ o_sampled, float *__restrict__ o_ok) {
  extern __shared__ int smem[];
  int *lists[3] = {smem, smem + R, smem + 2 * R};
  int *perm = smem + 3 * R;
  float *keyf = (float *)(smem + 4 * R);
  __shared__ int wave_cnt[3][kThreads / 64];
  __shared__ int base[3];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float *mo = max_iou + (size_t)b * R;
  if (tid < 3) base[tid] = 0;
  __syncthreads();
  for (int r0 = 0; r0 < R; r0 += kThreads) {
    const int r = r0 + tid;
    const bool in = r < R;
    const float m = in ? mo[r] : 0.f;
    bool f[3];
    f[0] = in & (m >= cfg.fg_thresh);
    f[1] = in & (m < cfg.cls_bg_lo);
    f[2] = in & (m < cfg.reg_fg) & (m >= cfg.cls_bg_lo);
    int prefix[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      unsigned long long mask = __ballot(f[k]);
      prefix[k] = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[k][wave] = __popcll(mask);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int off = base[k];
      for (int w = 0; w < wave; ++w) off += wave_cnt[k][w];
      if (f[k]) lists[k][off + prefix[k]] = r;
    }
    __syncthreads();
    if (tid < 3) {
      int t = 0;
      for (int w = 0; w < kThreads / 64; ++w) t += wave_cnt[tid][w];
      base[tid] += t;
    }
    __syncthreads();
  }
  const int n_fg = base[0], n_easy = base[1], n_hard = base[2], n_bg = n_easy + n_hard;
  // random permutation of the foreground members (:153): order by the caller's uniform keys
  for (int i = tid; i < n_fg; i += kThreads) keyf[i] = u_perm[(size_t)b * R + lists[0][i]];
  __syncthreads();
  for (int i = tid; i < n_fg; i += kThreads) {
    const float ki = keyf[i];
    int rank = 0;
    for (int j = 0; j < n_fg; ++j) rank += (keyf[j] < ki) | ((keyf[j] == ki) & (j < i));
    perm[rank] = lists[0][i];
  }
  __syncthreads();
  const int S = cfg.roi_per_image;
  int fg_this = min(n_fg, cfg.fg_per_image);
  if (n_bg <= 0) fg_this = S;   // :167-172
  if (n_fg <= 0) fg_this = 0;   // :174-179
  const int bg_this = S - fg_this;
  int hard_num = min((int)((float)bg_this * cfg.hard_bg_ratio), n_hard);
  if (n_easy <= 0) hard_num = bg_this;  // :203-207
  if (n_hard <= 0) hard_num = 0;        // :208-212
  const bool ok = (n_fg + n_bg) > 0;
  if (tid == 0) o_ok[b] = ok ? 1.f : 0.f;
  const float two_pi = (float)(2.0 * M_PI), pi = (float)M_PI, half_pi = (float)(M_PI * 0.5),
              pi15 = (float)(M_PI * 1.5);
  for (int s = tid; s < S; s += kThreads) {
    const float u = u_pick[(size_t)b * S + s];
    auto pick = [&](const int *lst, int n) {
      if (n <= 0) return 0;
      long long i = (long long)(u * (float)n);
      return lst[i < n - 1 ? (int)i : n - 1];
    };
    int src;
    if (s < fg_this) {
      src = n_bg > 0 ? perm[min(s, R - 1)] : pick(lists[0], n_fg);
    } else {
      src = (s - fg_this) < hard_num ? pick(lists[2], n_hard) : pick(lists[1], n_easy);
    }
    const size_t o = (size_t)b * S + s;
    o_sampled[o] = src;
    float roi[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      float v = rois[((size_t)b * R + src) * 7 + c];
      v = (v != v) ? 0.f : v;
      roi[c] = ok ? v : 0.f;
      o_rois[o * 7 + c] = roi[c];
    }
    const float iou = ok ? mo[src] : 0.f;
    o_iou[o] = iou;
    o_score[o] = ok ? roi_scores[(size_t)b * R + src] : 0.f;
    o_label[o] = ok ? roi_labels[(size_t)b * R + src] : 0;
    o_reg_valid[o] = iou > cfg.reg_fg ? 1 : 0;
    const bool is_fg = iou > cfg.cls_fg, is_bg = iou < cfg.cls_bg;
    o_cls_label[o] = (!is_fg && !is_bg) ? (iou - cfg.cls_bg) / cfg.cls_span : (is_fg ? 1.f : 0.f);
    const int ga = gt_idx[(size_t)b * R + src];
    const float *g = gt + ((size_t)b * G + ga) * gtc;
    float gv[7];
    for (int c = 0; c < gtc; ++c) {
      float v = ok ? g[c] : 0.f;
      o_gt_src[o * gtc + c] = v;
      if (c < 7) gv[c] = v;
      else o_gt_ct[o * gtc + c] = v;
    }
    // canonical frame of the RoI (roi_head_template.py:112-133)
    const float ry = py_mod(roi[6], two_pi);
    const float x = gv[0] - roi[0], y = gv[1] - roi[1], z = gv[2] - roi[2];
    const float ca = cosf(-ry), sa = sinf(-ry);
    float h = py_mod(gv[6] - ry, two_pi);
    if (h > half_pi && h < pi15) h = py_mod(h + pi, two_pi);
    if (h > pi) h -= two_pi;
    h = fminf(fmaxf(h, -half_pi), half_pi);
    float *oc = o_gt_ct + o * gtc;
    oc[0] = x * ca - y * sa;
    oc[1] = x * sa + y * ca;
    oc[2] = z;
    oc[3] = gv[3];
    oc[4] = gv[4];
    oc[5] = gv[5];
    oc[6] = h;
  }
}

// ---------------------------------------------------------------------------------------------
struct RcnnLossCfg {
  float w_cls, w_reg, w_corner, beta;
  float cw[7];
  int corner;
};

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *red) {  // fixed-order tree over the block
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

// one block: losses [cls, reg, corner] and the gradients w.r.t. rcnn_cls / rcnn_reg
__global__ __launch_bounds__(kThreads) void rcnn_loss_kernel(
    const float *__restrict__ rcnn_cls, const float *__restrict__ rcnn_reg, const float *__restrict__ rois,
    const float *__restrict__ gt_ct, const float *__restrict__ gt_src, const long long *__restrict__ reg_valid,
    const float *__restrict__ cls_labels, int N, int gtc, RcnnLossCfg c, float *__restrict__ out,
    float *__restrict__ g_cls, float *__restrict__ g_sl1, float *__restrict__ g_corner) {
  __shared__ double red[kThreads];
  const int tid = threadIdx.x;
  double n_valid = 0.0, n_fg = 0.0;
  for (int i = tid; i < N; i += kThreads) {
    n_valid += cls_labels[i] >= 0.f ? 1.0 : 0.0;
    n_fg += reg_valid[i] > 0 ? 1.0 : 0.0;
  }
  n_valid = block_sum(n_valid, red);
  n_fg = block_sum(n_fg, red);
  const float inv_valid = 1.f / fmaxf((float)n_valid, 1.f), inv_fg = 1.f / fmaxf((float)n_fg, 1.f);
  double s_cls = 0.0, s_reg = 0.0, s_cor = 0.0;
  for (int i = tid; i < N; i += kThreads) {
    // --- classification: BCE on sigmoid(rcnn_cls) (:200-218)
    {
      const float x = rcnn_cls[i], l = cls_labels[i];
      const bool valid = l >= 0.f;
      const float p = 1.f / (1.f + expf(-x));
      const float bce = -(l * fmaxf(logf(p), -100.f) + (1.f - l) * fmaxf(logf(1.f - p), -100.f));
      if (valid) s_cls += (double)bce;
      const float pq = p * (1.f - p);
      g_cls[i] = valid ? (p - l) / fmaxf(pq, 1e-12f) * pq * inv_valid * c.w_cls : 0.f;
    }
    const bool fg = reg_valid[i] > 0;
    float gs[7] = {0, 0, 0, 0, 0, 0, 0}, gc[7] = {0, 0, 0, 0, 0, 0, 0};
    if (fg) {
      const float *roi = rois + (size_t)i * 7, *t = rcnn_reg + (size_t)i * 7;
      const float *g = gt_ct + (size_t)i * gtc;
      // --- smooth-l1 on the residual encoding against the RoI at the origin (:149-163)
      const float dxa = fmaxf(roi[3], 1e-5f), dya = fmaxf(roi[4], 1e-5f), dza = fmaxf(roi[5], 1e-5f);
      const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
      const float diag = sqrtf(dxa * dxa + dya * dya);
      float tgt[7] = {g[0] / diag, g[1] / diag, g[2] / dza, logf(dxg / dxa), logf(dyg / dya), logf(dzg / dza),
                      g[6]};
      float row = 0.f;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float tk = tgt[k];
        if (tk != tk) tk = t[k];
        const float d = (t[k] - tk) * c.cw[k];
        const float n = fabsf(d);
        float l, dl;
        if (c.beta < 1e-5f) {
          l = n;
          dl = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else if (n < c.beta) {
          l = 0.5f * n * n / c.beta;
          dl = d / c.beta;
        } else {
          l = n - 0.5f * c.beta;
          dl = d > 0.f ? 1.f : -1.f;
        }
        row += l;
        gs[k] = dl * c.cw[k] * inv_fg * c.w_reg;
      }
      s_reg += (double)row;
      // --- corner regularisation (:165-196, loss_utils.py:209-233)
      if (c.corner) {
        const float *q = gt_src + (size_t)i * gtc;
        const float rdx = roi[3], rdy = roi[4], rdz = roi[5], ra = roi[6];
        const float rdiag = sqrtf(rdx * rdx + rdy * rdy);
        const float xg = t[0] * rdiag, yg = t[1] * rdiag, zg = t[2] * rdz;
        const float pdx = expf(t[3]) * rdx, pdy = expf(t[4]) * rdy, pdz = expf(t[5]) * rdz;
        const float pr = t[6] + ra;
        const float cr = cosf(ra), sr = sinf(ra);
        const float px = xg * cr - yg * sr + roi[0], py = xg * sr + yg * cr + roi[1], pz = zg + roi[2];
        const float cp = cosf(pr), sp = sinf(pr);
        const float cq = cosf(q[6]), sq = sinf(q[6]);
        const float cf = cosf(q[6] + (float)M_PI), sf = sinf(q[6] + (float)M_PI);
        float gpx = 0.f, gpy = 0.f, gpz = 0.f, gdx = 0.f, gdy = 0.f, gdz = 0.f, gr = 0.f, acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float tx = ((k & 3) == 0 || (k & 3) == 1) ? 0.5f : -0.5f;
          const float ty = ((k & 3) == 0 || (k & 3) == 3) ? 0.5f : -0.5f;
          const float tz = k < 4 ? -0.5f : 0.5f;
          const float lx = pdx * tx, ly = pdy * ty, lz = pdz * tz;
          const float cxp = lx * cp - ly * sp + px, cyp = lx * sp + ly * cp + py, czp = lz + pz;
          const float mx = q[3] * tx, my = q[4] * ty, mz = q[5] * tz;
          const float ax = cxp - (mx * cq - my * sq + q[0]), ay = cyp - (mx * sq + my * cq + q[1]),
                      az = czp - (mz + q[2]);
          const float bx = cxp - (mx * cf - my * sf + q[0]), by = cyp - (mx * sf + my * cf + q[1]);
          const float na = sqrtf(ax * ax + ay * ay + az * az), nb = sqrtf(bx * bx + by * by + az * az);
          const bool first = na <= nb;
          const float n = first ? na : nb;
          const float ex = first ? ax : bx, ey = first ? ay : by, ez = az;
          acc += n < 1.f ? 0.5f * n * n : n - 0.5f;
          const float sc = n < 1.f ? 1.f : (n > 0.f ? 1.f / n : 0.f);  // h'(n) / n
          const float ux = ex * sc, uy = ey * sc, uz = ez * sc;
          gpx += ux;
          gpy += uy;
          gpz += uz;
          gdx += (ux * cp + uy * sp) * tx;
          gdy += (-ux * sp + uy * cp) * ty;
          gdz += uz * tz;
          gr += ux * (-lx * sp - ly * cp) + uy * (lx * cp - ly * sp);
        }
        s_cor += (double)(acc / 8.f);
        const float k = inv_fg * c.w_corner / 8.f;
        gc[0] = (gpx * cr + gpy * sr) * rdiag * k;
        gc[1] = (-gpx * sr + gpy * cr) * rdiag * k;
        gc[2] = gpz * rdz * k;
        gc[3] = gdx * pdx * k;
        gc[4] = gdy * pdy * k;
        gc[5] = gdz * pdz * k;
        gc[6] = gr * k;
      }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      g_sl1[(size_t)i * 7 + k] = gs[k];
      g_corner[(size_t)i * 7 + k] = gc[k];
    }
  }
  s_cls = block_sum(s_cls, red);
  s_reg = block_sum(s_reg, red);
  s_cor = block_sum(s_cor, red);
  if (tid == 0) {
    out[0] = (float)s_cls * inv_valid * c.w_cls;
    out[1] = (float)s_reg * inv_fg * c.w_reg;
    out[2] = (float)s_cor * inv_fg * c.w_corner;
  }
}

__global__ __launch_bounds__(kThreads) void rcnn_loss_backward_kernel(
    const float *__restrict__ upstream, const float *__restrict__ g_cls, const float *__restrict__ g_sl1,
    const float *__restrict__ g_corner, int N, float *__restrict__ d_cls, float *__restrict__ d_reg) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const float u0 = upstream[0], u1 = upstream[1], u2 = upstream[2];
  if (i < N) d_cls[i] = g_cls[i] * u0;
  if (i < N * 7) d_reg[i] = g_sl1[i] * u1 + g_corner[i] * u2;
}

// ---------------------------------------------------------------------------------------------
// key-point segmentation labels: 1 / class inside a box, -1 in the enlarged margin, 0 elsewhere
__global__ __launch_bounds__(kThreads) void point_targets_kernel(
    const float *__restrict__ pts, int pt_stride, const float *__restrict__ gt, int P, int G, int gtc,
    float ex, float ey, float ez, int num_class, long long *__restrict__ labels) {
  __shared__ float sg[kGtChunk * 8];
  __shared__ float2 scs[kGtChunk];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int p = blockIdx.x * kThreads + tid;
  const float *gb = gt + (size_t)b * G * gtc;
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    const float *pp = pts + ((size_t)b * P + p) * pt_stride;
    x = pp[0];
    y = pp[1];
    z = pp[2];
  }
  int in_box = -1, in_ext = -1;
  for (int g0 = 0; g0 < G; g0 += kGtChunk) {
    const int ng = min(kGtChunk, G - g0);
    __syncthreads();
    for (int e = tid; e < ng * 8; e += kThreads) {
      int g = e >> 3, c = e & 7;
      sg[e] = gb[(size_t)(g0 + g) * gtc + (c < 7 ? c : gtc - 1)];
    }
    for (int g = tid; g < ng; g += kThreads) {
      double rz = (double)gb[(size_t)(g0 + g) * gtc + 6];
      scs[g] = make_float2((float)cos(-rz), (float)sin(-rz));  // roiaware_pool3d_kernel.cu:18
    }
    __syncthreads();
    if (p < P && (in_box < 0 || in_ext < 0)) {
      for (int g = 0; g < ng; ++g) {
        const float *q = sg + g * 8;
        const float sx = x - q[0], sy = y - q[1], az = fabsf(z - q[2]);
        const float cosa = scs[g].x, sina = scs[g].y;
        const float lx = fabsf(sx * cosa + sy * (-sina)), ly = fabsf(sx * sina + sy * cosa);
        const double m = (double)1e-5f;
        if (in_box < 0 && !((double)az > (double)q[5] / 2.0) && (double)lx < (double)q[3] / 2.0 + m &&
            (double)ly < (double)q[4] / 2.0 + m)
          in_box = g0 + g;
        const float wx = q[3] + ex, wy = q[4] + ey, wz = q[5] + ez;
        if (in_ext < 0 && !((double)az > (double)wz / 2.0) && (double)lx < (double)wx / 2.0 + m &&
            (double)ly < (double)wy / 2.0 + m)
          in_ext = g0 + g;
      }
    }
  }
  if (p < P) {
    const bool fg = in_box >= 0, ignore = fg != (in_ext >= 0);
    long long l = ignore ? -1 : 0;
    if (fg) l = num_class == 1 ? 1 : (long long)gb[(size_t)in_box * gtc + gtc - 1];
    labels[(size_t)b * P + p] = l;
  }
}

// one block: sigmoid focal loss (alpha 0.25, gamma 2) summed over points x classes, weights
// 1 / max(#positive, 1) on the labelled points; out = [loss, #positive]; grad (N, C)
__global__ __launch_bounds__(1024) void point_focal_kernel(const float *__restrict__ preds,
                                                            const long long *__restrict__ labels, int N, int C,
                                                            float alpha, float weight, float *__restrict__ out,
                                                            float *__restrict__ grad) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double npos = 0.0;
  for (int i = tid; i < N; i += 1024) npos += labels[i] > 0 ? 1.0 : 0.0;
  npos = block_sum(npos, red);
  const float w = 1.f / fmaxf((float)npos, 1.f);
  double s = 0.0;
  for (int e = tid; e < N * C; e += 1024) {
    const int i = e / C, cls = e - i * C;
    const long long l = labels[i];
    const float x = preds[e];
    const float t = l == cls + 1 ? 1.f : 0.f;
    const float p = 1.f / (1.f + expf(-x));
    const float aw = t * alpha + (1.f - t) * (1.f - alpha);
    const float pt = t * (1.f - p) + (1.f - t) * p;
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float wi = l >= 0 ? w : 0.f;
    s += (double)(aw * pt * pt * bce * wi);
    const float dpt = (t > 0.5f ? -1.f : 1.f) * p * (1.f - p);
    grad[e] = aw * (2.f * pt * dpt * bce + pt * pt * (p - t)) * wi * weight;
  }
  s = block_sum(s, red);
  if (tid == 0) {
    out[0] = (float)s * weight;
    out[1] = (float)npos;
  }
}

}  // namespace

extern "C" size_t dm_roi_targets_workspace_bytes(int batch, int n_rois) {
  if (batch <= 0 || n_rois <= 0) return 0;
  return dm_align((size_t)batch * n_rois * sizeof(float)) + dm_align((size_t)batch * n_rois * sizeof(int));
}

extern "C" int dm_roi_targets(const float *rois, const float *roi_scores, const long long *roi_labels,
                              const float *gt_boxes, int batch, int n_rois, int n_gt, int gt_cols,
                              const float *u_perm, const float *u_pick, int roi_per_image, int fg_per_image,
                              float reg_fg_thresh, float cls_fg_thresh, float cls_bg_thresh,
                              float cls_bg_thresh_lo, float hard_bg_ratio, float *out_rois, float *out_gt_src,
                              float *out_gt_canonical, float *out_iou, float *out_scores, long long *out_labels,
                              long long *out_reg_valid, float *out_cls_labels, long long *out_sampled,
                              float *out_ok, void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || n_rois <= 0 || n_gt <= 0 || roi_per_image <= 0 || gt_cols < 8 || gt_cols > 16)
    return DM_ERR_INVALID_ARG;
  if (!rois || !roi_scores || !roi_labels || !gt_boxes || !u_perm || !u_pick || !out_rois || !out_gt_src ||
      !out_gt_canonical || !out_iou || !out_scores || !out_labels || !out_reg_valid || !out_cls_labels ||
      !out_sampled || !out_ok || !workspace)
    return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_roi_targets_workspace_bytes(batch, n_rois)) return DM_ERR_WORKSPACE;
  const size_t lds = (size_t)5 * n_rois * sizeof(int);
  if (lds > 60 * 1024) return DM_ERR_INVALID_ARG;
  DmArena arena(workspace, workspace_bytes);
  float *max_iou = arena.take<float>((size_t)batch * n_rois);
  int *gt_idx = arena.take<int>((size_t)batch * n_rois);
  RoiTargetCfg c;
  c.roi_per_image = roi_per_image;
  c.fg_per_image = fg_per_image;
  c.reg_fg = reg_fg_thresh;
  c.cls_fg = cls_fg_thresh;
  c.cls_bg = cls_bg_thresh;
  c.cls_bg_lo = cls_bg_thresh_lo;
  c.hard_bg_ratio = hard_bg_ratio;
  c.fg_thresh = reg_fg_thresh < cls_fg_thresh ? reg_fg_thresh : cls_fg_thresh;
  c.cls_span = cls_fg_thresh - cls_bg_thresh;
  roi_gt_match_kernel<<<dim3(dm_ceil_div(n_rois, 16), batch), kThreads, 0, st>>>(
      rois, roi_labels, gt_boxes, n_rois, n_gt, gt_cols, max_iou, gt_idx);
  DM_CHECK_LAUNCH();
  roi_sample_kernel<<<batch, kThreads, lds, st>>>(rois, roi_scores, roi_labels, gt_boxes, max_iou, gt_idx,
                                                  u_perm, u_pick, c, n_rois, n_gt, gt_cols, out_rois,
                                                  out_gt_src, out_gt_canonical, out_iou, out_scores,
                                                  out_labels, out_reg_valid, out_cls_labels, out_sampled,
                                                  out_ok);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_rcnn_loss_forward(const float *rcnn_cls, const float *rcnn_reg, const float *rois,
                                    const float *gt_canonical, const float *gt_src,
                                    const long long *reg_valid, const float *cls_labels, int n, int gt_cols,
                                    const float *loss_weights3, const float *code_weights7, float beta,
                                    int corner_loss, float *out3, float *g_cls, float *g_sl1,
                                    float *g_corner, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || gt_cols < 7) return DM_ERR_INVALID_ARG;
  if (!rcnn_cls || !rcnn_reg || !rois || !gt_canonical || !gt_src || !reg_valid || !cls_labels ||
      !loss_weights3 || !code_weights7 || !out3 || !g_cls || !g_sl1 || !g_corner)
    return DM_ERR_INVALID_ARG;
  RcnnLossCfg c;
  c.w_cls = loss_weights3[0];
  c.w_reg = loss_weights3[1];
  c.w_corner = loss_weights3[2];
  c.beta = beta;
  for (int k = 0; k < 7; ++k) c.cw[k] = code_weights7[k];
  c.corner = corner_loss;
  rcnn_loss_kernel<<<1, kThreads, 0, st>>>(rcnn_cls, rcnn_reg, rois, gt_canonical, gt_src, reg_valid,
                                           cls_labels, n, gt_cols, c, out3, g_cls, g_sl1, g_corner);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_rcnn_loss_backward(const float *upstream3, const float *g_cls, const float *g_sl1,
                                     const float *g_corner, int n, float *d_rcnn_cls, float *d_rcnn_reg,
                                     dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || !upstream3 || !g_cls || !g_sl1 || !g_corner || !d_rcnn_cls || !d_rcnn_reg)
    return DM_ERR_INVALID_ARG;
  rcnn_loss_backward_kernel<<<dm_ceil_div((long long)n * 7, kThreads), kThreads, 0, st>>>(
      upstream3, g_cls, g_sl1, g_corner, n, d_rcnn_cls, d_rcnn_reg);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_point_targets(const float *points, int point_stride, const float *gt_boxes, int batch,
                                int n_points, int n_gt, int gt_cols, const float *extra_width3,
                                int num_class, long long *labels, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || n_points <= 0 || n_gt <= 0 || gt_cols < 7 || point_stride < 3) return DM_ERR_INVALID_ARG;
  if (!points || !gt_boxes || !extra_width3 || !labels) return DM_ERR_INVALID_ARG;
  point_targets_kernel<<<dim3(dm_ceil_div(n_points, kThreads), batch), kThreads, 0, st>>>(
      points, point_stride, gt_boxes, n_points, n_gt, gt_cols, extra_width3[0], extra_width3[1],
      extra_width3[2], num_class, labels);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_point_focal_loss(const float *preds, const long long *labels, int n, int n_cls, float alpha,
                                   float loss_weight, float *out2, float *grad, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || n_cls <= 0 || !preds || !labels || !out2 || !grad) return DM_ERR_INVALID_ARG;
  point_focal_kernel<<<1, 1024, 0, st>>>(preds, labels, n, n_cls, alpha, loss_weight, out2, grad);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
