// Rotated BEV overlap / IoU matrices and device-resident rotated NMS for gfx950.
//
// Replaces pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-359 and the host side of
// pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:50-185 (cudaMalloc + whole-mask D2H + serial
// host greedy).  Differences that are design, not semantics:
//   * cos/sin of every box heading are evaluated ONCE per box (a pre-pass) instead of
//     up to 20 times per pair; the values are identical because they are the same
//     function of the same input.  Transcendentals are "double libm rounded to float"
//     — the rule the CPU oracle uses too (oracle/dm_oracle.c) — so that keep masks
//     agree bit for bit; the fp32 op order of box_overlap is kept as written
//     (compiled with -ffp-contract=off).
//   * only the upper-triangular 64x64 tiles of the suppression mask are computed (the
//     greedy pass never reads the others, iou3d_nms.cpp:122-133);
//   * the greedy pass runs on the device (one wave, remv[] words spread over lanes);
//     nothing is copied to the host, num_keep stays in device memory.
#include <cstdlib>

#include "dm_common.h"

#include "box_geom.h"

namespace {

// Several independent NMS problems of equal size in ONE launch chain (blockIdx.z = problem): element strides of the
// per-problem arrays; all zero for a single problem.  The greedy pass is one wave per problem, so B problems
// issued one after the other leave the chip idle B times as long.
struct NmsBatch {
  long long boxes, cs, mask, keep;
};

__global__ __launch_bounds__(256) void heading_cos_sin(const float *boxes, int n, float2 *cs) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double h = (double)boxes[(size_t)i * 7 + 6];
  cs[i] = make_float2((float)cos(h), (float)sin(h));
}

// one 16x16 tile of the (na, nb) matrix per 256-thread block, boxes staged in LDS
template <bool IOU>
__global__ __launch_bounds__(256) void pair_matrix(const float *boxes_a, const float2 *cs_a, int na,
                                                   const float *boxes_b, const float2 *cs_b, int nb,
                                                   float *out) {
  __shared__ float sa[16 * 7], sb[16 * 7];
  __shared__ float2 ca[16], cb[16];
  int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
  if (threadIdx.x < 112) {
    int i = threadIdx.x / 7;
    if (a0 + i < na) sa[threadIdx.x] = boxes_a[(size_t)a0 * 7 + threadIdx.x];
    if (b0 + i < nb) sb[threadIdx.x] = boxes_b[(size_t)b0 * 7 + threadIdx.x];
  } else if (threadIdx.x < 128) {
    int i = threadIdx.x - 112;
    if (a0 + i < na) ca[i] = cs_a[a0 + i];
    if (b0 + i < nb) cb[i] = cs_b[b0 + i];
  }
  __syncthreads();
  int ai = a0 + ty, bi = b0 + tx;
  if (ai >= na || bi >= nb) return;
  float v = IOU ? iou_bev(sa + ty * 7, ca[ty], sb + tx * 7, cb[tx])
                : box_overlap(sa + ty * 7, ca[ty], sb + tx * 7, cb[tx]);
  out[(size_t)ai * nb + bi] = v;
}

// mmcv.ops.nms IoU on (x1, y1, x2, y2) boxes: inter / (area_a + area_b - inter), offset 0
__device__ __forceinline__ float iou_xyxy(const float *a, const float *b) {
  float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  float inter = w * h;
  float sa = (a[2] - a[0]) * (a[3] - a[1]);
  float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(64) void nms_mask_2d(const float *boxes, int n, float thresh,
                                                  unsigned long long *mask, int col_begin,
                                                  const int *num_keep, int max_keep, NmsBatch bt) {
  boxes += blockIdx.z * bt.boxes, mask += blockIdx.z * bt.mask;
  if (num_keep) num_keep += blockIdx.z;
  const int row_start = blockIdx.y, col_start = blockIdx.x + col_begin;
  if (col_start < row_start) return;
  if (num_keep && *num_keep >= max_keep) return;   // second phase not needed (see nms_two_phase)
  const int col_blocks = (n + 63) / 64;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float bb[64 * 4];
  int t = threadIdx.x;
  if (t < col_size) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bb[t * 4 + j] = boxes[(size_t)(64 * col_start + t) * 4 + j];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = 64 * row_start + t;
    float cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = boxes[(size_t)cur * 4 + j];
    unsigned long long bits = 0;
    int start = (row_start == col_start) ? t + 1 : 0;
    for (int i = start; i < col_size; i++)
      if (iou_xyxy(cb, bb + i * 4) > thresh) bits |= 1ULL << i;
    mask[(size_t)cur * col_blocks + col_start] = bits;
  }
}

// mask[r][c] bit i = iou(box_r, box_{64c+i}) > thresh; upper-triangular tiles only
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask(const float *boxes, const float2 *cs, int n,
                                               float thresh, unsigned long long *mask, int col_begin,
                                               const int *num_keep, int max_keep, NmsBatch bt) {
  boxes += blockIdx.z * bt.boxes, cs += blockIdx.z * bt.cs, mask += blockIdx.z * bt.mask;
  if (num_keep) num_keep += blockIdx.z;
  const int row_start = blockIdx.y, col_start = blockIdx.x + col_begin;
  if (col_start < row_start) return;
  if (num_keep && *num_keep >= max_keep) return;   // second phase not needed (see nms_two_phase)
  const int col_blocks = (n + 63) / 64;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float bb[64 * 7];
  __shared__ float2 bcs[64];
  int t = threadIdx.x;
  if (t < col_size) {
#pragma unroll
    for (int j = 0; j < 7; ++j) bb[t * 7 + j] = boxes[(size_t)(64 * col_start + t) * 7 + j];
    if (!NORMAL) bcs[t] = cs[64 * col_start + t];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = 64 * row_start + t;
    float cb[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cb[j] = boxes[(size_t)cur * 7 + j];
    float2 ccs = NORMAL ? make_float2(1.f, 0.f) : cs[cur];
    unsigned long long bits = 0;
    int start = (row_start == col_start) ? t + 1 : 0;
    for (int i = start; i < col_size; i++) {
      float v = NORMAL ? iou_normal(cb, bb + i * 7) : iou_bev(cb, ccs, bb + i * 7, bcs[i]);
      if (v > thresh) bits |= 1ULL << i;
    }
    mask[(size_t)cur * col_blocks + col_start] = bits;
  }
}

// Rotated variant with a load-balanced tile.  A score-sorted tile holds boxes from all over the scene:
// of its 4096 pairs only the few whose bounding circles touch can overlap at all, but in the plain loop
// above a wave pays for the polygon clipping of column i as soon as ONE of its 64 rows is near it.
// Stage 1: every lane tests its row against the 64 columns by centre distance (exact: pairs farther
// apart than r_a + r_b + 5 cm have no intersection point and no corner inside the other box even with
// the reference's 1 cm containment margin, so their overlap is 0 and the bit is 0).  Stage 2: the
// surviving (row, column) candidates of the whole tile are dealt out evenly to the 64 lanes.
__device__ __forceinline__ int select_bit(unsigned long long word, int k) {   // index of the k-th set bit
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll(word & (((1ull << w) - 1ull) << pos));
    if (k >= c) {
      k -= c;
      pos += w;
    }
  }
  return pos;
}

__global__ __launch_bounds__(64) void nms_mask_rot(const float *boxes, const float2 *cs, int n, float thresh,
                                                   unsigned long long *mask, int col_begin,
                                                   const int *num_keep, int max_keep, NmsBatch bt) {
  boxes += blockIdx.z * bt.boxes, cs += blockIdx.z * bt.cs, mask += blockIdx.z * bt.mask;
  if (num_keep) num_keep += blockIdx.z;
  const int row_start = blockIdx.y, col_start = blockIdx.x + col_begin;
  if (col_start < row_start) return;
  if (num_keep && *num_keep >= max_keep) return;
  const int col_blocks = (n + 63) / 64;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float rb[64 * 7], cbx[64 * 7];
  __shared__ float2 rcs[64], ccs[64];
  __shared__ float rrad[64], crad[64];
  __shared__ unsigned long long cand[64], res[64];
  __shared__ int pre[65];
  const int t = threadIdx.x;
  if (t < row_size) {
#pragma unroll
    for (int j = 0; j < 7; ++j) rb[t * 7 + j] = boxes[(size_t)(64 * row_start + t) * 7 + j];
    rcs[t] = cs[64 * row_start + t];
    rrad[t] = 0.5f * sqrtf(rb[t * 7 + 3] * rb[t * 7 + 3] + rb[t * 7 + 4] * rb[t * 7 + 4]);
  }
  if (t < col_size) {
#pragma unroll
    for (int j = 0; j < 7; ++j) cbx[t * 7 + j] = boxes[(size_t)(64 * col_start + t) * 7 + j];
    ccs[t] = cs[64 * col_start + t];
    crad[t] = 0.5f * sqrtf(cbx[t * 7 + 3] * cbx[t * 7 + 3] + cbx[t * 7 + 4] * cbx[t * 7 + 4]);
  }
  res[t] = 0ull;
  __syncthreads();
  unsigned long long c = 0ull;
  if (t < row_size) {
    const float x = rb[t * 7], y = rb[t * 7 + 1], r0 = rrad[t] + 0.05f;
    const int start = (row_start == col_start) ? t + 1 : 0;
    for (int i = start; i < col_size; i++) {
      const float dx = x - cbx[i * 7], dy = y - cbx[i * 7 + 1], rr = r0 + crad[i];
      if (!(dx * dx + dy * dy > rr * rr)) c |= 1ull << i;     // NaN-safe: unordered -> candidate
    }
  }
  cand[t] = c;
  int incl = __popcll(c);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (t >= o) incl += up;
  }
  pre[t + 1] = incl;
  if (t == 0) pre[0] = 0;
  __syncthreads();
  const int total = pre[64];
  for (int p = t; p < total; p += 64) {
    int lo = 0, hi = 63;                         // largest row with pre[row] <= p
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pre[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int col = select_bit(cand[lo], p - pre[lo]);
    const float v = iou_bev(rb + lo * 7, rcs[lo], cbx + col * 7, ccs[col]);
    if (v > thresh) atomicOr(&res[lo], 1ull << col);
  }
  __syncthreads();
  if (t < row_size) mask[(size_t)(64 * row_start + t) * col_blocks + col_start] = res[t];
}

// Greedy pass (iou3d_nms.cpp:117-133) by ONE wave: lane l owns remv words l, l+64, ...
// Boxes are resolved 64 at a time.  Inside a block of 64 the decisions depend only on the
// block's diagonal mask word of each box (one coalesced load, one word per lane) and are taken
// with scalar bit operations + readlane — no memory access in the dependent chain.  The full mask
// rows of the survivors of the block are then OR-ed into `remv` with up to 8 independent row loads
// in flight.  Stops after max_keep survivors (callers slice keep[:post_max] anyway).
__device__ __forceinline__ unsigned long long wave_bcast64(unsigned long long v, int src_lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src_lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src_lane);
  return ((unsigned long long)hi << 32) | lo;
}

// The pass can be split in two (nms_two_phase): boxes arrive sorted by score and callers keep only
// the first max_keep survivors, which are almost always found among the first few thousand
// candidates — so the mask is first computed for the leading blk_end blocks only, and the rest of the
// (quadratic) mask + the rest of this pass run only if the budget was not met (`resume`).
__global__ __launch_bounds__(64) void nms_greedy(const unsigned long long *__restrict__ mask, int n,
                                                 int max_keep, long long *keep, int *num_keep,
                                                 int blk_begin, int blk_end, int resume, NmsBatch bt) {
  mask += blockIdx.z * bt.mask, keep += blockIdx.z * bt.keep, num_keep += blockIdx.z;
  const int col_blocks = (n + 63) / 64;
  const int lane = threadIdx.x;
  constexpr int MAXW = 16;  // up to 64*16 words = 65536 boxes
  unsigned long long remv[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) remv[w] = 0ull;
  int kept = 0;
  if (resume) {
    kept = *num_keep;
    if (kept >= max_keep) return;
    // suppression state of the remaining blocks: OR the rows of the survivors found so far
    for (int i0 = 0; i0 < kept; i0 += 8) {
      const unsigned long long *rows[8];
      const int cnt = min(8, kept - i0);
#pragma unroll
      for (int u = 0; u < 8; ++u) rows[u] = mask + (size_t)keep[i0 + (u < cnt ? u : 0)] * col_blocks;
#pragma unroll
      for (int w = 0; w < MAXW; ++w) {
        if (w * 64 >= col_blocks) break;
        const int j = w * 64 + lane;
        const bool in = j >= blk_begin && j < col_blocks;
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (in && u < cnt) ? rows[u][j] : 0ull;
        remv[w] |= ((t[0] | t[1]) | (t[2] | t[3])) | ((t[4] | t[5]) | (t[6] | t[7]));
      }
    }
  }
  for (int blk = blk_begin; blk < blk_end && kept < max_keep; ++blk) {
    // the word that decides boxes [64*blk, 64*blk+64) lives on lane blk%64, slot blk/64
    const int slot = blk >> 6, owner = blk & 63;
    unsigned long long word = 0;
#pragma unroll
    for (int w = 0; w < MAXW; ++w)
      if (w == slot) word = remv[w];
    word = wave_bcast64(word, owner);
    const int lim = min(64, n - blk * 64);
    const unsigned long long diag =
        lane < lim ? mask[(size_t)(blk * 64 + lane) * col_blocks + blk] : 0ull;
    // Word `lane` (slot 0: blocks 0..63, where the survivor budget is normally met) of ALL 64 rows of
    // this block, requested before the decisions are known: the loads do not depend on the resolve
    // loop below and complete underneath it, instead of one gather round per 8 survivors afterwards.
    unsigned long long t0[64];
    {
      const bool in0 = lane > blk && lane < blk_end && lane < col_blocks;
#pragma unroll
      for (int u = 0; u < 64; ++u)
        t0[u] = (in0 && u < lim) ? mask[(size_t)(blk * 64 + u) * col_blocks + lane] : 0ull;
    }
    unsigned long long alive = ~word & (lim == 64 ? ~0ull : ((1ull << lim) - 1ull));
    unsigned long long survivors = 0ull;
    while (alive != 0ull && kept < max_keep) {
      const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
      survivors |= 1ull << b;
      if (lane == 0) keep[kept] = blk * 64 + b;
      ++kept;
      alive &= ~(wave_bcast64(diag, b) | (1ull << b));
    }
    {
      unsigned long long acc = 0ull;
#pragma unroll
      for (int u = 0; u < 64; ++u) acc |= ((survivors >> u) & 1ull) ? t0[u] : 0ull;
      remv[0] |= acc;
    }
    // slots >= 1 (boxes beyond 4096: only reached when the budget was not met among the leading
    // blocks): gather the survivors' rows, 8 at a time
    if (col_blocks > 64 && blk_end > 64) {
      while (survivors != 0ull) {
        const unsigned long long *rows[8];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          rows[u] = mask;
          if (survivors != 0ull) {
            const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(survivors));
            survivors &= survivors - 1ull;
            rows[u] = mask + (size_t)(blk * 64 + b) * col_blocks;
            cnt = u + 1;
          }
        }
#pragma unroll
        for (int w = 1; w < MAXW; ++w) {
          if (w * 64 >= col_blocks) break;
          const int j = w * 64 + lane;
          const bool in = j > blk && j < blk_end;
          unsigned long long t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = (in && u < cnt) ? rows[u][j] : 0ull;
          remv[w] |= ((t[0] | t[1]) | (t[2] | t[3])) | ((t[4] | t[5]) | (t[6] | t[7]));
        }
      }
    }
  }
  if (lane == 0) *num_keep = kept;
}

}  // namespace

extern "C" size_t dm_iou3d_workspace_bytes(int na, int nb) {
  size_t n = (size_t)(na > 0 ? na : 0) + (nb > 0 ? nb : 0);
  return dm_align(n * sizeof(float2)) + 256;
}

static int pair_launch(bool iou, const float *a, int na, const float *b, int nb, float *out,
                       void *ws, size_t ws_bytes, hipStream_t st) {
  if (na < 0 || nb < 0) return DM_ERR_INVALID_ARG;
  if (na == 0 || nb == 0) return DM_OK;
  if (!a || !b || !out || !ws) return DM_ERR_INVALID_ARG;
  if (ws_bytes < dm_iou3d_workspace_bytes(na, nb)) return DM_ERR_WORKSPACE;
  float2 *csa = (float2 *)ws, *csb = csa + na;
  heading_cos_sin<<<dm_ceil_div(na, 256), 256, 0, st>>>(a, na, csa);
  heading_cos_sin<<<dm_ceil_div(nb, 256), 256, 0, st>>>(b, nb, csb);
  DM_CHECK_LAUNCH();
  dim3 grid(dm_ceil_div(nb, 16), dm_ceil_div(na, 16));
  if (iou) pair_matrix<true><<<grid, 256, 0, st>>>(a, csa, na, b, csb, nb, out);
  else pair_matrix<false><<<grid, 256, 0, st>>>(a, csa, na, b, csb, nb, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// Exact rotated-rectangle intersection area (no containment margin, double precision): rectangle A
// clipped by the four half-planes of rectangle B (Sutherland-Hodgman on convex polygons, <= 8 vertices),
// shoelace area.  For the consumers that compare overlaps with thresholds or with zero — KITTI
// evaluation (the reference: rotate_iou_gpu_eval, mmdet3d/core/evaluation/kitti_utils/rotate_iou.py)
// and the collision test of the GT-paste sampler (box_np_ops.box_collision_test) — where the 1 cm
// corner margin of the NMS kernel above would turn near-touching boxes into overlapping ones.
__global__ __launch_bounds__(256) void pair_overlap_exact(const float *__restrict__ boxes_a, int na,
                                                          const float *__restrict__ boxes_b, int nb,
                                                          float *__restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)na * nb) return;
  const int ia = (int)(e / nb), ib = (int)(e % nb);
  const float *A = boxes_a + (size_t)ia * 7, *B = boxes_b + (size_t)ib * 7;
  double px[10], py[10], qx[10], qy[10];
  {
    const double c = cos((double)A[6]), s = sin((double)A[6]), hx = (double)A[3] / 2, hy = (double)A[4] / 2;
    const double sx[4] = {-hx, hx, hx, -hx}, sy[4] = {-hy, -hy, hy, hy};
    for (int k = 0; k < 4; ++k) px[k] = (double)A[0] + sx[k] * c - sy[k] * s, py[k] = (double)A[1] + sx[k] * s + sy[k] * c;
  }
  int n = 4;
  const double cb = cos((double)B[6]), sb = sin((double)B[6]);
  const double hbx = (double)B[3] / 2, hby = (double)B[4] / 2;
  // the four inward half-planes of B in its own frame: +-x <= hbx, +-y <= hby
  for (int side = 0; side < 4 && n > 0; ++side) {
    const double nx = side == 0 ? cb : (side == 1 ? -cb : (side == 2 ? -sb : sb));
    const double ny = side == 0 ? sb : (side == 1 ? -sb : (side == 2 ? cb : -cb));
    const double lim = (side < 2 ? hbx : hby);
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const int k2 = (k + 1 == n) ? 0 : k + 1;
      const double d1 = (px[k] - (double)B[0]) * nx + (py[k] - (double)B[1]) * ny - lim;
      const double d2 = (px[k2] - (double)B[0]) * nx + (py[k2] - (double)B[1]) * ny - lim;
      if (d1 <= 0.0) qx[m] = px[k], qy[m] = py[k], ++m;
      if ((d1 < 0.0 && d2 > 0.0) || (d1 > 0.0 && d2 < 0.0)) {
        const double t = d1 / (d1 - d2);
        qx[m] = px[k] + t * (px[k2] - px[k]), qy[m] = py[k] + t * (py[k2] - py[k]), ++m;
      }
    }
    n = m;
    for (int k = 0; k < n; ++k) px[k] = qx[k], py[k] = qy[k];
  }
  double area = 0.0;
  for (int k = 0; k < n; ++k) {
    const int k2 = (k + 1 == n) ? 0 : k + 1;
    area += px[k] * py[k2] - px[k2] * py[k];
  }
  out[e] = n >= 3 ? (float)(fabs(area) * 0.5) : 0.f;
}

extern "C" int dm_boxes_overlap_bev_exact(const float *boxes_a, int na, const float *boxes_b, int nb,
                                          float *ans_overlap, dm_stream_t stream) {
  if (na < 0 || nb < 0) return DM_ERR_INVALID_ARG;
  if (na == 0 || nb == 0) return DM_OK;
  if (!boxes_a || !boxes_b || !ans_overlap) return DM_ERR_INVALID_ARG;
  pair_overlap_exact<<<dm_ceil_div((long long)na * nb, 256), 256, 0, (hipStream_t)stream>>>(boxes_a, na, boxes_b, nb,
                                                                                          ans_overlap);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_boxes_overlap_bev(const float *boxes_a, int na, const float *boxes_b, int nb,
                                    float *ans_overlap, void *workspace, size_t workspace_bytes,
                                    dm_stream_t stream) {
  return pair_launch(false, boxes_a, na, boxes_b, nb, ans_overlap, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

extern "C" int dm_boxes_iou_bev(const float *boxes_a, int na, const float *boxes_b, int nb,
                                float *ans_iou, void *workspace, size_t workspace_bytes,
                                dm_stream_t stream) {
  return pair_launch(true, boxes_a, na, boxes_b, nb, ans_iou, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

extern "C" size_t dm_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  size_t col_blocks = ((size_t)n + 63) / 64;
  return dm_align((size_t)n * col_blocks * 8) + dm_align((size_t)n * sizeof(float2)) + 256;
}

// Blocks (of 64 score-sorted boxes) whose mask is computed up front: 4x the survivor budget, at
// least 1024 boxes.  Everything when no budget applies.
static int nms_lead_blocks(int n, int max_keep) {
  const int col_blocks = (n + 63) / 64;
  if (max_keep <= 0 || max_keep >= n) return col_blocks;
  long long lead_boxes = 4ll * max_keep;
  if (lead_boxes < 1024) lead_boxes = 1024;
  int lead = (int)((lead_boxes + 63) / 64);
  return lead < col_blocks ? lead : col_blocks;
}

static bool g_nms_plain = false;   // tuning aid (DM_NMS_PLAIN=1): the one-row-per-lane mask kernel

static int nms_launch(bool normal, const float *boxes, int n, float thresh, int max_keep,
                      long long *keep, int *num_keep, void *ws, size_t ws_bytes, hipStream_t st,
                      int batch = 1, long long keep_stride = 0) {
  if (n < 0 || batch < 1 || !num_keep) return DM_ERR_INVALID_ARG;
  if (n == 0) {
    DM_HIP(hipMemsetAsync(num_keep, 0, batch * sizeof(int), st));
    return DM_OK;
  }
  if (n > 64 * 64 * 16 || batch > 65535) return DM_ERR_UNSUPPORTED;
  if (!boxes || !keep || !ws) return DM_ERR_INVALID_ARG;
  {
    static const bool plain = getenv("DM_NMS_PLAIN") && getenv("DM_NMS_PLAIN")[0] == '1';
    g_nms_plain = plain;
  }
  if (ws_bytes < dm_nms_workspace_bytes(n) * (size_t)batch) return DM_ERR_WORKSPACE;
  if (max_keep <= 0 || max_keep > n) max_keep = n;
  int col_blocks = (n + 63) / 64;
  const size_t mask_bytes = dm_align((size_t)n * col_blocks * 8);
  unsigned long long *mask = (unsigned long long *)ws;
  float2 *cs = (float2 *)((char *)ws + mask_bytes * batch);
  NmsBatch bt{0, 0, 0, 0};
  if (batch > 1) bt = NmsBatch{(long long)n * 7, (long long)n, (long long)(mask_bytes / 8), keep_stride};
  if (!normal) {      // the problems' boxes are contiguous: one launch over batch * n headings
    heading_cos_sin<<<dm_ceil_div(n * batch, 256), 256, 0, st>>>(boxes, n * batch, cs);
    DM_CHECK_LAUNCH();
  }
  const int lead = nms_lead_blocks(n, max_keep);
  for (int phase = 0; phase < (lead < col_blocks ? 2 : 1); ++phase) {
    // phase 0: tiles (r <= c < lead), greedy over the leading blocks; phase 1 (skipped on the
    // device when the budget is already met): tiles with c >= lead, greedy over the rest
    const int c0 = phase == 0 ? 0 : lead, c1 = phase == 0 ? lead : col_blocks;
    dim3 g(c1 - c0, c1, batch);
    const int *flag = phase == 0 ? nullptr : num_keep;
    if (normal) nms_mask<true><<<g, 64, 0, st>>>(boxes, cs, n, thresh, mask, c0, flag, max_keep, bt);
    else if (g_nms_plain) nms_mask<false><<<g, 64, 0, st>>>(boxes, cs, n, thresh, mask, c0, flag, max_keep, bt);
    else nms_mask_rot<<<g, 64, 0, st>>>(boxes, cs, n, thresh, mask, c0, flag, max_keep, bt);
    DM_CHECK_LAUNCH();
    nms_greedy<<<dim3(1, 1, batch), 64, 0, st>>>(mask, n, max_keep, keep, num_keep, c0, c1, phase, bt);
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_nms(const float *boxes, int n, float thresh, int max_keep, long long *keep,
                      int *num_keep, void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  return nms_launch(false, boxes, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" int dm_nms_normal(const float *boxes, int n, float thresh, int max_keep,
                             long long *keep, int *num_keep, void *workspace,
                             size_t workspace_bytes, dm_stream_t stream) {
  return nms_launch(true, boxes, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" int dm_nms_batch(const float *boxes, int batch, int n, float thresh, int max_keep, int normal,
                            long long *keep, long long keep_stride, int *num_keep, void *workspace,
                            size_t workspace_bytes, dm_stream_t stream) {
  if (keep_stride < (max_keep > 0 && max_keep < n ? max_keep : n)) return DM_ERR_INVALID_ARG;
  return nms_launch(normal != 0, boxes, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes,
                    (hipStream_t)stream, batch, keep_stride);
}

// 2-D axis-aligned NMS on (n, 4) xyxy boxes sorted by descending score — replaces
// mmcv.ops.nms (mmcv-full 1.3.16) as called through batched_nms at
// mmdet3d/models/ssl_modules/bbox_utils.py:97 and by the Faster R-CNN RPN / bbox head.
static int nms_2d_launch(const float *boxes_xyxy, int n, float thresh, int max_keep, long long *keep,
                         int *num_keep, void *workspace, size_t workspace_bytes, hipStream_t st, int batch,
                         long long keep_stride) {
  if (n < 0 || batch < 1 || !num_keep) return DM_ERR_INVALID_ARG;
  if (n == 0) {
    DM_HIP(hipMemsetAsync(num_keep, 0, batch * sizeof(int), st));
    return DM_OK;
  }
  if (n > 64 * 64 * 16 || batch > 65535) return DM_ERR_UNSUPPORTED;
  if (!boxes_xyxy || !keep || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_nms_workspace_bytes(n) * (size_t)batch) return DM_ERR_WORKSPACE;
  if (max_keep <= 0 || max_keep > n) max_keep = n;
  int col_blocks = (n + 63) / 64;
  const size_t mask_bytes = dm_align((size_t)n * col_blocks * 8);
  unsigned long long *mask = (unsigned long long *)workspace;
  NmsBatch bt{0, 0, 0, 0};
  if (batch > 1) bt = NmsBatch{(long long)n * 4, 0, (long long)(mask_bytes / 8), keep_stride};
  const int lead = nms_lead_blocks(n, max_keep);
  for (int phase = 0; phase < (lead < col_blocks ? 2 : 1); ++phase) {
    const int c0 = phase == 0 ? 0 : lead, c1 = phase == 0 ? lead : col_blocks;
    nms_mask_2d<<<dim3(c1 - c0, c1, batch), 64, 0, st>>>(boxes_xyxy, n, thresh, mask, c0,
                                                          phase == 0 ? nullptr : num_keep, max_keep, bt);
    DM_CHECK_LAUNCH();
    nms_greedy<<<dim3(1, 1, batch), 64, 0, st>>>(mask, n, max_keep, keep, num_keep, c0, c1, phase, bt);
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_nms_2d(const float *boxes_xyxy, int n, float thresh, int max_keep,
                         long long *keep, int *num_keep, void *workspace,
                         size_t workspace_bytes, dm_stream_t stream) {
  return nms_2d_launch(boxes_xyxy, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes,
                       (hipStream_t)stream, 1, 0);
}

extern "C" int dm_nms_2d_batch(const float *boxes_xyxy, int batch, int n, float thresh, int max_keep,
                               long long *keep, long long keep_stride, int *num_keep, void *workspace,
                               size_t workspace_bytes, dm_stream_t stream) {
  if (keep_stride < (max_keep > 0 && max_keep < n ? max_keep : n)) return DM_ERR_INVALID_ARG;
  return nms_2d_launch(boxes_xyxy, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes,
                       (hipStream_t)stream, batch, keep_stride);
}
