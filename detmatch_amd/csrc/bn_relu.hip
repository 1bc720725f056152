// Fused training-mode BatchNorm (+ReLU) over row-major (N, C) activations.
//
// Every BatchNorm on the 3D side of the DetMatch step normalises such a matrix: the sparse backbone
// (BatchNorm1d after each sparse conv: pcdet/models/backbones_3d/spconv_backbone.py:9-28), the
// set-abstraction MLPs in row layout (pointnet2_modules.py:31-40: Conv2d 1x1 + BatchNorm2d + ReLU on
// M*nsample grouped rows) and the FC heads.  torch runs collect_statistics, transform_input and a
// separate ReLU forward (5 passes over the matrix) and threshold_backward, backward_reduce,
// backward_elemt backward (8 passes); here: statistics (1 read), normalise+ReLU (1 read, 1 write),
// and backward reduce (2 reads) + input gradient (2 reads, 1 write) with the ReLU mask recomputed
// from x — 3 + 5 passes, HBM bound.
//
// (Measured and dropped: folding the partials in the LAST workgroup to arrive at a counter — the
// threadFenceReduction pattern, one launch instead of two per reduction.  The device-wide fence every
// workgroup needs before it counts itself in writes back / invalidates the XCD's L2: the BN kernels of a
// training step took twice as long, 118 -> 153 ms per iteration.)
//
// Statistics: per-block shifted sums (shift = the block's first row, so no catastrophic cancellation),
// turned into (n, mean, M2) and combined over blocks in a fixed order (Chan et al.) -> deterministic.
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

constexpr int BN_ROWS = 512;   // most rows one workgroup folds

// Rows per workgroup: the reductions are a few MB at most and latency bound, so a launch wants ~1 k
// workgroups (4 per CU) before it wants long per-thread loops — (17.6 k voxels, C = 16) on 35 workgroups
// of 512 rows took 69 us, the same matrix on 550 workgroups of 32 rows is an HBM-speed read.  A fixed
// function of (n, c): the summation order, hence the result, depends on the shape only.
__host__ __device__ inline int bn_rows_per_block(long long n, int c) {
  const int tpr = c >= 4 ? c / 4 : 1;
  const int rpi = tpr <= 256 ? 256 / tpr : 1;           // rows one pass of the 256 threads covers
  long long r = (n + 1023) / 1024;
  r = (r + rpi - 1) / rpi * rpi;
  if (r < rpi) r = rpi;
  if (r > BN_ROWS) r = BN_ROWS;
  return (int)r;
}

// block layout: C/4 threads per row (float4 channel groups), 256 / (C/4) rows per iteration
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, long long n, int c,
                                                       float *__restrict__ partial /*[blocks][2][c]*/) {
  extern __shared__ float sm[];
  const int tpr = c / 4, rpi = 256 / tpr;          // threads per row, rows per iteration
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int rpb = bn_rows_per_block(n, c);
  const long long row0 = (long long)blockIdx.x * rpb;
  const int rows = (int)min((long long)rpb, n - row0);
  const float4 k = *(const float4 *)(x + row0 * c + 4 * cg);   // shift: first row of the block
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < rpi)
    for (int r = rl; r < rows; r += rpi) {
      const float4 v = *(const float4 *)(x + (row0 + r) * c + 4 * cg);
      const float dx = v.x - k.x, dy = v.y - k.y, dz = v.z - k.z, dw = v.w - k.w;
      s1.x += dx, s1.y += dy, s1.z += dz, s1.w += dw;
      s2.x += dx * dx, s2.y += dy * dy, s2.z += dz * dz, s2.w += dw * dw;
    }
  float *a1 = sm, *a2 = sm + 256 * 4;
  ((float4 *)a1)[threadIdx.x] = s1;
  ((float4 *)a2)[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < tpr) {          // fixed-order sum over the row lanes
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int j = 0; j < rpi; ++j) {
      const float4 u1 = ((float4 *)a1)[j * tpr + cg], u2 = ((float4 *)a2)[j * tpr + cg];
      t1.x += u1.x, t1.y += u1.y, t1.z += u1.z, t1.w += u1.w;
      t2.x += u2.x, t2.y += u2.y, t2.z += u2.z, t2.w += u2.w;
    }
    const float inv = 1.0f / (float)rows;
    float4 mean, m2;
    mean.x = k.x + t1.x * inv, mean.y = k.y + t1.y * inv, mean.z = k.z + t1.z * inv, mean.w = k.w + t1.w * inv;
    m2.x = t2.x - t1.x * t1.x * inv, m2.y = t2.y - t1.y * t1.y * inv;
    m2.z = t2.z - t1.z * t1.z * inv, m2.w = t2.w - t1.w * t1.w * inv;
    // partials are stored channel-major ([2][c][blocks]) so the finalising wave reads them contiguously
    const size_t nb = gridDim.x;
    float *p0 = partial + (size_t)(4 * cg) * nb + blockIdx.x, *p1 = p0 + (size_t)c * nb;
    p0[0] = mean.x, p0[nb] = mean.y, p0[2 * nb] = mean.z, p0[3 * nb] = mean.w;
    p1[0] = m2.x, p1[nb] = m2.y, p1[2 * nb] = m2.z, p1[3 * nb] = m2.w;
  }
}

// one wave per channel: every lane folds the blocks lane, lane+64, ... with Chan's update, then the 64
// lane results meet in a fixed butterfly (deterministic)
__device__ __forceinline__ void chan_merge(float &cnt, float &mean, float &m2, float nb, float mb, float qb) {
  if (nb == 0.f) return;
  const float tot = cnt + nb, delta = mb - mean;
  mean += delta * (nb / tot);
  m2 += qb + delta * delta * (cnt * nb / tot);
  cnt = tot;
}

// counts != nullptr: rows behind partial b (partials produced elsewhere, e.g. by the GEMM that wrote x:
// dm_rowgemm_stats); else the regular tiling of bn_stats_kernel
__global__ __launch_bounds__(64) void bn_finalize_kernel(
    const float *__restrict__ partial, int blocks, long long n, int c, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ save_mean,
    float *__restrict__ save_invstd, const float *__restrict__ counts = nullptr) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  const float *pm = partial + (size_t)ch * blocks, *pq = partial + ((size_t)c + ch) * blocks;
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  const int rpb = bn_rows_per_block(n, c);
  for (int b0 = lane; b0 < blocks; b0 += 64 * 8) {      // eight partial pairs in flight, folded in block order
    float vm[8], vq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = min(b0 + 64 * u, blocks - 1);       // clamped, unconditional loads
      vm[u] = pm[b];
      vq[u] = pq[b];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 64 * u;
      if (b < blocks) {
        const float nb = counts ? counts[b] : (float)min((long long)rpb, n - (long long)b * rpb);
        chan_merge(cnt, mean, m2, nb, vm[u], vq[u]);
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float on = __shfl_xor(cnt, off), om = __shfl_xor(mean, off), oq = __shfl_xor(m2, off);
    // both partners must compute the same merged value: merge (lower lane) <- (upper lane)
    float a_n = (lane & off) ? on : cnt, a_m = (lane & off) ? om : mean, a_q = (lane & off) ? oq : m2;
    const float b_n = (lane & off) ? cnt : on, b_m = (lane & off) ? mean : om, b_q = (lane & off) ? m2 : oq;
    chan_merge(a_n, a_m, a_q, b_n, b_m, b_q);
    cnt = a_n, mean = a_m, m2 = a_q;
  }
  if (lane != 0) return;
  const float var = m2 / cnt;
  save_mean[ch] = mean;
  save_invstd[ch] = 1.0f / sqrtf(var + eps);
  if (running_mean) {   // torch: running = (1 - momentum) * running + momentum * stat, unbiased variance
    const float unbiased = cnt > 1.f ? m2 / (cnt - 1.f) : var;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, long long n, int c,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta,
                                                       const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, int relu,
                                                       float *__restrict__ y) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * c) return;
  const int ch = (int)((threadIdx.x * 4u) % (unsigned)c);   // c divides the 1024 floats of a block
  const float4 v = *(const float4 *)(x + i);
  const float4 m = *(const float4 *)(mean + ch), s = *(const float4 *)(invstd + ch);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 o;
  o.x = (v.x - m.x) * s.x * g.x + b.x, o.y = (v.y - m.y) * s.y * g.y + b.y;
  o.z = (v.z - m.z) * s.z * g.z + b.z, o.w = (v.w - m.w) * s.w * g.w + b.w;
  if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
  *(float4 *)(y + i) = o;
}

// evaluation mode (the EMA teacher): y = relu?((x - running_mean) / sqrt(running_var + eps) * gamma + beta)
// in one launch — torch runs batch_norm_calc_invstd, the transform and a separate ReLU
__global__ __launch_bounds__(256) void bn_eval_kernel(const float *__restrict__ x, long long n, int c,
                                                      const float *__restrict__ gamma,
                                                      const float *__restrict__ beta,
                                                      const float *__restrict__ mean,
                                                      const float *__restrict__ var, float eps, int relu,
                                                      float *__restrict__ y) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * c) return;
  const int ch = (int)((threadIdx.x * 4u) % (unsigned)c);
  const float4 v = *(const float4 *)(x + i);
  const float4 m = *(const float4 *)(mean + ch), q = *(const float4 *)(var + ch);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float sx = 1.0f / sqrtf(q.x + eps), sy = 1.0f / sqrtf(q.y + eps), sz = 1.0f / sqrtf(q.z + eps),
              sw = 1.0f / sqrtf(q.w + eps);
  float4 o;
  o.x = (v.x - m.x) * sx * g.x + b.x, o.y = (v.y - m.y) * sy * g.y + b.y;
  o.z = (v.z - m.z) * sz * g.z + b.z, o.w = (v.w - m.w) * sw * g.w + b.w;
  if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
  *(float4 *)(y + i) = o;
}

// backward pass 1: per block sum(dy_r) and sum(dy_r * xhat), dy_r = dy masked by the ReLU
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, long long n, int c,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, int relu, float *__restrict__ partial /*[blocks][2][c]*/) {
  extern __shared__ float sm[];
  const int tpr = c / 4, rpi = 256 / tpr;
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int rpb = bn_rows_per_block(n, c);
  const long long row0 = (long long)blockIdx.x * rpb;
  const int rows = (int)min((long long)rpb, n - row0);
  const float4 m = *(const float4 *)(mean + 4 * cg), s = *(const float4 *)(invstd + 4 * cg);
  const float4 g = gamma ? *(const float4 *)(gamma + 4 * cg) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < rpi)
    for (int r = rl; r < rows; r += rpi) {
      const float4 v = *(const float4 *)(x + (row0 + r) * c + 4 * cg);
      float4 d = *(const float4 *)(dy + (row0 + r) * c + 4 * cg);
      const float hx = (v.x - m.x) * s.x, hy = (v.y - m.y) * s.y, hz = (v.z - m.z) * s.z, hw = (v.w - m.w) * s.w;
      if (relu) {
        if (hx * g.x + b.x <= 0.f) d.x = 0.f;
        if (hy * g.y + b.y <= 0.f) d.y = 0.f;
        if (hz * g.z + b.z <= 0.f) d.z = 0.f;
        if (hw * g.w + b.w <= 0.f) d.w = 0.f;
      }
      s1.x += d.x, s1.y += d.y, s1.z += d.z, s1.w += d.w;
      s2.x += d.x * hx, s2.y += d.y * hy, s2.z += d.z * hz, s2.w += d.w * hw;
    }
  float *a1 = sm, *a2 = sm + 256 * 4;
  ((float4 *)a1)[threadIdx.x] = s1;
  ((float4 *)a2)[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < tpr) {
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int j = 0; j < rpi; ++j) {
      const float4 u1 = ((float4 *)a1)[j * tpr + cg], u2 = ((float4 *)a2)[j * tpr + cg];
      t1.x += u1.x, t1.y += u1.y, t1.z += u1.z, t1.w += u1.w;
      t2.x += u2.x, t2.y += u2.y, t2.z += u2.z, t2.w += u2.w;
    }
    const size_t nb = gridDim.x;
    float *p0 = partial + (size_t)(4 * cg) * nb + blockIdx.x, *p1 = p0 + (size_t)c * nb;
    p0[0] = t1.x, p0[nb] = t1.y, p0[2 * nb] = t1.z, p0[3 * nb] = t1.w;
    p1[0] = t2.x, p1[nb] = t2.y, p1[2 * nb] = t2.z, p1[3 * nb] = t2.w;
  }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int blocks,
                                                             int c, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  const float *p1 = partial + (size_t)ch * blocks, *p2 = partial + ((size_t)c + ch) * blocks;
  float s1 = 0.f, s2 = 0.f;
  for (int b0 = lane; b0 < blocks; b0 += 64 * 8) {      // eight partial pairs in flight, summed in block order
    float v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = min(b0 + 64 * u, blocks - 1);       // clamped, unconditional loads
      v1[u] = p1[b];
      v2[u] = p2[b];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (b0 + 64 * u < blocks) s1 += v1[u], s2 += v2[u];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float o1 = __shfl_xor(s1, off), o2 = __shfl_xor(s2, off);
    s1 = (lane & off) ? o1 + s1 : s1 + o1;     // same operand order on both partners
    s2 = (lane & off) ? o2 + s2 : s2 + o2;
  }
  if (lane == 0) {
    dbeta[ch] = s1;
    dgamma[ch] = s2;
  }
}

// backward pass 2: dx = gamma * invstd * (dy_r - mean(dy_r) - xhat * mean(dy_r * xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, long long n, int c,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ dgamma, const float *__restrict__ dbeta,
    int relu, float *__restrict__ dx) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * c) return;
  const int ch = (int)((threadIdx.x * 4u) % (unsigned)c);
  const float inv_n = 1.0f / (float)n;
  const float4 v = *(const float4 *)(x + i);
  float4 d = *(const float4 *)(dy + i);
  const float4 m = *(const float4 *)(mean + ch), s = *(const float4 *)(invstd + ch);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 sg = *(const float4 *)(dgamma + ch), sb = *(const float4 *)(dbeta + ch);
  const float hx = (v.x - m.x) * s.x, hy = (v.y - m.y) * s.y, hz = (v.z - m.z) * s.z, hw = (v.w - m.w) * s.w;
  if (relu) {
    if (hx * g.x + b.x <= 0.f) d.x = 0.f;
    if (hy * g.y + b.y <= 0.f) d.y = 0.f;
    if (hz * g.z + b.z <= 0.f) d.z = 0.f;
    if (hw * g.w + b.w <= 0.f) d.w = 0.f;
  }
  float4 o;
  o.x = g.x * s.x * (d.x - sb.x * inv_n - hx * sg.x * inv_n);
  o.y = g.y * s.y * (d.y - sb.y * inv_n - hy * sg.y * inv_n);
  o.z = g.z * s.z * (d.z - sb.z * inv_n - hz * sg.z * inv_n);
  o.w = g.w * s.w * (d.w - sb.w * inv_n - hw * sg.w * inv_n);
  *(float4 *)(dx + i) = o;
}

// ---- BatchNorm + ReLU + max over the nsample rows of a group --------------------------------------
// The last layer of a shared MLP is followed by a max over the nsample grouped references of each query
// (pointnet2_modules.py:86-90): the normalised tensor (M * nsample, C) is written only to be read once by
// the pooling, and in the backward pass the pooled gradient is scattered into a dense zero tensor of the
// same size that the BatchNorm backward then reads twice.  Here the apply pass keeps a running maximum
// (and its row) per group and channel and writes only (M, C); the backward reduces over the M arg-max
// rows and writes the dense input gradient directly from (pooled gradient, arg-max, x).
// thread = 4 channels of one group; a block covers 256 / (C/4) groups.
// EVAL: `invstd` holds the running variance, eps is added and inverted here (the EMA teacher)
template <bool EVAL>
__global__ __launch_bounds__(256) void bn_apply_max_kernel(const float *__restrict__ x, long long m, int ns, int c,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, float eps,
                                                           float *__restrict__ pooled,
                                                           unsigned char *__restrict__ arg) {
  const int tpr = c / 4, gpb = 256 / tpr;
  const int cg = threadIdx.x % tpr, gl = threadIdx.x / tpr;
  const long long grp = (long long)blockIdx.x * gpb + gl;
  if (gl >= gpb || grp >= m) return;
  const int ch = 4 * cg;
  const float4 mu = *(const float4 *)(mean + ch);
  float4 s = *(const float4 *)(invstd + ch);
  if (EVAL) s.x = 1.0f / sqrtf(s.x + eps), s.y = 1.0f / sqrtf(s.y + eps), s.z = 1.0f / sqrtf(s.z + eps), s.w = 1.0f / sqrtf(s.w + eps);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float *src = x + (size_t)grp * ns * c + ch;
  float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);      // relu output is >= 0: the first row always wins
  int ix = 0, iy = 0, iz = 0, iw = 0;
  for (int j = 0; j < ns; ++j) {
    const float4 v = *(const float4 *)(src + (size_t)j * c);
    const float ox = fmaxf((v.x - mu.x) * s.x * g.x + b.x, 0.f), oy = fmaxf((v.y - mu.y) * s.y * g.y + b.y, 0.f);
    const float oz = fmaxf((v.z - mu.z) * s.z * g.z + b.z, 0.f), ow = fmaxf((v.w - mu.w) * s.w * g.w + b.w, 0.f);
    if (ox > best.x) best.x = ox, ix = j;
    if (oy > best.y) best.y = oy, iy = j;
    if (oz > best.z) best.z = oz, iz = j;
    if (ow > best.w) best.w = ow, iw = j;
  }
  *(float4 *)(pooled + (size_t)grp * c + ch) = best;
  if (arg)
    *(uchar4 *)(arg + (size_t)grp * c + ch) = make_uchar4((unsigned char)ix, (unsigned char)iy, (unsigned char)iz,
                                                          (unsigned char)iw);
}

// backward pass 1 over the M arg-max rows only: per block sum(dy_r) and sum(dy_r * xhat)
__global__ __launch_bounds__(256) void bn_max_bwd_reduce_kernel(
    const float *__restrict__ gp, const unsigned char *__restrict__ arg, const float *__restrict__ x, long long m,
    int ns, int c, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ mean, const float *__restrict__ invstd, int groups_per_block,
    float *__restrict__ partial /*[2][c][blocks]*/, long long ldg /* row pitch of gp, floats */) {
  extern __shared__ float sm[];
  const int tpr = c / 4, rpi = 256 / tpr;
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const long long g0 = (long long)blockIdx.x * groups_per_block;
  const int groups = (int)min((long long)groups_per_block, m - g0);
  const int ch = 4 * cg;
  const float4 mu = *(const float4 *)(mean + ch), s = *(const float4 *)(invstd + ch);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < rpi)
    for (int q = rl; q < groups; q += rpi) {
      const long long grp = g0 + q;
      const uchar4 a = *(const uchar4 *)(arg + (size_t)grp * c + ch);
      float4 d = *(const float4 *)(gp + (size_t)grp * ldg + ch);
      const float *base = x + (size_t)grp * ns * c + ch;
      const float hx = (base[(size_t)a.x * c + 0] - mu.x) * s.x, hy = (base[(size_t)a.y * c + 1] - mu.y) * s.y;
      const float hz = (base[(size_t)a.z * c + 2] - mu.z) * s.z, hw = (base[(size_t)a.w * c + 3] - mu.w) * s.w;
      if (hx * g.x + b.x <= 0.f) d.x = 0.f;
      if (hy * g.y + b.y <= 0.f) d.y = 0.f;
      if (hz * g.z + b.z <= 0.f) d.z = 0.f;
      if (hw * g.w + b.w <= 0.f) d.w = 0.f;
      s1.x += d.x, s1.y += d.y, s1.z += d.z, s1.w += d.w;
      s2.x += d.x * hx, s2.y += d.y * hy, s2.z += d.z * hz, s2.w += d.w * hw;
    }
  float *a1 = sm, *a2 = sm + 256 * 4;
  ((float4 *)a1)[threadIdx.x] = s1;
  ((float4 *)a2)[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < tpr) {
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int j = 0; j < rpi; ++j) {
      const float4 u1 = ((float4 *)a1)[j * tpr + cg], u2 = ((float4 *)a2)[j * tpr + cg];
      t1.x += u1.x, t1.y += u1.y, t1.z += u1.z, t1.w += u1.w;
      t2.x += u2.x, t2.y += u2.y, t2.z += u2.z, t2.w += u2.w;
    }
    const size_t nb = gridDim.x;
    float *p0 = partial + (size_t)(4 * cg) * nb + blockIdx.x, *p1 = p0 + (size_t)c * nb;
    p0[0] = t1.x, p0[nb] = t1.y, p0[2 * nb] = t1.z, p0[3 * nb] = t1.w;
    p1[0] = t2.x, p1[nb] = t2.y, p1[2 * nb] = t2.z, p1[3 * nb] = t2.w;
  }
}

// backward pass 2: dx = gamma * invstd * (dy_r - mean(dy_r) - xhat * mean(dy_r * xhat)), dy_r non-zero only
// on the arg-max row of each (group, channel)
__global__ __launch_bounds__(256) void bn_max_bwd_apply_kernel(
    const float *__restrict__ gp, const unsigned char *__restrict__ arg, const float *__restrict__ x, long long m,
    int ns, int c, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ dgamma,
    const float *__restrict__ dbeta, float *__restrict__ dx, long long ldg) {
  const long long n = m * ns;
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * c) return;
  const int ch = (int)((threadIdx.x * 4u) % (unsigned)c);
  const long long row = i / c;
  const long long grp = row / ns;
  const int j = (int)(row - grp * ns);
  const float inv_n = 1.0f / (float)n;
  const float4 v = *(const float4 *)(x + i);
  const uchar4 a = *(const uchar4 *)(arg + (size_t)grp * c + ch);
  const float4 gv = *(const float4 *)(gp + (size_t)grp * ldg + ch);
  float4 d = make_float4(a.x == j ? gv.x : 0.f, a.y == j ? gv.y : 0.f, a.z == j ? gv.z : 0.f, a.w == j ? gv.w : 0.f);
  const float4 mu = *(const float4 *)(mean + ch), s = *(const float4 *)(invstd + ch);
  const float4 g = gamma ? *(const float4 *)(gamma + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = beta ? *(const float4 *)(beta + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 sg = *(const float4 *)(dgamma + ch), sb = *(const float4 *)(dbeta + ch);
  const float hx = (v.x - mu.x) * s.x, hy = (v.y - mu.y) * s.y, hz = (v.z - mu.z) * s.z, hw = (v.w - mu.w) * s.w;
  if (hx * g.x + b.x <= 0.f) d.x = 0.f;
  if (hy * g.y + b.y <= 0.f) d.y = 0.f;
  if (hz * g.z + b.z <= 0.f) d.z = 0.f;
  if (hw * g.w + b.w <= 0.f) d.w = 0.f;
  float4 o;
  o.x = g.x * s.x * (d.x - sb.x * inv_n - hx * sg.x * inv_n);
  o.y = g.y * s.y * (d.y - sb.y * inv_n - hy * sg.y * inv_n);
  o.z = g.z * s.z * (d.z - sb.z * inv_n - hz * sg.z * inv_n);
  o.w = g.w * s.w * (d.w - sb.w * inv_n - hw * sg.w * inv_n);
  *(float4 *)(dx + i) = o;
}

bool bn_shape_ok(long long n, int c) { return n > 0 && c >= 4 && c <= 1024 && (c % 4) == 0 && 256 % (c / 4) == 0; }

}  // namespace

extern "C" size_t dm_bn_rows_workspace_bytes(long long n, int c) {
  if (n <= 0 || c <= 0) return 256;
  const int rpb = bn_rows_per_block(n, c);
  return dm_align((size_t)((n + rpb - 1) / rpb) * 2 * c * sizeof(float)) + 256;
}

extern "C" int dm_bn_rows_forward(const float *x, long long n, int c, const float *gamma,
                                  const float *beta, float eps, float momentum, float *running_mean,
                                  float *running_var, int relu, float *y, float *save_mean,
                                  float *save_invstd, void *workspace, size_t workspace_bytes,
                                  dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !y || !save_mean || !save_invstd || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_bn_rows_workspace_bytes(n, c)) return DM_ERR_WORKSPACE;
  const int rpb = bn_rows_per_block(n, c);
  const int blocks = (int)((n + rpb - 1) / rpb);
  float *partial = (float *)workspace;
  bn_stats_kernel<<<blocks, 256, 2 * 256 * 4 * sizeof(float), st>>>(x, n, c, partial);
  DM_CHECK_LAUNCH();
  bn_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, n, c, eps, momentum,
                                                        running_mean, running_var, save_mean, save_invstd);
  DM_CHECK_LAUNCH();
  const long long quads = n * c / 4;
  bn_apply_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, st>>>(x, n, c, gamma, beta, save_mean,
                                                                   save_invstd, relu, y);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// The same with the per-column statistics already reduced to `blocks` partials (mean, M2 — layout
// (2, c, blocks)) over `counts[b]` rows each: no statistics pass over x.
extern "C" int dm_bn_rows_forward_pre(const float *x, long long n, int c, const float *gamma, const float *beta,
                                      float eps, float momentum, float *running_mean, float *running_var,
                                      int relu, float *y, float *save_mean, float *save_invstd,
                                      const float *partial, const float *counts, int blocks,
                                      dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !y || !save_mean || !save_invstd || !partial || !counts || blocks < 1) return DM_ERR_INVALID_ARG;
  bn_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, n, c, eps, momentum, running_mean, running_var,
                                       save_mean, save_invstd, counts);
  DM_CHECK_LAUNCH();
  const long long quads = n * c / 4;
  bn_apply_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, st>>>(x, n, c, gamma, beta, save_mean,
                                                                   save_invstd, relu, y);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_max_forward_pre(const float *x, long long m, int ns, int c, const float *gamma,
                                          const float *beta, float eps, float momentum, float *running_mean,
                                          float *running_var, float *pooled, unsigned char *argmax,
                                          float *save_mean, float *save_invstd, const float *partial,
                                          const float *counts, int blocks, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const long long n = m * ns;
  if (ns < 1 || ns > 255 || m < 0) return DM_ERR_INVALID_ARG;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !pooled || !argmax || !save_mean || !save_invstd || !partial || !counts || blocks < 1)
    return DM_ERR_INVALID_ARG;
  bn_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, n, c, eps, momentum, running_mean, running_var,
                                       save_mean, save_invstd, counts);
  DM_CHECK_LAUNCH();
  const int gpb = 256 / (c / 4);
  bn_apply_max_kernel<false><<<(unsigned)((m + gpb - 1) / gpb), 256, 0, st>>>(
      x, m, ns, c, gamma, beta, save_mean, save_invstd, 0.f, pooled, argmax);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_max_forward(const float *x, long long m, int ns, int c, const float *gamma,
                                      const float *beta, float eps, float momentum, float *running_mean,
                                      float *running_var, float *pooled, unsigned char *argmax,
                                      float *save_mean, float *save_invstd, void *workspace,
                                      size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const long long n = m * ns;
  if (ns < 1 || ns > 255 || m < 0) return DM_ERR_INVALID_ARG;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !pooled || !argmax || !save_mean || !save_invstd || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_bn_rows_workspace_bytes(n, c)) return DM_ERR_WORKSPACE;
  const int rpb = bn_rows_per_block(n, c);
  const int blocks = (int)((n + rpb - 1) / rpb);
  float *partial = (float *)workspace;
  bn_stats_kernel<<<blocks, 256, 2 * 256 * 4 * sizeof(float), st>>>(x, n, c, partial);
  DM_CHECK_LAUNCH();
  bn_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, n, c, eps, momentum, running_mean, running_var,
                                       save_mean, save_invstd);
  DM_CHECK_LAUNCH();
  const int gpb = 256 / (c / 4);
  bn_apply_max_kernel<false><<<(unsigned)((m + gpb - 1) / gpb), 256, 0, st>>>(
      x, m, ns, c, gamma, beta, save_mean, save_invstd, 0.f, pooled, argmax);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_eval_max(const float *x, long long m, int ns, int c, const float *gamma,
                                   const float *beta, const float *running_mean, const float *running_var,
                                   float eps, float *pooled, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (ns < 1 || ns > 255 || m < 0) return DM_ERR_INVALID_ARG;
  if (!bn_shape_ok(m * ns, c)) return m * ns == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !pooled || !running_mean || !running_var) return DM_ERR_INVALID_ARG;
  const int gpb = 256 / (c / 4);
  bn_apply_max_kernel<true><<<(unsigned)((m + gpb - 1) / gpb), 256, 0, st>>>(
      x, m, ns, c, gamma, beta, running_mean, running_var, eps, pooled, nullptr);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_max_backward_ld(const float *grad_pooled, long long ldg, const unsigned char *argmax,
                                          const float *x, long long m, int ns, int c, const float *gamma,
                                          const float *beta, const float *save_mean, const float *save_invstd,
                                          float *grad_x, float *grad_gamma, float *grad_beta, void *workspace,
                                          size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const long long n = m * ns;
  if (ns < 1 || ns > 255 || m < 0 || ldg < c || (ldg & 3) || ((size_t)grad_pooled & 15)) return DM_ERR_INVALID_ARG;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!grad_pooled || !argmax || !x || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta ||
      !workspace)
    return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_bn_rows_workspace_bytes(n, c)) return DM_ERR_WORKSPACE;
  const int gpb = bn_rows_per_block(m, c);              // groups per block of the arg-max reduction
  const int blocks = (int)((m + gpb - 1) / gpb);
  float *partial = (float *)workspace;
  bn_max_bwd_reduce_kernel<<<blocks, 256, 2 * 256 * 4 * sizeof(float), st>>>(
      grad_pooled, argmax, x, m, ns, c, gamma, beta, save_mean, save_invstd, gpb, partial, ldg);
  DM_CHECK_LAUNCH();
  bn_bwd_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, c, grad_gamma, grad_beta);
  DM_CHECK_LAUNCH();
  const long long quads = n * c / 4;
  bn_max_bwd_apply_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, st>>>(
      grad_pooled, argmax, x, m, ns, c, gamma, beta, save_mean, save_invstd, grad_gamma, grad_beta, grad_x, ldg);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_max_backward(const float *grad_pooled, const unsigned char *argmax, const float *x,
                                       long long m, int ns, int c, const float *gamma, const float *beta,
                                       const float *save_mean, const float *save_invstd, float *grad_x,
                                       float *grad_gamma, float *grad_beta, void *workspace,
                                       size_t workspace_bytes, dm_stream_t stream) {
  return dm_bn_rows_max_backward_ld(grad_pooled, c, argmax, x, m, ns, c, gamma, beta, save_mean, save_invstd, grad_x,
                                    grad_gamma, grad_beta, workspace, workspace_bytes, stream);
}

extern "C" int dm_bn_rows_eval(const float *x, long long n, int c, const float *gamma, const float *beta,
                               const float *running_mean, const float *running_var, float eps, int relu,
                               float *y, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!x || !y || !running_mean || !running_var) return DM_ERR_INVALID_ARG;
  const long long quads = n * c / 4;
  bn_eval_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, st>>>(x, n, c, gamma, beta, running_mean,
                                                                  running_var, eps, relu, y);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bn_rows_backward(const float *grad_out, const float *x, long long n, int c,
                                   const float *gamma, const float *beta, const float *save_mean,
                                   const float *save_invstd, int relu, float *grad_x,
                                   float *grad_gamma, float *grad_beta, void *workspace,
                                   size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!bn_shape_ok(n, c)) return n == 0 ? DM_OK : DM_ERR_UNSUPPORTED;
  if (!grad_out || !x || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta || !workspace)
    return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_bn_rows_workspace_bytes(n, c)) return DM_ERR_WORKSPACE;
  const int rpb = bn_rows_per_block(n, c);
  const int blocks = (int)((n + rpb - 1) / rpb);
  float *partial = (float *)workspace;
  bn_bwd_reduce_kernel<<<blocks, 256, 2 * 256 * 4 * sizeof(float), st>>>(
      grad_out, x, n, c, gamma, beta, save_mean, save_invstd, relu, partial);
  DM_CHECK_LAUNCH();
  bn_bwd_finalize_kernel<<<c, 64, 0, st>>>(partial, blocks, c, grad_gamma, grad_beta);
  DM_CHECK_LAUNCH();
  const long long quads = n * c / 4;
  bn_bwd_apply_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, st>>>(
      grad_out, x, n, c, gamma, beta, save_mean, save_invstd, grad_gamma, grad_beta, relu, grad_x);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
