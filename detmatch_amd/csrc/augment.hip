// Fused 3D augmentation of a batch of LiDAR point clouds for gfx950 (the point part of
// the TS_SSL_Dataset pipelines, on the device).
//
// Replaces, per "view" (one sample as one of the student / teacher pipelines sees it), the chain
//   RandomFlip3D            mmdet3d/datasets/pipelines/transforms_3d.py:102-127 -> lidar_points.py:28-33
//   GlobalRotScaleTrans     transforms_3d.py:566-690 -> base_points.py:139-205,263-269
//   PointsRangeFilter       transforms_3d.py:783-797 -> base_points.py:207-229 (strict inequalities)
//   PointShuffle            transforms_3d.py:695-712 -> base_points.py:129-137
// which the reference runs on the host as ~8 whole-tensor torch ops per sample inside the data-loader
// workers.  Here ONE pair of launches handles every view of the batch: each point is read once,
// flipped, rotated (p' = p @ M, products accumulated in k order with fused multiply-adds, as the
// BLAS kernel behind the reference's (N,3)@(3,3) does), scaled, translated, range-tested and written
// to its compacted slot.  Several views may read the same source segment (student + teacher).
//
// Compaction is stable in SLOT order; slot j of a view reads source row perm[j] (or j): a uniformly
// random `perm` therefore yields a uniformly random order of the kept points, which is what
// filter-then-shuffle produces.  Two launches: per-256-slot chunk keep counts, then scatter with the
// chunk base recomputed from the counts (no atomics, deterministic).  HBM-bound: 2 reads + 1 write of
// a row per slot; at KITTI size (8 views x 20 k points) it is launch-latency sized.
#include "dm_common.h"

namespace {

constexpr int kChunk = 256;

struct AugViews {
  int n_views;
  int src_off[DM_AUG_MAX_VIEWS];
  int len[DM_AUG_MAX_VIEWS];
  int dst_off[DM_AUG_MAX_VIEWS];
  int first_chunk[DM_AUG_MAX_VIEWS + 1];
};

__device__ __forceinline__ bool aug_point(const float *__restrict__ P, float &x, float &y, float &z) {
  if (P[0] != 0.f) y = -y;                       // 'HF': lidar_points.py:31
  if (P[1] != 0.f) x = -x;                       // 'VF': lidar_points.py:33
  float rx = __fmul_rn(x, P[2]);                 // p @ M, M row-major at P[2..10]
  rx = fmaf(y, P[5], rx);
  rx = fmaf(z, P[8], rx);
  float ry = __fmul_rn(x, P[3]);
  ry = fmaf(y, P[6], ry);
  ry = fmaf(z, P[9], ry);
  float rz = __fmul_rn(x, P[4]);
  rz = fmaf(y, P[7], rz);
  rz = fmaf(z, P[10], rz);
  x = __fadd_rn(__fmul_rn(rx, P[11]), P[12]);    // *= scale ; += trans
  y = __fadd_rn(__fmul_rn(ry, P[11]), P[13]);
  z = __fadd_rn(__fmul_rn(rz, P[11]), P[14]);
  return x > P[15] && y > P[16] && z > P[17] && x < P[18] && y < P[19] && z < P[20];
}

template <bool SCATTER>
__global__ __launch_bounds__(kChunk) void aug_kernel(const float *__restrict__ points, int n_feat,
                                                     AugViews views, const float *__restrict__ params,
                                                     const int *__restrict__ perm,
                                                     int *__restrict__ chunk_counts, float *__restrict__ out,
                                                     int *__restrict__ out_counts) {
  __shared__ int s_view;
  __shared__ int s_wave[kChunk / DM_WAVE];
  __shared__ int s_red[kChunk / DM_WAVE];
  __shared__ float s_par[DM_AUG_PARAMS];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int v = 0;
    while (v + 1 < views.n_views && (int)blockIdx.x >= views.first_chunk[v + 1]) ++v;
    s_view = v;
  }
  __syncthreads();
  const int v = s_view;
  if (tid < DM_AUG_PARAMS) s_par[tid] = params[v * DM_AUG_PARAMS + tid];
  const int chunk = blockIdx.x - views.first_chunk[v];
  const int j = chunk * kChunk + tid;              // slot inside the view
  const int len = views.len[v];
  // base of this chunk = kept slots of the view's earlier chunks
  int base = 0;
  if (SCATTER) {
    int part = 0;
    for (int c = tid; c < chunk; c += kChunk) part += chunk_counts[views.first_chunk[v] + c];
    for (int o = DM_WAVE / 2; o > 0; o >>= 1) part += __shfl_down(part, o, DM_WAVE);
    if ((tid & (DM_WAVE - 1)) == 0) s_red[tid / DM_WAVE] = part;
  }
  __syncthreads();
  if (SCATTER) {
    for (int w = 0; w < kChunk / DM_WAVE; ++w) base += s_red[w];
  }
  bool keep = false;
  float x = 0.f, y = 0.f, z = 0.f, w4 = 0.f;
  const float *row = nullptr;
  if (j < len) {
    const int src = perm ? perm[views.dst_off[v] + j] : j;
    row = points + (size_t)(views.src_off[v] + src) * n_feat;
    if (n_feat == 4) {                               // KITTI rows [x,y,z,intensity]: one 16-byte load
      const float4 r = *(const float4 *)row;
      x = r.x, y = r.y, z = r.z, w4 = r.w;
    } else {
      x = row[0], y = row[1], z = row[2];
    }
    keep = aug_point(s_par, x, y, z);
  }
  const unsigned long long ballot = __ballot(keep);
  const int lane = tid & (DM_WAVE - 1), wave = tid / DM_WAVE;
  if (lane == 0) s_wave[wave] = __popcll(ballot);
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < kChunk / DM_WAVE; ++w) {
    if (w < wave) before += s_wave[w];
    total += s_wave[w];
  }
  if (!SCATTER) {
    if (tid == 0) chunk_counts[blockIdx.x] = total;
    return;
  }
  if (keep) {
    const int rank = base + before + __popcll(ballot & ((1ull << lane) - 1ull));
    float *dst = out + (size_t)(views.dst_off[v] + rank) * n_feat;
    if (n_feat == 4) {
      *(float4 *)dst = make_float4(x, y, z, w4);
    } else {
      dst[0] = x, dst[1] = y, dst[2] = z;
      for (int c = 3; c < n_feat; ++c) dst[c] = row[c];
    }
  }
  if (tid == 0 && chunk == (len + kChunk - 1) / kChunk - 1) out_counts[v] = base + total;
}

}  // namespace

extern "C" size_t dm_points_augment_workspace_bytes(int n_views, const int *src_len) {
  long long chunks = 0;
  for (int v = 0; v < n_views; ++v) chunks += (src_len[v] + kChunk - 1) / kChunk;
  return dm_align((size_t)(chunks + 1) * sizeof(int));
}

extern "C" int dm_points_augment(const float *points, int n_feat, int n_views, const int *src_off,
                                 const int *src_len, const int *dst_off, const float *params,
                                 const int *perm, float *out, int *out_counts, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  if (n_views < 0 || n_views > DM_AUG_MAX_VIEWS || n_feat < 3) return DM_ERR_INVALID_ARG;
  if (n_views == 0) return DM_OK;
  if (!points || !src_off || !src_len || !dst_off || !params || !out || !out_counts || !workspace)
    return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_points_augment_workspace_bytes(n_views, src_len)) return DM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  AugViews views;
  views.n_views = n_views;
  int chunks = 0;
  for (int v = 0; v < n_views; ++v) {
    if (src_len[v] < 0 || src_off[v] < 0 || dst_off[v] < 0) return DM_ERR_INVALID_ARG;
    views.src_off[v] = src_off[v];
    views.len[v] = src_len[v];
    views.dst_off[v] = dst_off[v];
    views.first_chunk[v] = chunks;
    chunks += (src_len[v] + kChunk - 1) / kChunk;
  }
  for (int v = n_views; v <= DM_AUG_MAX_VIEWS; ++v) views.first_chunk[v] = chunks;
  // views without points never run a block: their count is written here
  DM_HIP(hipMemsetAsync(out_counts, 0, sizeof(int) * n_views, st));
  if (chunks == 0) return DM_OK;
  int *chunk_counts = (int *)workspace;
  hipLaunchKernelGGL(aug_kernel<false>, dim3(chunks), dim3(kChunk), 0, st, points, n_feat, views, params,
                     perm, chunk_counts, out, out_counts);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(aug_kernel<true>, dim3(chunks), dim3(kChunk), 0, st, points, n_feat, views, params,
                     perm, chunk_counts, out, out_counts);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
