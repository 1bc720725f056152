// Hard voxelization (+ fused MeanVFE) for gfx950, whole batch in one pass.
//
// Replaces the reference's per-sample GPU path
//   mmdet3d/ops/voxel/src/voxelization_cuda.cu:184-326
// (O(N^2) duplicate scan :106-147, serial <<<1,1>>> voxel numbering :150-180,
// four device syncs + a D2H per sample) with a sort/scan formulation that is
// bit-exact with the CPU semantics (voxelization_cpu.cpp:44-103):
//   1. key_i   = sample*vol + (z*Y + y)*X + x of floor((p - min)/size)  (fp32 DIVISION)
//   2. stable radix sort of (key, i)  -> points of a voxel contiguous, ascending i
//   3. segment heads: first point of every voxel; slot-in-voxel = rank in segment
//   4. exclusive scan of "is first point" in POINT order -> voxel id = order of
//      first occurrence; voxels >= max_voxels (per sample) are dropped entirely
//   5. scatter points / coors / counts / mean features
// No dense coor_to_voxelidx grid (360 MB in the reference), no host sync.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "dm_common.h"

namespace {

struct VoxGeom {
  float vsize[3];
  float rmin[3];
  int grid[3];  // x, y, z
  uint32_t vol;
  int batch;
  int max_points;
  int max_voxels;
  DmBatchOffsets offs;
};

__device__ __forceinline__ int sample_of(const DmBatchOffsets &o, int batch, int i) {
  int b = 0;
  while (b + 1 < batch && i >= o.off[b + 1]) ++b;
  return b;
}

__global__ __launch_bounds__(256) void vox_keys(const float *points, int n, int c, VoxGeom g,
                                                uint32_t invalid_key, uint32_t *keys,
                                                int32_t *idx) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *p = points + (size_t)i * c;
  int cc[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // voxelization_cpu.cpp:23 / voxelization_cuda.cu:37,43,50 — division, then floor
    float q = __fdiv_rn(__fsub_rn(p[j], g.rmin[j]), g.vsize[j]);
    int v = (int)floorf(q);
    if (v < 0 || v >= g.grid[j]) ok = false;
    cc[j] = v;
  }
  uint32_t key = invalid_key;
  if (ok) {
    int b = sample_of(g.offs, g.batch, i);
    key = (uint32_t)b * g.vol + ((uint32_t)cc[2] * g.grid[1] + cc[1]) * g.grid[0] + cc[0];
  }
  keys[i] = key;
  idx[i] = i;
}

// head flags in sorted order; seg_head[p] = p if p starts a voxel else 0 (max-scanned later);
// is_first[point] = 1 for the first point (lowest index) of every voxel.
__global__ __launch_bounds__(256) void vox_heads(const uint32_t *keys_s, const int32_t *idx_s,
                                                 int n, uint32_t invalid_key, int32_t *seg_head,
                                                 int32_t *is_first) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  uint32_t k = keys_s[p];
  bool head = (k != invalid_key) && (p == 0 || keys_s[p - 1] != k);
  seg_head[p] = head ? p : 0;
  if (head) is_first[idx_s[p]] = 1;
}

// per-sample voxel counts and output bases (batch is tiny: one thread)
__global__ void vox_counts(const int32_t *first_rank /*exclusive scan, n+1 entries*/,
                           VoxGeom g, int32_t *sample_base_rank, int32_t *sample_out_base,
                           int32_t *voxel_counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int total = 0;
  for (int b = 0; b < g.batch; ++b) {
    int r0 = first_rank[g.offs.off[b]];
    int r1 = first_rank[g.offs.off[b + 1]];
    int cnt = r1 - r0;
    if (g.max_voxels >= 0 && cnt > g.max_voxels) cnt = g.max_voxels;
    sample_base_rank[b] = r0;
    sample_out_base[b] = total;
    voxel_counts[b] = cnt;
    total += cnt;
  }
  voxel_counts[g.batch] = total;
}

__global__ __launch_bounds__(256) void vox_scatter(const float *points, int n, int c, VoxGeom g,
                                                   uint32_t invalid_key, const uint32_t *keys_s,
                                                   const int32_t *idx_s, const int32_t *seg_start,
                                                   const int32_t *first_rank,
                                                   const int32_t *sample_base_rank,
                                                   const int32_t *sample_out_base, int coor_dim,
                                                   float *voxels, int32_t *coors,
                                                   int32_t *num_points, float *mean_feats) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  uint32_t key = keys_s[p];
  if (key == invalid_key) return;
  int s = seg_start[p];
  int h = idx_s[s];  // first point of this voxel
  uint32_t b = key / g.vol;
  int vid = first_rank[h] - sample_base_rank[b];
  if (g.max_voxels >= 0 && vid >= g.max_voxels) return;  // voxelization_cpu.cpp:76-81
  int row = sample_out_base[b] + vid;
  int slot = p - s;
  int i = idx_s[p];
  if (slot < g.max_points) {  // voxelization_cpu.cpp:89-95
    const float *src = points + (size_t)i * c;
    float *dst = voxels + ((size_t)row * g.max_points + slot) * c;
    for (int j = 0; j < c; ++j) dst[j] = src[j];
  }
  if (slot == 0) {
    uint32_t cell = key % g.vol;
    int x = cell % g.grid[0];
    int t = cell / g.grid[0];
    int y = t % g.grid[1];
    int z = t / g.grid[1];
    int32_t *co = coors + (size_t)row * coor_dim;
    if (coor_dim == 4) {
      co[0] = (int)b;
      co[1] = z;
      co[2] = y;
      co[3] = x;
    } else {
      co[0] = z;
      co[1] = y;
      co[2] = x;
    }
  }
  bool last = (p + 1 == n) || (keys_s[p + 1] != key);
  if (last) {
    int cnt = slot + 1;
    if (cnt > g.max_points) cnt = g.max_points;
    num_points[row] = cnt;
    if (mean_feats) {  // MeanVFE: sum over the kept points / clamp_min(num, 1)
      for (int j = 0; j < c; ++j) {
        float acc = 0.f;
        for (int q = 0; q < cnt; ++q) acc += points[(size_t)idx_s[s + q] * c + j];
        mean_feats[(size_t)row * c + j] = acc / (float)cnt;
      }
    }
  }
}

struct VoxWorkspace {
  uint32_t *keys, *keys_s;
  int32_t *idx, *idx_s, *seg_head, *seg_start, *is_first, *first_rank;
  int32_t *sample_base_rank, *sample_out_base;
  void *tmp;
  size_t tmp_bytes;
  size_t total;
};

VoxWorkspace carve(void *ws, size_t bytes, int n) {
  VoxWorkspace w;
  DmArena a(ws, bytes);
  size_t m = (size_t)n + 1;
  w.keys = a.take<uint32_t>(m);
  w.keys_s = a.take<uint32_t>(m);
  w.idx = a.take<int32_t>(m);
  w.idx_s = a.take<int32_t>(m);
  w.seg_head = a.take<int32_t>(m);
  w.seg_start = a.take<int32_t>(m);
  w.is_first = a.take<int32_t>(m);
  w.first_rank = a.take<int32_t>(m);
  w.sample_base_rank = a.take<int32_t>(DM_MAX_BATCH);
  w.sample_out_base = a.take<int32_t>(DM_MAX_BATCH);
  w.tmp_bytes = dm_align(m * 16 + (1u << 20));
  w.tmp = a.take<char>(w.tmp_bytes);
  w.total = a.off;
  return w;
}

}  // namespace

extern "C" size_t dm_hard_voxelize_workspace_bytes(int n_total, int batch) {
  (void)batch;
  if (n_total < 0) return 0;
  return carve(nullptr, 0, n_total).total;
}

extern "C" int dm_hard_voxelize(const float *points, int n_total, int c,
                                const int32_t *offsets_host, int batch,
                                const float *voxel_size_host, const float *coors_range_host,
                                int max_points, int max_voxels, int coor_dim, float *voxels,
                                int32_t *coors, int32_t *num_points, float *mean_feats,
                                int32_t *voxel_counts, void *workspace, size_t workspace_bytes,
                                dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_total < 0 || c < 3 || batch <= 0 || batch > DM_MAX_BATCH || max_points <= 0 ||
      max_voxels <= 0 || (coor_dim != 3 && coor_dim != 4) || !offsets_host || !voxel_counts ||
      !voxel_size_host || !coors_range_host)
    return DM_ERR_INVALID_ARG;
  if (offsets_host[0] != 0 || offsets_host[batch] != n_total) return DM_ERR_INVALID_ARG;
  VoxGeom g;
  unsigned long long vol = 1;
  for (int i = 0; i < 3; ++i) {
    g.vsize[i] = voxel_size_host[i];
    g.rmin[i] = coors_range_host[i];
    // voxelization_cpu.cpp:120-123: grid = round((max - min) / size) in fp32
    g.grid[i] = (int)roundf((coors_range_host[3 + i] - coors_range_host[i]) / voxel_size_host[i]);
    if (g.grid[i] <= 0) return DM_ERR_INVALID_ARG;
    vol *= (unsigned long long)g.grid[i];
  }
  if (vol * batch >= 0xFFFFFFFFull) return DM_ERR_INT32_RANGE;
  g.vol = (uint32_t)vol;
  g.batch = batch;
  g.max_points = max_points;
  g.max_voxels = max_voxels;
  for (int b = 0; b <= batch; ++b) {
    if (b > 0 && offsets_host[b] < offsets_host[b - 1]) return DM_ERR_INVALID_ARG;
    g.offs.off[b] = offsets_host[b];
  }
  size_t cap = (size_t)batch * max_voxels;
  if (!voxels || !coors || !num_points) return DM_ERR_INVALID_ARG;
  DM_HIP(hipMemsetAsync(voxels, 0, cap * max_points * c * sizeof(float), st));
  DM_HIP(hipMemsetAsync(num_points, 0, cap * sizeof(int32_t), st));
  DM_HIP(hipMemsetAsync(coors, 0, cap * coor_dim * sizeof(int32_t), st));
  if (mean_feats) DM_HIP(hipMemsetAsync(mean_feats, 0, cap * c * sizeof(float), st));
  if (n_total == 0) {
    DM_HIP(hipMemsetAsync(voxel_counts, 0, (batch + 1) * sizeof(int32_t), st));
    return DM_OK;
  }
  if (!points || !workspace) return DM_ERR_INVALID_ARG;
  VoxWorkspace w = carve(workspace, workspace_bytes, n_total);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  const int n = n_total;
  const uint32_t invalid_key = (uint32_t)(vol * batch);  // one past the largest real key
  int end_bit = 1;
  while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)invalid_key) ++end_bit;
  int nb = dm_ceil_div(n, 256);
  vox_keys<<<nb, 256, 0, st>>>(points, n, c, g, invalid_key, w.keys, w.idx);
  DM_CHECK_LAUNCH();
  size_t need = 0;
  DM_HIP(rocprim::radix_sort_pairs(nullptr, need, w.keys, w.keys_s, w.idx, w.idx_s, (size_t)n, 0,
                                   end_bit, st));
  if (need > w.tmp_bytes) return DM_ERR_WORKSPACE;
  need = w.tmp_bytes;
  DM_HIP(rocprim::radix_sort_pairs(w.tmp, need, w.keys, w.keys_s, w.idx, w.idx_s, (size_t)n, 0,
                                   end_bit, st));
  DM_HIP(hipMemsetAsync(w.is_first, 0, ((size_t)n + 1) * sizeof(int32_t), st));
  vox_heads<<<nb, 256, 0, st>>>(w.keys_s, w.idx_s, n, invalid_key, w.seg_head, w.is_first);
  DM_CHECK_LAUNCH();
  // seg_start = running max of head positions (sorted order)
  need = 0;
  DM_HIP(rocprim::inclusive_scan(nullptr, need, w.seg_head, w.seg_start, (size_t)n,
                                 rocprim::maximum<int32_t>(), st));
  if (need > w.tmp_bytes) return DM_ERR_WORKSPACE;
  need = w.tmp_bytes;
  DM_HIP(rocprim::inclusive_scan(w.tmp, need, w.seg_head, w.seg_start, (size_t)n,
                                 rocprim::maximum<int32_t>(), st));
  // voxel id = number of earlier first-points (point order); n+1 entries so that
  // first_rank[offsets[b+1]] is defined for the last sample
  need = 0;
  DM_HIP(rocprim::exclusive_scan(nullptr, need, w.is_first, w.first_rank, (int32_t)0,
                                 (size_t)n + 1, rocprim::plus<int32_t>(), st));
  if (need > w.tmp_bytes) return DM_ERR_WORKSPACE;
  need = w.tmp_bytes;
  DM_HIP(rocprim::exclusive_scan(w.tmp, need, w.is_first, w.first_rank, (int32_t)0,
                                 (size_t)n + 1, rocprim::plus<int32_t>(), st));
  vox_counts<<<1, 64, 0, st>>>(w.first_rank, g, w.sample_base_rank, w.sample_out_base,
                               voxel_counts);
  DM_CHECK_LAUNCH();
  vox_scatter<<<nb, 256, 0, st>>>(points, n, c, g, invalid_key, w.keys_s, w.idx_s, w.seg_start,
                                  w.first_rank, w.sample_base_rank, w.sample_out_base, coor_dim,
                                  voxels, coors, num_points, mean_feats);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
