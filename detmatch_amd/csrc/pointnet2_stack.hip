// Stacked-batch PointNet++ operators for gfx950: ball query, grouping (fwd / grad),
// furthest point sampling.
//
// Replaces pcdet/ops/pointnet2/pointnet2_stack/src/{ball_query,group_points,sampling}_gpu.cu.
// Index semantics are the reference's, bit for bit:
//   ball query  first `nsample` points of the query's sample with d2 < r^2 in STORAGE
//               order, remaining slots padded with the first hit (ball_query_gpu.cu:48-65)
//   FPS         start at index 0; next = argmax of the running min distance; ties go to
//               the point that the reference's block reduction would pick for its block
//               size min(2^floor(log2 n), 1024): smallest bit-reversed (k mod bs), then
//               smallest k
//               (sampling_gpu.cu:9-21,55-70)
// Distances are the FMA chain the device compiler makes of a*a+b*b+c*c, written out with
// fmaf() (the file is compiled with -ffp-contract=off, as is the CPU oracle).
#include "dm_common.h"

namespace {

__device__ __forceinline__ float dist2_fma(float dx, float dy, float dz) {
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

struct BatchCnt {
  int n[DM_MAX_BATCH];
};

// ---- ball query ---------------------------------------------------------------
// grid (query chunks, samples); a block stages 1024-point tiles of its sample's xyz in
// LDS and every thread scans them for its own query until it has nsample hits.
constexpr int BQ_TILE = 1024;

__global__ __launch_bounds__(256) void ball_query_kernel(int batch, float radius2, int nsample,
                                                         const float *__restrict__ new_xyz,
                                                         const int *__restrict__ new_cnt,
                                                         const float *__restrict__ xyz,
                                                         const int *__restrict__ xyz_cnt,
                                                         int *__restrict__ idx,
                                                         unsigned char *__restrict__ empty_mask) {
  __shared__ float tile[BQ_TILE * 3];
  __shared__ int alive_s;
  const int b = blockIdx.y;
  int q_start = 0, p_start = 0;
  for (int k = 0; k < b; ++k) {
    q_start += new_cnt[k];
    p_start += xyz_cnt[k];
  }
  const int mq = new_cnt[b], n = xyz_cnt[b];
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= mq) return;
  const bool has_q = q < mq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  int *out = idx;
  if (has_q) {
    const float *p = new_xyz + (size_t)(q_start + q) * 3;
    qx = p[0];
    qy = p[1];
    qz = p[2];
    out = idx + (size_t)(q_start + q) * nsample;
  }
  int cnt = 0;
  bool done = !has_q;
  const float *base = xyz + (size_t)p_start * 3;
  for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
    int tn = min(BQ_TILE, n - t0);
    __syncthreads();
    if (threadIdx.x == 0) alive_s = 0;
    for (int e = threadIdx.x; e < tn * 3; e += 256) tile[e] = base[(size_t)t0 * 3 + e];
    __syncthreads();
    if (!done) {
      for (int k = 0; k < tn; ++k) {
        float d2 = dist2_fma(qx - tile[k * 3 + 0], qy - tile[k * 3 + 1], qz - tile[k * 3 + 2]);
        if (d2 < radius2) {
          int pk = t0 + k;
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[l] = pk;
          out[cnt] = pk;
          ++cnt;
          if (cnt >= nsample) {
            done = true;
            break;
          }
        }
      }
      if (!done) alive_s = 1;  // benign race: any writer stores 1
    }
    __syncthreads();
    if (alive_s == 0) break;
  }
  if (has_q) {
    if (empty_mask) {  // fused post-processing of pointnet2_utils.py:36-37
      empty_mask[q_start + q] = cnt == 0;
      if (cnt == 0)
        for (int l = 0; l < nsample; ++l) out[l] = 0;
    } else if (cnt == 0) {
      out[0] = -1;  // ball_query_gpu.cu:65
    }
  }
}

// Tiled ball query: a block of four waves stages the sample's points 1024 at a time in LDS
// (coalesced dword loads, the next tile prefetched into registers while the current one is
// scanned) and every wave scans the tile 64 points at a time for its Q queries.  Hits are
// ranked with a ballot + prefix popcount, so "first nsample in storage order" is kept
// exactly.  M/Q waves instead of M/256 blocks fill 256 CUs when M = 4096 keypoints, and the
// global-load latency is paid once per 1024 points instead of once per 64 (the scan of a
// wave that loads its own 64 points is one dependent ~300 ns load per step: 94 us for 20 k
// points, against ~10 us of distance arithmetic).
template <int Q>
__global__ __launch_bounds__(256) void ball_query_wave(int batch, int m, float radius2,
                                                       int nsample,
                                                       const float *__restrict__ new_xyz,
                                                       const int *__restrict__ new_cnt,
                                                       const float *__restrict__ xyz,
                                                       const int *__restrict__ xyz_cnt,
                                                       int *__restrict__ idx,
                                                       unsigned char *__restrict__ empty_mask) {
  __shared__ float s_pts[BQ_TILE * 3];
  const int lane = threadIdx.x & 63;
  const int bq0 = blockIdx.x * 4 * Q;                          // first query of this block
  const int wq = bq0 + (threadIdx.x >> 6) * Q;                 // first query of this wave
  float qx[Q], qy[Q], qz[Q];
  int cnt[Q], first[Q], qb[Q];
  bool live[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    int q = wq + u;
    live[u] = q < m;
    int qq = live[u] ? q : m - 1;
    int b = 0, acc = new_cnt[0];
    for (int k = 1; k < batch; ++k) {
      if (qq < acc) break;
      acc += new_cnt[k];
      b = k;
    }
    qb[u] = b;
    qx[u] = new_xyz[(size_t)qq * 3 + 0];
    qy[u] = new_xyz[(size_t)qq * 3 + 1];
    qz[u] = new_xyz[(size_t)qq * 3 + 2];
    cnt[u] = 0;
    first[u] = 0;
  }
  // queries are stacked by sample: the block's queries span samples b_lo..b_hi (one, except
  // at a sample boundary)
  int b_lo = 0, b_hi = 0;
  {
    int q_last = bq0 + 4 * Q - 1 < m ? bq0 + 4 * Q - 1 : m - 1;
    int acc = 0;
    for (int k = 0; k < batch; ++k) {
      if (bq0 >= acc) b_lo = k;
      if (q_last >= acc) b_hi = k;
      acc += new_cnt[k];
    }
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  int pstart = 0;
  for (int k = 0; k < b_lo; ++k) pstart += xyz_cnt[k];
  for (int b = b_lo; b <= b_hi; ++b) {
    const int n = xyz_cnt[b];
    const float *base = xyz + (size_t)pstart * 3;
    const int n3 = n * 3;
    float r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      int e = j * 256 + (int)threadIdx.x;
      r[j] = e < n3 ? base[e] : 0.f;
    }
    for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
#pragma unroll
      for (int j = 0; j < 12; ++j) s_pts[j * 256 + threadIdx.x] = r[j];
      __syncthreads();
      if (t0 + BQ_TILE < n) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          int e = (t0 + BQ_TILE) * 3 + j * 256 + (int)threadIdx.x;
          r[j] = e < n3 ? base[e] : 0.f;
        }
      }
      bool open = false;
#pragma unroll
      for (int u = 0; u < Q; ++u) open = open || (live[u] && qb[u] == b && cnt[u] < nsample);
      if (open) {
        const int lim = n - t0 < BQ_TILE ? n - t0 : BQ_TILE;
        for (int c0 = 0; c0 < lim; c0 += 64) {
          int kk = c0 + lane;
          bool in = kk < lim;
          float x = s_pts[kk * 3 + 0], y = s_pts[kk * 3 + 1], z = s_pts[kk * 3 + 2];
          bool still = false;
#pragma unroll
          for (int u = 0; u < Q; ++u) {
            if (!(live[u] && qb[u] == b) || cnt[u] >= nsample) continue;
            float d2 = dist2_fma(qx[u] - x, qy[u] - y, qz[u] - z);
            bool hit = in && d2 < radius2;
            unsigned long long mk = __ballot(hit);
            if (mk != 0ull) {
              if (cnt[u] == 0) first[u] = t0 + c0 + __ffsll((long long)mk) - 1;
              int pos = cnt[u] + __popcll(mk & lt);
              if (hit && pos < nsample) idx[(size_t)(wq + u) * nsample + pos] = t0 + kk;
              cnt[u] += __popcll(mk);
            }
            still = still || cnt[u] < nsample;
          }
          if (!still) break;
        }
        open = false;
#pragma unroll
        for (int u = 0; u < Q; ++u) open = open || (live[u] && qb[u] == b && cnt[u] < nsample);
      }
      if (!__syncthreads_or(open ? 1 : 0)) break;
    }
    pstart += n;
  }
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    if (!live[u]) continue;
    int *out = idx + (size_t)(wq + u) * nsample;
    int c = cnt[u] < nsample ? cnt[u] : nsample;
    if (c > 0) {
      for (int l = c + lane; l < nsample; l += 64) out[l] = first[u];  // pad with first hit
      if (empty_mask && lane == 0) empty_mask[wq + u] = 0;
    } else if (empty_mask) {
      for (int l = lane; l < nsample; l += 64) out[l] = 0;
      if (lane == 0) empty_mask[wq + u] = 1;
    } else if (lane == 0) {
      out[0] = -1;  // ball_query_gpu.cu:65
    }
  }
}

// Two radii in ONE scan (the two groupers of a set-abstraction source query the same points around the
// same centres): the distance of a (query, point) pair is computed once and ranked into both hit lists.
struct BallPair {
  float radius2[2];
  int nsample[2];
  int *idx[2];
  unsigned char *empty[2];
};

template <int Q>
__global__ __launch_bounds__(256) void ball_query_wave2(int batch, int m, BallPair bp,
                                                       const float *__restrict__ new_xyz,
                                                       const int *__restrict__ new_cnt,
                                                       const float *__restrict__ xyz,
                                                       const int *__restrict__ xyz_cnt) {
  __shared__ float s_pts[BQ_TILE * 3];
  const int lane = threadIdx.x & 63;
  const int bq0 = blockIdx.x * 4 * Q;                          // first query of this block
  const int wq = bq0 + (threadIdx.x >> 6) * Q;                 // first query of this wave
  float qx[Q], qy[Q], qz[Q];
  int cnt[Q][2], first[Q][2], qb[Q];
  bool live[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    int q = wq + u;
    live[u] = q < m;
    int qq = live[u] ? q : m - 1;
    int b = 0, acc = new_cnt[0];
    for (int k = 1; k < batch; ++k) {
      if (qq < acc) break;
      acc += new_cnt[k];
      b = k;
    }
    qb[u] = b;
    qx[u] = new_xyz[(size_t)qq * 3 + 0];
    qy[u] = new_xyz[(size_t)qq * 3 + 1];
    qz[u] = new_xyz[(size_t)qq * 3 + 2];
    cnt[u][0] = cnt[u][1] = 0;
    first[u][0] = first[u][1] = 0;
  }
  // queries are stacked by sample: the block's queries span samples b_lo..b_hi (one, except
  // at a sample boundary)
  int b_lo = 0, b_hi = 0;
  {
    int q_last = bq0 + 4 * Q - 1 < m ? bq0 + 4 * Q - 1 : m - 1;
    int acc = 0;
    for (int k = 0; k < batch; ++k) {
      if (bq0 >= acc) b_lo = k;
      if (q_last >= acc) b_hi = k;
      acc += new_cnt[k];
    }
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  int pstart = 0;
  for (int k = 0; k < b_lo; ++k) pstart += xyz_cnt[k];
  for (int b = b_lo; b <= b_hi; ++b) {
    const int n = xyz_cnt[b];
    const float *base = xyz + (size_t)pstart * 3;
    const int n3 = n * 3;
    float r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      int e = j * 256 + (int)threadIdx.x;
      r[j] = e < n3 ? base[e] : 0.f;
    }
    for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
#pragma unroll
      for (int j = 0; j < 12; ++j) s_pts[j * 256 + threadIdx.x] = r[j];
      __syncthreads();
      if (t0 + BQ_TILE < n) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          int e = (t0 + BQ_TILE) * 3 + j * 256 + (int)threadIdx.x;
          r[j] = e < n3 ? base[e] : 0.f;
        }
      }
      bool open = false;
#pragma unroll
      for (int u = 0; u < Q; ++u)
        open = open || (live[u] && qb[u] == b && (cnt[u][0] < bp.nsample[0] || cnt[u][1] < bp.nsample[1]));
      if (open) {
        const int lim = n - t0 < BQ_TILE ? n - t0 : BQ_TILE;
        for (int c0 = 0; c0 < lim; c0 += 64) {
          int kk = c0 + lane;
          bool in = kk < lim;
          float x = s_pts[kk * 3 + 0], y = s_pts[kk * 3 + 1], z = s_pts[kk * 3 + 2];
          bool still = false;
#pragma unroll
          for (int u = 0; u < Q; ++u) {
            if (!(live[u] && qb[u] == b) || (cnt[u][0] >= bp.nsample[0] && cnt[u][1] >= bp.nsample[1])) continue;
            float d2 = dist2_fma(qx[u] - x, qy[u] - y, qz[u] - z);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              if (cnt[u][rr] >= bp.nsample[rr]) continue;
              bool hit = in && d2 < bp.radius2[rr];
              unsigned long long mk = __ballot(hit);
              if (mk != 0ull) {
                if (cnt[u][rr] == 0) first[u][rr] = t0 + c0 + __ffsll((long long)mk) - 1;
                int pos = cnt[u][rr] + __popcll(mk & lt);
                if (hit && pos < bp.nsample[rr]) bp.idx[rr][(size_t)(wq + u) * bp.nsample[rr] + pos] = t0 + kk;
                cnt[u][rr] += __popcll(mk);
              }
              still = still || cnt[u][rr] < bp.nsample[rr];
            }
          }
          if (!still) break;
        }
        open = false;
#pragma unroll
        for (int u = 0; u < Q; ++u)
          open = open || (live[u] && qb[u] == b && (cnt[u][0] < bp.nsample[0] || cnt[u][1] < bp.nsample[1]));
      }
      if (!__syncthreads_or(open ? 1 : 0)) break;
    }
    pstart += n;
  }
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    if (!live[u]) continue;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int nsample = bp.nsample[rr];
      int *out = bp.idx[rr] + (size_t)(wq + u) * nsample;
      int c = cnt[u][rr] < nsample ? cnt[u][rr] : nsample;
      if (c > 0) {
        for (int l = c + lane; l < nsample; l += 64) out[l] = first[u][rr];  // pad with first hit
        if (lane == 0) bp.empty[rr][wq + u] = 0;
      } else {
        for (int l = lane; l < nsample; l += 64) out[l] = 0;
        if (lane == 0) bp.empty[rr][wq + u] = 1;
      }
    }
  }
}

// ---- grouping -------------------------------------------------------------------
// One wave per query point: rows are read coalesced (lanes over channels), transposed
// through a wave-private LDS tile and written as contiguous (C, nsample) blocks.
constexpr int GP_MAX_ELEMS = 4352;  // (C+1)*nsample floats of LDS per wave, e.g. C=135, ns=32

__global__ __launch_bounds__(256) void group_points_kernel(int batch, int m, int c, int nsample,
                                                           const float *__restrict__ feats,
                                                           const int *__restrict__ feats_cnt,
                                                           const int *__restrict__ idx,
                                                           const int *__restrict__ idx_cnt,
                                                           const unsigned char *__restrict__ empty,
                                                           float *__restrict__ out) {
  extern __shared__ float gp_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pt = blockIdx.x * 4 + wave;
  if (pt >= m) return;
  float *tile = gp_lds + (size_t)wave * (c + 1) * nsample;
  int bs = 0, acc = idx_cnt[0];
  for (int k = 1; k < batch; ++k) {
    if (pt < acc) break;
    acc += idx_cnt[k];
    bs = k;
  }
  int start = 0;
  for (int k = 0; k < bs; ++k) start += feats_cnt[k];
  const bool zero = empty && empty[pt];
  for (int s = 0; s < nsample; ++s) {
    const float *row = feats + (size_t)(start + idx[(size_t)pt * nsample + s]) * c;
    for (int ci = lane; ci < c; ci += 64) tile[s * (c + 1) + ci] = zero ? 0.f : row[ci];
  }
  // wave-private tile: LDS ops of one wave complete in order; the fence only pins the
  // compiler's ordering of the cross-lane hand-off
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float *o = out + (size_t)pt * c * nsample;
  for (int e = lane; e < c * nsample; e += 64) {
    int ci = e / nsample, s = e % nsample;
    o[e] = tile[s * (c + 1) + ci];
  }
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int batch, int m, int c,
                                                                int nsample,
                                                                const float *__restrict__ grad_out,
                                                                const int *__restrict__ idx,
                                                                const int *__restrict__ idx_cnt,
                                                                const int *__restrict__ feats_cnt,
                                                                float *__restrict__ grad_feats) {
  extern __shared__ float gp_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pt = blockIdx.x * 4 + wave;
  if (pt >= m) return;
  float *tile = gp_lds + (size_t)wave * (c + 1) * nsample;
  int bs = 0, acc = idx_cnt[0];
  for (int k = 1; k < batch; ++k) {
    if (pt < acc) break;
    acc += idx_cnt[k];
    bs = k;
  }
  int start = 0;
  for (int k = 0; k < bs; ++k) start += feats_cnt[k];
  const float *g = grad_out + (size_t)pt * c * nsample;
  for (int e = lane; e < c * nsample; e += 64) {
    int ci = e / nsample, s = e % nsample;
    tile[s * (c + 1) + ci] = g[e];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // one 256-byte-contiguous atomic wave-instruction per (sample slot, 64 channels)
  for (int s = 0; s < nsample; ++s) {
    float *row = grad_feats + (size_t)(start + idx[(size_t)pt * nsample + s]) * c;
    for (int ci = lane; ci < c; ci += 64) atomicAdd(row + ci, tile[s * (c + 1) + ci]);
  }
}

// Large query sets whose neighbourhoods overlap heavily (RoI-grid pooling: 216 grid points of one
// RoI share a few dozen key-points; every RoI of a scene clusters on the same objects) make the
// per-sample global atomics above contention-bound: float atomics execute at the memory side and
// thousands of adders per row serialise (MI355X_MICROARCH.md "Global float atomics": one row
// 14x slower).  This variant sums everything that shares a destination ON CHIP first: a workgroup
// takes `chunk` consecutive queries, hashes their source rows into an LDS table (<= GPC_ACC_BYTES
// of fp32 accumulators), adds with LDS atomics, and flushes each distinct row once.
bool g_gp_grad_combine = true;
constexpr int GPC_ACC_BYTES = 120 * 1024;
constexpr int GPC_MAX_SLOTS = 512;

__global__ __launch_bounds__(256) void group_points_grad_combine(int batch, int m, int c, int nsample,
                                                                 int chunk, int n_slots,
                                                                 const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx,
                                                                 const int *__restrict__ idx_cnt,
                                                                 const int *__restrict__ feats_cnt,
                                                                 float *__restrict__ grad_feats) {
  extern __shared__ float gpc_lds[];
  float *acc = gpc_lds;                                   // [n_slots][c]
  int *keys = (int *)(acc + (size_t)n_slots * c);         // [n_slots] global source row or -1
  unsigned short *slot_of = (unsigned short *)(keys + n_slots);   // [chunk * nsample]
  const int q0 = blockIdx.x * chunk;
  const int nq = min(chunk, m - q0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < n_slots * c; i += 256) acc[i] = 0.0f;
  for (int i = tid; i < n_slots; i += 256) keys[i] = -1;
  __syncthreads();
  // ---- 1. hash every reference of the chunk to a slot --------------------------------------
  for (int r = tid; r < nq * nsample; r += 256) {
    const int q = q0 + r / nsample;
    int bs = 0, upto = idx_cnt[0], start = 0;
    for (int k = 1; k < batch && q >= upto; ++k) {
      start += feats_cnt[k - 1];
      upto += idx_cnt[k];
      bs = k;
    }
    (void)bs;
    const int src = start + idx[(size_t)q0 * nsample + r];
    unsigned h = ((unsigned)src * 2654435761u) % (unsigned)n_slots;
    int slot = 0xFFFF;
    for (int probe = 0; probe < n_slots; ++probe) {
      const int old = atomicCAS(&keys[h], -1, src);
      if (old == -1 || old == src) {
        slot = (int)h;
        break;
      }
      h = h + 1 == (unsigned)n_slots ? 0u : h + 1;
    }
    slot_of[r] = (unsigned short)slot;
  }
  __syncthreads();
  // ---- 2. accumulate: one wave per query, coalesced reads of its (c, nsample) tile ----------
  for (int ql = wave; ql < nq; ql += 4) {
    const float *g = grad_out + (size_t)(q0 + ql) * c * nsample;
    const unsigned short *so = slot_of + ql * nsample;
    for (int e = lane; e < c * nsample; e += 64) {
      const int ci = e / nsample, s = e % nsample;
      const int slot = so[s];
      const float v = g[e];
      if (slot != 0xFFFF) {
        unsafeAtomicAdd(&acc[slot * c + ci], v);
      } else {   // table full (not expected for overlapping neighbourhoods): straight to memory
        const int q = q0 + ql;
        int upto = idx_cnt[0], start = 0;
        for (int k = 1; k < batch && q >= upto; ++k) {
          start += feats_cnt[k - 1];
          upto += idx_cnt[k];
        }
        unsafeAtomicAdd(grad_feats + (size_t)(start + idx[(size_t)q * nsample + s]) * c + ci, v);
      }
    }
  }
  __syncthreads();
  // ---- 3. flush each distinct row once: contiguous row segments per wave instruction ---------
  for (int f = tid; f < n_slots * c; f += 256) {
    const int slot = f / c, ci = f - slot * c;
    const int key = keys[slot];
    if (key >= 0) unsafeAtomicAdd(grad_feats + (size_t)key * c + ci, acc[f]);
  }
}

// ---- fused QueryAndGroup in row layout ---------------------------------------------------------
// pointnet2_utils.py:119-156 produces (M, 3+C, nsample) with three launches + cat and the shared
// MLP then runs as 1x1 Conv2d on a (1, C, M, nsample) view (a strided copy first).  In row layout
// out (M, nsample, 3+C) a reference is one contiguous row: the gather is a coalesced row copy, the
// MLP a plain GEMM on (M*nsample, C) rows and BatchNorm a column reduction — no transposes, no cat.
__device__ __forceinline__ int sample_start(int q, int batch, const int *__restrict__ q_cnt,
                                            const int *__restrict__ src_cnt) {
  int upto = q_cnt[0], start = 0;
  for (int k = 1; k < batch && q >= upto; ++k) {
    start += src_cnt[k - 1];
    upto += q_cnt[k];
  }
  return start;
}

__global__ __launch_bounds__(256) void query_group_rows_kernel(
    int batch, int m, int c, int nsample, int use_xyz, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ feats,
    const int *__restrict__ xyz_cnt, const int *__restrict__ new_cnt, const int *__restrict__ idx,
    const unsigned char *__restrict__ empty, float *__restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + wave;     // reference = (query, sample)
  if (r >= (long long)m * nsample) return;
  const int q = (int)(r / nsample);
  const int xoff = use_xyz ? 4 : 0;        // xyz occupies a 16-byte slot: [dx, dy, dz, 0]
  const int width = xoff + c;
  float *o = out + r * width;
  if (empty && empty[q]) {
    for (int e = lane; e < width; e += 64) o[e] = 0.0f;
    return;
  }
  const int src = sample_start(q, batch, new_cnt, xyz_cnt) + idx[r];
  if (use_xyz && lane < 4)
    o[lane] = lane < 3 ? xyz[(size_t)src * 3 + lane] - new_xyz[(size_t)q * 3 + lane] : 0.0f;
  const float *f = feats + (size_t)src * c;
  float *of = o + xoff;
  if ((c & 3) == 0) {                      // rows and feature blocks are 16-byte aligned
    for (int e = lane; e < c / 4; e += 64) ((float4 *)of)[e] = ((const float4 *)f)[e];
  } else {
    for (int e = lane; e < c; e += 64) of[e] = f[e];
  }
}

// The same for rows made of float4 quads (use_xyz, C % 4 == 0: [dx dy dz 0][C/4 quads]): a half-wave per
// reference, 16 references per wave — two rows per load / store instruction and eight dependent
// idx -> row chains in flight per wave instead of one (one wave per 528-byte row: 120 us for the 467 MB of
// RoI-grid pooling).
__global__ __launch_bounds__(256) void query_group_rows_quads_kernel(
    int batch, int m, int cq /*C / 4*/, int nsample, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ feats, const int *__restrict__ xyz_cnt,
    const int *__restrict__ new_cnt, const int *__restrict__ idx, const unsigned char *__restrict__ empty,
    float *__restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l = lane & 31;
  const long long refs = (long long)m * nsample;
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 16;
  const int quads = cq + 1;
#pragma unroll 4
  for (int i = 0; i < 8; ++i) {
    const long long r = r0 + 2 * i + half;
    if (r >= refs) continue;
    const int q = (int)(r / nsample);
    float4 *o = (float4 *)(out + r * (size_t)(4 * quads));
    if (empty && empty[q]) {
      for (int j = l; j < quads; j += 32) o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const int src = sample_start(q, batch, new_cnt, xyz_cnt) + idx[r];
    const float4 *f = (const float4 *)(feats + (size_t)src * 4 * cq);
    for (int j = l; j < quads; j += 32) {
      float4 v;
      if (j == 0) {
        v.x = xyz[(size_t)src * 3 + 0] - new_xyz[(size_t)q * 3 + 0];
        v.y = xyz[(size_t)src * 3 + 1] - new_xyz[(size_t)q * 3 + 1];
        v.z = xyz[(size_t)src * 3 + 2] - new_xyz[(size_t)q * 3 + 2];
        v.w = 0.f;
      } else {
        v = f[j - 1];
      }
      o[j] = v;
    }
  }
}

// backward: grad_feats[src, :] += grad_out[r, col_off : col_off + c].  Small query sets: one
// coalesced row of global atomics per reference.  Large ones (RoI-grid pooling): per-workgroup LDS
// hash of the distinct source rows first (see group_points_grad_combine), flush once per row.
__global__ __launch_bounds__(256) void group_rows_grad_direct(
    int batch, int m, int c, int nsample, int width, int col_off, const float *__restrict__ gout,
    const int *__restrict__ idx, const int *__restrict__ q_cnt, const int *__restrict__ src_cnt,
    const unsigned char *__restrict__ empty, float *__restrict__ gfeats) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= (long long)m * nsample) return;
  const int q = (int)(r / nsample);
  if (empty && empty[q]) return;
  const int src = sample_start(q, batch, q_cnt, src_cnt) + idx[r];
  const float *g = gout + r * width + col_off;
  float *dst = gfeats + (size_t)src * c;
  for (int e = lane; e < c; e += 64) unsafeAtomicAdd(dst + e, g[e]);
}

__global__ __launch_bounds__(256) void group_rows_grad_combine(
    int batch, int m, int c, int nsample, int width, int col_off, int chunk, int n_slots,
    const float *__restrict__ gout, const int *__restrict__ idx, const int *__restrict__ q_cnt,
    const int *__restrict__ src_cnt, const unsigned char *__restrict__ empty,
    float *__restrict__ gfeats) {
  // Counting sort of the chunk's references by source row (LDS hash -> slot, per-slot counts,
  // prefix sum, scatter), then every wave sums the rows of "its" slots in REGISTERS and issues one
  // global atomic row per distinct source.  (A first version accumulated in LDS with ds_add_f32:
  // LDS float atomics retire lane-serially and were the bottleneck.)
  extern __shared__ int gps_lds[];
  int *keys = gps_lds;                       // [n_slots] source row or -1
  int *cnt = keys + n_slots;                 // [n_slots] references per slot, then scatter cursor
  int *first = cnt + n_slots;                // [n_slots + 1] start of the slot's run in `sorted`
  short *slot_of = (short *)(first + n_slots + 1);          // [chunk * nsample]
  unsigned short *sorted = (unsigned short *)(slot_of + chunk * nsample);   // [chunk * nsample]
  const int q0 = blockIdx.x * chunk;
  const int nq = min(chunk, m - q0);
  const int n_refs = nq * nsample;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < n_slots; i += 256) keys[i] = -1, cnt[i] = 0;
  __syncthreads();
  for (int rl = tid; rl < n_refs; rl += 256) {
    const int q = q0 + rl / nsample;
    int slot = -2;                                        // -2: skip (empty ball)
    if (!(empty && empty[q])) {
      const int src = sample_start(q, batch, q_cnt, src_cnt) + idx[(long long)q0 * nsample + rl];
      unsigned h = ((unsigned)src * 2654435761u) % (unsigned)n_slots;
      slot = -1;                                          // -1: table full
      for (int probe = 0; probe < n_slots; ++probe) {
        const int old = atomicCAS(&keys[h], -1, src);
        if (old == -1 || old == src) {
          slot = (int)h;
          break;
        }
        h = h + 1 == (unsigned)n_slots ? 0u : h + 1;
      }
      if (slot >= 0) atomicAdd(&cnt[slot], 1);
    }
    slot_of[rl] = (short)slot;
  }
  __syncthreads();
  if (wave == 0) {                           // exclusive prefix sum of cnt over the slots
    const int per = (n_slots + 63) / 64;
    int sum = 0;
    for (int j = 0; j < per; ++j) {
      const int sidx = lane * per + j;
      if (sidx < n_slots) sum += cnt[sidx];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    int run = incl - sum;
    for (int j = 0; j < per; ++j) {
      const int sidx = lane * per + j;
      if (sidx < n_slots) {
        first[sidx] = run;
        run += cnt[sidx];
      }
    }
    if (lane == 63) first[n_slots] = incl;
  }
  __syncthreads();
  for (int i = tid; i < n_slots; i += 256) cnt[i] = first[i];   // scatter cursors
  __syncthreads();
  for (int rl = tid; rl < n_refs; rl += 256) {
    const int slot = slot_of[rl];
    if (slot >= 0) {
      sorted[atomicAdd(&cnt[slot], 1)] = (unsigned short)rl;
    } else if (slot == -1) {     // table full (incoherent neighbourhoods): straight to memory
      const int q = q0 + rl / nsample;
      const long long r = (long long)q0 * nsample + rl;
      float *dst = gfeats + (size_t)(sample_start(q, batch, q_cnt, src_cnt) + idx[r]) * c;
      const float *g = gout + r * width + col_off;
      for (int e = 0; e < c; ++e) unsafeAtomicAdd(dst + e, g[e]);
    }
  }
  __syncthreads();
  const float *gbase = gout + (long long)q0 * nsample * width + col_off;
  for (int slot = wave; slot < n_slots; slot += 4) {
    const int lo = first[slot], hi = first[slot + 1];
    if (hi == lo) continue;
    float *dst = gfeats + (size_t)keys[slot] * c;
    if (((c | width | col_off) & 3) == 0) {
      // 16-byte aligned rows: one float4 per lane covers 256 channels per sweep
      for (int e0 = 0; e0 < c / 4; e0 += 64) {
        const int e = e0 + lane;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < c / 4)
          for (int i = lo; i < hi; i += 4) {             // four rows in flight
            const float4 v0 = ((const float4 *)(gbase + (long long)sorted[i] * width))[e];
            const float4 v1 = ((const float4 *)(gbase + (long long)sorted[min(i + 1, hi - 1)] * width))[e];
            const float4 v2 = ((const float4 *)(gbase + (long long)sorted[min(i + 2, hi - 1)] * width))[e];
            const float4 v3 = ((const float4 *)(gbase + (long long)sorted[min(i + 3, hi - 1)] * width))[e];
            const float w1 = i + 1 < hi ? 1.f : 0.f, w2 = i + 2 < hi ? 1.f : 0.f, w3 = i + 3 < hi ? 1.f : 0.f;
            a.x += v0.x + w1 * v1.x + (w2 * v2.x + w3 * v3.x);
            a.y += v0.y + w1 * v1.y + (w2 * v2.y + w3 * v3.y);
            a.z += v0.z + w1 * v1.z + (w2 * v2.z + w3 * v3.z);
            a.w += v0.w + w1 * v1.w + (w2 * v2.w + w3 * v3.w);
          }
        if (e < c / 4) {
          unsafeAtomicAdd(dst + 4 * e, a.x);
          unsafeAtomicAdd(dst + 4 * e + 1, a.y);
          unsafeAtomicAdd(dst + 4 * e + 2, a.z);
          unsafeAtomicAdd(dst + 4 * e + 3, a.w);
        }
      }
      continue;
    }
    for (int e0 = 0; e0 < c; e0 += 256) {              // <= 4 channels per lane per sweep
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const int e = e0 + lane;
      for (int i = lo; i < hi; i += 4) {               // four rows in flight
        const float *g0 = gbase + (long long)sorted[i] * width;
        const float *g1 = gbase + (long long)sorted[min(i + 1, hi - 1)] * width;
        const float *g2 = gbase + (long long)sorted[min(i + 2, hi - 1)] * width;
        const float *g3 = gbase + (long long)sorted[min(i + 3, hi - 1)] * width;
        const float w1 = i + 1 < hi ? 1.f : 0.f, w2 = i + 2 < hi ? 1.f : 0.f, w3 = i + 3 < hi ? 1.f : 0.f;
        if (e < c) a0 += g0[e] + w1 * g1[e] + (w2 * g2[e] + w3 * g3[e]);
        if (e + 64 < c) a1 += g0[e + 64] + w1 * g1[e + 64] + (w2 * g2[e + 64] + w3 * g3[e + 64]);
        if (e + 128 < c) a2 += g0[e + 128] + w1 * g1[e + 128] + (w2 * g2[e + 128] + w3 * g3[e + 128]);
        if (e + 192 < c) a3 += g0[e + 192] + w1 * g1[e + 192] + (w2 * g2[e + 192] + w3 * g3[e + 192]);
      }
      if (e < c) unsafeAtomicAdd(dst + e, a0);
      if (e + 64 < c) unsafeAtomicAdd(dst + e + 64, a1);
      if (e + 128 < c) unsafeAtomicAdd(dst + e + 128, a2);
      if (e + 192 < c) unsafeAtomicAdd(dst + e + 192, a3);
    }
  }
}

// ---- furthest point sampling ----------------------------------------------------
// One 1024-thread workgroup per sample; each thread keeps its points (xyz + running min
// distance) in registers for all m-1 rounds, so a round is pure VALU + one cross-wave
// reduction — no global traffic inside the loop.
struct FpsBest {
  float d;
  int k;
};

__device__ __forceinline__ bool fps_better(float d2, int k2, float d1, int k1, int bs_mask) {
  // reference tie rule: larger distance; then the stride class (k mod block_size) that the
  // shared-memory tree keeps — at the level where two classes first meet the one in the
  // lower half survives, i.e. the smaller BIT-REVERSED class index; then smaller k
  if (d2 != d1) return d2 > d1;
  unsigned int c2 = __brev((unsigned int)(k2 & bs_mask)), c1 = __brev((unsigned int)(k1 & bs_mask));
  if (c2 != c1) return c2 < c1;
  return k2 < k1;
}

struct FpsSamples {
  int off[DM_MAX_BATCH + 1];  // sample b owns points [off[b], off[b+1]) of the stacked xyz
};

// 512 threads = 2 waves per SIMD = 256 registers per lane: up to 48 points per thread
// (n <= 24576) stay in registers without spilling.
constexpr int FPS_T = 512;

// ---- round 4: the same sampling with the per-round critical path cut to the arithmetic --------------------------
// The kernel of rounds 1-3 (removed in round 5) carried (distance, index) pairs through the scan (a compare + two
// selects per point), through twelve ds_bpermute steps per wave and a second shuffle tree on wave 0, and fetched the
// winner's coordinates from global memory behind three barriers: 3.3 us per round at KITTI size (20 k points), of
// which the arithmetic is a third.  Here a round reduces the VALUE only:
//   * the scan updates two points per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: the same IEEE operations per
//     element as dist2_fma) and keeps a running maximum with one v_max3 per pair: 4.5 instead of ~10 instructions
//     per point;
//   * the wave maximum is six DPP steps (no LDS), the workgroup maximum is one LDS exchange: every wave reads all
//     eight wave maxima and reduces them itself — one barrier;
//   * only then the winning INDEX is recovered: the threads whose maximum equals the workgroup's look up which of
//     their points it was and compete with the reference's tie rule as a 64-bit key through one LDS atomic min
//     (normally one thread takes part) — second barrier; everybody reads the key and fetches the winner's
//     coordinates with a wave-uniform (scalar) load.
// Same total order as the reference (larger distance; bit-reversed stride class; smaller index), so the indices are
// the oracle's bit for bit (tests/test_ops_gpu.py).
typedef float fps_f2 __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float fps_dpp_max(float v) {
  const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, o));
}
// maximum over the wave, valid in lane 63 (row butterflies, then the last lane of each row broadcast downstream)
__device__ __forceinline__ float fps_wave_max(float v) {
  v = fps_dpp_max<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
  v = fps_dpp_max<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
  v = fps_dpp_max<0x141, 0xf>(v);    // row_half_mirror
  v = fps_dpp_max<0x140, 0xf>(v);    // row_mirror: every lane of a row holds the row's maximum
  v = fps_dpp_max<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v = fps_dpp_max<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int PPT>
__global__ __launch_bounds__(FPS_T) void fps_kernel2(FpsSamples smp, int m, const float *__restrict__ xyz,
                                                     float *__restrict__ temp, int *__restrict__ idxs) {
  static_assert(PPT % 2 == 0, "points are scanned in pairs");
  constexpr int NW = FPS_T / 64;
  __shared__ __attribute__((aligned(16))) float s_d[2][NW];
  __shared__ unsigned long long s_key[3];
  const int b = blockIdx.x;
  const int n = smp.off[b + 1] - smp.off[b];
  const float *data = xyz + (size_t)smp.off[b] * 3;
  float *tmp = temp + (size_t)smp.off[b];
  int *out = idxs + (size_t)b * m;
  if (n <= 0) return;
  int bs = 1;
  while (bs * 2 <= n && bs < 1024) bs *= 2;
  const unsigned bs_mask = (unsigned)(bs - 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fps_f2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], pt[PPT / 2];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * FPS_T;
    const bool ok = k < n;
    px[i / 2][i & 1] = ok ? data[(size_t)k * 3 + 0] : 0.f;
    py[i / 2][i & 1] = ok ? data[(size_t)k * 3 + 1] : 0.f;
    pz[i / 2][i & 1] = ok ? data[(size_t)k * 3 + 2] : 0.f;
    pt[i / 2][i & 1] = ok ? tmp[k] : -1.f;      // padding: min(d, -1) = -1 never wins (real distances are >= 0)
  }
  if (tid == 0) {
    out[0] = 0;
    s_key[0] = s_key[1] = s_key[2] = ~0ull;
  }
  float x1 = data[0], y1 = data[1], z1 = data[2];
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const fps_f2 vx = {x1, x1}, vy = {y1, y1}, vz = {z1, z1};
    // distances are >= 0 (or the padding's -1): their bit patterns order like the values, so min / max run on
    // the integer pipe's forms (no NaN canonicalisation: one v_min_i32 per point, one v_max3_i32 per pair)
    int besti = __float_as_int(-1.f);
#pragma unroll
    for (int i = 0; i < PPT / 2; ++i) {
      const fps_f2 dx = px[i] - vx, dy = py[i] - vy, dz = pz[i] - vz;
      const fps_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const float d0 = d[0], d1 = d[1], p0 = pt[i][0], p1 = pt[i][1];     // (bit casts of vector ELEMENTS in place read element 0)
      const int a0 = min(__float_as_int(d0), __float_as_int(p0));
      const int a1 = min(__float_as_int(d1), __float_as_int(p1));
      fps_f2 nv;
      nv[0] = __int_as_float(a0);
      nv[1] = __int_as_float(a1);
      pt[i] = nv;
      besti = max(besti, max(a0, a1));
    }
    const float best = __int_as_float(besti);
    const float wbest = fps_wave_max(best);
    const int buf = j & 1, kb = j % 3;
    if (lane == 0) s_d[buf][wave] = wbest;
    if (tid == 0) s_key[(j + 1) % 3] = ~0ull;       // the key slot of the NEXT round (last read two rounds ago)
    __syncthreads();
    const float4 a = *(const float4 *)&s_d[buf][0], c = *(const float4 *)&s_d[buf][4];
    const float g = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)));
    if (wbest == g) {                                // wave-uniform: normally one wave gets here
      if (best == g) {
        // which of this thread's points: among its own candidates (k = tid + 512 i) the stride class k & bs_mask
        // differs only for block size 1024, where the odd slots' class has its lowest reversed bit set — they lose
        // to the even slots; otherwise the smaller index wins.  Walked downwards: the last hit is the smallest.
        int se = PPT, so = PPT;
#pragma unroll
        for (int i = PPT - 1; i >= 0; --i) {
          const bool hit = pt[i / 2][i & 1] == g;
          if (i & 1) so = hit ? i : so;
          else se = hit ? i : se;
        }
        const int slot = bs_mask == 1023u ? (se < PPT ? se : so) : (se < so ? se : so);
        const unsigned k = (unsigned)(tid + slot * FPS_T);
        atomicMin(&s_key[kb], ((unsigned long long)__brev(k & bs_mask) << 32) | k);
      }
    }
    __syncthreads();
    const int k = __builtin_amdgcn_readfirstlane((int)(unsigned)(s_key[kb] & 0xffffffffull));
    if (tid == 0) out[j] = k;
    x1 = data[(size_t)k * 3 + 0];
    y1 = data[(size_t)k * 3 + 1];
    z1 = data[(size_t)k * 3 + 2];
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * FPS_T;
    if (k < n) tmp[k] = pt[i / 2][i & 1];
  }
}

// fallback for point counts whose per-thread share does not fit the register file:
// the running minima live in `temp` (global), as in the reference
__global__ __launch_bounds__(1024) void fps_kernel_global(FpsSamples smp, int m,
                                                          const float *__restrict__ xyz,
                                                          float *__restrict__ temp,
                                                          int *__restrict__ idxs) {
  __shared__ float s_d[16];
  __shared__ int s_k[16];
  __shared__ int s_old;
  const int b = blockIdx.x;
  const int n = smp.off[b + 1] - smp.off[b];
  const float *data = xyz + (size_t)smp.off[b] * 3;
  float *tmp = temp + (size_t)smp.off[b];
  int *out = idxs + (size_t)b * m;
  if (n <= 0) return;
  int bs = 1;
  while (bs * 2 <= n && bs < 1024) bs *= 2;
  const int bs_mask = bs - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    out[0] = 0;
    s_old = 0;
  }
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    int old = s_old;
    const float x1 = data[(size_t)old * 3], y1 = data[(size_t)old * 3 + 1],
                z1 = data[(size_t)old * 3 + 2];
    float best = -1.f;
    int besti = 0;
    for (int k = tid; k < n; k += 1024) {
      float d = dist2_fma(data[(size_t)k * 3] - x1, data[(size_t)k * 3 + 1] - y1,
                          data[(size_t)k * 3 + 2] - z1);
      float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      if (best < 0.f || fps_better(d2, k, best, besti, bs_mask)) {
        best = d2;
        besti = k;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float od = __shfl_xor(best, off);
      int ok = __shfl_xor(besti, off);
      if (od >= 0.f && (best < 0.f || fps_better(od, ok, best, besti, bs_mask))) {
        best = od;
        besti = ok;
      }
    }
    __syncthreads();
    if (lane == 0) {
      s_d[wave] = best;
      s_k[wave] = besti;
    }
    __syncthreads();
    if (wave == 0) {
      float d = lane < 16 ? s_d[lane] : -1.f;
      int k = lane < 16 ? s_k[lane] : 0;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        float od = __shfl_xor(d, off);
        int ok = __shfl_xor(k, off);
        if (od >= 0.f && (d < 0.f || fps_better(od, ok, d, k, bs_mask))) {
          d = od;
          k = ok;
        }
      }
      if (lane == 0) {
        out[j] = k;
        s_old = k;
      }
    }
    __syncthreads();
  }
}

// ---- furthest point sampling of LARGE clouds: several workgroups per sample ----------------------
// A Waymo-sized frame (~200 k points -> 4096 key points) does not fit one workgroup's registers; the
// single-workgroup fallback above streams all points from memory in every round (167 ms per call on
// MI355X).  Here G workgroups own an interleaved share of the sample each (points and running minima
// in registers, as in fps_kernel2); per round every workgroup publishes its best candidate
// (distance, index, coordinates) in a slot of the scratch buffer, tagged with the round number, and
// reads everybody's slots back: a grid-wide exchange through L2 without atomics (release store of
// the tag, acquire polls; slots double-buffered by round parity).  The winner is chosen with the
// same total order (fps_better), so the indices are those of the one-workgroup kernels bit for bit.
// All G * batch workgroups must be co-resident: the launcher keeps them to a fraction of the chip.
// candidate 2 replaces candidate 1?  (a negative distance marks "no candidate")
__device__ __forceinline__ bool fps_take(float d2, int k2, float d1, int k1, int bs_mask) {
  if (!(d2 >= 0.f)) return false;
  if (d1 < 0.f) return true;
  return fps_better(d2, k2, d1, k1, bs_mask);
}

// One 64-bit word per (round parity, workgroup): {distance bits : 32 | round & 255 : 8 | index : 24},
// published with ONE relaxed read-modify-write and polled with read-modify-writes: atomics execute at the
// device's coherence point, so no release / acquire fences (L2 write-back + invalidate per round) are
// needed; the winner's coordinates are re-read from the (read-only) point array.
constexpr int FPS_MT = 1024;

__device__ __forceinline__ unsigned long long fps_word(float d, int k, int round) {
  return ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)(round & 255) << 24) |
         (unsigned long long)(k & 0xFFFFFF);
}

template <int PPT>
__global__ __launch_bounds__(FPS_MT) void fps_kernel_multi(FpsSamples smp, int m, const float *__restrict__ xyz,
                                                           float *__restrict__ temp, int *__restrict__ idxs) {
  __shared__ float s_d[16];
  __shared__ int s_k[16];
  __shared__ float s_win[3];
  const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int n = smp.off[b + 1] - smp.off[b];
  const float *data = xyz + (size_t)smp.off[b] * 3;
  // [2][G] words at the first 8-byte boundary of the sample's scratch range (tagged 0xFF.. by the launcher)
  unsigned long long *slots = (unsigned long long *)(((uintptr_t)(temp + (size_t)smp.off[b]) + 7) & ~(uintptr_t)7);
  int *out = idxs + (size_t)b * m;
  if (n <= 0) return;
  int bs = 1;
  while (bs * 2 <= n && bs < 1024) bs *= 2;
  const int bs_mask = bs - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float px[PPT], py[PPT], pz[PPT], pd[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = (i * G + g) * FPS_MT + tid;
    const bool in = k < n;
    const int kc = in ? k : 0;
    px[i] = data[(size_t)kc * 3], py[i] = data[(size_t)kc * 3 + 1], pz[i] = data[(size_t)kc * 3 + 2];
    pd[i] = in ? 1e10f : -1.f;
  }
  float x1 = data[0], y1 = data[1], z1 = data[2];
  if (g == 0 && tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    float best = -1.f;
    int bslot = 0;   // register slot of the running best; its point index is (bslot * G + g) * FPS_MT + tid
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = dist2_fma(px[i] - x1, py[i] - y1, pz[i] - z1);
      const float d2 = fminf(d, pd[i]);      // slots beyond the sample hold -1 and stay -1
      pd[i] = d2;
      bool take = d2 > best;                 // strict '>' is the common path; exact ties take the full rule
      if (d2 == best && d2 >= 0.f)
        take = fps_better(d2, (i * G + g) * FPS_MT + tid, best, (bslot * G + g) * FPS_MT + tid, bs_mask);
      best = take ? d2 : best;
      bslot = take ? i : bslot;
    }
    int besti = (bslot * G + g) * FPS_MT + tid;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float od = __shfl_xor(best, off);
      const int ok = __shfl_xor(besti, off);
      if (fps_take(od, ok, best, besti, bs_mask)) {
        best = od;
        besti = ok;
      }
    }
    if (lane == 0) s_d[wave] = best, s_k[wave] = besti;
    __syncthreads();
    if (wave == 0) {
      float d = lane < 16 ? s_d[lane] : -1.f;
      int k = lane < 16 ? s_k[lane] : 0;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const int ok = __shfl_xor(k, off);
        if (fps_take(od, ok, d, k, bs_mask)) {
          d = od;
          k = ok;
        }
      }
      unsigned long long *row = slots + (size_t)(j & 1) * G;
      if (lane == 0)
        __hip_atomic_exchange(row + g, fps_word(d, k, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d = -1.f, k = 0;
      for (int q = lane; q < G; q += 64) {
        unsigned long long w;
        do {
          w = __hip_atomic_fetch_or(row + q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (((w >> 24) & 255ull) != (unsigned long long)(j & 255));
        const float od = __uint_as_float((unsigned int)(w >> 32));
        const int ok = (int)(w & 0xFFFFFFull);
        if (fps_take(od, ok, d, k, bs_mask)) {
          d = od;
          k = ok;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const int ok = __shfl_xor(k, off);
        if (fps_take(od, ok, d, k, bs_mask)) {
          d = od;
          k = ok;
        }
      }
      if (lane < 3) s_win[lane] = data[(size_t)k * 3 + lane];
      if (lane == 0 && g == 0) out[j] = k;
    }
    __syncthreads();
    x1 = s_win[0], y1 = s_win[1], z1 = s_win[2];
  }
}

}  // namespace

extern "C" int dm_ball_query_stack2(int batch, int m, float radius_a, int nsample_a, float radius_b,
                                    int nsample_b, const float *new_xyz, const int *new_xyz_batch_cnt,
                                    const float *xyz, const int *xyz_batch_cnt, int *idx_a, int *idx_b,
                                    unsigned char *empty_a, unsigned char *empty_b, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || batch > DM_MAX_BATCH || m < 0 || nsample_a <= 0 || nsample_b <= 0) return DM_ERR_INVALID_ARG;
  if (m == 0) return DM_OK;
  if (!new_xyz || !new_xyz_batch_cnt || !xyz_batch_cnt || !idx_a || !idx_b || !empty_a || !empty_b)
    return DM_ERR_INVALID_ARG;
  BallPair bp;
  bp.radius2[0] = radius_a * radius_a, bp.radius2[1] = radius_b * radius_b;   // ball_query_gpu.cu:43
  bp.nsample[0] = nsample_a, bp.nsample[1] = nsample_b;
  bp.idx[0] = idx_a, bp.idx[1] = idx_b;
  bp.empty[0] = empty_a, bp.empty[1] = empty_b;
  if (m >= 16384)
    ball_query_wave2<4><<<dm_ceil_div(m, 16), 256, 0, st>>>(batch, m, bp, new_xyz, new_xyz_batch_cnt, xyz,
                                                            xyz_batch_cnt);
  else
    ball_query_wave2<2><<<dm_ceil_div(m, 8), 256, 0, st>>>(batch, m, bp, new_xyz, new_xyz_batch_cnt, xyz,
                                                           xyz_batch_cnt);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_ball_query_stack(int batch, int m, float radius, int nsample,
                                   const float *new_xyz, const int *new_xyz_batch_cnt,
                                   const float *xyz, const int *xyz_batch_cnt, int max_m_per_sample,
                                   int *idx, unsigned char *empty_mask, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || batch > DM_MAX_BATCH || m < 0 || nsample <= 0 || max_m_per_sample < 0)
    return DM_ERR_INVALID_ARG;
  if (m == 0) return DM_OK;
  if (!new_xyz || !new_xyz_batch_cnt || !xyz_batch_cnt || !idx) return DM_ERR_INVALID_ARG;
  float radius2 = radius * radius;  // ball_query_gpu.cu:43
  (void)max_m_per_sample;
  if (m >= 16384)
    ball_query_wave<4><<<dm_ceil_div(m, 16), 256, 0, st>>>(batch, m, radius2, nsample, new_xyz,
                                                           new_xyz_batch_cnt, xyz, xyz_batch_cnt,
                                                           idx, empty_mask);
  else
    ball_query_wave<2><<<dm_ceil_div(m, 8), 256, 0, st>>>(batch, m, radius2, nsample, new_xyz,
                                                          new_xyz_batch_cnt, xyz, xyz_batch_cnt,
                                                          idx, empty_mask);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_group_points_stack(int batch, int m, int c, int nsample, const float *features,
                                     const int *features_batch_cnt, const int *idx,
                                     const int *idx_batch_cnt, const unsigned char *empty_mask,
                                     float *out, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || m < 0 || c <= 0 || nsample <= 0) return DM_ERR_INVALID_ARG;
  if (m == 0) return DM_OK;
  if ((c + 1) * nsample > GP_MAX_ELEMS) return DM_ERR_UNSUPPORTED;
  if (!features || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return DM_ERR_INVALID_ARG;
  size_t smem = 4ull * (c + 1) * nsample * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)group_points_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    DM_HIP(hipFuncSetAttribute((const void *)group_points_grad_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr = true;
  }
  group_points_kernel<<<dm_ceil_div(m, 4), 256, smem, st>>>(batch, m, c, nsample, features,
                                                            features_batch_cnt, idx, idx_batch_cnt,
                                                            empty_mask, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_group_points_grad_stack(int batch, int m, int c, int n, int nsample,
                                          const float *grad_out, const int *idx,
                                          const int *idx_batch_cnt, const int *features_batch_cnt,
                                          float *grad_features, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || m < 0 || c <= 0 || nsample <= 0 || n < 0) return DM_ERR_INVALID_ARG;
  if (n > 0) {
    if (!grad_features) return DM_ERR_INVALID_ARG;
    DM_HIP(hipMemsetAsync(grad_features, 0, (size_t)n * c * sizeof(float), st));
  }
  if (m == 0 || n == 0) return DM_OK;
  if ((c + 1) * nsample > GP_MAX_ELEMS) return DM_ERR_UNSUPPORTED;
  if (!grad_out || !idx || !idx_batch_cnt || !features_batch_cnt) return DM_ERR_INVALID_ARG;
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)group_points_grad_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    DM_HIP(hipFuncSetAttribute((const void *)group_points_grad_combine,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  if (m >= 16384 && g_gp_grad_combine) {
    // one chunk per CU-sized share of the queries (216 = one RoI's grid for the PV-RCNN head)
    int chunk = dm_ceil_div(m, 256);
    chunk = chunk < 32 ? 32 : (chunk > 256 ? 256 : chunk);
    int n_slots = GPC_ACC_BYTES / (4 * c);
    if (n_slots > GPC_MAX_SLOTS) n_slots = GPC_MAX_SLOTS;
    size_t lds = (size_t)n_slots * c * 4 + (size_t)n_slots * 4 + (size_t)chunk * nsample * 2;
    lds = (lds + 15) & ~(size_t)15;
    if (n_slots >= 64 && lds <= 160 * 1024) {
      group_points_grad_combine<<<dm_ceil_div(m, chunk), 256, lds, st>>>(
          batch, m, c, nsample, chunk, n_slots, grad_out, idx, idx_batch_cnt, features_batch_cnt,
          grad_features);
      DM_CHECK_LAUNCH();
      return DM_OK;
    }
  }
  size_t smem = 4ull * (c + 1) * nsample * sizeof(float);
  group_points_grad_kernel<<<dm_ceil_div(m, 4), 256, smem, st>>>(
      batch, m, c, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_query_group_rows(int batch, int m, int c, int nsample, int use_xyz,
                                   const float *xyz, const float *new_xyz, const float *features,
                                   const int *xyz_batch_cnt, const int *new_xyz_batch_cnt,
                                   const int *idx, const unsigned char *empty_mask, float *out,
                                   dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || m < 0 || c < 0 || nsample <= 0 || (!use_xyz && c == 0)) return DM_ERR_INVALID_ARG;
  if (m == 0) return DM_OK;
  if (!xyz || !new_xyz || !xyz_batch_cnt || !new_xyz_batch_cnt || !idx || !out || (c > 0 && !features))
    return DM_ERR_INVALID_ARG;
  long long refs = (long long)m * nsample;
  if (use_xyz && c > 0 && (c & 3) == 0 && refs >= 65536) {
    query_group_rows_quads_kernel<<<(unsigned)((refs + 63) / 64), 256, 0, st>>>(
        batch, m, c / 4, nsample, xyz, new_xyz, features, xyz_batch_cnt, new_xyz_batch_cnt, idx, empty_mask, out);
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  query_group_rows_kernel<<<(unsigned)((refs + 3) / 4), 256, 0, st>>>(
      batch, m, c, nsample, use_xyz, xyz, new_xyz, features, xyz_batch_cnt, new_xyz_batch_cnt, idx,
      empty_mask, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_group_rows_grad(int batch, int m, int c, int n, int nsample, int row_width,
                                  int col_offset, const float *grad_out, const int *idx,
                                  const int *idx_batch_cnt, const int *features_batch_cnt,
                                  const unsigned char *empty_mask, float *grad_features,
                                  dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch <= 0 || m < 0 || c <= 0 || nsample <= 0 || n < 0 || col_offset < 0 ||
      col_offset + c > row_width)
    return DM_ERR_INVALID_ARG;
  if (n > 0) {
    if (!grad_features) return DM_ERR_INVALID_ARG;
    DM_HIP(hipMemsetAsync(grad_features, 0, (size_t)n * c * sizeof(float), st));
  }
  if (m == 0 || n == 0) return DM_OK;
  if (!grad_out || !idx || !idx_batch_cnt || !features_batch_cnt) return DM_ERR_INVALID_ARG;
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)group_rows_grad_combine,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  if (m >= 16384 && g_gp_grad_combine) {
    // Queries per workgroup.  Round 6: ~4 096 workgroups instead of 256 (one per CU): a wave keeps four 512-byte rows in
    // flight, so 1 024 waves hold 2 MB — the RoI-grid-pooling call (55 296 x 16 references of 128 floats) read its rows at
    // 0.9-1.4 TB/s; with 14 queries per workgroup 149 instead of 513 us on a synthetic RoI geometry (3.0 TB/s,
    // tools/bench_group_grad.py), more atomic rows per key point notwithstanding.  DM_GRG_WGS: A/B switch.
    static const int target_wgs = [] { const char *e = getenv("DM_GRG_WGS"); const int v = e ? atoi(e) : 4096; return v < 1 ? 1 : v; }();
    int chunk = dm_ceil_div(m, target_wgs);
    chunk = chunk < 8 ? 8 : (chunk > 256 ? 256 : chunk);
    int n_slots = 1024;                       // open-addressing table of distinct source rows
    size_t lds = ((size_t)(3 * n_slots + 1) * 4 + (size_t)chunk * nsample * 4 + 15) & ~(size_t)15;
    if (chunk * nsample <= 65535 && lds <= 64 * 1024) {
      group_rows_grad_combine<<<dm_ceil_div(m, chunk), 256, lds, st>>>(
          batch, m, c, nsample, row_width, col_offset, chunk, n_slots, grad_out, idx, idx_batch_cnt,
          features_batch_cnt, empty_mask, grad_features);
      DM_CHECK_LAUNCH();
      return DM_OK;
    }
  }
  long long refs = (long long)m * nsample;
  group_rows_grad_direct<<<(unsigned)((refs + 3) / 4), 256, 0, st>>>(
      batch, m, c, nsample, row_width, col_offset, grad_out, idx, idx_batch_cnt, features_batch_cnt,
      empty_mask, grad_features);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// ---- voxel centres + rows per sample of a sparse level (the set abstraction's xyz / xyz_batch_cnt) ----------------
// common_utils.get_voxel_centers (pcdet/utils/common_utils.py:65-82: (idx[x,y,z] + 0.5) * voxel * stride +
// range_min on the flipped coordinate columns — five element-wise launches) and the per-sample row counts
// (voxel_set_abstraction.py:209-214 counts them with a Python loop of `.sum()` read-backs) in ONE launch: thread i
// converts row i with the same fp32 operations in the same order (bit-identical); the first `batch` threads find
// the row range of their sample by binary search in the batch column (rows of a sparse level are sample-major,
// which the stacked operators behind it rely on anyway).
__global__ __launch_bounds__(256) void voxel_centers_kernel(const int32_t *__restrict__ coords, int n, int batch,
                                                            float vx, float vy, float vz, float rx, float ry, float rz,
                                                            float *__restrict__ xyz, int32_t *__restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int4 c = *(const int4 *)(coords + (size_t)i * 4);      // [b, z, y, x]
    xyz[(size_t)i * 3 + 0] = ((float)c.w + 0.5f) * vx + rx;
    xyz[(size_t)i * 3 + 1] = ((float)c.z + 0.5f) * vy + ry;
    xyz[(size_t)i * 3 + 2] = ((float)c.y + 0.5f) * vz + rz;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < batch) {
    auto lower = [&](int key) {      // first row whose sample index is >= key
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (coords[(size_t)mid * 4] < key) lo = mid + 1;
        else hi = mid;
      }
      return lo;
    };
    const int b = threadIdx.x;
    counts[b] = lower(b + 1) - lower(b);
  }
}

extern "C" int dm_voxel_centers(const int32_t *coords, int n, int batch, float vx, float vy, float vz, float rx,
                                float ry, float rz, float *xyz, int32_t *counts, dm_stream_t stream) {
  if (n < 0 || batch <= 0 || batch > 256) return DM_ERR_INVALID_ARG;
  if (!counts || (n > 0 && (!coords || !xyz))) return DM_ERR_INVALID_ARG;
  voxel_centers_kernel<<<dm_ceil_div(n > 0 ? n : 1, 256), 256, 0, (hipStream_t)stream>>>(coords, n, batch, vx, vy, vz, rx, ry,
                                                                                 rz, xyz, counts);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// tuning / test aid: 0 auto, 1 one workgroup per sample even for large clouds
static int g_fps_variant = 0;

extern "C" int dm_fps_set_variant(int v) {
  if (v != 0 && v != 1) return DM_ERR_INVALID_ARG;
  g_fps_variant = v;
  return DM_OK;
}

static int fps_launch(const FpsSamples &smp, int batch, int max_n, int m, const float *xyz,
                      float *temp, int *idxs, hipStream_t st) {
  int ppt = dm_ceil_div(max_n, FPS_T);
  if (ppt <= 48) {
#define DM_FPS2(P) fps_kernel2<P><<<batch, FPS_T, 0, st>>>(smp, m, xyz, temp, idxs)
    if (ppt <= 4) DM_FPS2(4);
    else if (ppt <= 8) DM_FPS2(8);
    else if (ppt <= 16) DM_FPS2(16);
    else if (ppt <= 24) DM_FPS2(24);
    else if (ppt <= 32) DM_FPS2(32);
    else if (ppt <= 40) DM_FPS2(40);
    else DM_FPS2(48);
#undef DM_FPS2
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  {
    // large clouds: G workgroups per sample (see fps_kernel_multi).  The exchange slots live at the
    // start of each sample's `temp` range (n floats >= 2*G*8); all G * batch workgroups spin on each
    // other, so they are kept to a quarter of the chip's 512 slots of 1024 threads.
    int G = 128 / batch;
    G = G > 64 ? 64 : (G < 1 ? 1 : G);
    const int ppt_m = dm_ceil_div(max_n, G * FPS_MT);
    bool aligned = g_fps_variant != 1 && G >= 2 && ppt_m <= 8 && max_n < (1 << 24);
    for (int b = 0; b < batch && aligned; ++b) aligned = (smp.off[b + 1] - smp.off[b]) >= 4 * G + 4;
    if (aligned) {
      for (int b = 0; b < batch; ++b)
        DM_HIP(hipMemsetAsync(temp + smp.off[b], 0xFF, ((size_t)2 * G + 1) * sizeof(unsigned long long), st));
      if (ppt_m <= 4) fps_kernel_multi<4><<<dim3(G, batch), FPS_MT, 0, st>>>(smp, m, xyz, temp, idxs);
      else fps_kernel_multi<8><<<dim3(G, batch), FPS_MT, 0, st>>>(smp, m, xyz, temp, idxs);
    } else {
      fps_kernel_global<<<batch, 1024, 0, st>>>(smp, m, xyz, temp, idxs);
    }
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_furthest_point_sampling(int batch, int n, int m, const float *xyz, float *temp,
                                          int *idxs, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch < 0 || batch > DM_MAX_BATCH || n <= 0 || m < 0) return DM_ERR_INVALID_ARG;
  if (batch == 0 || m == 0) return DM_OK;
  if (!xyz || !temp || !idxs) return DM_ERR_INVALID_ARG;
  FpsSamples smp;
  for (int b = 0; b <= batch; ++b) smp.off[b] = b * n;
  return fps_launch(smp, batch, n, m, xyz, temp, idxs, st);
}

extern "C" int dm_furthest_point_sampling_stack(int batch, const int *offsets_host, int m,
                                                const float *xyz, float *temp, int *idxs,
                                                dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch < 0 || batch > DM_MAX_BATCH || m < 0 || !offsets_host) return DM_ERR_INVALID_ARG;
  if (batch == 0 || m == 0) return DM_OK;
  if (!xyz || !temp || !idxs) return DM_ERR_INVALID_ARG;
  FpsSamples smp;
  int max_n = 0;
  for (int b = 0; b <= batch; ++b) {
    smp.off[b] = offsets_host[b];
    if (b > 0) {
      int nb = offsets_host[b] - offsets_host[b - 1];
      if (nb <= 0) return DM_ERR_INVALID_ARG;
      if (nb > max_n) max_n = nb;
    }
  }
  return fps_launch(smp, batch, max_n, m, xyz, temp, idxs, st);
}
