// Y (R x N) = X (R x K) . W^T for R in the hundreds of thousands and K, N <= 136: the shared 1x1 layers of
// RoI-grid pooling and the set abstraction (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:31-40,
// Conv2d 1x1 over M * nsample grouped rows) and their input gradients.
//
// A BLAS call (and the K-tiled implicit-GEMM kernel of conv2d.hip) re-stages the K x N weights for every
// 64-row tile and pays a prologue / epilogue per tile: 281 / 139 us for 884 736 x 132 / 64 -> 64 against
// an fp32-MFMA floor of ~120 / 60 us.  Here a workgroup keeps ALL of W in LDS, walks row tiles of 64
// (persistent: grid = 2 workgroups per CU), fetches the next tile's rows into registers while the
// current one is multiplied (v_mfma_f32_32x32x2_f32, 2 x 2 waves, the k order inside an 8-block permuted
// identically for both operands as in conv2d.hip), and writes the tile back through LDS as whole rows
// (accumulators stored straight from the MFMA layout were measured 1.5x slower here, as in conv2d.hip;
// weights in registers instead of LDS: 5 % faster, not kept).  Output rows may be wider than N (`ldy`)
// and start at column `col0` (columns [0, col0) are written as zeros): the input gradient of a first
// layer skips the xyz / padding columns nobody differentiates (QueryGroupRows.backward).
// (The weight gradient dW = G^T X stays a split batched BLAS GEMM + sum: a persistent kernel with the N x K
// accumulator blocks in registers and one row pair per MFMA k-step was measured at 502 / 161 us against
// 250 / 96 us for 884 736 x 132 / 64 x 64.)
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RG_ROWS = 64;       // rows per tile
constexpr int RG_MAXK = 136;      // K padded to a multiple of 8
constexpr int RG_MAXLD = RG_MAXK + 4;

// NBW: 32-column blocks per wave (the two column-waves interleave blocks: wn, wn + 2, wn + 4)
template <int NBW>
__global__ __launch_bounds__(256) void rowgemm_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                      float *__restrict__ Y, int R, int K, int N, int KP,
                                                      int ldy, int col0, float *__restrict__ st_partial,
                                                      float *__restrict__ st_counts, int wt_ld) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ld = KP + 4;                         // row stride of both operand tiles (floats)
  float *Ws = lds;                               // [NBW * 64][ld]   (row n: W[n][0..K), zero padded)
  float *Xs = lds + NBW * 64 * ld;               // [64][ld]; reused as the output tile [64][ldc]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int kq4 = KP / 4;                        // float4 per staged row
  // ---- stage W once (rows >= N and columns >= K are zero) ------------------------------------
  for (int e = tid; e < NBW * 64 * kq4; e += 256) {
    const int n = e / kq4, q = e - n * kq4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N && wt_ld > 0) {                    // W given transposed: element (n, k) at W[k * wt_ld + n] (dm_rowgemm_wt)
      const float *src = W + (size_t)(4 * q) * wt_ld + n;
      if (4 * q + 3 < K) v = make_float4(src[0], src[wt_ld], src[2 * (size_t)wt_ld], src[3 * (size_t)wt_ld]);
    } else if (n < N) {
      const float *src = W + (size_t)n * K + 4 * q;
      if (4 * q + 3 < K) v = *(const float4 *)src;
      else if (4 * q < K) {                      // K % 4 == 0: never partial, kept for safety
        v.x = src[0];
      }
    }
    *(float4 *)(Ws + n * ld + 4 * q) = v;
  }
  const int n_tiles = (R + RG_ROWS - 1) / RG_ROWS;
  constexpr int XP = (RG_ROWS * (RG_MAXK / 4) + 255) / 256;   // float4 fetches per thread and tile (max)
  float4 xr[XP];
  auto fetch = [&](int tile) {
    const int row0 = tile * RG_ROWS;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int e = p * 256 + tid;
      const int r = e / kq4, q = e - r * kq4;
      const int row = row0 + r;
      const bool ok = r < RG_ROWS && row < R && 4 * q < K;
      // unconditional load from a clamped address, masked afterwards (a conditional load serialises)
      const float4 v = *(const float4 *)(X + (ok ? (size_t)row * K + 4 * q : 0));
      xr[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int e = p * 256 + tid;
      const int r = e / kq4, q = e - r * kq4;
      if (r < RG_ROWS) *(float4 *)(Xs + r * ld + 4 * q) = xr[p];
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  const int ldc = NBW * 64 + 4;
  // optional per-column statistics of the output for the BatchNorm that follows (st_partial != nullptr):
  // thread = (column, row slice); shifted sums (shift = the workgroup's first output row) over every row
  // this workgroup produces, folded into (mean, M2) per column at the end — one partial per workgroup,
  // merged by bn_finalize in workgroup order.  Replaces a full read of Y by bn_stats_kernel.
  const int st_slices = N <= 256 ? 256 / N : 0;
  const int st_col = tid % N, st_slice = tid / N;
  const bool st_on = st_partial != nullptr && st_slices > 0 && st_slice < st_slices;
  const int st_rows = st_slices > 0 ? (RG_ROWS + st_slices - 1) / st_slices : 0;
  float st_k = 0.f, st_s1 = 0.f, st_s2 = 0.f;
  bool st_first = true;
  int st_cnt = 0;
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                              // previous tile's output rows have left Xs
    stage();
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);   // in flight under the MFMAs
    f32x16 acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const float *As = Xs + (wm * 32 + lr) * ld + lh * 4;
    const float *Bs = Ws + (wn * 32 + lr) * ld + lh * 4;
    for (int kb = 0; kb < KP / 8; ++kb) {
      const float4 a = *(const float4 *)(As + kb * 8);
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        const float4 w = *(const float4 *)(Bs + b * 64 * ld + kb * 8);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc[b], 0, 0, 0);
      }
    }
    __syncthreads();                              // every wave is done with Xs
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float *cs = Xs;
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ldc + (wn + 2 * b) * 32 + lr] = acc[b][r];
    __syncthreads();
    if (st_partial != nullptr) {
      const int valid = min(RG_ROWS, R - tile * RG_ROWS);
      if (st_on) {
        if (st_first) st_k = cs[st_col], st_first = false;          // row 0 of the first tile
        const int r_lo = st_slice * st_rows, r_hi = min(r_lo + st_rows, valid);
        for (int r = r_lo; r < r_hi; ++r) {
          const float d = cs[r * ldc + st_col] - st_k;
          st_s1 += d, st_s2 += d * d;
        }
      }
      st_cnt += valid;
    }
    // output rows are `ldy` floats wide and start at column col0; columns [0, col0) are zeros
    const int nq = N / 4, cq = col0 / 4;
    const int row0 = tile * RG_ROWS;
    for (int e = tid; e < RG_ROWS * (nq + cq); e += 256) {
      const int r = e / (nq + cq), q = e - r * (nq + cq);
      if (row0 + r >= R) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q >= cq) v = *(const float4 *)(cs + r * ldc + 4 * (q - cq));
      *(float4 *)(Y + (size_t)(row0 + r) * ldy + 4 * q) = v;
    }
  }
  if (st_partial != nullptr) {                     // fold the row slices of each column, slice order
    __syncthreads();
    float *f1 = Xs, *f2 = Xs + 256;
    f1[tid] = st_s1, f2[tid] = st_s2;
    __syncthreads();
    if (tid < N) {
      float a1 = 0.f, a2 = 0.f;
      for (int sl = 0; sl < st_slices; ++sl) a1 += f1[sl * N + tid], a2 += f2[sl * N + tid];
      const float cnt = (float)st_cnt;
      const size_t nb = gridDim.x;
      st_partial[(size_t)tid * nb + blockIdx.x] = cnt > 0.f ? st_k + a1 / cnt : 0.f;
      st_partial[((size_t)N + tid) * nb + blockIdx.x] = cnt > 0.f ? a2 - a1 * a1 / cnt : 0.f;
      if (tid == 0) st_counts[blockIdx.x] = cnt;
    }
  }
}

size_t rowgemm_smem(int K, int nbw) {
  const int KP = (K + 7) / 8 * 8;
  const int ld = KP + 4, ldc = nbw * 64 + 4;
  return ((size_t)nbw * 64 * ld + 64 * (size_t)(ld > ldc ? ld : ldc)) * sizeof(float);
}

int rowgemm_grid(int R, size_t smem) {
  const int n_tiles = (R + RG_ROWS - 1) / RG_ROWS;
  int per_cu = (int)((150 * 1024) / smem);       // workgroups resident per CU (LDS bound)
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  return n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu;
}

template <int NBW>
int launch(const float *X, const float *W, float *Y, int R, int K, int N, int ldy, int col0, hipStream_t st,
           float *st_partial = nullptr, float *st_counts = nullptr, int wt_ld = 0) {
  const int KP = (K + 7) / 8 * 8;
  const size_t smem = rowgemm_smem(K, NBW);
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)rowgemm_kernel<NBW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    attr = true;
  }
  if (smem > 160 * 1024) return DM_ERR_UNSUPPORTED;
  const int grid = rowgemm_grid(R, smem);
  rowgemm_kernel<NBW><<<grid, 256, smem, st>>>(X, W, Y, R, K, N, KP, ldy, col0, st_partial, st_counts, wt_ld);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

}  // namespace

extern "C" int dm_rowgemm_supported(int k, int n) {
  return k >= 4 && k <= RG_MAXK && (k & 3) == 0 && n >= 4 && n <= 192 && (n & 3) == 0;
}

extern "C" int dm_rowgemm_strided(const float *x, const float *w, float *y, long long rows, int k, int n,
                                  int ldy, int col0, dm_stream_t stream) {
  if (rows < 0 || rows > 0x7fffffffLL / 256) return DM_ERR_INT32_RANGE;
  if (!dm_rowgemm_supported(k, n) || (ldy & 3) || (col0 & 3) || col0 < 0 || ldy < col0 + n)
    return DM_ERR_UNSUPPORTED;
  if (rows == 0) return DM_OK;
  if (!x || !w || !y) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 64) return launch<1>(x, w, y, (int)rows, k, n, ldy, col0, st);
  if (n <= 128) return launch<2>(x, w, y, (int)rows, k, n, ldy, col0, st);
  return launch<3>(x, w, y, (int)rows, k, n, ldy, col0, st);
}

// y = x . w^T plus the statistics of y's columns for the BatchNorm that follows: partial
// (2, n, dm_rowgemm_parts(rows, k, n)) = per-workgroup (mean, M2) of every column, counts (parts) = rows per
// workgroup — what dm_bn_rows_forward_pre / dm_bn_rows_max_forward_pre take instead of reading y again.
extern "C" int dm_rowgemm_parts(long long rows, int k, int n) {
  if (rows <= 0 || !dm_rowgemm_supported(k, n)) return 0;
  return rowgemm_grid((int)rows, rowgemm_smem(k, n <= 64 ? 1 : (n <= 128 ? 2 : 3)));
}

extern "C" int dm_rowgemm_stats(const float *x, const float *w, float *y, long long rows, int k, int n,
                                float *partial, float *counts, dm_stream_t stream) {
  if (rows <= 0 || rows > 0x7fffffffLL / 256) return rows == 0 ? DM_OK : DM_ERR_INT32_RANGE;
  if (!dm_rowgemm_supported(k, n) || n > 256) return DM_ERR_UNSUPPORTED;
  if (!x || !w || !y || !partial || !counts) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 64) return launch<1>(x, w, y, (int)rows, k, n, n, 0, st, partial, counts);
  if (n <= 128) return launch<2>(x, w, y, (int)rows, k, n, n, 0, st, partial, counts);
  return launch<3>(x, w, y, (int)rows, k, n, n, 0, st, partial, counts);
}

// y[:, col0:col0+n] = x (rows, k) . wt, wt (k, wt_ld) row-major with the n columns starting at wt: the INPUT gradient of
// a linear layer straight from its stored (out, in) weight — x = dY, k = out, wt = weight + first live input column,
// wt_ld = in — without a transposed copy of the weight per call
extern "C" int dm_rowgemm_wt(const float *x, const float *wt, int wt_ld, float *y, long long rows, int k, int n,
                             int ldy, int col0, dm_stream_t stream) {
  if (rows < 0 || rows > 0x7fffffffLL / 256) return DM_ERR_INT32_RANGE;
  if (!dm_rowgemm_supported(k, n) || (ldy & 3) || (col0 & 3) || col0 < 0 || ldy < col0 + n || wt_ld < n)
    return DM_ERR_UNSUPPORTED;
  if (rows == 0) return DM_OK;
  if (!x || !wt || !y) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 64) return launch<1>(x, wt, y, (int)rows, k, n, ldy, col0, st, nullptr, nullptr, wt_ld);
  if (n <= 128) return launch<2>(x, wt, y, (int)rows, k, n, ldy, col0, st, nullptr, nullptr, wt_ld);
  return launch<3>(x, wt, y, (int)rows, k, n, ldy, col0, st, nullptr, nullptr, wt_ld);
}

extern "C" int dm_rowgemm(const float *x, const float *w, float *y, long long rows, int k, int n,
                          dm_stream_t stream) {
  return dm_rowgemm_strided(x, w, y, rows, k, n, n, 0, stream);
}
