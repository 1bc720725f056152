// Fully connected layers of the two RoI heads, the key-point fusion layer and the point head
// (pcdet/models/roi_heads/pvrcnn_head.py:25-52, pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:107-111,
// pcdet/models/dense_heads/point_head_template.py:34-47): SMALL fp32 GEMMs — a few hundred to a few thousand rows,
// 128 .. 1024 columns, one skinny giant (256 RoIs x 27 648 -> 256) — forward, input gradient and weight gradient.
//
// The reference (and rounds 1-5 of this repository) hands them to the vendor BLAS.  On gfx950 every fp32 GEMM of
// hipBLASLt is a Tensile Stream-K kernel: 63 us for 256 x 256 x 256 (0.03 GFLOP: launch + fix-up protocol, not
// arithmetic), 125 us for the forward of the giant, and two of them in flight from one library handle dead-lock the
// device.  Here: one kernel, three operand forms, exact fp32 on v_mfma_f32_32x32x2_f32.
//   form 0  C[M][N] = A[M][K] . B[N][K]^T     forward            y  = x w^T
//   form 1  C[M][N] = A[M][K] . B[K][N]       input gradient     gx = gy w
//   form 2  C[M][N] = A[K][M]^T . B[K][N]     weight gradient    gw = gy^T x
// (all row-major with leading dimensions; K = contraction length).
// Workgroup = 64 x 64 output tile, 2 x 2 waves of 32 x 32; the contraction walks 32 at a time: both operand tiles are
// staged in LDS as [row][k] with row stride 33 (the transposing stores of forms 1 / 2 and the fragment reads are then at
// most 2-way bank conflicted), the next tile is fetched into registers while the current one is multiplied.  A GEMM
// with few output tiles and a long contraction (the giant: 16 tiles x 864 steps) is split along K over blockIdx.z;
// the partial products go to a workspace and are summed in split order by a second kernel that also adds the bias and
// applies the ReLU — the result is a function of the shapes only (no atomics).
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FC_T = 64;        // output tile (rows and columns)
constexpr int FC_BK = 32;       // contraction step
constexpr int FC_LD = FC_BK + 1;

struct FcArgs {
  const float *A, *B, *bias;
  float *C;          // output, or the partial buffer [splits][M][N] when splits > 1
  int M, N, K, lda, ldb, ldc;
  int steps_per_split, relu, splits;
  int tiles_m, tiles_n, per_xcd;
};

// One 64 x 32 operand tile, one float4 (4 consecutive elements of the operand's contiguous dimension) per thread pair.
// ROWS_ALONG_K = false: the operand is [row][k] in memory (k contiguous): thread -> (row = t / 8, k4 = t % 8).
// ROWS_ALONG_K = true:  the operand is [k][row] in memory (row contiguous): thread -> (k = t / 16, r4 = t % 16), two per thread.
template <bool ROWS_ALONG_K>
struct TileRegs {
  float4 v[ROWS_ALONG_K ? 2 : 2];
};

template <bool ROWS_ALONG_K>
__device__ __forceinline__ void fetch_tile(TileRegs<ROWS_ALONG_K> &t, const float *P, int ld, int row0, int n_rows,
                                           int k0, int k_hi, int tid, bool vec) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int r, k;
    if (!ROWS_ALONG_K) {
      r = (tid >> 3) + 32 * h;          // 64 rows, 8 float4 of k each
      k = k0 + 4 * (tid & 7);
    } else {
      k = k0 + (tid >> 4) + 16 * h;     // 32 k-rows, 16 float4 of rows each
      r = 4 * (tid & 15);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ROWS_ALONG_K) {
      const int row = row0 + r;
      if (row < n_rows && k < k_hi) {
        const float *p = P + (size_t)row * ld + k;
        if (vec && k + 3 < k_hi) v = *(const float4 *)p;
        else {
          v.x = p[0];
          if (k + 1 < k_hi) v.y = p[1];
          if (k + 2 < k_hi) v.z = p[2];
          if (k + 3 < k_hi) v.w = p[3];
        }
      }
    } else {
      const int row = row0 + r;
      if (k < k_hi && row < n_rows) {
        const float *p = P + (size_t)k * ld + row;
        if (vec && row + 3 < n_rows) v = *(const float4 *)p;
        else {
          v.x = p[0];
          if (row + 1 < n_rows) v.y = p[1];
          if (row + 2 < n_rows) v.z = p[2];
          if (row + 3 < n_rows) v.w = p[3];
        }
      }
    }
    t.v[h] = v;
  }
}

template <bool ROWS_ALONG_K>
__device__ __forceinline__ void stage_tile(const TileRegs<ROWS_ALONG_K> &t, float *S, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 v = t.v[h];
    if (!ROWS_ALONG_K) {
      float *d = S + ((tid >> 3) + 32 * h) * FC_LD + 4 * (tid & 7);
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    } else {
      const int k = (tid >> 4) + 16 * h, r = 4 * (tid & 15);
      S[(r + 0) * FC_LD + k] = v.x;
      S[(r + 1) * FC_LD + k] = v.y;
      S[(r + 2) * FC_LD + k] = v.z;
      S[(r + 3) * FC_LD + k] = v.w;
    }
  }
}

// FORM 0: A [M][K], B [N][K];  FORM 1: A [M][K], B [K][N];  FORM 2: A [K][M], B [K][N]
template <int FORM>
__global__ __launch_bounds__(256) void fc_gemm_kernel(FcArgs a, int vec_a, int vec_b) {
  constexpr bool A_T = FORM == 2, B_T = FORM != 0;
  __shared__ float As[FC_T * FC_LD];
  __shared__ float Bs[FC_T * FC_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the ids one XCD sees are
  // L, L + 8, L + 16, ...; they are mapped to CONSECUTIVE logical tiles = (split, n tile, m tile) with m fastest — the
  // workgroups that re-read the same B columns (skinny M: 4 m tiles) or the same K slice of both operands (split
  // contraction: all tiles of a split) run next to each other on ONE XCD and hit in its L2 instead of going to HBM again.
  const int logical = (int)(blockIdx.x % 8) * a.per_xcd + (int)(blockIdx.x / 8);
  const int n_tiles = a.tiles_m * a.tiles_n;
  if (logical >= n_tiles * a.splits) return;
  const int split = logical / n_tiles, tile = logical - split * n_tiles;
  const int m0 = (tile % a.tiles_m) * FC_T, n0 = (tile / a.tiles_m) * FC_T;
  const int steps = (a.K + FC_BK - 1) / FC_BK;
  const int s_lo = split * a.steps_per_split;
  const int s_hi = min(steps, s_lo + a.steps_per_split);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  TileRegs<A_T> ra;
  TileRegs<B_T> rb;
  if (s_lo < s_hi) {
    fetch_tile<A_T>(ra, a.A, a.lda, m0, a.M, s_lo * FC_BK, a.K, tid, vec_a != 0);
    fetch_tile<B_T>(rb, a.B, a.ldb, n0, a.N, s_lo * FC_BK, a.K, tid, vec_b != 0);
  }
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();                               // the previous step's fragments have been read
    stage_tile<A_T>(ra, As, tid);
    stage_tile<B_T>(rb, Bs, tid);
    __syncthreads();
    if (s + 1 < s_hi) {                            // in flight under the MFMAs
      fetch_tile<A_T>(ra, a.A, a.lda, m0, a.M, (s + 1) * FC_BK, a.K, tid, vec_a != 0);
      fetch_tile<B_T>(rb, a.B, a.ldb, n0, a.N, (s + 1) * FC_BK, a.K, tid, vec_b != 0);
    }
    const float *pa = As + (wm * 32 + lr) * FC_LD + lh;
    const float *pb = Bs + (wn * 32 + lr) * FC_LD + lh;
#pragma unroll
    for (int k = 0; k < FC_BK; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
  }
  // C layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int col = n0 + wn * 32 + lr;
  if (col >= a.N) return;
  const bool direct = a.splits == 1;
  const float bv = (direct && a.bias) ? a.bias[col] : 0.f;
  float *out = direct ? a.C : a.C + (size_t)split * a.M * a.N;
  const int ldo = direct ? a.ldc : a.N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (row < a.M) {
      float v = acc[r] + bv;
      if (direct && a.relu) v = fmaxf(v, 0.f);
      out[(size_t)row * ldo + col] = v;
    }
  }
}

// C[m][n] = relu?(sum_s partial[s][m][n] + bias[n]), splits summed in index order
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float *__restrict__ partial, int splits, long long mn, int N,
                                                        const float *__restrict__ bias, int relu, float *__restrict__ C,
                                                        int ldc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float v = partial[i];
  for (int s = 1; s < splits; ++s) v += partial[(size_t)s * mn + i];
  const int n = (int)(i % N);
  if (bias) v += bias[n];
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)(i / N) * ldc + n] = v;
}

// how the contraction is split: a function of the shape only
int fc_splits(int M, int N, int K) {
  const long long tiles = (long long)dm_ceil_div(M, FC_T) * dm_ceil_div(N, FC_T);
  const int steps = dm_ceil_div(K, FC_BK);
  if (tiles >= 256 || steps < 16) return 1;
  long long want = (512 + tiles - 1) / tiles;            // ~2 workgroups per compute unit
  const int most = steps / 8 > 0 ? steps / 8 : 1;         // at least 8 steps (256 of K) per split
  if (want > most) want = most;
  if (want > 64) want = 64;
  return want < 1 ? 1 : (int)want;
}

}  // namespace

extern "C" size_t dm_fc_gemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = fc_splits(M, N, K);
  return s > 1 ? dm_align((size_t)s * M * N * sizeof(float)) : 0;
}

extern "C" int dm_fc_gemm(int form, const float *A, const float *B, const float *bias, float *C, int M, int N, int K,
                          int lda, int ldb, int ldc, int relu, void *workspace, size_t workspace_bytes,
                          dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (form < 0 || form > 2 || M < 0 || N < 0 || K < 0 || ldc < N) return DM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return DM_OK;
  if (!A || !B || !C) return DM_ERR_INVALID_ARG;
  const int min_lda = form == 2 ? M : K, min_ldb = form == 0 ? K : N;
  if (K > 0 && (lda < min_lda || ldb < min_ldb)) return DM_ERR_INVALID_ARG;
  FcArgs a;
  a.A = A, a.B = B, a.bias = bias, a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.relu = relu;
  const int steps = dm_ceil_div(K > 0 ? K : 1, FC_BK);
  a.splits = K > 0 ? fc_splits(M, N, K) : 1;
  a.steps_per_split = dm_ceil_div(steps, a.splits);
  a.splits = dm_ceil_div(steps, a.steps_per_split);       // no empty split
  if (K == 0) a.steps_per_split = 0;
  a.C = C;
  if (a.splits > 1) {
    const size_t need = (size_t)a.splits * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need) return DM_ERR_WORKSPACE;
    a.C = (float *)workspace;
  }
  const int vec_a = (lda % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0;
  const int vec_b = (ldb % 4 == 0 && ((uintptr_t)B & 15) == 0) ? 1 : 0;
  a.tiles_m = dm_ceil_div(M, FC_T), a.tiles_n = dm_ceil_div(N, FC_T);
  const long long total = (long long)a.tiles_m * a.tiles_n * a.splits;
  if (total > 0x3fffffff) return DM_ERR_INVALID_ARG;
  a.per_xcd = (int)((total + 7) / 8);
  dim3 grid((unsigned)(a.per_xcd * 8), 1, 1);
  if (form == 0) fc_gemm_kernel<0><<<grid, 256, 0, st>>>(a, vec_a, vec_b);
  else if (form == 1) fc_gemm_kernel<1><<<grid, 256, 0, st>>>(a, vec_a, vec_b);
  else fc_gemm_kernel<2><<<grid, 256, 0, st>>>(a, vec_a, vec_b);
  DM_CHECK_LAUNCH();
  if (a.splits > 1) {
    const long long mn = (long long)M * N;
    fc_reduce_kernel<<<(unsigned)((mn + 255) / 256), 256, 0, st>>>((const float *)workspace, a.splits, mn, N, bias, relu,
                                                                 C, ldc);
    DM_CHECK_LAUNCH();
  }
  return DM_OK;
}
