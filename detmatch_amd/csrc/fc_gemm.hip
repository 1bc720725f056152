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
#include <stdlib.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FC_T = 64;        // output tile rows (columns: 64 * NB)
constexpr int FC_DEFAULT_BK = 32;
constexpr int FC_DEFAULT_NB_LARGE = 1;

struct FcArgs {
  const float *A, *B, *bias;
  float *C;          // output, or the partial buffer [splits][M][N] when splits > 1
  int M, N, K, lda, ldb, ldc;
  int steps_per_split, relu, splits;
  int tiles_m, tiles_n, per_xcd;
};

// One 64 x BK operand tile in registers: BK / 16 float4 per thread (4 consecutive elements of the operand's contiguous
// dimension each).
// ROWS_ALONG_K = false: the operand is [row][k] in memory (k contiguous): float4 i -> (row = i / (BK / 4), k4 = i % (BK / 4)).
// ROWS_ALONG_K = true:  the operand is [k][row] in memory (row contiguous): float4 i -> (k = i / 16, r4 = i % 16).
template <int BK>
struct TileRegs {
  float4 v[BK / 16];
};

template <bool ROWS_ALONG_K, int BK>
__device__ __forceinline__ void fetch_tile(TileRegs<BK> &t, const float *P, int ld, int row0, int n_rows,
                                           int k0, int k_hi, int tid, bool vec) {
#pragma unroll
  for (int h = 0; h < BK / 16; ++h) {
    const int i = tid + 256 * h;
    int r, k;
    if (!ROWS_ALONG_K) {
      r = i / (BK / 4);
      k = k0 + 4 * (i % (BK / 4));
    } else {
      k = k0 + i / 16;
      r = 4 * (i % 16);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int row = row0 + r;
    if (!ROWS_ALONG_K) {
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

template <bool ROWS_ALONG_K, int BK>
__device__ __forceinline__ void stage_tile(const TileRegs<BK> &t, float *S, int tid) {
  constexpr int LD = BK + 1;
#pragma unroll
  for (int h = 0; h < BK / 16; ++h) {
    const int i = tid + 256 * h;
    const float4 v = t.v[h];
    if (!ROWS_ALONG_K) {
      float *d = S + (i / (BK / 4)) * LD + 4 * (i % (BK / 4));
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    } else {
      const int k = i / 16, r = 4 * (i % 16);
      S[(r + 0) * LD + k] = v.x;
      S[(r + 1) * LD + k] = v.y;
      S[(r + 2) * LD + k] = v.z;
      S[(r + 3) * LD + k] = v.w;
    }
  }
}

// FORM 0: A [M][K], B [N][K];  FORM 1: A [M][K], B [K][N];  FORM 2: A [K][M], B [K][N]
// Workgroup tile 64 x (64 * NB): wave (wm, wn) owns rows wm * 32 .. + 32 and NB column blocks of 32 at
// wn * 32 * NB (one A fragment feeds NB matrix instructions); the contraction walks BK at a time.
template <int FORM, int BK, int NB>
__global__ __launch_bounds__(256) void fc_gemm_kernel(FcArgs a, int vec_a, int vec_b) {
  constexpr bool A_T = FORM == 2, B_T = FORM != 0;
  constexpr int LD = BK + 1;
  extern __shared__ float fc_lds[];
  float *As = fc_lds;                    // [64][LD]
  float *Bs = fc_lds + FC_T * LD;        // [64 * NB][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the ids one XCD sees are
  // L, L + 8, L + 16, ...; they are mapped to CONSECUTIVE logical tiles = (split, n tile, m tile) with m fastest — the
  // workgroups that re-read the same B columns (skinny M: 4 m tiles) or the same K slice of both operands (split
  // contraction: all tiles of a split) run next to each other on ONE XCD.  (Measured neutral on the shapes of the step:
  // the re-reads already hit in the memory-side cache; kept because it costs nothing.)
  const int logical = (int)(blockIdx.x % 8) * a.per_xcd + (int)(blockIdx.x / 8);
  const int n_tiles = a.tiles_m * a.tiles_n;
  if (logical >= n_tiles * a.splits) return;
  const int split = logical / n_tiles, tile = logical - split * n_tiles;
  const int m0 = (tile % a.tiles_m) * FC_T, n0 = (tile / a.tiles_m) * FC_T * NB;
  const int steps = (a.K + BK - 1) / BK;
  const int s_lo = split * a.steps_per_split;
  const int s_hi = min(steps, s_lo + a.steps_per_split);
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  TileRegs<BK> ra, rb[NB];
  if (s_lo < s_hi) {
    fetch_tile<A_T, BK>(ra, a.A, a.lda, m0, a.M, s_lo * BK, a.K, tid, vec_a != 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) fetch_tile<B_T, BK>(rb[b], a.B, a.ldb, n0 + 64 * b, a.N, s_lo * BK, a.K, tid, vec_b != 0);
  }
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();                               // the previous step's fragments have been read
    stage_tile<A_T, BK>(ra, As, tid);
#pragma unroll
    for (int b = 0; b < NB; ++b) stage_tile<B_T, BK>(rb[b], Bs + 64 * b * LD, tid);
    __syncthreads();
    if (s + 1 < s_hi) {                            // in flight under the MFMAs
      fetch_tile<A_T, BK>(ra, a.A, a.lda, m0, a.M, (s + 1) * BK, a.K, tid, vec_a != 0);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        fetch_tile<B_T, BK>(rb[b], a.B, a.ldb, n0 + 64 * b, a.N, (s + 1) * BK, a.K, tid, vec_b != 0);
    }
    const float *pa = As + (wm * 32 + lr) * LD + lh;
    const float *pb = Bs + (wn * 32 * NB + lr) * LD + lh;
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
      const float av = pa[k];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb[b * 32 * LD + k], acc[b], 0, 0, 0);
    }
  }
  // C layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const bool direct = a.splits == 1;
  float *out = direct ? a.C : a.C + (size_t)split * a.M * a.N;
  const int ldo = direct ? a.ldc : a.N;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = n0 + wn * 32 * NB + b * 32 + lr;
    if (col >= a.N) continue;
    const float bv = (direct && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < a.M) {
        float v = acc[b][r] + bv;
        if (direct && a.relu) v = fmaxf(v, 0.f);
        out[(size_t)row * ldo + col] = v;
      }
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
int fc_splits(int M, int N, int K, int bk, int nb) {
  const long long tiles = (long long)dm_ceil_div(M, FC_T) * dm_ceil_div(N, FC_T * nb);
  const int steps = dm_ceil_div(K, bk);
  if (tiles >= 256 || steps * bk < 512) return 1;
  long long want = (512 + tiles - 1) / tiles;            // ~2 workgroups per compute unit
  const int most = steps * bk / 256 > 0 ? steps * bk / 256 : 1;   // at least 256 of K per split
  if (want > most) want = most;
  if (want > 64) want = 64;
  return want < 1 ? 1 : (int)want;
}

// (contraction step, column blocks per wave) by shape; DM_FC_VARIANT=<bk><nb> (e.g. 642) forces one for tools/bench_fc.py
void fc_variant(int M, int N, int K, int *bk, int *nb) {
  static int forced = -1;
  if (forced < 0) {
    const char *e = getenv("DM_FC_VARIANT");
    forced = e ? atoi(e) : 0;
  }
  if (forced > 0) {
    *bk = forced / 10 == 64 ? 64 : 32;
    *nb = forced % 10 == 2 ? 2 : 1;
    return;
  }
  *bk = FC_DEFAULT_BK;
  *nb = (N >= 128 && (long long)dm_ceil_div(M, FC_T) * dm_ceil_div(N, 128) >= 128) ? FC_DEFAULT_NB_LARGE : 1;
  (void)K;
}

template <int FORM>
void fc_launch(const FcArgs &a, int bk, int nb, int vec_a, int vec_b, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)FC_T * (1 + nb) * (bk + 1) * sizeof(float);
  if (bk == 32 && nb == 1) fc_gemm_kernel<FORM, 32, 1><<<grid, 256, lds, st>>>(a, vec_a, vec_b);
  else if (bk == 32 && nb == 2) fc_gemm_kernel<FORM, 32, 2><<<grid, 256, lds, st>>>(a, vec_a, vec_b);
  else if (bk == 64 && nb == 1) fc_gemm_kernel<FORM, 64, 1><<<grid, 256, lds, st>>>(a, vec_a, vec_b);
  else fc_gemm_kernel<FORM, 64, 2><<<grid, 256, lds, st>>>(a, vec_a, vec_b);
}

}  // namespace

extern "C" size_t dm_fc_gemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int bk, nb;
  fc_variant(M, N, K, &bk, &nb);
  const int s = fc_splits(M, N, K, bk, nb);
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
  int bk, nb;
  fc_variant(M, N, K, &bk, &nb);
  const int steps = dm_ceil_div(K > 0 ? K : 1, bk);
  a.splits = K > 0 ? fc_splits(M, N, K, bk, nb) : 1;
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
  a.tiles_m = dm_ceil_div(M, FC_T), a.tiles_n = dm_ceil_div(N, FC_T * nb);
  const long long total = (long long)a.tiles_m * a.tiles_n * a.splits;
  if (total > 0x3fffffff) return DM_ERR_INVALID_ARG;
  a.per_xcd = (int)((total + 7) / 8);
  dim3 grid((unsigned)(a.per_xcd * 8), 1, 1);
  if (form == 0) fc_launch<0>(a, bk, nb, vec_a, vec_b, grid, st);
  else if (form == 1) fc_launch<1>(a, bk, nb, vec_a, vec_b, grid, st);
  else fc_launch<2>(a, bk, nb, vec_a, vec_b, grid, st);
  DM_CHECK_LAUNCH();
  if (a.splits > 1) {
    const long long mn = (long long)M * N;
    fc_reduce_kernel<<<(unsigned)((mn + 255) / 256), 256, 0, st>>>((const float *)workspace, a.splits, mn, N, bias, relu,
                                                                 C, ldc);
    DM_CHECK_LAUNCH();
  }
  return DM_OK;
}
