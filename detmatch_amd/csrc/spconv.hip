// Fused sparse-conv gather-GEMM-scatter for gfx950 (fp32, exact).
//
// Replaces the reference's 27 x {gather kernel, cuBLAS SGEMM, scatter-add kernel}
//   mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-360 (indiceConv)
//   mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456 (indiceConvBackward)
//   mmdet3d/ops/spconv/include/spconv/reordering.cu.h:22-160
// with ONE output-stationary launch per layer:
//   out[o,:] = sum_k feat[nbr[k][o],:] @ W[k]
// A workgroup owns 16*WAVES consecutive output rows; every wave owns 16 of
// them and all output channels.  Per kernel offset k the (cin x cout) weight
// slice is staged once per workgroup in LDS (double-buffered, 16-byte
// conflict-free reads), the gathered input rows go global -> VGPR as float4
// directly in the v_mfma_f32_16x16x4_f32 A-operand layout (no LDS round trip,
// no intermediate buffers in HBM), and the accumulators stay in registers until
// the single final store — no atomics, no read-modify-write of `out`.
// Kernel offsets that feed none of the workgroup's rows are skipped entirely
// (no weight staging, no barrier).
//
// Algorithmic bytes per pair: (cin + cout)*4 + 8.
#include "dm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- weight packing -------------------------------------------------------
// Packed layout per kernel offset k (cin x cout floats):
//   wp[k][t][nb][kq][n][j] = B_k[16t + 4kq + j][16nb + n]
// i.e. exactly the order in which a wave's lanes (n = lane&15, kq = lane>>4)
// consume B operands, 4 consecutive k-steps (j) per float4.
// B_k = W[k] (forward) or W[kk]^T with kk = flip ? kvol-1-k : k (input grad).
// For cin_eff == 4 (first layer) the layout degenerates to t = 0, kq = row.
__global__ __launch_bounds__(256) void pack_weights(const float *w, float *wp, int kvol, int ci,
                                                    int co, int transpose_w, int flip_k) {
  // B_k is (ci x co).  forward: W is (kvol, ci, co); transposed: W is (kvol, co, ci).
  int per_k = ci * co;
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kvol * per_k) return;
  int k = e / per_k;
  int r = e % per_k;
  int c, col;
  if (ci >= 16) {  // r = (((t*NB + nb)*4 + kq)*16 + n)*4 + j
    int j = r & 3, n = (r >> 2) & 15, kq = (r >> 6) & 3;
    int nbs = co / 16;
    int nb = (r >> 8) % nbs, t = (r >> 8) / nbs;
    c = 16 * t + 4 * kq + j;
    col = 16 * nb + n;
  } else {  // ci == 4: r = (nb*4 + kq)*16 + n
    int n = r & 15, kq = (r >> 4) & 3, nb = r >> 6;
    c = kq;
    col = 16 * nb + n;
  }
  int kk = flip_k ? kvol - 1 - k : k;
  wp[e] = transpose_w ? w[((size_t)kk * co + col) * ci + c] : w[((size_t)kk * ci + c) * co + col];
}

// ---- main kernel ----------------------------------------------------------
template <int CIN, int COUT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spconv_gg(const float *__restrict__ feat,
                                                        const float *__restrict__ wpack,
                                                        const int32_t *__restrict__ nbr,
                                                        int n_out, int kvol,
                                                        float *__restrict__ out) {
  constexpr int ROWS = 16 * WAVES;
  constexpr int NB = COUT / 16;
  constexpr int CT = CIN >= 16 ? CIN / 16 : 1;
  constexpr int WSZ = CIN * COUT;          // floats per kernel offset
  constexpr int NT = WAVES * 64;
  constexpr int W4 = WSZ / 4;              // float4 per offset
  constexpr int W4_PER_T = (W4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *wl = (float *)smem;                           // 2 * WSZ floats
  int32_t *nb_l = (int32_t *)(wl + 2 * WSZ);           // kvol * ROWS
  unsigned int *active_mask_p = (unsigned int *)(nb_l + kvol * ROWS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * ROWS;

  if (tid == 0) *active_mask_p = 0u;
  __syncthreads();
  // stage this tile's slice of the gather table; build the active-offset mask
  unsigned int my_mask = 0u;
  for (int e = tid; e < kvol * ROWS; e += NT) {
    int k = e / ROWS, rr = e % ROWS;
    int row = row0 + rr;
    int v = row < n_out ? nbr[(size_t)k * n_out + row] : -1;
    nb_l[e] = v;
    if (v >= 0) my_mask |= 1u << k;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) my_mask |= __shfl_xor(my_mask, d);
  if (lane == 0 && my_mask) atomicOr(active_mask_p, my_mask);
  __syncthreads();
  unsigned int active = *active_mask_p;

  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (active != 0u) {
    // prologue: first active offset -> LDS buffer 0
    int k = __ffs(active) - 1;
    unsigned int rest = active & (active - 1u);
    {
      const f32x4 *src = (const f32x4 *)(wpack + (size_t)k * WSZ);
      f32x4 *dst = (f32x4 *)wl;
#pragma unroll
      for (int i = 0; i < W4_PER_T; ++i) {
        int e = tid + i * NT;
        if (e < W4) dst[e] = src[e];
      }
    }
    int buf = 0;
    // A rows of the current offset
    f32x4 a[CT];
    {
      int idx = nb_l[k * ROWS + wave * 16 + r];
#pragma unroll
      for (int t = 0; t < CT; ++t) a[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (idx >= 0) {
        if (CIN >= 16) {
#pragma unroll
          for (int t = 0; t < CT; ++t)
            a[t] = *(const f32x4 *)(feat + (size_t)idx * CIN + 16 * t + 4 * kq);
        } else {
          a[0][0] = feat[(size_t)idx * CIN + kq];
        }
      }
    }
    while (true) {
      __syncthreads();  // LDS[buf] holds W[k]; nobody still reads LDS[buf^1]
      // prefetch next active offset: weights -> registers, A rows -> registers
      int kn = rest ? __ffs(rest) - 1 : -1;
      f32x4 wreg[W4_PER_T];
      f32x4 an[CT];
      if (kn >= 0) {
        const f32x4 *src = (const f32x4 *)(wpack + (size_t)kn * WSZ);
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) {
          int e = tid + i * NT;
          if (e < W4) wreg[i] = src[e];
        }
        int idx = nb_l[kn * ROWS + wave * 16 + r];
#pragma unroll
        for (int t = 0; t < CT; ++t) an[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx >= 0) {
          if (CIN >= 16) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
              an[t] = *(const f32x4 *)(feat + (size_t)idx * CIN + 16 * t + 4 * kq);
          } else {
            an[0][0] = feat[(size_t)idx * CIN + kq];
          }
        }
      }
      // compute offset k (skip if none of this wave's 16 rows has a neighbour)
      {
        int idx = nb_l[k * ROWS + wave * 16 + r];
        if (__ballot(idx >= 0) != 0ull) {
          const float *wb = wl + buf * WSZ;
          if (CIN >= 16) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
              f32x4 b[NB];
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                b[nb] = *(const f32x4 *)(wb + (((t * NB + nb) * 4 + kq) * 16 + r) * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                  acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[nb][j], acc[nb], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              float b = wb[(nb * 4 + kq) * 16 + r];
              acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][0], b, acc[nb], 0, 0, 0);
            }
          }
        }
      }
      if (kn < 0) break;
      // stash next weights into the other LDS buffer
      {
        f32x4 *dst = (f32x4 *)(wl + (buf ^ 1) * WSZ);
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) {
          int e = tid + i * NT;
          if (e < W4) dst[e] = wreg[i];
        }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) a[t] = an[t];
      k = kn;
      rest &= rest - 1u;
      buf ^= 1;
    }
  }
  // epilogue: D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int row = row0 + wave * 16 + kq * 4 + reg;
    if (row < n_out) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) out[(size_t)row * COUT + 16 * nb + r] = acc[nb][reg];
    }
  }
}

template <int CIN, int COUT, int WAVES>
int launch_gg_w(const float *feat, const float *wpack, const int32_t *nbr, int n_out, int kvol,
                float *out, hipStream_t st) {
  size_t smem = 2ull * CIN * COUT * sizeof(float) + (size_t)kvol * 16 * WAVES * sizeof(int32_t) + 16;
  static bool attr_set = false;  // > 64 KB of dynamic LDS needs the opt-in
  if (!attr_set) {
    DM_HIP(hipFuncSetAttribute((const void *)spconv_gg<CIN, COUT, WAVES>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set = true;
  }
  spconv_gg<CIN, COUT, WAVES><<<dm_ceil_div(n_out, 16 * WAVES), WAVES * 64, smem, st>>>(
      feat, wpack, nbr, n_out, kvol, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

template <int CIN, int COUT>
int launch_gg(const float *feat, const float *wpack, const int32_t *nbr, int n_out, int kvol,
              float *out, hipStream_t st) {
  // 64-row tiles when they still fill the chip (256 CUs), else 32-row tiles
  if (n_out >= 64 * 512) return launch_gg_w<CIN, COUT, 4>(feat, wpack, nbr, n_out, kvol, out, st);
  return launch_gg_w<CIN, COUT, 2>(feat, wpack, nbr, n_out, kvol, out, st);
}

// ---- weight gradient --------------------------------------------------------
// One wave per (kernel offset k, 16-channel block of cin, chunk of pairs):
// dW[k][16cb..][:] += X[in[s]][16cb..]^T (x) dY[out[s]][:] over the chunk,
// 4 pairs per MFMA step.  Partials go to a slab, a second kernel sums the
// slabs in chunk order (bitwise reproducible, no float atomics).
template <int COUT>
__global__ __launch_bounds__(64) void spconv_wgrad_partial(const float *__restrict__ feat,
                                                           const float *__restrict__ ograd,
                                                           const int32_t *__restrict__ pairs,
                                                           const int32_t *__restrict__ indice_num,
                                                           int pair_stride, int cin, int chunk,
                                                           float *__restrict__ slab) {
  constexpr int NB = COUT / 16;
  const int k = blockIdx.z, cb = blockIdx.y, ch = blockIdx.x;
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  const int kvol = gridDim.z;
  const int npairs = indice_num[k];
  const int s_begin = ch * chunk;
  int s_end = s_begin + chunk;
  if (s_end > npairs) s_end = npairs;
  const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pair_stride;
  const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pair_stride;
  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool col_ok = (cin >= 16) || (m < cin);
  for (int s0 = s_begin; s0 < s_end; s0 += 16) {
    float a[4];
    float b[4][NB];
    int ii[4], oo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int s = s0 + 4 * u + kq;
      bool ok = s < s_end;
      ii[u] = ok ? pin[s] : -1;
      oo[u] = ok ? pout[s] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = (ii[u] >= 0 && col_ok) ? feat[(size_t)ii[u] * cin + 16 * cb + m] : 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        b[u][nb] = oo[u] >= 0 ? ograd[(size_t)oo[u] * COUT + 16 * nb + m] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][nb], acc[nb], 0, 0, 0);
  }
  // D: row (c within block) = 4*kq + reg, col (n) = m
  float *dst = slab + ((size_t)ch * kvol + k) * (size_t)cin * COUT;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int c = 16 * cb + 4 * kq + reg;
    if (c < cin) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dst[(size_t)c * COUT + 16 * nb + m] = acc[nb][reg];
    }
  }
}

__global__ __launch_bounds__(256) void spconv_wgrad_reduce(const float *slab, int nchunks,
                                                           size_t per_chunk, float *filt_grad) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += slab[(size_t)c * per_chunk + e];
  filt_grad[e] = s;
}

int wgrad_chunks(int n_in, int *chunk) {
  // ~16 chunks per offset at KITTI sizes; multiples of 4 pairs
  int c = 512;
  while ((long long)c * 64 < n_in) c *= 2;
  *chunk = c;
  return dm_ceil_div(n_in > 0 ? n_in : 1, c);
}

bool chan_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

}  // namespace

extern "C" size_t dm_spconv_workspace_bytes(int kvol, int cin, int cout) {
  if (kvol <= 0 || cin <= 0 || cout <= 0) return 0;
  return dm_align((size_t)kvol * cin * cout * sizeof(float));
}

#define DM_GG_CASE(CI, CO)                                                              \
  if (ci == CI && co == CO) return launch_gg<CI, CO>(feat, wp, nbr, n_rows_out, kvol, out, st);

extern "C" int dm_spconv_gather_gemm(const float *feat, int n_rows_in, const float *filters,
                                     const int32_t *nbr, int n_rows_out, int kvol, int cin,
                                     int cout, int transpose_w, int flip_k, float *out,
                                     void *workspace, size_t workspace_bytes,
                                     dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_rows_in < 0 || n_rows_out < 0 || kvol <= 0 || kvol > 32) return DM_ERR_INVALID_ARG;
  // effective B operand dims: (ci x co)
  int ci = transpose_w ? cout : cin;
  int co = transpose_w ? cin : cout;
  if (!((ci == 4 || chan_ok(ci)) && chan_ok(co))) return DM_ERR_UNSUPPORTED;
  if (n_rows_out == 0) return DM_OK;
  if (!filters || !nbr || !out || !workspace || (n_rows_in > 0 && !feat)) return DM_ERR_INVALID_ARG;
  size_t need = dm_spconv_workspace_bytes(kvol, cin, cout);
  if (workspace_bytes < need) return DM_ERR_WORKSPACE;
  float *wp = (float *)workspace;
  int total = kvol * ci * co;
  // pack_weights indexes W as (kvol, ci, co) when !transpose_w and as
  // (kvol, co_w = co.., ) transposed otherwise: W is (kvol, cin, cout) = (kvol, co, ci)
  pack_weights<<<dm_ceil_div(total, 256), 256, 0, st>>>(filters, wp, kvol, ci, co, transpose_w,
                                                        flip_k);
  DM_CHECK_LAUNCH();
  DM_GG_CASE(4, 16)
  DM_GG_CASE(16, 16)
  DM_GG_CASE(16, 32)
  DM_GG_CASE(32, 16)
  DM_GG_CASE(32, 32)
  DM_GG_CASE(32, 64)
  DM_GG_CASE(64, 32)
  DM_GG_CASE(64, 64)
  DM_GG_CASE(64, 128)
  DM_GG_CASE(128, 64)
  return DM_ERR_UNSUPPORTED;
}

extern "C" size_t dm_spconv_wgrad_workspace_bytes(int n_in, int kvol, int cin, int cout) {
  if (kvol <= 0 || cin <= 0 || cout <= 0 || n_in < 0) return 0;
  int chunk;
  int nchunks = wgrad_chunks(n_in, &chunk);
  return dm_align((size_t)nchunks * kvol * cin * cout * sizeof(float));
}

extern "C" int dm_spconv_wgrad(const float *feat, const float *out_grad,
                               const int32_t *indice_pairs, const int32_t *indice_num,
                               int pair_stride, int kvol, int cin, int cout, float *filt_grad,
                               void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kvol <= 0 || pair_stride < 0 || !filt_grad) return DM_ERR_INVALID_ARG;
  if (!((cin == 4 || chan_ok(cin)) && chan_ok(cout))) return DM_ERR_UNSUPPORTED;
  size_t per_chunk = (size_t)kvol * cin * cout;
  if (pair_stride == 0) {
    DM_HIP(hipMemsetAsync(filt_grad, 0, per_chunk * sizeof(float), st));
    return DM_OK;
  }
  if (!feat || !out_grad || !indice_pairs || !indice_num || !workspace) return DM_ERR_INVALID_ARG;
  int chunk;
  int nchunks = wgrad_chunks(pair_stride, &chunk);
  if (workspace_bytes < dm_spconv_wgrad_workspace_bytes(pair_stride, kvol, cin, cout))
    return DM_ERR_WORKSPACE;
  float *slab = (float *)workspace;
  dim3 grid(nchunks, dm_ceil_div(cin, 16), kvol);
  switch (cout) {
    case 16:
      spconv_wgrad_partial<16><<<grid, 64, 0, st>>>(feat, out_grad, indice_pairs, indice_num,
                                                    pair_stride, cin, chunk, slab);
      break;
    case 32:
      spconv_wgrad_partial<32><<<grid, 64, 0, st>>>(feat, out_grad, indice_pairs, indice_num,
                                                    pair_stride, cin, chunk, slab);
      break;
    case 64:
      spconv_wgrad_partial<64><<<grid, 64, 0, st>>>(feat, out_grad, indice_pairs, indice_num,
                                                    pair_stride, cin, chunk, slab);
      break;
    default:
      spconv_wgrad_partial<128><<<grid, 64, 0, st>>>(feat, out_grad, indice_pairs, indice_num,
                                                     pair_stride, cin, chunk, slab);
      break;
  }
  DM_CHECK_LAUNCH();
  spconv_wgrad_reduce<<<dm_ceil_div((long long)per_chunk, 256), 256, 0, st>>>(slab, nchunks,
                                                                              per_chunk, filt_grad);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
