// Sparse gather-GEMM on gfx950's 16-bit matrix instructions (v_mfma_f32_16x16x32_{bf16,f16}), fp32
// accumulation.  Counterpart of the reference's half-precision bindings
//   mmdet3d/ops/spconv/src/all.cc:35-36      indice_conv_half / indice_conv_backward_half
//   mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-456 instantiated for at::Half
// and the sparse half of the mixed-precision mode that serves the reference's fp16 configs
// (mmdet3d/apis/ssl_train.py:100-105, BASELINE configs[4]).  Three storage modes, one kernel:
//   DM_SP16_F32ROWS  rows, filters and output fp32 in HBM (nothing else of the framework changes);
//                    multiplicands rounded to bf16 (RNE) on their way into the matrix pipe
//   DM_SP16_F16      rows / filters / output IEEE half (the reference's at::Half path)
//   DM_SP16_BF16     rows / filters / output bfloat16
//   DM_SP16_F32SPLIT rows, filters and output fp32; fp32-CLASS arithmetic: every multiplicand is split into three
//                    bf16 numbers (h + m + l = the 24 significand bits), the six cross products of weight 2^-16
//                    and larger run on the bf16 instruction with fp32 accumulation — more accurate against
//                    float64 than v_mfma_f32_16x16x4_f32 (see conv2d.hip / tools/probe_bf16_split.py) at 3/8 of
//                    its matrix-pipe time, which is what bounds the fp32 kernel
// Same output-stationary structure as spconv_gr (spconv.hip): a workgroup = 4 waves = one tile of 16
// output rows, the tile's active kernel offsets dealt round-robin to the waves, the (cin x cout) B operand
// of the current offset in registers, one LDS meeting of the four fp32 partial tiles summed in wave order
// (bitwise reproducible), one store.  One 16x16x32 instruction does the work of eight fp32 16x16x4 ones
// in half the cycles: the matrix pipe, which bounds the fp32 kernel, drops out and the
// launch is paced by the table -> row fetch; with 16-bit rows the gathered bytes halve as well.
//
// Algorithmic bytes per pair: (cin + cout) * T + 8, T = 4 (F32ROWS) or 2.
#include <hip/hip_ext.h>

#include "dm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename MT>
struct Math;
template <>
struct Math<__bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Math<_Float16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// 8 consecutive channels of a row as the math type
template <typename ST, typename MT>
__device__ __forceinline__ typename Math<MT>::v8 load8(const ST *p) {
  if constexpr (sizeof(ST) == 4) {
    const f32x4 lo = *(const f32x4 *)p, hi = *(const f32x4 *)(p + 4);
    const f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_convertvector(v, typename Math<MT>::v8);       // v_cvt_pk_bf16_f32: round to nearest even
  } else {
    return *(const typename Math<MT>::v8 *)p;
  }
}

template <typename ST, typename MT>
__device__ __forceinline__ void store4(ST *p, f32x4 v) {
  if constexpr (sizeof(ST) == 4) {
    *(f32x4 *)p = v;
  } else {
    *(typename Math<MT>::v4 *)p = __builtin_convertvector(v, typename Math<MT>::v4);
  }
}

// ---- weight packing ---------------------------------------------------------------------------
// wp[k][t][nb][lane][j] = B_k[32 t + 8 (lane >> 4) + j][16 nb + (lane & 15)], zero beyond ci rows:
// the order in which a wave's lanes consume B operands of v_mfma_f32_16x16x32.  B_k = W[k] (forward) or
// W[kk]^T, kk = flip ? kvol-1-k : k (input gradient).  W: (kvol, cin_w, cout_w) of type WT.
// SPLIT = 3: three planes per (k, t, nb) block, wp[k][t][nb][plane][lane][j], plane 0 = h, 1 = m, 2 = l.
template <typename WT, typename MT, int SPLIT>
__global__ __launch_bounds__(256) void pack_weights16(const WT *__restrict__ w, MT *__restrict__ wp, int kvol,
                                                      int ci, int co, int transpose_w, int flip_k) {
  const int ct = (ci + 31) / 32, nbs = co / 16;
  const int per_k = ct * nbs * 512;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kvol * per_k) return;
  const int k = e / per_k, r = e % per_k;
  const int j = r & 7, lane = (r >> 3) & 63, nb = (r >> 9) % nbs, t = (r >> 9) / nbs;
  const int c = 32 * t + 8 * (lane >> 4) + j, col = 16 * nb + (lane & 15);
  const int kk = flip_k ? kvol - 1 - k : k;
  float v = 0.f;
  if (c < ci) v = (float)(transpose_w ? w[((size_t)kk * co + col) * ci + c] : w[((size_t)kk * ci + c) * co + col]);
  if constexpr (SPLIT == 3) {
    const MT h = (MT)v;
    const float r1 = v - (float)h;
    const MT m = (MT)r1;
    const MT l = (MT)(r1 - (float)m);
    MT *dst = wp + ((size_t)(k * ct + t) * nbs + nb) * (3 * 512) + lane * 8 + j;
    dst[0] = h, dst[512] = m, dst[1024] = l;
  } else {
    wp[e] = (MT)v;
  }
}

// ---- main kernel ------------------------------------------------------------------------------
template <int CIN, int COUT, typename ST, typename MT, int SPLIT>
__global__ __launch_bounds__(256) void spconv_gr16(const ST *__restrict__ feat, const MT *__restrict__ wpack,
                                                   const int32_t *__restrict__ nbr,
                                                   const int32_t *__restrict__ perm, int n_out, int kvol,
                                                   int cout_full, ST *__restrict__ out,
                                                   const int32_t *__restrict__ tile_order) {
  typedef typename Math<MT>::v8 v8;
  constexpr int NB = COUT / 16;
  constexpr int CT = (CIN + 31) / 32;          // 32-channel k blocks per kernel offset
  constexpr int CTS = (CT < 2 || SPLIT == 3) ? 1 : 2;   // ... per pipeline step (the split keeps 3 planes of B live)
  constexpr int S = CT / CTS;
  constexpr int LDP = COUT + 4;
  __shared__ int32_t tbl[4][32][16];
  __shared__ __attribute__((aligned(16))) float part[4][16][LDP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int row0 = (tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x) * 16;
  const int nb_full = cout_full / 16;
  const int nb0 = blockIdx.y * NB;
  // lanes whose 8-channel slice lies beyond CIN (CIN == 16: kq >= 2) contribute zeros
  const bool chan_live = 8 * kq < CIN;

  unsigned int active = 0u;
  {
    int v[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const int k = 4 * k4 + kq;
      const bool in = (k < kvol) && (row0 + r < n_out);
      v[k4] = nbr[in ? (size_t)k * n_out + row0 + r : 0];
      if (!in) v[k4] = -1;
    }
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      tbl[wave][4 * k4 + kq][r] = v[k4];
      const unsigned long long m = __ballot(v[k4] >= 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((m >> (16 * q)) & 0xFFFFull) active |= 1u << (4 * k4 + q);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  unsigned int mine = 0u;
  {
    int rank = 0;
    for (unsigned int a = active; a; a &= a - 1u) {
      if ((rank & 3) == wave) mine |= a & (0u - a);
      ++rank;
    }
  }
  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // w: [t][nb][plane], a: [t][plane]
  auto load_step = [&](int k, int s, v8 *w, v8 *a) {
    const int idx = tbl[wave][k][r];
    const bool ok = (idx >= 0) && chan_live;
    const ST *src = feat + (size_t)(idx >= 0 ? idx : 0) * CIN + 32 * (s * CTS) + (chan_live ? 8 * kq : 0);
    v8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (MT)0.f;
#pragma unroll
    for (int t = 0; t < CTS; ++t) {
      if constexpr (SPLIT == 3) {
        const f32x4 lo = *(const f32x4 *)(src + 32 * t), hi = *(const f32x4 *)(src + 32 * t + 4);
        const f32x8 x = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const v8 h = __builtin_convertvector(x, v8);
        const f32x8 r1 = x - __builtin_convertvector(h, f32x8);
        const v8 m = __builtin_convertvector(r1, v8);
        const v8 l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x8), v8);
        a[3 * t + 0] = ok ? h : z, a[3 * t + 1] = ok ? m : z, a[3 * t + 2] = ok ? l : z;
      } else {
        const v8 raw = load8<ST, MT>(src + 32 * t);          // unconditional (index clamped), then masked
        a[t] = ok ? raw : z;
      }
    }
    const v8 *wk = (const v8 *)wpack + (size_t)k * (CT * (size_t)nb_full * 64 * SPLIT);
#pragma unroll
    for (int t = 0; t < CTS; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          w[(t * NB + nb) * SPLIT + pl] = wk[(((s * CTS + t) * nb_full + nb0 + nb) * SPLIT + pl) * 64 + lane];
  };
  auto compute = [&](const v8 *w, const v8 *a) {
#pragma unroll
    for (int t = 0; t < CTS; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const v8 *wb = w + (t * NB + nb) * SPLIT;
        if constexpr (SPLIT == 3) {     // smallest terms first: l h, h l, m m, m h, h m, h h
          acc[nb] = Math<MT>::mfma(a[3 * t + 2], wb[0], acc[nb]);
          acc[nb] = Math<MT>::mfma(a[3 * t + 0], wb[2], acc[nb]);
          acc[nb] = Math<MT>::mfma(a[3 * t + 1], wb[1], acc[nb]);
          acc[nb] = Math<MT>::mfma(a[3 * t + 1], wb[0], acc[nb]);
          acc[nb] = Math<MT>::mfma(a[3 * t + 0], wb[1], acc[nb]);
          acc[nb] = Math<MT>::mfma(a[3 * t + 0], wb[0], acc[nb]);
        } else {
          acc[nb] = Math<MT>::mfma(a[t], wb[0], acc[nb]);
        }
      }
  };

  if (mine != 0u) {
    const int n_steps = __popc(mine) * S;
    unsigned int rest = mine;
    int k = __ffs(rest) - 1, s = 0;
    auto advance = [&]() {
      if (s + 1 < S) {
        ++s;
      } else if (rest & (rest - 1u)) {
        rest &= rest - 1u;
        k = __ffs(rest) - 1;
        s = 0;
      }
    };
    v8 w0[CTS * NB * SPLIT], w1[CTS * NB * SPLIT], a0[CTS * SPLIT], a1[CTS * SPLIT];
    load_step(k, s, w0, a0);
    int i = 0;
    // pairs of steps as one straight-line loop body, the odd step behind the loop (see spconv_gr)
    for (; i + 2 <= n_steps; i += 2) {
      advance();
      load_step(k, s, w1, a1);
      compute(w0, a0);
      advance();
      load_step(k, s, w0, a0);
      compute(w1, a1);
    }
    if (i < n_steps) compute(w0, a0);
  }
  // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) part[wave][4 * kq + reg][16 * nb + r] = acc[nb][reg];
  __syncthreads();
  constexpr int F4_PER_ROW = COUT / 4;
  if (tid < 16 * F4_PER_ROW) {
    const int rr = tid / F4_PER_ROW, c4 = tid % F4_PER_ROW;
    const int prow = row0 + rr;
    if (prow < n_out) {
      const f32x4 v0 = *(const f32x4 *)&part[0][rr][4 * c4];
      const f32x4 v1 = *(const f32x4 *)&part[1][rr][4 * c4];
      const f32x4 v2 = *(const f32x4 *)&part[2][rr][4 * c4];
      const f32x4 v3 = *(const f32x4 *)&part[3][rr][4 * c4];
      const f32x4 sum = (v0 + v1) + (v2 + v3);
      const int row = perm ? perm[prow] : prow;
      store4<ST, MT>(out + (size_t)row * cout_full + blockIdx.y * COUT + 4 * c4, sum);
    }
  }
}

template <int CIN, int COUT_FULL, typename ST, typename MT, int SPLIT>
int launch16(const void *feat, const void *wpack, const int32_t *nbr, const int32_t *perm,
             const int32_t *tile_order, int n_out, int kvol, void *out, int mode, hipStream_t st) {
  constexpr int COUT = COUT_FULL > 64 ? 64 : COUT_FULL;
  dim3 grid(dm_ceil_div(n_out, 16), COUT_FULL / COUT);
  hipEvent_t e0, e1;
  // profile tag c = 16 + storage mode: bench.py tells these launches from the fp32 kernels
  if (dm_prof_open(DM_PROF_SPCONV_GG, CIN, COUT_FULL, 16 + mode, n_out, kvol, nbr, &e0, &e1) >= 0)
    hipExtLaunchKernelGGL((spconv_gr16<CIN, COUT, ST, MT, SPLIT>), grid, dim3(256), 0, st, e0, e1, 0, (const ST *)feat,
                          (const MT *)wpack, nbr, perm, n_out, kvol, (int)COUT_FULL, (ST *)out, tile_order);
  else
    spconv_gr16<CIN, COUT, ST, MT, SPLIT><<<grid, 256, 0, st>>>((const ST *)feat, (const MT *)wpack, nbr, perm, n_out,
                                                         kvol, COUT_FULL, (ST *)out, tile_order);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

bool chan_ok16(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

template <typename ST, typename WT, typename MT, int SPLIT>
int run16(const void *feat, const void *filters, const int32_t *nbr, int n_rows_out, int kvol, int cin, int cout,
          int transpose_w, int flip_k, void *out, const int32_t *tile_order, const int32_t *row_perm, void *workspace,
          int mode, hipStream_t st) {
  const int ci = transpose_w ? cout : cin, co = transpose_w ? cin : cout;
  const int total = kvol * ((ci + 31) / 32) * (co / 16) * 512;
  pack_weights16<WT, MT, SPLIT><<<dm_ceil_div(total, 256), 256, 0, st>>>((const WT *)filters, (MT *)workspace, kvol, ci, co,
                                                                 transpose_w, flip_k);
  DM_CHECK_LAUNCH();
#define DM_CASE16(CI, CO)   \
  if (ci == CI && co == CO) \
    return launch16<CI, CO, ST, MT, SPLIT>(feat, workspace, nbr, row_perm, tile_order, n_rows_out, kvol, out, mode, st);
  DM_CASE16(16, 16)
  DM_CASE16(16, 32)
  DM_CASE16(32, 16)
  DM_CASE16(32, 32)
  DM_CASE16(32, 64)
  DM_CASE16(64, 32)
  DM_CASE16(64, 64)
  DM_CASE16(64, 128)
  DM_CASE16(128, 64)
#undef DM_CASE16
  return DM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" size_t dm_spconv16_workspace_bytes(int kvol, int cin, int cout) {
  if (kvol <= 0 || cin <= 0 || cout <= 0) return 0;
  const size_t a = (size_t)kvol * ((cin + 31) / 32) * 32 * cout, b = (size_t)kvol * ((cout + 31) / 32) * 32 * cin;
  return dm_align((a > b ? a : b) * 2 * 3);        // three planes in the split mode
}

extern "C" int dm_spconv_gather_gemm16(const void *feat, int n_rows_in, const void *filters, int storage,
                                       const int32_t *nbr, int n_rows_out, int kvol, int cin, int cout,
                                       int transpose_w, int flip_k, void *out, const int32_t *tile_order,
                                       const int32_t *row_perm, void *workspace, size_t workspace_bytes,
                                       dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_rows_in < 0 || n_rows_out < 0 || kvol <= 0 || kvol > 32) return DM_ERR_INVALID_ARG;
  const int ci = transpose_w ? cout : cin, co = transpose_w ? cin : cout;
  if (!(chan_ok16(ci) && chan_ok16(co))) return DM_ERR_UNSUPPORTED;
  if (n_rows_out == 0) return DM_OK;
  if (!filters || !nbr || !out || !workspace || (n_rows_in > 0 && !feat)) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_spconv16_workspace_bytes(kvol, cin, cout)) return DM_ERR_WORKSPACE;
  switch (storage) {
    case DM_SP16_F32ROWS:
      return run16<float, float, __bf16, 1>(feat, filters, nbr, n_rows_out, kvol, cin, cout, transpose_w, flip_k, out,
                                         tile_order, row_perm, workspace, storage, st);
    case DM_SP16_F16:
      return run16<_Float16, _Float16, _Float16, 1>(feat, filters, nbr, n_rows_out, kvol, cin, cout, transpose_w,
                                                 flip_k, out, tile_order, row_perm, workspace, storage, st);
    case DM_SP16_BF16:
      return run16<__bf16, __bf16, __bf16, 1>(feat, filters, nbr, n_rows_out, kvol, cin, cout, transpose_w, flip_k,
                                           out, tile_order, row_perm, workspace, storage, st);
    case DM_SP16_F32SPLIT:
      return run16<float, float, __bf16, 3>(feat, filters, nbr, n_rows_out, kvol, cin, cout, transpose_w, flip_k, out,
                                            tile_order, row_perm, workspace, storage, st);
    default:
      return DM_ERR_INVALID_ARG;
  }
}
