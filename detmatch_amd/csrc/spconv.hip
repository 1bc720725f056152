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
#include <hip/hip_ext.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "dm_common.h"

// Input channels per pipeline step of spconv_gr = 16 * DM_GR_CTS.  2 (32 channels, 8 KiB of weights
// per step) keeps the kernel at 106 VGPRs = 4 waves/SIMD; 4 needs 166 (3 waves/SIMD) and measures
// 8 % slower on the 64->64 layers (tools/bench_spconv_layers.py).
#ifndef DM_GR_CTS
#define DM_GR_CTS 2
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// mask a gathered value with an all-ones / all-zeros word (no branch, no NaN leak)
__device__ __forceinline__ float mask_bits(float a, unsigned int m) {
  return __uint_as_float(__float_as_uint(a) & m);
}

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
// Workgroup = 4 waves = RT row tiles of 16 output rows x NS column splits
// (RT * NS = 4).  Wave w owns row tile w / NS and output channels
// [ (w % NS) * COUT/NS, +COUT/NS ).  Splitting the channels rather than giving a
// wave more rows keeps the per-wave MFMA chain short (the layer has only ~1-2 k
// row tiles for 1024 SIMDs, so work-unit granularity decides the tail) while all
// four waves still share one LDS copy of W[k].
// `perm` (optional): processing order -> output row; `nbr` is indexed in
// processing order.  Lets the rulebook group rows with equal neighbour masks
// into the same tile without changing the row order of `out`.
template <int CIN, int COUT, int NS>
__global__ __launch_bounds__(256) void spconv_gg(const float *__restrict__ feat,
                                                 const float *__restrict__ wpack,
                                                 const int32_t *__restrict__ nbr,
                                                 const int32_t *__restrict__ perm, int n_out,
                                                 int kvol, float *__restrict__ out,
                                                 unsigned long long *__restrict__ stamps) {
  constexpr int RT = 4 / NS;
  constexpr int ROWS = 16 * RT;
  constexpr int NB = COUT / 16;            // 16-col blocks of the whole B operand
  constexpr int NBW = NB / NS;             // ... owned by one wave
  constexpr int CT = CIN >= 16 ? CIN / 16 : 1;
  constexpr int WSZ = CIN * COUT;          // floats per kernel offset
  constexpr int NT = 256;
  constexpr int W4 = WSZ / 4;              // float4 per offset
  constexpr int W4_PER_T = (W4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *wl = (float *)smem;                           // 2 * WSZ floats
  int32_t *nb_l = (int32_t *)(wl + 2 * WSZ);           // kvol * ROWS
  unsigned int *active_mask_p = (unsigned int *)(nb_l + kvol * ROWS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int rt = wave / NS, ns = wave % NS;
  const int row0 = blockIdx.x * ROWS;

  unsigned long long st0 = 0, st1 = 0, rt0 = 0;
  if (stamps) {  // diagnostic only (dm_spconv_debug_stamps); never set in production
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  if (tid == 0) *active_mask_p = 0u;
  __syncthreads();
  // stage this tile's slice of the gather table (4 independent loads in flight per
  // thread); build the active-offset mask
  unsigned int my_mask = 0u;
  const int n_ent = kvol * ROWS;
  for (int base = 0; base < n_ent; base += NT * 4) {
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int e = base + u * NT + tid;
      int k = e / ROWS, rr = e % ROWS;
      int row = row0 + rr;
      bool ok = (e < n_ent) && (row < n_out);
      v[u] = nbr[ok ? (size_t)k * n_out + row : 0];
      if (!ok) v[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int e = base + u * NT + tid;
      if (e < n_ent) {
        nb_l[e] = v[u];
        if (v[u] >= 0) my_mask |= 1u << (e / ROWS);
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) my_mask |= __shfl_xor(my_mask, d);
  if (lane == 0 && my_mask) atomicOr(active_mask_p, my_mask);
  __syncthreads();
  unsigned int active = *active_mask_p;

  f32x4 acc[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // branch-free gather of this lane's slice of the 16 input rows feeding offset k
  // (rows without a neighbour read row 0 and are zeroed by a select)
  // Gather of this lane's slice of the 16 input rows feeding offset k.  The load is
  // unconditional (rows without a neighbour read row 0) and is masked only when it is
  // consumed, by an AND with an all-ones / all-zeros word: no exec-masked
  // branch around the loads, so the compiler can keep them in flight across the MFMA
  // block of the previous offset with exact vmcnt counts.
  auto load_a = [&](int k, f32x4 *dst, unsigned int *okf) {
    int idx = nb_l[k * ROWS + rt * 16 + r];
    *okf = idx >= 0 ? 0xFFFFFFFFu : 0u;
    size_t base = (size_t)(idx >= 0 ? idx : 0) * CIN;
    if (CIN >= 16) {
#pragma unroll
      for (int t = 0; t < CT; ++t) dst[t] = *(const f32x4 *)(feat + base + 16 * t + 4 * kq);
    } else {
      dst[0] = (f32x4){feat[base + kq], 0.f, 0.f, 0.f};
    }
  };
  static_assert(W4 % NT == 0 || W4 < NT, "weight slice must tile the workgroup");
  auto load_w = [&](int k, f32x4 *dst) {
    const f32x4 *src = (const f32x4 *)(wpack + (size_t)k * WSZ);
#pragma unroll
    for (int i = 0; i < W4_PER_T; ++i) {
      int e = tid + i * NT;
      dst[i] = src[W4 % NT == 0 ? e : (e < W4 ? e : 0)];
    }
  };
  auto store_w = [&](int b, const f32x4 *srcv) {
    f32x4 *dst = (f32x4 *)(wl + b * WSZ);
#pragma unroll
    for (int i = 0; i < W4_PER_T; ++i) {
      int e = tid + i * NT;
      if (W4 % NT == 0 || e < W4) dst[e] = srcv[i];
    }
  };

  auto compute = [&](int k, const f32x4 *a, unsigned int okf, int buf) {
    // skip if none of this wave's 16 rows has a neighbour through offset k
    int idx = nb_l[k * ROWS + rt * 16 + r];
    if (__ballot(idx >= 0) == 0ull) return;
    const float *wb = wl + buf * WSZ;
    if (CIN >= 16) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        f32x4 b[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          b[nb] = *(const f32x4 *)(wb + (((t * NB + ns * NBW + nb) * 4 + kq) * 16 + r) * 4);
        float av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = mask_bits(a[t][j], okf);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[nb][j], acc[nb], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        float b = wb[((ns * NBW + nb) * 4 + kq) * 16 + r];
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(mask_bits(a[0][0], okf), b, acc[nb], 0, 0, 0);
      }
    }
  };

  if (stamps) st1 = __builtin_amdgcn_s_memtime();
  if (active != 0u && n_out > 0) {
    // Software pipeline, prefetch distance 2: in iteration i the loads for offset
    // k_{i+2} are issued, offset k_i is computed from LDS[buf] and registers loaded two
    // iterations ago, and W[k_{i+1}] (loaded one iteration ago) moves from registers to
    // LDS[buf^1].  A global-load latency is thus covered by a full iteration, and the
    // per-offset chain is barrier + MFMA block + LDS write.
    const int n_act = __popc(active);
    unsigned int rest = active;
    auto pop = [&]() {  // next active offset; repeats the last one when exhausted
      int k = __ffs(rest) - 1;
      if (rest & (rest - 1u)) rest &= rest - 1u;
      return k;
    };
    int kc = pop();
    int k1 = pop();
    f32x4 wr0[W4_PER_T], wr1[W4_PER_T];
    f32x4 a0[CT], a1[CT], a2[CT];
    unsigned int ok0, ok1, ok2;
    load_w(kc, wr0);
    load_a(kc, a0, &ok0);
    load_w(k1, wr1);
    load_a(k1, a1, &ok1);
    store_w(0, wr0);
    int buf = 0;
    int i = 0;
    auto iter = [&](f32x4 *w_new, f32x4 *w_old) {
      __syncthreads();  // LDS[buf] holds W[k_i]; nobody still reads LDS[buf^1]
      int k2 = pop();
      load_w(k2, w_new);
      load_a(k2, a2, &ok2);
      compute(kc, a0, ok0, buf);
      if (i + 1 < n_act) store_w(buf ^ 1, w_old);
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        a0[t] = a1[t];
        a1[t] = a2[t];
      }
      ok0 = ok1;
      ok1 = ok2;
      kc = k1;
      k1 = k2;
      buf ^= 1;
      ++i;
    };
    while (true) {
      iter(wr0, wr1);
      if (i >= n_act) break;
      iter(wr1, wr0);
      if (i >= n_act) break;
    }
  }
  if (stamps && tid == 0) {
    unsigned long long st2 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = stamps + (size_t)blockIdx.x * 6;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = rt0; o[4] = rt1; o[5] = __popc(active);
  }
  // epilogue: D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int prow = row0 + rt * 16 + kq * 4 + reg;
    if (prow < n_out) {
      int row = perm ? perm[prow] : prow;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        out[(size_t)row * COUT + 16 * (ns * NBW + nb) + r] = acc[nb][reg];
    }
  }
}

unsigned long long *g_debug_stamps = nullptr;
int g_gg_variant = -1;  // -1 auto, 0 LDS-staged weights (spconv_gg), 1 register weights, 16-row tiles (spconv_gr);
                        // auto = gr for cin >= 32, else gg

// ---- main kernel, register-resident weights ----------------------------------
// (Measured and dropped, round 2: EIGHT waves per tile for the layers with fewer than ~1 k tiles — each
// wave then walks at most 4 offsets: 19.6 us against 19.4 us on the 6.7 k-row 64 -> 64 layers, 31 against
// 29 us on the 16.8 k-row ones.  The short layers are not bound by the length of a wave's offset chain.)
// Workgroup = 4 waves = ONE tile of 16 output rows; the tile's active kernel
// offsets are dealt round-robin to the four waves.  A wave keeps the whole
// (cin x cout) B operand of its current offset in registers (loaded straight from
// the packed weights, 1 KiB per wave-instruction) next to the gathered A rows, so
// the main loop has no LDS traffic and no barrier; the four partial accumulators
// meet once, in LDS, are summed in a fixed order (bitwise reproducible) and leave
// as whole 16-byte-per-lane row segments.  Work units are a quarter of a tile's
// offsets, which is what lets ~1-2 k tiles balance over 1024 SIMDs.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_gr(const float *__restrict__ feat,
                                                 const float *__restrict__ wpack,
                                                 const int32_t *__restrict__ nbr,
                                                 const int32_t *__restrict__ perm, int n_out,
                                                 int kvol, int cout_full,
                                                 float *__restrict__ out,
                                                 unsigned long long *__restrict__ stamps,
                                                 const int32_t *__restrict__ tile_order) {
  constexpr int NB = COUT / 16;
  constexpr int CT = CIN / 16;
  constexpr int CTS = CT < DM_GR_CTS ? CT : DM_GR_CTS;   // k-chunks of <= 16*DM_GR_CTS input channels per step
  constexpr int S = CT / CTS;              // steps per kernel offset
  constexpr int LDP = COUT + 4;            // padded row stride of the partial tiles
  __shared__ int32_t tbl[4][32][16];
  __shared__ __attribute__((aligned(16))) float part[4][16][LDP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int row0 = (tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x) * 16;
  const int nb_full = cout_full / 16;
  const int nb0 = blockIdx.y * NB;
  unsigned long long st0 = 0, rt0 = 0, st1 = 0;
  if (stamps) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }

  // this wave's private copy of the tile's gather table + the active-offset mask
  unsigned int active = 0u;
  {
    int v[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {  // all eight loads in flight before the first use
      int k = 4 * k4 + kq;
      bool in = (k < kvol) && (row0 + r < n_out);
      v[k4] = nbr[in ? (size_t)k * n_out + row0 + r : 0];
      if (!in) v[k4] = -1;
    }
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      tbl[wave][4 * k4 + kq][r] = v[k4];
      unsigned long long m = __ballot(v[k4] >= 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((m >> (16 * q)) & 0xFFFFull) active |= 1u << (4 * k4 + q);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // tbl[wave] is read across lanes
  __builtin_amdgcn_wave_barrier();
  // deal the active offsets round-robin to the four waves
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

  auto load_step = [&](int k, int s, f32x4 *w, f32x4 *a, unsigned int *ok) {
    int idx = tbl[wave][k][r];
    *ok = idx >= 0 ? 0xFFFFFFFFu : 0u;
    const float *src = feat + (size_t)(idx >= 0 ? idx : 0) * CIN + 16 * (s * CTS) + 4 * kq;
#pragma unroll
    for (int t = 0; t < CTS; ++t) a[t] = *(const f32x4 *)(src + 16 * t);
    const f32x4 *wk = (const f32x4 *)wpack + (size_t)k * (CIN * (size_t)cout_full / 4);
#pragma unroll
    for (int t = 0; t < CTS; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        w[t * NB + nb] = wk[((s * CTS + t) * nb_full + nb0 + nb) * 64 + lane];
  };
  auto compute = [&](const f32x4 *w, const f32x4 *a, unsigned int ok) {
#pragma unroll
    for (int t = 0; t < CTS; ++t) {
      float av[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = mask_bits(a[t][j], ok);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[t * NB + nb][j], acc[nb], 0, 0, 0);
    }
  };

  if (stamps) st1 = __builtin_amdgcn_s_memtime();
  if (mine != 0u) {
    const int n_steps = __popc(mine) * S;
    unsigned int rest = mine;
    int k = __ffs(rest) - 1, s = 0;
    auto advance = [&]() {  // next (offset, k-chunk); sticks at the last one
      if (s + 1 < S) {
        ++s;
      } else if (rest & (rest - 1u)) {
        rest &= rest - 1u;
        k = __ffs(rest) - 1;
        s = 0;
      }
    };
    f32x4 w0[CTS * NB], w1[CTS * NB], a0[CTS], a1[CTS];
    unsigned int ok0, ok1;
    load_step(k, s, w0, a0, &ok0);
    int i = 0;
    // pairs of steps as ONE straight-line loop body, the odd step behind the loop: with an exit between the two
    // halves the accumulators of the two paths get different registers and are copied (through VGPRs) every turn.
    // The last pair's second load re-reads the last step (advance() sticks there).
    for (; i + 2 <= n_steps; i += 2) {
      advance();
      load_step(k, s, w1, a1, &ok1);
      compute(w0, a0, ok0);
      advance();
      load_step(k, s, w0, a0, &ok0);
      compute(w1, a1, ok1);
    }
    if (i < n_steps) compute(w0, a0, ok0);
  }
  // meet in LDS: D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) part[wave][4 * kq + reg][16 * nb + r] = acc[nb][reg];
  __syncthreads();
  if (stamps && tid == 0) {
    unsigned long long *o = stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6;
    o[0] = st0; o[1] = st1; o[2] = __builtin_amdgcn_s_memtime(); o[3] = rt0;
    o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = __popc(active);
  }
  constexpr int F4_PER_ROW = COUT / 4;
  if (tid < 16 * F4_PER_ROW) {
    int rr = tid / F4_PER_ROW, c4 = tid % F4_PER_ROW;
    int prow = row0 + rr;
    if (prow < n_out) {
      f32x4 v0 = *(const f32x4 *)&part[0][rr][4 * c4];
      f32x4 v1 = *(const f32x4 *)&part[1][rr][4 * c4];
      f32x4 v2 = *(const f32x4 *)&part[2][rr][4 * c4];
      f32x4 v3 = *(const f32x4 *)&part[3][rr][4 * c4];
      f32x4 sum = (v0 + v1) + (v2 + v3);
      int row = perm ? perm[prow] : prow;
      *(f32x4 *)(out + (size_t)row * cout_full + blockIdx.y * COUT + 4 * c4) = sum;
    }
  }
}

// (Measured and dropped, round 3 / pruned round 5: 32-row tiles on v_mfma_f32_32x32x2_f32, eight waves per tile
// sharing the offsets — half the weight traffic per flop, 30.7 against 29.0 us on the 64 -> 64 layer and 22.5 against
// 19.5 us on the small ones: the weight traffic is not what limits spconv_gr.)

template <int CIN, int COUT_FULL>
int launch_gr(const float *feat, const float *wpack, const int32_t *nbr, const int32_t *perm,
              const int32_t *tile_order, int n_out, int kvol, float *out, hipStream_t st) {
  constexpr int COUT = COUT_FULL > 64 ? 64 : COUT_FULL;  // columns per workgroup
  dim3 grid(dm_ceil_div(n_out, 16), COUT_FULL / COUT);
  hipEvent_t e0, e1;
  if (dm_prof_open(DM_PROF_SPCONV_GG, CIN, COUT_FULL, 0, n_out, kvol, nbr, &e0, &e1) >= 0)
    hipExtLaunchKernelGGL((spconv_gr<CIN, COUT>), grid, dim3(256), 0, st, e0, e1, 0, feat, wpack, nbr, perm,
                          n_out, kvol, (int)COUT_FULL, out, g_debug_stamps, tile_order);
  else
    spconv_gr<CIN, COUT><<<grid, 256, 0, st>>>(feat, wpack, nbr, perm, n_out, kvol, COUT_FULL, out,
                                               g_debug_stamps, tile_order);
  DM_CHECK_LAUNCH();
  return DM_OK;
}


template <int CIN, int COUT>
int launch_gg(const float *feat, const float *wpack, const int32_t *nbr, const int32_t *perm,
              int n_out, int kvol, float *out, hipStream_t st) {
  // column splits per row tile: keep >= 16 channels per wave
  constexpr int NS = COUT >= 128 ? 4 : (COUT >= 32 ? 2 : 1);
  constexpr int ROWS = 16 * (4 / NS);
  size_t smem = 2ull * CIN * COUT * sizeof(float) + (size_t)kvol * ROWS * sizeof(int32_t) + 16;
  static bool attr_set = false;  // > 64 KB of dynamic LDS needs the opt-in
  if (!attr_set) {
    DM_HIP(hipFuncSetAttribute((const void *)spconv_gg<CIN, COUT, NS>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set = true;
  }
  hipEvent_t e0, e1;
  if (dm_prof_open(DM_PROF_SPCONV_GG, CIN, COUT, NS, n_out, kvol, nbr, &e0, &e1) >= 0)
    hipExtLaunchKernelGGL((spconv_gg<CIN, COUT, NS>), dim3(dm_ceil_div(n_out, ROWS)), dim3(256), (uint32_t)smem,
                          st, e0, e1, 0, feat, wpack, nbr, perm, n_out, kvol, out, g_debug_stamps);
  else
    spconv_gg<CIN, COUT, NS><<<dm_ceil_div(n_out, ROWS), 256, smem, st>>>(
        feat, wpack, nbr, perm, n_out, kvol, out, g_debug_stamps);
  DM_CHECK_LAUNCH();
  return DM_OK;
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

// Round 2: one wave per (kernel offset, chunk of pairs) covers ALL channels of both operands.
// A lane (j = lane & 15, pair slot kq = lane >> 4) fetches CIN/16 CONSECUTIVE input channels of its
// pair's X row and COUT/16 consecutive channels of its dY row with one vector load each (a wave
// instruction = 4 whole rows, 16 lanes x 16 B each for 64 channels) — every gathered row leaves
// memory exactly once per pair, where the round-1 kernel re-read dY once per 16-channel block of
// cin and X in 64-byte pieces (84 MB per launch).  Component q of the X vector
// and component r of the dY vector feed MFMA (q, r): its 16x16 output tile holds channel pairs
// (cin = VA*i + q, cout = VB*j + r), i.e. the channel order inside an MFMA is a permutation that
// only the final store has to know.  Loads are unconditional (index clamped, X masked): a
// conditional load makes hipcc branch and wait per element.
template <int CIN, int COUT, bool CIN4 = false>      // CIN4: the rows of `feat` hold 4 channels (the input layer), padded with zeros to 16 here
__device__ __forceinline__ void spconv_wgrad_rows_body(const float *__restrict__ feat,
                                                       const float *__restrict__ ograd,
                                                       const int32_t *__restrict__ pairs,
                                                       const int32_t *__restrict__ indice_num,
                                                       int pair_stride, int chunk,
                                                       float *__restrict__ slab, int kvol, int k, int ch,
                                                       float *tile) {
  constexpr int VA = CIN / 16, VB = COUT / 16;
  typedef float vecA __attribute__((ext_vector_type(VA)));
  typedef float vecB __attribute__((ext_vector_type(VB)));
  // a workgroup = 4 waves = one chunk of pairs, a quarter each; their partial tiles are summed
  // through LDS (`tile`, CIN * COUT floats) in wave order (deterministic) and written once: 4x less slab
  // traffic per pair.  Chunks beyond this offset's pair count exit at once and are never read by the reduce.
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
  const int npairs = indice_num[k];
  if (ch * chunk >= npairs) return;
  const int quarter = chunk / 4;
  const int s_begin = ch * chunk + wave * quarter;
  const int s_end = min(s_begin + quarter, npairs);
  const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pair_stride;
  const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pair_stride;
  f32x4 acc[VA][VB];
#pragma unroll
  for (int q = 0; q < VA; ++q)
#pragma unroll
    for (int r = 0; r < VB; ++r) acc[q][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;                      // groups of 4 pairs in flight
  // Three-stage software pipeline over steps of 16 pairs: the pair INDICES of step n+2 and the ROWS of
  // step n+1 are requested before the MFMAs of step n are issued.  (Round 2 fetched indices and rows of
  // step n+1 together: the wave then sat out the index latency inside every fetch, with nothing to
  // issue — the rows depend on the indices.)  Loads past the wave's range are clamped, not branched.
  vecA a[2][U];
  vecB b[2][U];
  bool ok[2][U];
  int ii[2][U], oo[2][U];
  bool inr[2][U];
  auto fetch_idx = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + 4 * u + kq;
      const bool in = s < s_end;
      const int sc = in ? s : s_begin;
      ii[buf][u] = pin[sc];
      oo[buf][u] = pout[sc];
      inr[buf][u] = in;
    }
  };
  auto fetch_rows = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i0 = ii[buf][u], o0 = oo[buf][u];
      ok[buf][u] = inr[buf][u] & (i0 >= 0) & (o0 >= 0);
      if constexpr (CIN4) {
        static_assert(CIN == 16, "the 4-channel layer rides on the 16-row tile");
        a[buf][u] = (vecA)(feat[(size_t)(i0 >= 0 ? i0 : 0) * 4 + (j & 3)]);
        ok[buf][u] = ok[buf][u] & (j < 4);
      } else {
        a[buf][u] = *(const vecA *)(feat + (size_t)(i0 >= 0 ? i0 : 0) * CIN + VA * j);
      }
      b[buf][u] = *(const vecB *)(ograd + (size_t)(o0 >= 0 ? o0 : 0) * COUT + VB * j);
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < VA; ++q) {
        const float av = ok[buf][u] ? a[buf][u][q] : 0.0f;
#pragma unroll
        for (int r = 0; r < VB; ++r)
          acc[q][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[buf][u][r], acc[q][r], 0, 0, 0);
      }
  };
  if (s_begin < s_end) {
    const int nsteps = (s_end - s_begin + 4 * U - 1) / (4 * U);
    fetch_idx(0, s_begin);
    fetch_idx(1, s_begin + 4 * U);
    fetch_rows(0);
    for (int n = 0; n < nsteps; n += 2) {          // two steps per trip: the buffers alternate
      fetch_idx(0, s_begin + (n + 2) * 4 * U);     // step n's indices are consumed (rows requested)
      fetch_rows(1);                               // rows of step n+1
      mma(0);                                      // step n
      if (n + 1 >= nsteps) break;
      fetch_idx(1, s_begin + (n + 3) * 4 * U);
      fetch_rows(0);                               // rows of step n+2
      mma(1);                                      // step n+1
    }
  }
  // D of MFMA (q, r): row i = 4*kq + reg -> cin channel VA*i + q; col j -> cout channels VB*j + r
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int q = 0; q < VA; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int c = VA * (4 * kq + reg) + q;
          vecB *p = (vecB *)(tile + c * COUT + VB * j);
          vecB v = w == 0 ? vecB(0.0f) : *p;
#pragma unroll
          for (int r = 0; r < VB; ++r) v[r] += acc[q][r][reg];
          *p = v;
        }
    }
    __syncthreads();
  }
  float *dst = slab + ((size_t)ch * kvol + k) * (size_t)CIN * COUT;
  for (int e = threadIdx.x * 4; e < CIN * COUT; e += 1024)
    *(float4 *)(dst + e) = *(const float4 *)(tile + e);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_rows(const float *__restrict__ feat,
                                                        const float *__restrict__ ograd,
                                                        const int32_t *__restrict__ pairs,
                                                        const int32_t *__restrict__ indice_num,
                                                        int pair_stride, int chunk,
                                                        float *__restrict__ slab, int nchunks, int kvol,
                                                        int order) {
  __shared__ float tile[CIN * COUT];
  // (chunk, offset) of this workgroup.  order 0: chunks fastest; 1: offsets fastest (workgroups that run
  // at the same time read the same stretch of rows); 2: offsets fastest AND all offsets of a chunk on one XCD
  // (workgroup ids go round-robin over the 8 XCDs, each with its own L2)
  int k, ch;
  {
    const int id = blockIdx.x;
    if (order == 0) {
      ch = id % nchunks, k = id / nchunks;
    } else if (order == 1) {
      k = id % kvol, ch = id / kvol;
    } else {
      const int xcd = id & 7, slot = id >> 3;
      k = slot % kvol;
      ch = xcd + 8 * (slot / kvol);
      if (ch >= nchunks) return;
    }
  }
  spconv_wgrad_rows_body<CIN, COUT>(feat, ograd, pairs, indice_num, pair_stride, chunk, slab, kvol, k, ch, tile);
}

// ---- the weight gradients of a whole backward pass in ONE launch pair --------------------------------------
// A layer's weight gradient needs its saved input rows and the gradient of its output rows — both exist once
// the input-gradient chain has passed the layer, and nothing downstream reads dW.  So the caller queues the
// layers of a backward pass (spconv/ops.py: deferred weight gradients) and hands all of them over at its end:
// one grid over every (layer, kernel offset, chunk of pairs) unit, one reduce over every weight element.  The
// small layers (<= 90 k pairs: two launches of ~10 us each whatever they move) ride along underneath the big
// ones, and 22 launches per pass become 2.
#define DM_WGRAD_MAX_JOBS 16
struct WgradBatch {
  const float *feat[DM_WGRAD_MAX_JOBS], *ograd[DM_WGRAD_MAX_JOBS];
  const int32_t *pairs[DM_WGRAD_MAX_JOBS], *indice_num[DM_WGRAD_MAX_JOBS];
  float *filt_grad[DM_WGRAD_MAX_JOBS];
  unsigned long long slab_off[DM_WGRAD_MAX_JOBS];      // floats
  int pair_stride[DM_WGRAD_MAX_JOBS], kvol[DM_WGRAD_MAX_JOBS], cin[DM_WGRAD_MAX_JOBS], cout[DM_WGRAD_MAX_JOBS];
  int cin_rows[DM_WGRAD_MAX_JOBS];                     // channels the layer really has (4 for the input layer, else cin)
  int chunk[DM_WGRAD_MAX_JOBS], nchunks[DM_WGRAD_MAX_JOBS];
  int unit_base[DM_WGRAD_MAX_JOBS + 1];                // workgroups of the rows kernel
  int red_base[DM_WGRAD_MAX_JOBS + 1];                 // 256-thread blocks of the reduce kernel
  int n;
};

// One instantiation per REGISTER CLASS of the layer bodies (64 ... 404 VGPRs from 16 -> 16 to 64 -> 128): a kernel
// that switched over all of them would run every layer at the occupancy of the widest (measured: 1.4x slower
// than the per-layer launches).  Class 0: Cin, Cout <= 32 (and the 4-channel input layer on the 16-row tile),
// 2: 32 -> 64 and 64 -> 64, 3: 64 -> 128 (1: unused).
__host__ __device__ inline int wgrad_class(int cin, int cout) {
  const int key = cin * 1000 + cout;
  if (key == 16016 || key == 16032 || key == 32032) return 0;
  if (key == 32064) return 2;      // (136 VGPRs: rides in the 64 -> 64 kernel's 248 for free)
  if (key == 64064) return 2;
  if (key == 64128) return 3;
  return -1;
}

template <int CLS>
__global__ __launch_bounds__(256) void spconv_wgrad_batch_rows(const WgradBatch t, float *__restrict__ slab) {
  constexpr int TILE = CLS == 0 ? 32 * 32 : (CLS == 1 ? 32 * 64 : (CLS == 2 ? 64 * 64 : 64 * 128));
  __shared__ float tile[TILE];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.unit_base[j + 1]) ++j;
  const int id = blockIdx.x - t.unit_base[j];
  const int kvol = t.kvol[j];
  const int k = id % kvol, ch = id / kvol;            // offsets fastest (see spconv_wgrad_rows)
  float *sl = slab + t.slab_off[j];
#define DM_WB(CI, CO, C4)                                                                                    \
  spconv_wgrad_rows_body<CI, CO, C4>(t.feat[j], t.ograd[j], t.pairs[j], t.indice_num[j], t.pair_stride[j],  \
                                     t.chunk[j], sl, kvol, k, ch, tile)
  if constexpr (CLS == 0) {
    const int key = t.cin[j] * 1000 + t.cout[j];
    if (key == 16016) {
      if (t.cin_rows[j] == 4) DM_WB(16, 16, true);
      else DM_WB(16, 16, false);
    } else if (key == 16032) DM_WB(16, 32, false);
    else DM_WB(32, 32, false);
  } else if constexpr (CLS == 1) {
    DM_WB(32, 64, false);
  } else if constexpr (CLS == 2) {
    if (t.cin[j] == 32) DM_WB(32, 64, false);
    else DM_WB(64, 64, false);
  } else {
    DM_WB(64, 128, false);
  }
#undef DM_WB
}

__global__ __launch_bounds__(256) void spconv_wgrad_batch_reduce(const WgradBatch t, const float *__restrict__ slab,
                                                                 int accumulate) {
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.red_base[j + 1]) ++j;
  const int cin = t.cin[j], cout = t.cout[j], rows = t.cin_rows[j];
  const size_t per_offset = (size_t)cin * cout, per_chunk = per_offset * t.kvol[j];
  const size_t e = (size_t)(blockIdx.x - t.red_base[j]) * 256 + threadIdx.x;
  if (e >= per_chunk) return;
  const int k = (int)(e / per_offset);
  const int c = (int)(e % per_offset) / cout, nn = (int)(e % cout);
  if (c >= rows) return;                               // padding rows of the 4-channel layer's tile
  const int n = (t.indice_num[j][k] + t.chunk[j] - 1) / t.chunk[j];
  const float *sl = slab + t.slab_off[j];
  float s = 0.f;
  for (int cc = 0; cc < n; ++cc) s += sl[(size_t)cc * per_chunk + e];
  float *o = t.filt_grad[j] + ((size_t)k * rows + c) * cout + nn;
  *o = accumulate ? *o + s : s;
}

__global__ __launch_bounds__(256) void spconv_wgrad_reduce(const float *slab, int nchunks,
                                                           size_t per_chunk, float *filt_grad) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += slab[(size_t)c * per_chunk + e];
  filt_grad[e] = s;
}

// the same over the chunks that exist for each offset (spconv_wgrad_rows writes no others)
__global__ __launch_bounds__(256) void spconv_wgrad_reduce_active(const float *slab, int chunk,
                                                                  const int32_t *__restrict__ indice_num,
                                                                  size_t per_offset, size_t per_chunk,
                                                                  float *filt_grad) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= per_chunk) return;
  const int k = (int)(e / per_offset);
  const int n = (indice_num[k] + chunk - 1) / chunk;
  float s = 0.f;
  for (int c = 0; c < n; ++c) s += slab[(size_t)c * per_chunk + e];
  filt_grad[e] = s;
}

int g_wgrad_chunk = 0;   // 0 = heuristic; dm_spconv_set_wgrad_chunk (developer switch)
int g_wgrad_order = 1;   // workgroup -> (chunk, offset) map of spconv_wgrad_rows, see the kernel; offsets fastest:
                         // 58 -> 41 us on the 64 -> 64 layer with 16.8 k rows (order 2, XCD-aware: no better)

int wgrad_chunks(int n_in, int *chunk, bool rows_kernel = true) {
  // a chunk = one 4-wave workgroup (a quarter each); multiples of 64 pairs.  Measured (round 3,
  // tools/bench_spconv_layers.py --wgrad-chunk): 256 wins on the layers below ~12 k rows (more workgroups
  // than CUs), 512 above (the per-workgroup merge + slab write is a fixed cost)
  int c = g_wgrad_chunk > 0 ? g_wgrad_chunk : ((n_in < 12000 && rows_kernel) ? 256 : 512);
  while ((long long)c * 128 < n_in) c *= 2;
  *chunk = c;
  return dm_ceil_div(n_in > 0 ? n_in : 1, c);
}

bool chan_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

}  // namespace

// diagnostic: when set, every workgroup of spconv_gg writes 6 u64 stamps
// {t_start, t_after_prologue, t_end (s_memtime), rt_start, rt_end (s_memrealtime), n_active}
extern "C" int dm_spconv_debug_stamps(void *buf) {
  g_debug_stamps = (unsigned long long *)buf;
  return DM_OK;
}

// ---- launch order of the 16-row tiles ------------------------------------------------------
// A gather-GEMM launch fits the chip in about one round of workgroups, so it lasts as long as the
// compute unit that happened to receive the heaviest tiles (a tile walks 1..kvol active offsets;
// rocprofv3 SQ counters: 63 % of the wave cycles are issue stalls behind the fp32 MFMA pipe, whose
// busy time is 14 us per SIMD of a 34 us launch).  Handing the tiles out by descending work lets
// the dispatcher spread the heavy ones first: -13..17 % per launch on the 32/64-channel layers.
// The order is a property of the gather table: built once per rulebook (counting sort by active
// offsets, two tiny launches), reused by every forward / input-gradient launch on that table.
template <bool SCATTER, int ROWS>
__global__ __launch_bounds__(256) void tile_order_kernel(const int32_t *__restrict__ nbr, int n_rows,
                                                         int kvol, int n_tiles, int *__restrict__ hist,
                                                         int32_t *__restrict__ order) {
  constexpr int KPI = 64 / ROWS;                   // kernel offsets per wave iteration
  const int lane = threadIdx.x & 63, r = lane % ROWS, kq = lane / ROWS;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int row = tile * ROWS + r;
  unsigned int active = 0u;
#pragma unroll
  for (int k4 = 0; k4 < 32 / KPI; ++k4) {
    const int k = KPI * k4 + kq;
    const bool in = (k < kvol) && (row < n_rows);
    int v = nbr[in ? (size_t)k * n_rows + row : 0];
    if (!in) v = -1;
    const unsigned long long m = __ballot(v >= 0);
#pragma unroll
    for (int q = 0; q < KPI; ++q)
      if ((m >> (ROWS * q)) & ((1ull << ROWS) - 1ull)) active |= 1u << (KPI * k4 + q);
  }
  if (lane != 0) return;
  const int w = __popc(active);                    // 0..32
  if (!SCATTER) {
    atomicAdd(&hist[w], 1);
  } else {
    int base = 0;                                  // tiles with more work come first
    for (int c = w + 1; c <= 32; ++c) base += hist[c];
    order[base + atomicAdd(&hist[33 + w], 1)] = tile;
  }
}

extern "C" size_t dm_spconv_tile_order_workspace_bytes(void) { return dm_align(66 * sizeof(int)); }

extern "C" int dm_spconv_tile_order(const int32_t *nbr, int n_rows, int kvol, int32_t *order,
                                    void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_rows < 0 || kvol <= 0 || kvol > 32) return DM_ERR_INVALID_ARG;
  if (n_rows == 0) return DM_OK;
  if (!nbr || !order || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_spconv_tile_order_workspace_bytes()) return DM_ERR_WORKSPACE;
  int *hist = (int *)workspace;
  DM_HIP(hipMemsetAsync(hist, 0, 66 * sizeof(int), st));
  const int n_tiles = dm_ceil_div(n_rows, 16);
  tile_order_kernel<false, 16><<<dm_ceil_div(n_tiles, 4), 256, 0, st>>>(nbr, n_rows, kvol, n_tiles, hist, order);
  DM_CHECK_LAUNCH();
  tile_order_kernel<true, 16><<<dm_ceil_div(n_tiles, 4), 256, 0, st>>>(nbr, n_rows, kvol, n_tiles, hist, order);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// ---- packing rows with equal neighbour masks into the same tiles --------------------------------
// A 16-row tile pays a full 16 x cin x cout MFMA block for every kernel offset that ANY of its rows
// uses.  In rulebook (raster) order only 40-72 % of those rows exist; after a stable sort of the rows
// by their neighbour mask 72-89 % do (tools/spconv_pack_probe.py), i.e. 20-45 % fewer (tile, offset)
// units, their weights and MFMA work.  Like the tile order this is a property of the gather table:
// built once per rulebook (mask keys, one radix sort, one gather), reused by every launch on it; the
// kernels write row p of the packed order to out[perm[p]].
__global__ __launch_bounds__(256) void row_mask_kernel(const int32_t *__restrict__ nbr, int n_rows, int kvol,
                                                       uint32_t *__restrict__ keys, int32_t *__restrict__ idx) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  uint32_t m = 0u;
  for (int k = 0; k < kvol; ++k) m |= (nbr[(size_t)k * n_rows + r] >= 0 ? 1u : 0u) << k;
  keys[r] = m;
  idx[r] = r;
}

__global__ __launch_bounds__(256) void row_gather_kernel(const int32_t *__restrict__ nbr, int n_rows, int kvol,
                                                         const int32_t *__restrict__ perm,
                                                         int32_t *__restrict__ packed) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_rows) return;
  const int r = perm[p];
  for (int k = 0; k < kvol; ++k) packed[(size_t)k * n_rows + p] = nbr[(size_t)k * n_rows + r];
}

extern "C" size_t dm_spconv_pack_rows_workspace_bytes(int n_rows) {
  if (n_rows <= 0) return 0;
  size_t need = 0;
  uint32_t *k = nullptr;
  int32_t *v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, need, k, k, v, v, (size_t)n_rows, 0, 32, (hipStream_t)0);
  return 3 * dm_align((size_t)n_rows * 4) + dm_align(need) + dm_spconv_tile_order_workspace_bytes();
}

extern "C" int dm_spconv_pack_rows(const int32_t *nbr, int n_rows, int kvol, int32_t *perm,
                                   int32_t *nbr_packed, int32_t *tile_order_packed, void *workspace,
                                   size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_rows < 0 || kvol <= 0 || kvol > 32) return DM_ERR_INVALID_ARG;
  if (n_rows == 0) return DM_OK;
  if (!nbr || !perm || !nbr_packed || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_spconv_pack_rows_workspace_bytes(n_rows)) return DM_ERR_WORKSPACE;
  DmArena arena(workspace, workspace_bytes);
  uint32_t *keys = arena.take<uint32_t>(n_rows), *keys_s = arena.take<uint32_t>(n_rows);
  int32_t *idx = arena.take<int32_t>(n_rows);
  size_t need = 0;
  DM_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, keys_s, idx, perm, (size_t)n_rows, 0, kvol, st));
  void *tmp = arena.take<char>(need);
  if (!arena.ok()) return DM_ERR_WORKSPACE;
  row_mask_kernel<<<dm_ceil_div(n_rows, 256), 256, 0, st>>>(nbr, n_rows, kvol, keys, idx);
  DM_CHECK_LAUNCH();
  DM_HIP(rocprim::radix_sort_pairs(tmp, need, keys, keys_s, idx, perm, (size_t)n_rows, 0, kvol, st));
  row_gather_kernel<<<dm_ceil_div(n_rows, 256), 256, 0, st>>>(nbr, n_rows, kvol, perm, nbr_packed);
  DM_CHECK_LAUNCH();
  if (tile_order_packed) {   // the heavy-first launch order of the packed table, in the same call
    int *hist = arena.take<int>(66);
    if (!arena.ok()) return DM_ERR_WORKSPACE;
    return dm_spconv_tile_order(nbr_packed, n_rows, kvol, tile_order_packed, hist,
                                dm_spconv_tile_order_workspace_bytes(), stream);
  }
  return DM_OK;
}

// tuning aid: -1 auto, 0 force the LDS-staged kernel, 1 force the register-weights kernel
extern "C" int dm_spconv_set_wgrad_chunk(int pairs) {
  if (pairs < 0) {                 // -1 - order: developer switch for the workgroup map
    const int order = -1 - pairs;
    if (order > 2) return DM_ERR_INVALID_ARG;      // validated BEFORE it is stored
    g_wgrad_order = order;
    return DM_OK;
  }
  if (pairs != 0 && (pairs < 64 || pairs % 64)) return DM_ERR_INVALID_ARG;
  g_wgrad_chunk = pairs;
  return DM_OK;
}

extern "C" int dm_spconv_set_variant(int v) {
  if (v < -1 || v > 1) return DM_ERR_INVALID_ARG;
  g_gg_variant = v;
  return DM_OK;
}

extern "C" size_t dm_spconv_workspace_bytes(int kvol, int cin, int cout) {
  if (kvol <= 0 || cin <= 0 || cout <= 0) return 0;
  return dm_align((size_t)kvol * cin * cout * sizeof(float));
}

#define DM_GG_CASE(CI, CO)                                                              \
  if (ci == CI && co == CO) {                                                           \
    bool use_gr = g_gg_variant < 0 ? (CI >= 32) : (g_gg_variant >= 1);                  \
    if constexpr (CI >= 16) {                                                           \
      if (use_gr) return launch_gr<CI, CO>(feat, wp, nbr, row_perm, tile_order, n_rows_out, kvol, out, st); \
    }                                                                                   \
    return launch_gg<CI, CO>(feat, wp, nbr, row_perm, n_rows_out, kvol, out, st);        \
  }

extern "C" int dm_spconv_gather_gemm(const float *feat, int n_rows_in, const float *filters,
                                     const int32_t *nbr, int n_rows_out, int kvol, int cin,
                                     int cout, int transpose_w, int flip_k, float *out,
                                     const int32_t *tile_order, const int32_t *row_perm,
                                     void *workspace, size_t workspace_bytes, dm_stream_t stream) {
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
  pack_weights<<<dm_ceil_div(total, 256), 256, 0, st>>>(filters, wp, kvol, ci, co, transpose_w, flip_k);
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
  int nchunks = wgrad_chunks(pair_stride, &chunk, cin >= 16);
  if (workspace_bytes < dm_spconv_wgrad_workspace_bytes(pair_stride, kvol, cin, cout))
    return DM_ERR_WORKSPACE;
  float *slab = (float *)workspace;
  dim3 grid(nchunks, dm_ceil_div(cin, 16), kvol);
  int pi = dm_prof_begin(st, DM_PROF_SPCONV_WGRAD, cin, cout, 1, pair_stride, kvol, indice_pairs);
  bool rows_kernel = true;
  const int n_wg = g_wgrad_order >= 2 ? 8 * dm_ceil_div(nchunks, 8) * kvol : nchunks * kvol;
#define DM_WGRAD_ROWS(CI, CO)                                                                   \
  spconv_wgrad_rows<CI, CO><<<n_wg, 256, 0, st>>>(feat, out_grad, indice_pairs, indice_num,    \
                                                  pair_stride, chunk, slab, nchunks, kvol, g_wgrad_order)
  const int key = cin * 1000 + cout;
  if (key == 16016) DM_WGRAD_ROWS(16, 16);
  else if (key == 16032) DM_WGRAD_ROWS(16, 32);
  else if (key == 32032) DM_WGRAD_ROWS(32, 32);
  else if (key == 32064) DM_WGRAD_ROWS(32, 64);
  else if (key == 64064) DM_WGRAD_ROWS(64, 64);
  else if (key == 64128) DM_WGRAD_ROWS(64, 128);
  else { rows_kernel = false; switch (cout) {      // other channel pairs (the 4-channel input layer): one wave per cin block
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
  } }
#undef DM_WGRAD_ROWS
  DM_CHECK_LAUNCH();
  if (rows_kernel)
    spconv_wgrad_reduce_active<<<dm_ceil_div((long long)per_chunk, 256), 256, 0, st>>>(
        slab, chunk, indice_num, (size_t)cin * cout, per_chunk, filt_grad);
  else
    spconv_wgrad_reduce<<<dm_ceil_div((long long)per_chunk, 256), 256, 0, st>>>(slab, nchunks,
                                                                                per_chunk, filt_grad);
  dm_prof_end(pi, st);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

static bool wgrad_batchable(int cin, int cout) { return (cin == 4 && cout == 16) || wgrad_class(cin, cout) >= 0; }

// one job = the arguments of dm_spconv_wgrad (include/detmatch_hip.h: dm_spconv_wgrad_job)
extern "C" size_t dm_spconv_wgrad_batch_workspace_bytes(const dm_spconv_wgrad_job *jobs, int n_jobs) {
  if (!jobs || n_jobs <= 0) return 0;
  size_t total = 0, single = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const int cin = jobs[i].cin == 4 ? 16 : jobs[i].cin;      // the 4-channel layer's tiles are 16 rows high
    const size_t b = dm_spconv_wgrad_workspace_bytes(jobs[i].pair_stride, jobs[i].kvol, cin, jobs[i].cout);
    if (wgrad_batchable(jobs[i].cin, jobs[i].cout)) total += b;
    else single = b > single ? b : single;
  }
  return total + single;
}

extern "C" int dm_spconv_wgrad_batch(const dm_spconv_wgrad_job *jobs, int n_jobs, int accumulate,
                                     void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_jobs <= 0) return DM_OK;
  if (!jobs || !workspace) return DM_ERR_INVALID_ARG;
  if (workspace_bytes < dm_spconv_wgrad_batch_workspace_bytes(jobs, n_jobs)) return DM_ERR_WORKSPACE;
  WgradBatch all, cls[4];
  all.n = 0;
  all.red_base[0] = 0;
  for (int c = 0; c < 4; ++c) cls[c].n = 0, cls[c].unit_base[0] = 0;
  size_t off = 0;      // floats
  long long P = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const dm_spconv_wgrad_job &jb = jobs[i];
    if (jb.kvol <= 0 || jb.pair_stride < 0 || !jb.filt_grad) return DM_ERR_INVALID_ARG;
    if (!wgrad_batchable(jb.cin, jb.cout) || jb.pair_stride == 0) continue;
    if (!jb.feat || !jb.out_grad || !jb.indice_pairs || !jb.indice_num) return DM_ERR_INVALID_ARG;
    if (all.n == DM_WGRAD_MAX_JOBS) return DM_ERR_INVALID_ARG;
    const int cin = jb.cin == 4 ? 16 : jb.cin;
    int chunk;
    const int nchunks = wgrad_chunks(jb.pair_stride, &chunk, true);
    WgradBatch *tabs[2] = {&all, &cls[wgrad_class(cin, jb.cout)]};
    for (WgradBatch *t : tabs) {
      const int j = t->n++;
      t->feat[j] = jb.feat, t->ograd[j] = jb.out_grad, t->pairs[j] = jb.indice_pairs, t->indice_num[j] = jb.indice_num;
      t->filt_grad[j] = jb.filt_grad;
      t->pair_stride[j] = jb.pair_stride, t->kvol[j] = jb.kvol, t->cin[j] = cin, t->cout[j] = jb.cout;
      t->cin_rows[j] = jb.cin;
      t->chunk[j] = chunk, t->nchunks[j] = nchunks;
      t->slab_off[j] = off;
    }
    WgradBatch &tc = cls[wgrad_class(cin, jb.cout)];
    tc.unit_base[tc.n] = tc.unit_base[tc.n - 1] + nchunks * jb.kvol;
    all.red_base[all.n] = all.red_base[all.n - 1] + dm_ceil_div((long long)jb.kvol * cin * jb.cout, 256);
    off += dm_spconv_wgrad_workspace_bytes(jb.pair_stride, jb.kvol, cin, jb.cout) / sizeof(float);
    P += (long long)jb.pair_stride * jb.kvol;
  }
  if (all.n > 0) {
    int pi = dm_prof_begin(st, DM_PROF_SPCONV_WGRAD, -all.n, 0, 1, (int)(P > 0x7fffffffLL ? 0x7fffffffLL : P), 0, all.pairs[0]);
    float *slab = (float *)workspace;
    if (cls[0].n) spconv_wgrad_batch_rows<0><<<cls[0].unit_base[cls[0].n], 256, 0, st>>>(cls[0], slab);
    if (cls[1].n) spconv_wgrad_batch_rows<1><<<cls[1].unit_base[cls[1].n], 256, 0, st>>>(cls[1], slab);
    if (cls[2].n) spconv_wgrad_batch_rows<2><<<cls[2].unit_base[cls[2].n], 256, 0, st>>>(cls[2], slab);
    if (cls[3].n) spconv_wgrad_batch_rows<3><<<cls[3].unit_base[cls[3].n], 256, 0, st>>>(cls[3], slab);
    DM_CHECK_LAUNCH();
    spconv_wgrad_batch_reduce<<<all.red_base[all.n], 256, 0, st>>>(all, slab, accumulate);
    dm_prof_end(pi, st);
    DM_CHECK_LAUNCH();
  }
  // layers outside the batched channel pairs, and empty ones: one by one
  char *single_ws = (char *)workspace + off * sizeof(float);
  for (int i = 0; i < n_jobs; ++i) {
    const dm_spconv_wgrad_job &jb = jobs[i];
    if (wgrad_batchable(jb.cin, jb.cout) && jb.pair_stride != 0) continue;
    if (accumulate) return DM_ERR_UNSUPPORTED;      // (the caller adds such a layer's gradient itself)
    const int rc = dm_spconv_wgrad(jb.feat, jb.out_grad, jb.indice_pairs, jb.indice_num, jb.pair_stride, jb.kvol,
                                   jb.cin, jb.cout, jb.filt_grad, single_ws, workspace_bytes - off * sizeof(float),
                                   stream);
    if (rc != DM_OK) return rc;
  }
  return DM_OK;
}
