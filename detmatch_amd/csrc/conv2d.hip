// Dense 2-D convolution family for gfx950: implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// Replaces the cuDNN convolutions behind torch.nn.Conv2d / ConvTranspose2d on the DetMatch path:
//   pcdet/models/backbones_2d/base_bev_backbone.py:38-69,94-112   (BEV backbone, 3x3 / 3x3 s2 /
//                                                                  ConvT 1x1 / ConvT 2x2 s2)
//   pcdet/models/dense_heads/anchor_head_single.py:20-37          (1x1 heads)
//   mmdet ResNet-50 (caffe) + FPN + RPNHead as configured at
//   configs/detmatch/001/detmatch/split_0.py:39-99               (7x7 s2 stem, 1x1, 3x3)
//
// ONE gather-GEMM kernel covers forward, input gradient (any stride) and transposed convolution
// by describing the op as taps over an output LATTICE:
//
//   out[b, oy0 + i*oys, ox0 + j*oxs, n] (+bias[n]) (relu) =
//       sum_{t < T} sum_{c < Cin} in[b, i*iys + dy[t], j*ixs + dx[t], c] * W[ws[t]][n][c]
//
// (reads outside the input image are zero).  Rows of the GEMM are the lattice points m = (b,i,j),
// columns the output channels, K = T*Cin with the tap as the slow index.  Activations are NHWC, so a
// row's K-slice of one tap is contiguous; weights are pre-packed [slice][N][K] (dm_dconv_pack).
//   forward conv (k, stride s, pad p):  lattice = all outputs, iys = s, dy = kh - p
//   input gradient, stride 1:           lattice = all inputs,  dy = p - kh, W = packed transpose
//   input gradient, stride s:           s*s launches, one per residue class (oy0 = ry, oys = s), each
//                                       with the taps whose (ry + p - kh) is divisible by s
//   ConvTranspose2d(k = s):             s*s launches of one tap each
//
// Tile: BM x BN outputs per workgroup of 4 waves, K-step BK = 32, operands staged through LDS with
// 16-byte rows of BK+4 floats (conflict-free ds_read_b128: 36*n mod 64 hits 16 distinct 4-bank
// groups), double buffered with the next tile's global loads in flight under the MFMAs.  Each lane
// fetches 4 consecutive k of its row with one ds_read_b128 and feeds them to 4 MFMAs: the k order
// inside an 8-block is permuted identically for A and B (lane half h, element j -> k = 4h + j), which
// a dot product does not care about.  The weight gradient is a second kernel (pixels are the
// reduction index, split over workgroups, fixed-order reduce -> bitwise reproducible).
#include <type_traits>

#include "dm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DConvGeom {
  int B, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int LH, LW;
  int oy0, ox0, oys, oxs;
  int iys, ixs;
  int T;
  int relu;
  int dense_out;   // lattice == every output pixel in order: row m lives at y + m*Cout
  int M;           // B*LH*LW
  int Ktot;        // T*Cin
  const float *residual;   // optional: added to the output rows (same layout as y) before the ReLU
};

struct DConvTaps {   // 32-bit entries: a wave-uniform tap index then reads them with s_load_dword
  int dy[64], dx[64], ws[64];
};

#define DCONV_MAX_TAPS 64
#ifndef STAGGER_SLEEP
#define STAGGER_SLEEP 70   /* x64 cycles: about half a K-tile of a 128x128 tile with the pipe shared */
#endif

// LIMIT (1 or 2): pad the LDS allocation so that at most that many workgroups are resident per CU —
// used for tail launches (see dm_dconv_gemm), whose few tiles should spread over the chip instead
// of piling up on the CUs that happen to be free first.  0 = no limit.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool UNI, int LIMIT>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64) void dconv_gemm_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, const DConvGeom g, const DConvTaps tt, int n_tiles_m, int n_tiles_n,
    int m_lo, int kt_per_split, float *__restrict__ partial) {
#ifdef DCONV_STAMPS
  const float *bias_arg = bias;
  const long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDK = BK + 4;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int KQ = BK / 4;
  constexpr int RPP = NT / KQ;  // rows staged per pass
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM >= 1 && TN >= 1, "tile shape");
  constexpr int LDS_FLOATS = 2 * (BM + BN) * LDK;
  constexpr int LDS_MIN = LIMIT == 1 ? 21504 : (LIMIT == 2 ? 14336 : 0);   // 84 KB / 56 KB
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS < LDS_MIN ? LDS_MIN : LDS_FLOATS];

  // XCD-aware tile map: workgroups L, L+8, L+16 ... share an XCD (its L2); give them the column tiles
  // of the SAME row tile back to back, so the activation rows are fetched from HBM once.
  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int mt = (seq / n_tiles_n) * 8 + xcd;
  const int nt = seq % n_tiles_n;
  if (mt >= n_tiles_m) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int kq = tid % KQ, r0 = tid / KQ;
  const int m0 = m_lo + mt * BM, n0 = nt * BN;

  // per-thread row state; addresses are 32-bit BYTE offsets from the (wave-uniform) tensor base, so a
  // load is `global_load_dwordx4 v, v_off, s[base]` and a K-tile's address math is a few 32-bit adds
  int a_iy[AP], a_ix[AP];
  unsigned a_off[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + r0 + p * RPP;
    if (m < g.M) {
      const int j = m % g.LW, tmp = m / g.LW, i = tmp % g.LH, b = tmp / g.LH;
      a_iy[p] = i * g.iys;
      a_ix[p] = j * g.ixs;
      a_off[p] = (unsigned)(((b * g.Hin + a_iy[p]) * g.Win + a_ix[p]) * g.Cin + kq * 4) * 4u;
    } else {
      a_iy[p] = -(1 << 20);  // every tap lands outside the image
      a_ix[p] = 0;
      a_off[p] = 0;
    }
  }
  // weight rows beyond Cout are clamped to row 0: their products land in accumulator columns that
  // the epilogue never stores
  unsigned b_off[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int n = n0 + r0 + p * RPP;
    b_off[p] = (unsigned)((n < g.Cout ? n : 0) * g.Cin + kq * 4) * 4u;
  }
  const char *xb = (const char *)x, *wb = (const char *)w;
  const unsigned slice_bytes = (unsigned)g.Cout * g.Cin * 4u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 ra[AP], rb[BP];
  bool ra_ok[AP];
  // split-K: this workgroup owns K-tiles [kt0, kt0 + KT) of the reduction (blockIdx.y = split)
  const int KT_all = (g.Ktot + BK - 1) / BK;
  const int kt0 = blockIdx.y * kt_per_split;
  int tU = (kt0 * BK) / g.Cin, cU = (kt0 * BK) % g.Cin;  // UNI: tap / first channel of the K-tile being fetched

  // Loads are UNCONDITIONAL (offset clamped to a valid element; padding taps are zeroed later, in
  // sstore, so that nothing waits for the data before the MFMAs of the current tile): a conditional
  // load makes hipcc branch around it and drain vmcnt per element, which serialises the eight
  // fetches of a K-tile into eight dependent L2 round trips.
  // K-tile fetch, split into pieces (one global load each) so that the main loop can weave them
  // between groups of MFMAs: gprep() computes the tile's wave-uniform part, gA(p) / gB(p) issue one
  // load, sA(p) / sB(p) mask and store one piece to LDS.
  int g_dy = 0, g_dx = 0;
  unsigned g_shift = 0, g_wshift = 0;
  bool g_kv = true;
  auto gprep = [&](int kt) {
    if (UNI) {
      const int tr = __builtin_amdgcn_readfirstlane(tU);
      const int t = tr < g.T ? tr : g.T - 1;   // past the last tile: a valid, unused fetch
      const int c0 = __builtin_amdgcn_readfirstlane(cU);
      cU += BK;
      const int wrap = cU >= g.Cin;
      tU += wrap;
      cU = wrap ? 0 : cU;
      g_dy = tt.dy[t], g_dx = tt.dx[t];
      g_shift = (unsigned)((g_dy * g.Win + g_dx) * g.Cin + c0) * 4u;   // SALU
      g_wshift = (unsigned)tt.ws[t] * slice_bytes + (unsigned)c0 * 4u;
    } else {
      const int k = (kt0 + kt) * BK + kq * 4;
      const int t = k / g.Cin, c = k - t * g.Cin;
      g_kv = t < g.T;
      const int tc = g_kv ? t : 0;
      g_dy = tt.dy[tc], g_dx = tt.dx[tc];
      g_shift = (unsigned)((g_dy * g.Win + g_dx) * g.Cin + c - kq * 4) * 4u;
      g_wshift = (unsigned)tt.ws[tc] * slice_bytes + (unsigned)(c - kq * 4) * 4u;
    }
  };
  auto gA = [&](int p) {
    const int iy = a_iy[p] + g_dy, ix = a_ix[p] + g_dx;
    const bool ok = g_kv & ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
    ra[p] = *(const float4 *)(xb + (ok ? a_off[p] + g_shift : 0u));
    ra_ok[p] = ok;
  };
  auto gB = [&](int p) {
    rb[p] = *(const float4 *)(wb + (g_kv ? b_off[p] + g_wshift : 0u));
  };
  auto sA = [&](int buf, int p) {
    float4 v = ra[p];
    const bool ok = ra_ok[p];
    v.x = ok ? v.x : 0.0f, v.y = ok ? v.y : 0.0f, v.z = ok ? v.z : 0.0f, v.w = ok ? v.w : 0.0f;
    *(float4 *)(lds + buf * (BM + BN) * LDK + (r0 + p * RPP) * LDK + kq * 4) = v;
  };
  auto sB = [&](int buf, int p) {
    float4 v = rb[p];
    if (!UNI) {   // generic path (the 7x7 stem only): K padding must contribute zeros
      const bool kv = g_kv;
      v.x = kv ? v.x : 0.0f, v.y = kv ? v.y : 0.0f, v.z = kv ? v.z : 0.0f, v.w = kv ? v.w : 0.0f;
    }
    *(float4 *)(lds + buf * (BM + BN) * LDK + BM * LDK + (r0 + p * RPP) * LDK + kq * 4) = v;
  };
  auto gload = [&](int kt) {
    gprep(kt);
#pragma unroll
    for (int p = 0; p < AP; ++p) gA(p);
#pragma unroll
    for (int p = 0; p < BP; ++p) gB(p);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < AP; ++p) sA(buf, p);
#pragma unroll
    for (int p = 0; p < BP; ++p) sB(buf, p);
  };

  const int lr = lane & 31, lh = lane >> 5;
  // Fragments of the NEXT 8-deep k-block are fetched from LDS before the MFMAs of the current one
  // are issued, and the staging stores of the next K-tile sit in the middle of the MFMA stream: a
  // 32x32x2 f32 MFMA occupies the matrix pipe for 64 cycles but the wave's issue slot for 8, so all
  // of this runs in the MFMAs' shadow as long as it is in the same basic block (no branches in the
  // steady-state loop: the last K-tile is peeled).
  float4 af[2][TM], bf[2][TN];
  auto frag = [&](int buf, int kb, int slot) {
    const float *As = lds + buf * (BM + BN) * LDK + (wm * WM + lr) * LDK + lh * 4;
    const float *Bs = lds + buf * (BM + BN) * LDK + BM * LDK + (wn * WN + lr) * LDK + lh * 4;
#pragma unroll
    for (int a = 0; a < TM; ++a) af[slot][a] = *(const float4 *)(As + a * 32 * LDK + kb * 8);
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[slot][b] = *(const float4 *)(Bs + b * 32 * LDK + kb * 8);
  };
  auto mma_ab = [&](int slot, int a, int b) {
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a].x, bf[slot][b].x, acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a].y, bf[slot][b].y, acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a].z, bf[slot][b].z, acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a].w, bf[slot][b].w, acc[a][b], 0, 0, 0);
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) mma_ab(slot, a, b);
  };
  constexpr int NKB = BK / 8;
  static_assert(NKB == 4, "the K-tile schedule below is written for BK = 32");

  const int KT = min(kt_per_split, KT_all - kt0);
#ifdef DCONV_STAMPS
  long long st_t[6] = {0, 0, 0, 0, 0, 0};
  const long long st_real0 = __builtin_amdgcn_s_memrealtime();
  long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i)                                        \
  do {                                                  \
    const long long now_ = __builtin_amdgcn_s_memtime(); \
    st_t[i] += now_ - st_prev;                          \
    st_prev = now_;                                     \
  } while (0)
#else
#define STAMP(i)
#endif
  gload(0);
  sstore(0);
  __syncthreads();
  STAMP(0);
  // Steady state: the 4 x (TM*TN) groups of 4 MFMAs of a K-tile are issued in a fixed order
  // (sched_barrier pins) with ONE piece of memory work in front of each group: the next tile's
  // global loads in front of the first groups, its masked LDS stores in front of the last ones
  // (>= half a tile of MFMAs later), so the wave never stops feeding the matrix pipe except at the
  // barrier.
  constexpr int G = TM * TN, NG = 4 * G, NP = AP + BP;
  constexpr int HALF = NG / 2, PPG = (NP + HALF - 1) / HALF;   // pieces per MFMA group
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    frag(buf, 0, 0);
    frag(buf, 1, 1);
    gprep(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int kb = gi / G, ab = gi % G;
#pragma unroll
      for (int q = 0; q < PPG; ++q) {
        const int p = (gi % HALF) * PPG + q;
        if (p < NP) {
          if (gi < HALF) {                 // first half of the groups: the next tile's loads
            if (p < AP) gA(p); else gB(p - AP);
          } else {                         // second half: its masked LDS stores
            if (p < AP) sA(buf ^ 1, p); else sB(buf ^ 1, p - AP);
          }
        }
      }
      if (ab == 0 && kb >= 1 && kb + 1 < 4) frag(buf, kb + 1, (kb + 1) & 1);   // prefetch next k-block
      __builtin_amdgcn_sched_barrier(0);
      mma_ab(kb & 1, ab / TN, ab % TN);
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
  }
  {
    const int buf = (KT - 1) & 1;
    frag(buf, 0, 0);
    frag(buf, 1, 1);
    mma(0);
    frag(buf, 2, 0);
    mma(1);
    frag(buf, 3, 1);
    mma(0);
    mma(1);
  }
#ifdef DCONV_STAMPS
  if (lane == 0 && g.relu == 2) {   // debug build only: relu == 2 asks for the stamp dump
    long long *dbg = (long long *)bias_arg;   // caller passes a buffer of 6 * waves int64 as `bias`
    const int wv_id = (blockIdx.x * (NT / 64) + wave);
    for (int i = 0; i < 6; ++i) dbg[wv_id * 8 + i] = st_t[i];
    dbg[wv_id * 8 + 0] = __builtin_amdgcn_s_memrealtime() - st_real0;   // 100 MHz ticks, whole loop
    dbg[wv_id * 8 + 6] = st_real0 - st_entry;                          // entry -> loop start
  }
  bias = nullptr;
  const long long st_loop_end = __builtin_amdgcn_s_memrealtime();
  long long *dbg_end = (long long *)bias_arg;
#endif

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane
  // >> 5): a lane owns single floats of 16 rows, so storing straight from the accumulators issues 64
  // two-line stores per wave (measured: a quarter of the kernel at one round of tiles).  Instead each
  // wave transposes its WM x WN tile through its own slice of the (now idle) LDS and writes whole
  // rows 16 bytes per lane.
  constexpr int LDC = WN + 4;
  static_assert(WAVES_M * WAVES_N * WM * LDC <= 2 * (BM + BN) * LDK, "epilogue tile must fit the LDS");
  __syncthreads();   // every wave has finished reading the operand tiles
  float *cs = lds + wave * WM * LDC;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + wn * WN + b * 32 + lr;
    const float bv = (bias != nullptr && partial == nullptr && n < g.Cout) ? bias[n] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] + bv;
        if (g.relu == 1 && partial == nullptr && g.residual == nullptr) v = fmaxf(v, 0.0f);
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + b * 32 + lr] = v;
      }
  }
  // (same wave reads what it wrote: the compiler's lgkmcnt wait orders it, no barrier)
  constexpr int CQ = WN / 4;          // float4 per tile row
  constexpr int RPI = 64 / CQ;        // rows per wave-instruction
  const int cq = lane % CQ, rr = lane / CQ;
  const int ncol = n0 + wn * WN + cq * 4;
  const bool vec_ok = (g.Cout & 3) == 0;
#pragma unroll 4
  for (int it = 0; it < WM / RPI; ++it) {
    const int rl = it * RPI + rr;
    const int m = m0 + wm * WM + rl;
    if (m >= g.M || ncol >= g.Cout) continue;
    size_t row;
    float *obase = y;
    if (partial != nullptr) {       // split-K: raw partial sums, dense rows [split][M][Cout]
      row = (size_t)blockIdx.y * g.M + m;
      obase = partial;
    } else if (g.dense_out) {
      row = (size_t)m;
    } else {
      const int j = m % g.LW, tmp = m / g.LW, i = tmp % g.LH, bb = tmp / g.LH;
      row = ((size_t)bb * g.Hout + g.oy0 + i * g.oys) * g.Wout + g.ox0 + j * g.oxs;
    }
    float4 v = *(const float4 *)(cs + rl * LDC + cq * 4);
    float *dst = obase + row * g.Cout + ncol;
    if (g.residual != nullptr && partial == nullptr) {   // shortcut branch of a residual block, then the ReLU
      const float *rp = g.residual + row * g.Cout + ncol;
      if (vec_ok && ncol + 3 < g.Cout) {
        const float4 r = *(const float4 *)rp;
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
      } else {
        v.x += rp[0];
        if (ncol + 1 < g.Cout) v.y += rp[1];
        if (ncol + 2 < g.Cout) v.z += rp[2];
        if (ncol + 3 < g.Cout) v.w += rp[3];
      }
      if (g.relu == 1) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    }
    if (vec_ok && ncol + 3 < g.Cout) {
      *(float4 *)dst = v;
    } else {
      dst[0] = v.x;
      if (ncol + 1 < g.Cout) dst[1] = v.y;
      if (ncol + 2 < g.Cout) dst[2] = v.z;
      if (ncol + 3 < g.Cout) dst[3] = v.w;
    }
  }
#ifdef DCONV_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && g.relu == 2)
    dbg_end[(blockIdx.x * (NT / 64) + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - st_loop_end;
#endif
}

// ------------------------------------------------------------------------------------------------
// Mixed-precision variant (dm_dconv_set_math(1)): the same lattice gather-GEMM with bf16 multiplicands
// and fp32 accumulation / storage on v_mfma_f32_32x32x16_bf16 — what the reference's fp16 configs
// (BASELINE configs[4], torch autocast) compute in half precision.  Tensors stay fp32 in HBM: a
// K-tile of 64 is fetched as float4, rounded to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) on
// its way into LDS (rows of 64 bf16 + 16 bytes: 36-dword stride, conflict-free ds_read_b128), and
// a lane's fragment is the 8 consecutive k of its half (k = 16*step + 8*(lane >> 5) + j).  One MFMA
// now does the work of eight fp32 ones, so the kernel is bound by L2 -> LDS staging instead of the
// matrix pipe: plain double buffering (next tile's loads in flight under the current tile's MFMAs),
// no pinned schedule.  Cin % 64 == 0 only (every UNI layer of the path except Cin = 32 heads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
  union { bf16x2 h; unsigned u; } lo, hi;
  lo.h = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
  hi.h = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
  return make_uint2(lo.u, hi.u);
}

//
// SPLIT = 3 (dm_dconv_set_math(2)): fp32-CLASS arithmetic on the bf16 pipe.  Every fp32 operand is split on
// its way into LDS into three bf16 planes, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// together the 24 significand bits of x.  Products of two bf16 numbers are exact in fp32, so the six products
// of weight 2^-16 and larger — hh, hm, mh, hl, lh, mm — summed into the fp32 accumulator carry the fp32
// product to 2^-24 relative (the three dropped terms are below the rounding of the sum).  Measured against
// float64 (tools/probe_bf16_split.py): rms error 2.0e-7 of the output against 6.0e-7 for the native
// v_mfma_f32_32x32x2_f32 kernel — the matrix pipe's own fp32 instruction is the LESS accurate path — at 6
// bf16 instructions of 32 cycles per 16 k against 8 fp32 instructions of 64 (2.7x less matrix-pipe time).
// K-tile 16 (one instruction deep), three planes per buffer: 72 KB of LDS, two workgroups per CU.
// 18 VALU instructions per four elements: v_cvt_pk_bf16_f32 rounds a PAIR to nearest-even, the pair is
// widened back with one shift and one mask, the residual is one v_pk_add_f32 (the vector form of the same
// three conversions costs 30: it converts every element once more on its own to widen it).  Measured: 1-4 %
// on the split kernels — they are not bound by the VALU count alone.
__device__ __forceinline__ unsigned pack_bf16x2_rne(float a, float b) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  union { bf16x2v b; unsigned u; } r;
  const f32x2v x = {a, b};
  r.b = __builtin_convertvector(x, bf16x2v);
  return r.u;
}
__device__ __forceinline__ void split_bf16x3_pair(float a, float b, unsigned *h, unsigned *m, unsigned *l) {
  const unsigned ph = pack_bf16x2_rne(a, b);
  const float a1 = a - __uint_as_float(ph << 16), b1 = b - __uint_as_float(ph & 0xffff0000u);
  const unsigned pm = pack_bf16x2_rne(a1, b1);
  const float a2 = a1 - __uint_as_float(pm << 16), b2 = b1 - __uint_as_float(pm & 0xffff0000u);
  *h = ph, *m = pm, *l = pack_bf16x2_rne(a2, b2);
}
__device__ __forceinline__ void split_bf16x3(const float4 v, uint2 *h, uint2 *m, uint2 *l) {
  split_bf16x3_pair(v.x, v.y, &h->x, &m->x, &l->x);
  split_bf16x3_pair(v.z, v.w, &h->y, &m->y, &l->y);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int LIMIT, int SPLIT = 1>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(SPLIT == 3 ? 2 : 1)))
void dconv_gemm_bf16_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, const DConvGeom g, const DConvTaps tt, int n_tiles_m, int n_tiles_n,
    int m_lo, int kt_per_split, float *__restrict__ partial) {
  constexpr int BK = SPLIT == 3 ? 16 : 64;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDW = BK / 2 + 4;                      // dwords per LDS row
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int KQ = BK / 4;                           // float4 pieces per row
  constexpr int RPP = NT / KQ;
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM >= 1 && TN >= 1, "tile shape");
  constexpr int PLANE = (BM + BN) * LDW;               // dwords of one bf16 plane of one buffer
  constexpr int LDS_WORDS = 2 * SPLIT * PLANE;
  constexpr int LDS_MIN = LIMIT == 1 ? 21504 : (LIMIT == 2 ? 14336 : 0);
  __shared__ __attribute__((aligned(16))) unsigned ldsw[LDS_WORDS < LDS_MIN ? LDS_MIN : LDS_WORDS];

  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int mt = (seq / n_tiles_n) * 8 + xcd;
  const int nt = seq % n_tiles_n;
  if (mt >= n_tiles_m) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // SPLIT == 3 (rows of 48 bytes, 8-byte plane stores banked mod 128 bytes per 16 contiguous lanes): four lanes store
  // one row's 32 bytes, so a group of 16 lanes must take rows {r, r + 2, r + 4, r + 6} (offsets 0, 96, 64, 32 mod 128)
  // — with rows {r .. r + 3} the fourth lands on the first two (a 2-way conflict on every store: one third of the
  // kernel's LDS cycles by SQ_LDS_BANK_CONFLICT)
  const int kq = tid % KQ, rj = tid / KQ;
  const int r0 = SPLIT == 3 ? ((rj & ~7) | ((rj & 3) << 1) | ((rj >> 2) & 1)) : rj;
  const int m0 = m_lo + mt * BM, n0 = nt * BN;

  int a_iy[AP], a_ix[AP];
  unsigned a_off[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + r0 + p * RPP;
    if (m < g.M) {
      const int j = m % g.LW, tmp = m / g.LW, i = tmp % g.LH, b = tmp / g.LH;
      a_iy[p] = i * g.iys;
      a_ix[p] = j * g.ixs;
      a_off[p] = (unsigned)(((b * g.Hin + a_iy[p]) * g.Win + a_ix[p]) * g.Cin + kq * 4) * 4u;
    } else {
      a_iy[p] = -(1 << 20);
      a_ix[p] = 0;
      a_off[p] = 0;
    }
  }
  unsigned b_off[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int n = n0 + r0 + p * RPP;
    b_off[p] = (unsigned)((n < g.Cout ? n : 0) * g.Cin + kq * 4) * 4u;
  }
  const char *xb = (const char *)x, *wb = (const char *)w;
  const unsigned slice_bytes = (unsigned)g.Cout * g.Cin * 4u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // Register ring of fetched K-tiles: DEPTH tiles in flight (slot of tile kt = kt % DEPTH).  The mixed-precision
  // kernel (SPLIT 1, K-tile 64) keeps one.  The split kernel's K-tile is one instruction deep — 6 * TM * TN matrix
  // instructions per wave, a fraction of an L2 round trip — so it fetches DEPTH tiles ahead, and its memory work is
  // woven into the instruction stream in pieces (one global load, or one split + three LDS stores) in front of
  // each (a, b) group of six instructions, pinned by sched_barriers: the split's VALU work and the LDS traffic
  // run in the shadow of the matrix pipe instead of between its bursts.  Everything in the steady-state loop is
  // unconditional (fetches past the last tile are clamped to valid, unused data; the store of a tile past the
  // end goes to the idle buffer), so the loop body is one basic block.
  // (128 x 128 with a ring of three fits since the loops are straight-line — 214 VGPRs, no scratch — and measures the
  // same: 285 / 150 / 419 us against 281 / 149 / 415 us on BEV 256->128, 128->128 and FPN P2)
  constexpr int DEPTH = SPLIT == 3 ? (BM * BN > 64 * 64 ? 2 : 3) : 1;
  float4 ra[DEPTH][AP], rb[DEPTH][BP];
  bool ra_ok[DEPTH][AP];
  const int KT_all = g.Ktot / BK;
  const int kt0 = blockIdx.y * kt_per_split;
  int tU = (kt0 * BK) / g.Cin, cU = (kt0 * BK) % g.Cin;
  int g_dy = 0, g_dx = 0;
  unsigned g_shift = 0, g_wshift = 0;
  auto gprep = [&]() {
    const int tr = __builtin_amdgcn_readfirstlane(tU);
    const int t = tr < g.T ? tr : g.T - 1;
    const int c0 = __builtin_amdgcn_readfirstlane(cU);
    cU += BK;
    const int wrap = cU >= g.Cin;
    tU += wrap;
    cU = wrap ? 0 : cU;
    g_dy = tt.dy[t], g_dx = tt.dx[t];
    g_shift = (unsigned)((g_dy * g.Win + g_dx) * g.Cin + c0) * 4u;
    g_wshift = (unsigned)tt.ws[t] * slice_bytes + (unsigned)c0 * 4u;
  };
  auto gA = [&](auto slot, int p) {     // unconditional clamped loads, masked at the LDS store (see the fp32 kernel)
    constexpr int R = decltype(slot)::value;
    const int iy = a_iy[p] + g_dy, ix = a_ix[p] + g_dx;
    const bool ok = ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
    ra[R][p] = *(const float4 *)(xb + (ok ? a_off[p] + g_shift : 0u));
    ra_ok[R][p] = ok;
  };
  auto gB = [&](auto slot, int p) {
    constexpr int R = decltype(slot)::value;
    rb[R][p] = *(const float4 *)(wb + b_off[p] + g_wshift);
  };
  auto gload = [&](auto slot) {
    gprep();
#pragma unroll
    for (int p = 0; p < AP; ++p) gA(slot, p);
#pragma unroll
    for (int p = 0; p < BP; ++p) gB(slot, p);
  };
  auto sA = [&](int buf, auto slot, int p) {
    constexpr int R = decltype(slot)::value;
    float4 v = ra[R][p];
    const bool ok = ra_ok[R][p];
    v.x = ok ? v.x : 0.0f, v.y = ok ? v.y : 0.0f, v.z = ok ? v.z : 0.0f, v.w = ok ? v.w : 0.0f;
    unsigned *dst = ldsw + buf * SPLIT * PLANE + (r0 + p * RPP) * LDW + kq * 2;
    if constexpr (SPLIT == 3) {
#ifndef DCONV_PROBE_NO_ASPLIT
      uint2 h, m, l;
      split_bf16x3(v, &h, &m, &l);
      *(uint2 *)dst = h, *(uint2 *)(dst + PLANE) = m, *(uint2 *)(dst + 2 * PLANE) = l;
#else
      if (v.x == 123.456f) *(uint2 *)dst = make_uint2(1, 2);
#endif
    } else {
      *(uint2 *)dst = pack_bf16x4(v);
    }
  };
  auto sB = [&](int buf, auto slot, int p) {
    constexpr int R = decltype(slot)::value;
    unsigned *dst = ldsw + buf * SPLIT * PLANE + BM * LDW + (r0 + p * RPP) * LDW + kq * 2;
    if constexpr (SPLIT == 3) {
#ifndef DCONV_PROBE_NO_BSPLIT       // timing probe (tools/probe_dconv.sh): results are garbage
      uint2 h, m, l;
      split_bf16x3(rb[R][p], &h, &m, &l);
      *(uint2 *)dst = h, *(uint2 *)(dst + PLANE) = m, *(uint2 *)(dst + 2 * PLANE) = l;
#else
      if (rb[R][p].x == 123.456f) *(uint2 *)dst = make_uint2(1, 2);
#endif
    } else {
      *(uint2 *)dst = pack_bf16x4(rb[R][p]);
    }
  };
  auto sstore = [&](int buf, auto slot) {
#pragma unroll
    for (int p = 0; p < AP; ++p) sA(buf, slot, p);
#pragma unroll
    for (int p = 0; p < BP; ++p) sB(buf, slot, p);
  };
  const int lr = lane & 31, lh = lane >> 5;
  const int KT = min(kt_per_split, KT_all - kt0);
  bf16x8 af[SPLIT][TM], bfr[SPLIT][TN];
  auto frags = [&](int buf, int ks) {
    const unsigned *As = ldsw + buf * SPLIT * PLANE + (wm * WM + lr) * LDW + lh * 4;
    const unsigned *Bs = ldsw + buf * SPLIT * PLANE + BM * LDW + (wn * WN + lr) * LDW + lh * 4;
#pragma unroll
    for (int s = 0; s < SPLIT; ++s) {
#pragma unroll
      for (int a = 0; a < TM; ++a) af[s][a] = *(const bf16x8 *)(As + s * PLANE + a * 32 * LDW + ks * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b) bfr[s][b] = *(const bf16x8 *)(Bs + s * PLANE + b * 32 * LDW + ks * 8);
    }
  };
  auto mma_ab = [&](int a, int b) {
    if constexpr (SPLIT == 3) {      // smallest terms first: l h, h l, m m, m h, h m, h h
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bfr[0][b], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[2][b], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[0][b], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[1][b], acc[a][b], 0, 0, 0);
    }
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[0][b], acc[a][b], 0, 0, 0);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      frags(buf, ks);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) mma_ab(a, b);
    }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1 % DEPTH> S1;
  typedef std::integral_constant<int, 2 % DEPTH> S2;
  if constexpr (DEPTH == 1) {
    gload(S0());
    sstore(0, S0());
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < KT;
      if (more) gload(S0());
      compute(buf);
      if (more) sstore(buf ^ 1, S0());
      __syncthreads();
    }
  } else {
    gload(S0());
    gload(S1());
    if constexpr (DEPTH == 3) gload(S2());
    sstore(0, S0());
    __syncthreads();
    int kt = 0;
    constexpr int G = TM * TN, NP = AP + BP;
    constexpr int NGRP = 6;                            // one group per cross product, G independent accumulators each
    constexpr int PPG = (2 * NP + NGRP - 1) / NGRP;    // pieces of memory work per group
    // product order: smallest terms first (l h, h l, m m, m h, h m, h h); consecutive instructions of a group hit
    // DIFFERENT accumulators, so none waits for its predecessor's result
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    auto step = [&](auto cur, auto nxt) {      // cur: slot of tile kt (in LDS already), nxt: slot of tile kt + 1
      const int buf = kt & 1;
      frags(buf, 0);
      gprep();                                 // tile kt + DEPTH goes into the slot tile kt left
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gi = 0; gi < NGRP; ++gi) {
#pragma unroll
        for (int q = 0; q < PPG; ++q) {
          const int piece = gi * PPG + q;      // 0 .. NP-1: loads of tile kt + DEPTH; NP .. 2 NP-1: stores of tile kt + 1
          if (piece < NP) {
            if (piece < AP) gA(cur, piece); else gB(cur, piece - AP);
          } else if (piece < 2 * NP) {
            const int p = piece - NP;
            if (p < AP) sA(buf ^ 1, nxt, p); else sB(buf ^ 1, nxt, p - AP);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ab = 0; ab < G; ++ab)
          acc[ab / TN][ab % TN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[gi]][ab / TN], bfr[PB[gi]][ab % TN],
                                                                           acc[ab / TN][ab % TN], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      ++kt;
    };
    // whole turns of the ring as ONE straight-line loop body, the remainder behind the loop: with an exit between
    // the steps the accumulators of the two paths land in different registers and every turn copies all of them
    if constexpr (DEPTH == 3) {
      while (kt + 2 < KT) {
        step(S0(), S1());
        step(S1(), S2());
        step(S2(), S0());
      }
      if (kt < KT) step(S0(), S1());
      if (kt < KT) step(S1(), S2());
    } else {
      while (kt + 1 < KT) {
        step(S0(), S1());
        step(S1(), S0());
      }
      if (kt < KT) step(S0(), S1());
    }
  }

  // epilogue: as in the fp32 kernel (LDS-transposed tile, whole rows of 16 bytes per lane)
  constexpr int LDC = WN + 4;
  static_assert(WAVES_M * WAVES_N * WM * LDC <= LDS_WORDS, "epilogue tile must fit the LDS");
  float *cs = (float *)ldsw + wave * WM * LDC;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + wn * WN + b * 32 + lr;
    const float bv = (bias != nullptr && partial == nullptr && n < g.Cout) ? bias[n] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] + bv;
        if (g.relu == 1 && partial == nullptr && g.residual == nullptr) v = fmaxf(v, 0.0f);
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + b * 32 + lr] = v;
      }
  }
  constexpr int CQ = WN / 4;
  constexpr int RPI = 64 / CQ;
  const int cq = lane % CQ, rr = lane / CQ;
  const int ncol = n0 + wn * WN + cq * 4;
  const bool vec_ok = (g.Cout & 3) == 0;
#pragma unroll 4
  for (int it = 0; it < WM / RPI; ++it) {
    const int rl = it * RPI + rr;
    const int m = m0 + wm * WM + rl;
    if (m >= g.M || ncol >= g.Cout) continue;
    size_t row;
    float *obase = y;
    if (partial != nullptr) {
      row = (size_t)blockIdx.y * g.M + m;
      obase = partial;
    } else if (g.dense_out) {
      row = (size_t)m;
    } else {
      const int j = m % g.LW, tmp = m / g.LW, i = tmp % g.LH, bb = tmp / g.LH;
      row = ((size_t)bb * g.Hout + g.oy0 + i * g.oys) * g.Wout + g.ox0 + j * g.oxs;
    }
    float4 v = *(const float4 *)(cs + rl * LDC + cq * 4);
    float *dst = obase + row * g.Cout + ncol;
    if (g.residual != nullptr && partial == nullptr) {   // shortcut branch of a residual block, then the ReLU
      const float *rp = g.residual + row * g.Cout + ncol;
      if (vec_ok && ncol + 3 < g.Cout) {
        const float4 r = *(const float4 *)rp;
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
      } else {
        v.x += rp[0];
        if (ncol + 1 < g.Cout) v.y += rp[1];
        if (ncol + 2 < g.Cout) v.z += rp[2];
        if (ncol + 3 < g.Cout) v.w += rp[3];
      }
      if (g.relu == 1) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    }
    if (vec_ok && ncol + 3 < g.Cout) {
      *(float4 *)dst = v;
    } else {
      dst[0] = v.x;
      if (ncol + 1 < g.Cout) dst[1] = v.y;
      if (ncol + 2 < g.Cout) dst[2] = v.z;
      if (ncol + 3 < g.Cout) dst[3] = v.w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  G[t][u][v] = sum_m U[m][u] * V[b, i*vys + dy[t], j*vxs + dx[t]][v]
// (Conv2d: U = dy, V = x, G = dW[t][cout][cin]; ConvTranspose2d: U = x, V = dy, G = dW[t][cin][cout]).
// Grid = (tap, u-tile, v-tile) x nsplit chunks of the pixel range; partial tiles go to the workspace
// [nsplit][T][Cu][Cv], dconv_wgrad_reduce sums them in a fixed order and writes the caller's layout.
struct DWgradGeom {
  int B, LH, LW;         // lattice of U rows (dense: row m at U + m*Cu)
  int Cu, Cv;
  int Hv, Wv;            // V image
  int vys, vxs;
  int T;
  int M;                 // B*LH*LW
  int chunk;             // rows per split (multiple of BK)
};

template <int BU, int BV, int BK, int WAVES_U, int WAVES_V>
__global__ __launch_bounds__(WAVES_U *WAVES_V * 64) void dconv_wgrad_kernel(
    const float *__restrict__ U, const float *__restrict__ V, float *__restrict__ part,
    const DWgradGeom g, const DConvTaps tt, int n_tiles_u, int n_tiles_v) {
  constexpr int NT = WAVES_U * WAVES_V * 64;
  constexpr int LDU = BU + 4, LDV = BV + 4;
  constexpr int WU = BU / WAVES_U, WV = BV / WAVES_V, TU = WU / 32, TV = WV / 32;
  constexpr int UQ = BU / 4, VQ = BV / 4;           // float4 per staged row
  constexpr int URPP = NT / UQ, VRPP = NT / VQ;     // pixel rows per pass
  constexpr int UP = BK / URPP, VP = BK / VRPP;
  static_assert(BK % URPP == 0 && BK % VRPP == 0 && UP >= 1 && VP >= 1, "tile shape");
  constexpr int LDS_STAGE = 2 * BK * (LDU + LDV), LDS_EPI = WAVES_U * WAVES_V * WU * (WV + 4);
  __shared__ __attribute__((aligned(16))) float lds[LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI];

  int tile = blockIdx.x;
  const int vt = tile % n_tiles_v;
  tile /= n_tiles_v;
  const int ut = tile % n_tiles_u;
  const int t = tile / n_tiles_u;
  const int split = blockIdx.y;
  const int m_lo = split * g.chunk;
  const int m_hi = min(g.M, m_lo + g.chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wu = wave / WAVES_V, wv = wave % WAVES_V;
  const int u0 = ut * BU, v0 = vt * BV;
  const int uq = tid % UQ, ur0 = tid / UQ;
  const int vq = tid % VQ, vr0 = tid / VQ;
  const int dy = tt.dy[t], dx = tt.dx[t];
  const bool u_ok = u0 + uq * 4 < g.Cu, v_ok = v0 + vq * 4 < g.Cv;

  f32x16 acc[TU][TV];
#pragma unroll
  for (int a = 0; a < TU; ++a)
#pragma unroll
    for (int b = 0; b < TV; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 ru[UP], rv[VP];
  bool ru_ok[UP], rv_ok[VP];
  // unconditional loads from clamped addresses, masked on the way to LDS (see dconv_gemm_kernel);
  // one piece = one global load, woven between the MFMA groups of the main loop
  auto gU = [&](int kt, int p) {
    const int m = m_lo + kt * BK + ur0 + p * URPP;
    const bool ok = (m < m_hi) & u_ok;
    ru[p] = *(const float4 *)(U + (size_t)(ok ? m : m_lo) * g.Cu + (u_ok ? u0 + uq * 4 : 0));
    ru_ok[p] = ok;
  };
  auto gV = [&](int kt, int p) {
    const int m = m_lo + kt * BK + vr0 + p * VRPP;
    const int mm = m < m_hi ? m : m_lo;
    const int j = mm % g.LW, tmp = mm / g.LW, i = tmp % g.LH, b = tmp / g.LH;
    const int iy = i * g.vys + dy, ix = j * g.vxs + dx;
    const bool ok = (m < m_hi) & v_ok & ((unsigned)iy < (unsigned)g.Hv) & ((unsigned)ix < (unsigned)g.Wv);
    const int pix = ok ? (b * g.Hv + iy) * g.Wv + ix : 0;
    rv[p] = *(const float4 *)(V + (size_t)pix * g.Cv + (v_ok ? v0 + vq * 4 : 0));
    rv_ok[p] = ok;
  };
  auto sU = [&](int buf, int p) {
    float4 v = ru[p];
    const bool ok = ru_ok[p];
    v.x = ok ? v.x : 0.0f, v.y = ok ? v.y : 0.0f, v.z = ok ? v.z : 0.0f, v.w = ok ? v.w : 0.0f;
    *(float4 *)(lds + buf * BK * (LDU + LDV) + (ur0 + p * URPP) * LDU + uq * 4) = v;
  };
  auto sV = [&](int buf, int p) {
    float4 v = rv[p];
    const bool ok = rv_ok[p];
    v.x = ok ? v.x : 0.0f, v.y = ok ? v.y : 0.0f, v.z = ok ? v.z : 0.0f, v.w = ok ? v.w : 0.0f;
    *(float4 *)(lds + buf * BK * (LDU + LDV) + BK * LDU + (vr0 + p * VRPP) * LDV + vq * 4) = v;
  };
  const int lr = lane & 31, lh = lane >> 5;
  float af[2][TU], bf[2][TV];
  auto frag = [&](int buf, int kk, int slot) {
    const float *Us = lds + buf * BK * (LDU + LDV) + lh * LDU + wu * WU + lr;
    const float *Vs = lds + buf * BK * (LDU + LDV) + BK * LDU + lh * LDV + wv * WV + lr;
#pragma unroll
    for (int a = 0; a < TU; ++a) af[slot][a] = Us[kk * 2 * LDU + a * 32];
#pragma unroll
    for (int b = 0; b < TV; ++b) bf[slot][b] = Vs[kk * 2 * LDV + b * 32];
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TU; ++a)
#pragma unroll
      for (int b = 0; b < TV; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a], bf[slot][b], acc[a][b], 0, 0, 0);
  };

  const int KT = (m_hi - m_lo + BK - 1) / BK;
  constexpr int NG = BK / 2, HALF = NG / 2, NP = UP + VP, PPG = (NP + HALF - 1) / HALF;
  if (KT > 0) {
#pragma unroll
    for (int p = 0; p < UP; ++p) gU(0, p);
#pragma unroll
    for (int p = 0; p < VP; ++p) gV(0, p);
#pragma unroll
    for (int p = 0; p < UP; ++p) sU(0, p);
#pragma unroll
    for (int p = 0; p < VP; ++p) sV(0, p);
    __syncthreads();
    // same pinned schedule as dconv_gemm_kernel: one piece of memory work in front of each group of
    // MFMAs (loads of the next tile first, its LDS stores in the second half), fragments one k-pair
    // ahead
    for (int kt = 0; kt + 1 < KT; ++kt) {
      const int buf = kt & 1;
      frag(buf, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
        for (int q = 0; q < PPG; ++q) {
          const int p = (gi % HALF) * PPG + q;
          if (p < NP) {
            if (gi < HALF) {
              if (p < UP) gU(kt + 1, p); else gV(kt + 1, p - UP);
            } else {
              if (p < UP) sU(buf ^ 1, p); else sV(buf ^ 1, p - UP);
            }
          }
        }
        if (gi + 1 < NG) frag(buf, gi + 1, (gi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(gi & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    {
      const int buf = (KT - 1) & 1;
      frag(buf, 0, 0);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) frag(buf, gi + 1, (gi + 1) & 1);
        mma(gi & 1);
      }
    }
  }
  float *dst = part + ((size_t)split * g.T + t) * g.Cu * g.Cv;
  // transposed through LDS and stored 16 bytes per lane (see dconv_gemm_kernel's epilogue)
  constexpr int LDC = WV + 4;
  __syncthreads();
  float *cs = lds + wave * WU * LDC;
#pragma unroll
  for (int b = 0; b < TV; ++b)
#pragma unroll
    for (int a = 0; a < TU; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + b * 32 + lr] = acc[a][b][r];
  constexpr int CQ = WV / 4, RPI = 64 / CQ;
  const int cq = lane % CQ, rr = lane / CQ;
  const int vcol = v0 + wv * WV + cq * 4;
#pragma unroll 4
  for (int it = 0; it < WU / RPI; ++it) {
    const int rl = it * RPI + rr;
    const int u = u0 + wu * WU + rl;
    if (u < g.Cu && vcol < g.Cv)     // Cv % 4 == 0: whole float4 in range
      *(float4 *)(dst + (size_t)u * g.Cv + vcol) = *(const float4 *)(cs + rl * LDC + cq * 4);
  }
}

// ---- 3 x 3, stride 1: the split kernel on an input PATCH ------------------------------------------------------
// The lattice GEMM above stages, per k-block, the input pixels of ONE tap — so every input element is fetched,
// split into its three bf16 planes and written to LDS nine times (once per tap), and that VALU work, not the matrix
// pipe, paces dconv_gemm_bf16_kernel<.., 3> (rocprofv3 SQ counters, DESIGN §6.3: 8.3 vector instructions per
// matrix instruction).  For the 3 x 3 / stride-1 layers (BEV blocks, ResNet conv2, FPN and RPN 3 x 3, and their
// input gradients — 60 % of the dense flops) a workgroup here owns an 8 x 16 patch of output pixels x BN output
// channels and stages, per 16-channel block, the 10 x 18 input patch ONCE: the nine taps read their A fragments
// from the same LDS image at shifted rows.  A-side fetch + split work drops 6.4x; the weights (BN x 16 per tap)
// are staged per (tap, block) as before, double-buffered, fetched two tiles ahead.
// LDS: A 3 planes x 180 rows x 12 words (25.9 KB, single buffer: refreshed between channel blocks), B 2 buffers x 3
// planes x BN rows x 12 words (36.9 KB at BN = 128): 62.8 KB, two workgroups per CU.
// Which pixel of a 2 x 16 block the MFMA row i (= lane & 31 of the A operand) stands for.  ds_read_b128 serves a wave in
// four NON-contiguous groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32); with the
// plain map (i >> 4, i & 15) a group reads patch rows {0-3, 12-15, 22-29} + const of the 48-byte rows: rows 12/28 and
// 13/29 are 16 rows = 768 bytes = 3 bank rows apart, a 2-way conflict on every A fragment read (SQ_LDS_BANK_CONFLICT
// 35 % of the LDS cycles).  Here each group gets 16 rows that are distinct mod 16: 3 * row mod 16 then covers the
// sixteen 16-byte slots of the 256-byte bank row once (checked for every tap shift).  The epilogue places row i at
// the same pixel.
__device__ __forceinline__ void dm_patch_pixel(int i, int *py, int *px) {
  const int blk = i >> 2;                    // groups of four lanes move together
  // blk:      0  1  2  3  4  5  6  7
  // row:      0  0  0  0  1  1  1  1
  // column:   0  8 12  4 14  6 10  2   (+ i & 3; blk 4: 14, 15, 0, 1)
  const int col0 = (0x2A6E4C80u >> (4 * blk)) & 15;
  *py = i >> 4;
  *px = (col0 + (i & 3)) & 15;
}

template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void dconv_patch_split_kernel(const float *__restrict__ x, const float *__restrict__ w,
                              const float *__restrict__ bias, float *__restrict__ y, const DConvGeom g,
                              const DConvTaps tt, int tiles_y, int tiles_x, int n_tiles_m, int n_tiles_n) {
  static_assert(WAVES_M * WAVES_N == 4, "four waves");
  constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PR = PH * PW;     // 180 patch rows
  constexpr int LDW = 12;
  constexpr int A_PLANE = PR * LDW, B_PLANE = BN * LDW;
  constexpr int WM = 128 / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int ROWS_W = TH / WAVES_M;                 // patch rows of output pixels per wave
  constexpr int AP = (PR * 4 + 255) / 256;             // float4 pieces of the A patch per thread (3)
  constexpr int BP = BN * 4 / 256;                     // ... of a weight tile (2 at BN = 128)
  static_assert(TM >= 1 && TN >= 1 && BP >= 1 && ROWS_W * 16 == WM, "tile shape");
  constexpr int LDS_WORDS = 3 * A_PLANE + 2 * 3 * B_PLANE;
  __shared__ __attribute__((aligned(16))) unsigned ldsw[LDS_WORDS];
  unsigned *lds_a = ldsw, *lds_b = ldsw + 3 * A_PLANE;

  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int mt = (seq / n_tiles_n) * 8 + xcd;
  const int nt = seq % n_tiles_n;
  if (mt >= n_tiles_m) return;
  const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, bimg = mt / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  // A patch pieces of this thread: piece index q = tid + 256 p -> patch row q / 4, channel quad q % 4
  unsigned a_off[AP];
  bool a_ok[AP];
  int a_lds[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int q = tid + 256 * p;
    const int rj = q >> 2, kq = q & 3;
    // 16 contiguous lanes store rows {r, r + 2, r + 4, r + 6} of the 48-byte rows (see dconv_gemm_bf16_kernel)
    const int row = rj < (PR & ~7) ? ((rj & ~7) | ((rj & 3) << 1) | ((rj >> 2) & 1)) : rj;
    const int py = row / PW, px = row % PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = (row < PR) & ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
    a_ok[p] = ok;
    a_off[p] = ok ? (unsigned)(((bimg * g.Hin + iy) * g.Win + ix) * g.Cin + kq * 4) * 4u : 0u;
    a_lds[p] = row < PR ? row * LDW + kq * 2 : -1;
  }
  unsigned b_off[BP];
  int b_lds[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int q = tid + 256 * p;
    const int n = n0 + (q >> 2), kq = q & 3;
    b_off[p] = (unsigned)((n < g.Cout ? n : 0) * g.Cin + kq * 4) * 4u;
    b_lds[p] = (q >> 2) * LDW + kq * 2;
  }
  const char *xb = (const char *)x, *wb = (const char *)w;
  const unsigned slice_bytes = (unsigned)g.Cout * g.Cin * 4u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int NCB = g.Cin / 16, T = g.T, KT = NCB * T;
  float4 ra[AP], rb[2][BP];
  auto gA = [&](int cb) {
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *(const float4 *)(xb + a_off[p] + (a_ok[p] ? (unsigned)cb * 64u : 0u));
  };
  auto sA = [&]() {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      if (a_lds[p] < 0) continue;
      float4 v = ra[p];
      const bool ok = a_ok[p];
      v.x = ok ? v.x : 0.0f, v.y = ok ? v.y : 0.0f, v.z = ok ? v.z : 0.0f, v.w = ok ? v.w : 0.0f;
      unsigned *dst = lds_a + a_lds[p];
#ifndef DCONV_PROBE_NO_ASPLIT
      uint2 h, m, l;
      split_bf16x3(v, &h, &m, &l);
      *(uint2 *)dst = h, *(uint2 *)(dst + A_PLANE) = m, *(uint2 *)(dst + 2 * A_PLANE) = l;
#else
      if (v.x == 123.456f) *(uint2 *)dst = make_uint2(1, 2);
#endif
    }
  };
  int ktf = 0;           // k-tile the next gB() fetches: tap = ktf % T, channel block = ktf / T (clamped at the end)
  auto gB = [&](auto slot) {
    constexpr int R = decltype(slot)::value;
    const int kc = ktf < KT ? ktf : KT - 1;
    const int tap = __builtin_amdgcn_readfirstlane(kc % T), cb = __builtin_amdgcn_readfirstlane(kc / T);
    ++ktf;
    const unsigned wshift = (unsigned)tt.ws[tap] * slice_bytes + (unsigned)cb * 64u;
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[R][p] = *(const float4 *)(wb + b_off[p] + wshift);
  };
  auto sB = [&](int buf, auto slot, int p) {
    constexpr int R = decltype(slot)::value;
    unsigned *dst = lds_b + buf * 3 * B_PLANE + b_lds[p];
#ifndef DCONV_PROBE_NO_BSPLIT
    uint2 h, m, l;
    split_bf16x3(rb[R][p], &h, &m, &l);
    *(uint2 *)dst = h, *(uint2 *)(dst + B_PLANE) = m, *(uint2 *)(dst + 2 * B_PLANE) = l;
#else
    if (rb[R][p].x == 123.456f) *(uint2 *)dst = make_uint2(1, 2);
#endif
  };
  // fragment addresses: MFMA tile a of this wave covers patch rows ROWS_W*wm + 2a, +1 (16 pixels each)
  int apy, apx;
  dm_patch_pixel(lr, &apy, &apx);
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
    a_base[a] = ((ROWS_W * wm + 2 * a + apy + 1) * PW + apx + 1) * LDW + lh * 4;
  const int b_base = (wn * WN + lr) * LDW + lh * 4;

  bf16x8 af[3][TM], bfr[3][TN];
  auto frags = [&](int buf, int tap) {
    const int shift = (tt.dy[tap] * PW + tt.dx[tap]) * LDW;
    const unsigned *Bs = lds_b + buf * 3 * B_PLANE + b_base;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int a = 0; a < TM; ++a) af[s][a] = *(const bf16x8 *)(lds_a + s * A_PLANE + a_base[a] + shift);
#pragma unroll
      for (int b = 0; b < TN; ++b) bfr[s][b] = *(const bf16x8 *)(Bs + s * B_PLANE + b * 32 * LDW);
    }
  };
  auto mma_ab = [&](int a, int b) {      // smallest terms first: l h, h l, m m, m h, h m, h h
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bfr[0][b], acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[2][b], acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[0][b], acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[1][b], acc[a][b], 0, 0, 0);
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[0][b], acc[a][b], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;

  gA(0);
  gB(S0());
  gB(S1());
  sA();
#pragma unroll
  for (int p = 0; p < BP; ++p) sB(0, S0(), p);
  __syncthreads();
  int kt = 0;
  constexpr int G = TM * TN;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  auto step = [&](int tap, auto cur, auto nxt) {   // cur: ring slot of tile kt (in LDS already), nxt: of tile kt + 1
    const int buf = kt & 1;
    frags(buf, tap);
    gB(cur);                                   // tile kt + 2 into the slot tile kt left
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int gi = 0; gi < 6; ++gi) {           // one group per cross product: G independent accumulators
      if (gi < BP) sB(buf ^ 1, nxt, gi);       // split + store of tile kt + 1, one piece per group
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ab = 0; ab < G; ++ab)
        acc[ab / TN][ab % TN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[gi]][ab / TN], bfr[PB[gi]][ab % TN],
                                                                         acc[ab / TN][ab % TN], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    ++kt;
  };
  // (channel block, tap) steps in PAIRS as one straight-line loop body (a branch on the parity of kt puts the
  // accumulators of the two paths into different registers and copies all 64 of them every other tap)
  int cb = 0, tap = 0;
  auto one = [&](auto cur, auto nxt) {
    if (tap == 0 && cb + 1 < NCB) gA(cb + 1);
    step(tap, cur, nxt);
    if (++tap == T) {
      tap = 0;
      if (++cb < NCB) {        // every wave has passed the last tap's barrier: the A image is free
        sA();
        __syncthreads();
      }
    }
  };
  const int total = NCB * T;
  while (kt + 1 < total) {
    one(S0(), S1());
    one(S1(), S0());
  }
  if (kt < total) one(S0(), S1());

  // epilogue: per 32-column block the wave's WM x 32 tile through LDS (the B buffers are idle now), rows of 16 bytes
  constexpr int LDC = 36;
  static_assert(4 * WM * LDC <= 2 * 3 * B_PLANE + 3 * A_PLANE, "epilogue tile must fit the LDS");
  float *cs = (float *)ldsw + wave * WM * LDC;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int ncol0 = n0 + wn * WN + b * 32;
    const float bv = (bias != nullptr && ncol0 + lr < g.Cout) ? bias[ncol0 + lr] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] + bv;
        if (g.relu == 1) v = fmaxf(v, 0.0f);
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + lr] = v;
      }
    const int cq = lane & 7, rr = lane >> 3;
    const int ncol = ncol0 + cq * 4;
#pragma unroll
    for (int it = 0; it < WM / 8; ++it) {
      const int rl = it * 8 + rr;                       // pixel of the wave's tile: row rl / 16, column rl % 16
      int epy, epx;
      dm_patch_pixel(rl & 31, &epy, &epx);
      const int oy = y0 + ROWS_W * wm + 2 * (rl >> 5) + epy, ox = x0 + epx;
      if (oy < g.LH && ox < g.LW && ncol < g.Cout) {
        const float4 v = *(const float4 *)(cs + rl * LDC + cq * 4);
        float *dst = y + (((size_t)bimg * g.LH + oy) * g.LW + ox) * g.Cout + ncol;
        if (ncol + 3 < g.Cout) {
          *(float4 *)dst = v;
        } else {
          dst[0] = v.x;
          if (ncol + 1 < g.Cout) dst[1] = v.y;
          if (ncol + 2 < g.Cout) dst[2] = v.z;
        }
      }
    }
  }
}

// ---- the patch kernel with the weights through LDS-DMA ---------------------------------------------------------
// Timing probes (tools/probe_dconv.sh: the kernel above with the weight split + LDS stores compiled out) put the
// in-loop weight work at 15-20 % of the 3 x 3 layers' time: every workgroup re-fetches, re-splits and re-stores
// the same (tap, 16-channel) weight tile that every other workgroup of the layer handles too.  Here the pack
// kernels (`planes` copy: dconv_pack_planes) split the weights ONCE per weight update and lay the three bf16
// planes out in MFMA-fragment order — [slice][16-channel block][32-column block][plane][lane][8 bf16]: one
// (column block, plane) fragment is 1 KiB, a (tap, channel block) tile of BN columns BN/32 x 3 KiB, contiguous —
// and a workgroup moves a tile into LDS with BN/32 x 3 `global_load_lds_dwordx4` instructions (no VGPR, no VALU,
// no ds_write: the LDS image of such an instruction is lane-linear, which IS the fragment order).  Ring of four
// tiles: tile kt + 3 is requested at step kt into the slot step kt - 1 read; before the barrier that ends step kt
// a counted `s_waitcnt vmcnt` retires tile kt + 1 (tiles kt + 2, kt + 3 stay in flight across the barrier — raw
// s_barrier: __syncthreads() would drain the DMA).  The A patch is still fetched through registers (it needs the
// split), by inline-asm loads: the compiler's own wait insertion would answer the first use of an ordinary load
// with vmcnt(0) while a DMA is in flight; the loads are nine steps old when sA() reads them, long retired by the
// steps' counted waits (vmcnt retires in issue order).
// Fragment reads as inline asm: the compiler answers a C++ read of an LDS array that an LDS-DMA may be writing
// with vmcnt(0) — which would drain the ring at every step.  The reads are fenced by hand: one lgkmcnt(0) behind
// the last of them (dm_frags_ready), and the values pass through an empty asm so that no use moves above it.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ f32x4v dm_ds_read128(unsigned addr) {
  f32x4v v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ unsigned dm_lds_addr(const void *p) {     // byte offset inside the workgroup's LDS
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
template <int I, int N, typename F>
__device__ __forceinline__ void dm_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    dm_static_for<I + 1, N>(f);
  }
}
template <int N>
__device__ __forceinline__ void dm_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dm_lds_barrier() {      // LDS operations of this wave done, then the workgroup barrier
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void dconv_patch_gl_kernel(const float *__restrict__ x, const unsigned char *__restrict__ wpl,
                           const float *__restrict__ bias, float *__restrict__ y, const DConvGeom g,
                           const DConvTaps tt, int tiles_y, int tiles_x, int n_tiles_m, int n_tiles_n, int nnb) {
  static_assert(WAVES_M * WAVES_N == 4, "four waves");
  constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PR = PH * PW;     // 180 patch rows
  constexpr int LDW = 12;
  constexpr int A_PLANE = PR * LDW;
  constexpr int WM = 128 / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int ROWS_W = TH / WAVES_M;
  constexpr int AP = (PR * 4 + 255) / 256;
#ifndef DCONV_GL_NBUF
#define DCONV_GL_NBUF 4
#endif
  constexpr int NBUF = DCONV_GL_NBUF;
  constexpr int PIECES = BN / 32 * 3;          // 1 KiB fragments of a weight tile
  constexpr int PPW = (PIECES + 3) / 4;        // DMA instructions per wave and tile: piece j * 4 + wave (BN = 64: waves 2, 3 one less)
  constexpr int REM = PIECES % 4;              // 0: every wave issues PPW
  constexpr int SLOT_W = PIECES * 256;         // words of a ring slot
  static_assert(TM >= 1 && TN >= 1 && ROWS_W * 16 == WM, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned lds_a[3 * A_PLANE];
  __shared__ __attribute__((aligned(1024))) unsigned lds_b[NBUF * SLOT_W];

  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int mt = (seq / n_tiles_n) * 8 + xcd;
  const int nt = seq % n_tiles_n;
  if (mt >= n_tiles_m) return;
  const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, bimg = mt / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  unsigned a_off[AP];
  bool a_ok[AP];
  int a_lds[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int q = tid + 256 * p;
    const int rj = q >> 2, kq = q & 3;
    // 16 contiguous lanes store rows {r, r + 2, r + 4, r + 6} of the 48-byte rows (see dconv_gemm_bf16_kernel)
    const int row = rj < (PR & ~7) ? ((rj & ~7) | ((rj & 3) << 1) | ((rj >> 2) & 1)) : rj;
    const int py = row / PW, px = row % PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = (row < PR) & ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
    a_ok[p] = ok;
    a_off[p] = ok ? (unsigned)(((bimg * g.Hin + iy) * g.Win + ix) * g.Cin + kq * 4) * 4u : 0u;
    a_lds[p] = row < PR ? row * LDW + kq * 2 : -1;
  }
  const char *xb = (const char *)x;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int NCB = g.Cin / 16, T = g.T, KT = NCB * T;
  f32x4v ra[AP];
  auto gA = [&](int cb) {       // asm: invisible to the compiler's wait insertion (see the header comment)
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const char *src = xb + a_off[p] + (a_ok[p] ? (unsigned)cb * 64u : 0u);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[p]) : "v"(src) : "memory");
    }
  };
  auto sA = [&]() {
#pragma unroll
    for (int p = 0; p < AP; ++p) asm volatile("" : "+v"(ra[p])::"memory");     // not before the wait that retired the loads
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      if (a_lds[p] < 0) continue;
      const bool ok = a_ok[p];
      float4 v;
      v.x = ok ? ra[p][0] : 0.0f, v.y = ok ? ra[p][1] : 0.0f, v.z = ok ? ra[p][2] : 0.0f, v.w = ok ? ra[p][3] : 0.0f;
      uint2 h, m, l;
      split_bf16x3(v, &h, &m, &l);
      unsigned *dst = lds_a + a_lds[p];
      *(uint2 *)dst = h, *(uint2 *)(dst + A_PLANE) = m, *(uint2 *)(dst + 2 * A_PLANE) = l;
    }
  };
  // weight tiles: slice tt.ws[tap], channel block cb, column blocks n0/32 ... of the planes copy
  const unsigned char *wbase = wpl + (size_t)(n0 >> 5) * 3072 + lane * 16;
  const size_t cb_stride = (size_t)nnb * 3072;
  int ktf = 0;
  auto gB = [&]() {
    const int kc = ktf < KT ? ktf : KT - 1;
    const int tap = __builtin_amdgcn_readfirstlane(kc % T), cb = __builtin_amdgcn_readfirstlane(kc / T);
    const unsigned char *src = wbase + ((size_t)tt.ws[tap] * NCB + cb) * cb_stride;
    unsigned *dst = lds_b + (ktf % NBUF) * SLOT_W;
    ++ktf;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = j * 4 + wave;
      if (REM == 0 || j + 1 < PPW || wave < REM)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024),
                                         (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
    }
  };
  auto wait_tiles = [&](auto keep_) {      // all but the youngest `keep` tiles of this wave's requests have landed
    constexpr int keep = decltype(keep_)::value;
    if constexpr (REM == 0) {
      dm_wait_vmcnt<keep * PPW>();
    } else {
      if (wave < REM) dm_wait_vmcnt<keep * PPW>();
      else dm_wait_vmcnt<keep * (PPW - 1)>();
    }
  };
  int apy, apx;
  dm_patch_pixel(lr, &apy, &apx);
  unsigned a_base[TM];        // byte addresses inside LDS
#pragma unroll
  for (int a = 0; a < TM; ++a)
    a_base[a] = dm_lds_addr(lds_a) + (((ROWS_W * wm + 2 * a + apy + 1) * PW + apx + 1) * LDW + lh * 4) * 4u;
  const unsigned b_base = dm_lds_addr(lds_b) + ((wn * TN * 3) * 256 + lane * 4) * 4u;

  f32x4v afr[3][TM], bfrr[3][TN];
  bf16x8 af[3][TM], bfr[3][TN];
  auto frags = [&](int slot, int tap) {
    const unsigned shift = (unsigned)((tt.dy[tap] * PW + tt.dx[tap]) * LDW * 4);
    const unsigned bs = b_base + (unsigned)slot * (SLOT_W * 4u);
    dm_static_for<0, 3>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
#pragma unroll
      for (int a = 0; a < TM; ++a) afr[s][a] = dm_ds_read128<s * A_PLANE * 4>(a_base[a] + shift);
      dm_static_for<0, TN>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        bfrr[s][b] = dm_ds_read128<(b * 3 + s) * 1024>(bs);
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        asm volatile("" : "+v"(afr[s][a])::"memory");
        af[s][a] = __builtin_bit_cast(bf16x8, afr[s][a]);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        asm volatile("" : "+v"(bfrr[s][b])::"memory");
        bfr[s][b] = __builtin_bit_cast(bf16x8, bfrr[s][b]);
      }
    }
  };
  constexpr int G = TM * TN;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first

  gA(0);
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) gB();
  wait_tiles(std::integral_constant<int, NBUF - 2>());      // the A patch and tile 0
  sA();
  dm_lds_barrier();
  int cb = 0, tap = 0;
  for (int kt = 0; kt < KT; ++kt) {
#ifndef DCONV_GL_PROBE_NOA       // timing probes (tools/probe_dconv.sh): results are garbage
    if (tap == 0 && cb + 1 < NCB) gA(cb + 1);
#endif
#ifndef DCONV_GL_PROBE_NOB
    gB();                                   // tile kt + 3 into the slot step kt - 1 read
#endif
    frags(kt % NBUF, tap);
    __builtin_amdgcn_sched_barrier(0);
#ifndef DCONV_GL_PROBE_NOMFMA
#pragma unroll
#else
    for (int s_ = 0; s_ < 3; ++s_) { acc[0][0][s_] += (float)af[s_][0][0] + (float)bfr[s_][TN - 1][1] + (float)af[s_][TM - 1][2]; }
#pragma unroll 1
    for (int gi_ = 0; gi_ < 0; ++gi_)
#endif
    for (int gi = 0; gi < 6; ++gi) {
#pragma unroll
      for (int ab = 0; ab < G; ++ab)
        acc[ab / TN][ab % TN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[gi]][ab / TN], bfr[PB[gi]][ab % TN],
                                                                         acc[ab / TN][ab % TN], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_tiles(std::integral_constant<int, NBUF - 2>());      // tile kt + 1 has landed (this wave's pieces; the barrier covers the others')
    dm_lds_barrier();
    if (++tap == T) {
      tap = 0;
      if (++cb < NCB) {        // every wave has passed the last tap's barrier: the A image is free
#ifndef DCONV_GL_PROBE_NOA
        sA();
        dm_lds_barrier();
#endif
      }
    }
  }
  dm_wait_vmcnt<0>();          // the clamped tail requests: nothing may land in LDS after the epilogue took it over
  dm_lds_barrier();

  constexpr int LDC = 36;
  static_assert(4 * WM * LDC <= NBUF * SLOT_W, "epilogue tile must fit the weight ring");
  float *cs = (float *)lds_b + wave * WM * LDC;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int ncol0 = n0 + wn * WN + b * 32;
    const float bv = (bias != nullptr && ncol0 + lr < g.Cout) ? bias[ncol0 + lr] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] + bv;
        if (g.relu == 1) v = fmaxf(v, 0.0f);
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + lr] = v;
      }
    const int cq = lane & 7, rr = lane >> 3;
    const int ncol = ncol0 + cq * 4;
#pragma unroll
    for (int it = 0; it < WM / 8; ++it) {
      const int rl = it * 8 + rr;
      int epy, epx;
      dm_patch_pixel(rl & 31, &epy, &epx);
      const int oy = y0 + ROWS_W * wm + 2 * (rl >> 5) + epy, ox = x0 + epx;
      if (oy < g.LH && ox < g.LW && ncol < g.Cout) {
        const float4 v = *(const float4 *)(cs + rl * LDC + cq * 4);
        float *dst = y + (((size_t)bimg * g.LH + oy) * g.LW + ox) * g.Cout + ncol;
        if (ncol + 3 < g.Cout) {
          *(float4 *)dst = v;
        } else {
          dst[0] = v.x;
          if (ncol + 1 < g.Cout) dst[1] = v.y;
          if (ncol + 2 < g.Cout) dst[2] = v.z;
        }
      }
    }
  }
}

// Mixed-precision weight gradient (dm_dconv_set_math(1)), 128 x 128 tiles: both operands have the
// reduction index (pixels) as their ROW index in memory, so the MFMA fragments (8 consecutive pixels of
// one channel) are columns of the staged tile.  The tile is stored as it arrives — [pixel][channel]
// bf16 rows of 256 B, 16-byte chunks XOR-swizzled — and read with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: per 16 lanes a block of 4 pixel rows x 16 channels, delivered channel-major),
// two reads per v_mfma_f32_32x32x16_bf16 operand, conflict-free on this image.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wg_tile_off(int row, int chunk) {   // bytes; chunk = 16-byte piece of the row
  return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// SPLIT = 3: the fp32-class mode of dconv_gemm_bf16_kernel (three bf16 planes per operand, six products),
// K-tile of 16 pixels.
template <int SPLIT>
__global__ __launch_bounds__(256) void dconv_wgrad_bf16_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                               float *__restrict__ part, const DWgradGeom g,
                                                               const DConvTaps tt, int n_tiles_u, int n_tiles_v) {
  constexpr int BU = 128, BV = 128, BK = SPLIT == 3 ? 16 : 32, WU = 64, WV = 64, TU = 2, TV = 2;
  constexpr int NP = BK / 8;                                 // pixel rows per thread and tile
  constexpr int TILE_BYTES = BK * 256;                       // one operand, one plane, one buffer
  constexpr int LDS_STAGE = 4 * SPLIT * TILE_BYTES, LDS_EPI = 4 * WU * (WV + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI];

  int tile = blockIdx.x;
  const int vt = tile % n_tiles_v;
  tile /= n_tiles_v;
  const int ut = tile % n_tiles_u;
  const int t = tile / n_tiles_u;
  const int split = blockIdx.y;
  const int m_lo = split * g.chunk;
  const int m_hi = min(g.M, m_lo + g.chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wu = wave >> 1, wv = wave & 1;
  const int u0 = ut * BU, v0 = vt * BV;
  const int cq = tid & 31, r0 = tid >> 5;                    // channel quad, first pixel row of the thread
  const int dy = tt.dy[t], dx = tt.dx[t];
  const bool u_ok = u0 + cq * 4 < g.Cu, v_ok = v0 + cq * 4 < g.Cv;

  f32x16 acc[TU][TV];
#pragma unroll
  for (int a = 0; a < TU; ++a)
#pragma unroll
    for (int b = 0; b < TV; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // Two register sets: tile kt + 2 is requested while tile kt is multiplied and leaves for the LDS at the end of
  // step kt + 1 — a whole step (~1000 cycles of MFMAs) more than an L2 / HBM round trip under load, which a
  // one-step distance did not cover.  The pixel coordinates of a thread's rows advance by BK per tile: carried
  // along instead of two integer divisions per row and tile.
  float4 ru[2][NP], rv[2][NP];
  bool ru_ok[2][NP], rv_ok[2][NP];
  int pj[NP], pi[NP], pb[NP];            // (column, row, image) of this thread's pixel rows in the NEXT tile to fetch
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m = min(m_lo + r0 + p * 8, g.M - 1);
    pj[p] = m % g.LW;
    const int tmp = m / g.LW;
    pi[p] = tmp % g.LH, pb[p] = tmp / g.LH;
  }
  auto gload = [&](auto set, int kt) {   // called with kt = 0, 1, 2, ... (the coordinates follow)
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = m_lo + kt * BK + r0 + p * 8;
      const bool ok = (m < m_hi) & u_ok;
      ru[S][p] = *(const float4 *)(U + (size_t)(ok ? m : m_lo) * g.Cu + (u_ok ? u0 + cq * 4 : 0));
      ru_ok[S][p] = ok;
      const int iy = pi[p] * g.vys + dy, ix = pj[p] * g.vxs + dx;
      const bool okv = (m < m_hi) & v_ok & ((unsigned)iy < (unsigned)g.Hv) & ((unsigned)ix < (unsigned)g.Wv);
      const int pix = okv ? (pb[p] * g.Hv + iy) * g.Wv + ix : 0;
      rv[S][p] = *(const float4 *)(V + (size_t)pix * g.Cv + (v_ok ? v0 + cq * 4 : 0));
      rv_ok[S][p] = okv;
      pj[p] += BK;                        // the same rows of the next tile
      while (pj[p] >= g.LW) {
        pj[p] -= g.LW;
        if (++pi[p] >= g.LH) pi[p] = 0, ++pb[p];
      }
    }
  };
  auto sstore = [&](auto set, int buf) {
    constexpr int S = decltype(set)::value;
    unsigned char *ub = lds + buf * 2 * SPLIT * TILE_BYTES, *vb = ub + SPLIT * TILE_BYTES;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = r0 + p * 8;
      const int off = wg_tile_off(row, cq >> 1) + 8 * (cq & 1);
      float4 a = ru[S][p], b = rv[S][p];
      const bool oka = ru_ok[S][p], okb = rv_ok[S][p];
      a.x = oka ? a.x : 0.0f, a.y = oka ? a.y : 0.0f, a.z = oka ? a.z : 0.0f, a.w = oka ? a.w : 0.0f;
      b.x = okb ? b.x : 0.0f, b.y = okb ? b.y : 0.0f, b.z = okb ? b.z : 0.0f, b.w = okb ? b.w : 0.0f;
      if constexpr (SPLIT == 3) {
        uint2 h, m, l;
        split_bf16x3(a, &h, &m, &l);
        *(uint2 *)(ub + off) = h, *(uint2 *)(ub + TILE_BYTES + off) = m, *(uint2 *)(ub + 2 * TILE_BYTES + off) = l;
        split_bf16x3(b, &h, &m, &l);
        *(uint2 *)(vb + off) = h, *(uint2 *)(vb + TILE_BYTES + off) = m, *(uint2 *)(vb + 2 * TILE_BYTES + off) = l;
      } else {
        *(uint2 *)(ub + off) = pack_bf16x4(a);
        *(uint2 *)(vb + off) = pack_bf16x4(b);
      }
    }
  };
  // transposed fragment reads: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its block
  const int grp = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  const int cb = grp & 1, fh = grp >> 1;
  auto frag = [&](const unsigned char *base, int ks, int chan0) {   // chan0: first channel of the 32-block
    bf16x8 out;
    const int c0 = (chan0 + 16 * cb) / 8 + (fp >> 1);
    union { s16x4 s[2]; bf16x8 v; } u;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 16 * ks + 8 * fh + 4 * rr + fq;
      u.s[rr] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4 *)(base + wg_tile_off(row, c0) + 8 * (fp & 1)));
    }
    out = u.v;
    return out;
  };

  const int KT = (m_hi - m_lo + BK - 1) / BK;
  typedef std::integral_constant<int, 0> R0;
  typedef std::integral_constant<int, 1> R1;
  int kt = 0;
  auto step = [&](auto fetch_set, auto store_set) {      // fetch_set: free, its tile (kt) is in the LDS already
    const int buf = kt & 1;
    if (kt + 2 < KT) gload(fetch_set, kt + 2);
    const unsigned char *ub = lds + buf * 2 * SPLIT * TILE_BYTES, *vb = ub + SPLIT * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[SPLIT][TU], bfr[SPLIT][TV];
#pragma unroll
      for (int s = 0; s < SPLIT; ++s) {
#pragma unroll
        for (int a = 0; a < TU; ++a) af[s][a] = frag(ub + s * TILE_BYTES, ks, wu * WU + a * 32);
#pragma unroll
        for (int b = 0; b < TV; ++b) bfr[s][b] = frag(vb + s * TILE_BYTES, ks, wv * WV + b * 32);
      }
#pragma unroll
      for (int a = 0; a < TU; ++a)
#pragma unroll
        for (int b = 0; b < TV; ++b) {
          if constexpr (SPLIT == 3) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bfr[0][b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[2][b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bfr[0][b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[1][b], acc[a][b], 0, 0, 0);
          }
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bfr[0][b], acc[a][b], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) sstore(store_set, buf ^ 1);
    __syncthreads();
    ++kt;
  };
  if (KT > 0) {
    gload(R0(), 0);
    sstore(R0(), 0);
    if (KT > 1) gload(R1(), 1);
    __syncthreads();
    while (kt + 1 < KT) {                 // pairs as one straight-line body (see dconv_gemm_bf16_kernel)
      step(R0(), R1());
      step(R1(), R0());
    }
    if (kt < KT) step(R0(), R1());
  }
  float *dst = part + ((size_t)split * g.T + t) * g.Cu * g.Cv;
  constexpr int LDC = WV + 4;
  const int lr = lane & 31, lh = lane >> 5;
  float *cs = (float *)lds + wave * WU * LDC;
#pragma unroll
  for (int b = 0; b < TV; ++b)
#pragma unroll
    for (int a = 0; a < TU; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cs[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + b * 32 + lr] = acc[a][b][r];
  constexpr int CQ = WV / 4, RPI = 64 / CQ;
  const int ccq = lane % CQ, rr = lane / CQ;
  const int vcol = v0 + wv * WV + ccq * 4;
#pragma unroll 4
  for (int it = 0; it < WU / RPI; ++it) {
    const int rl = it * RPI + rr;
    const int u = u0 + wu * WU + rl;
    if (u < g.Cu && vcol < g.Cv) *(float4 *)(dst + (size_t)u * g.Cv + vcol) = *(const float4 *)(cs + rl * LDC + ccq * 4);
  }
}

// ---- 3 x 3 / stride 1 weight gradient with the nine taps in ONE workgroup ---------------------------------------
// dconv_wgrad_bf16_kernel gives every tap its own workgroups: the same 16 pixels of dY and (shifted) X are fetched
// from L2, split into planes and written to LDS nine times over, for 24 matrix instructions per wave and step.
// Here a workgroup owns a 64 x 64 (dY channels x X channels) tile of ALL nine taps (nine 32 x 32 accumulators per
// wave: 144 registers).  A step is 16 pixels of ONE image row, and a workgroup walks DOWN a column of such segments
// (steps are numbered (image, segment column, row) with the row fastest): the three X rows a step needs — 18 pixels
// each, one halo pixel either side — live in a ring of four row slots in the LDS, so each step fetches, splits and
// stores ONE new X row (row + 2, for the step after next) and the next step's 16 dY pixels: 2.1 float4 per thread for
// 54 matrix instructions per wave (1728 cycles), against 2 float4 per 24 instructions above.  The taps read their
// B fragments from ring slot (row - 1 + t / 3) & 3 at pixel offset t % 3.  A run (the part of a column inside the
// workgroup's step range) starts with three load-only iterations that fill the ring.
// Same planes and transposing LDS reads (ds_read_tr16_b64) as the kernel above; pixel rows of 128 bytes (64 bf16
// channels), the 64-byte halves swapped on every second row pair: the four pixel rows of a transposed read are
// 128 bytes apart (two of them per 256-byte bank span) and a wave reads one 64-byte half of each — without the
// swap rows r and r + 2 hit the same banks (measured: 3.5x on the whole kernel).
#ifndef W9_STORE_AT
#define W9_STORE_AT 5
#endif
__device__ __forceinline__ int wg9_off(int row, int chunk) {     // bytes inside one row slot; chunk = 16-byte piece
  return 128 * row + 16 * (chunk ^ (((row >> 1) & 1) << 2));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void dconv_wgrad9_kernel(const float *__restrict__ U, const float *__restrict__ V, float *__restrict__ part,
                         const DWgradGeom g, int n_tiles_u, int n_tiles_v, int njt, int kt_total, int kt_chunk) {
  constexpr int U_BUF = 16 * 128, V_SLOT = 18 * 128;         // one plane of one dY segment / of one X row
  constexpr int U_PLANE = 2 * U_BUF, V_PLANE = 4 * V_SLOT;
  constexpr int LDS_MAIN = 3 * (U_PLANE + V_PLANE);          // 39 KB
  constexpr int LDS_EPI = 4 * 32 * 36 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI];
  unsigned char *const ubase = lds, *const vbase = lds + 3 * U_PLANE;

  const int tile = blockIdx.x;
  const int vt = tile % n_tiles_v, ut = tile / n_tiles_v;
  const int kt_lo = blockIdx.y * kt_chunk;
  const int kt_hi = min(kt_total, kt_lo + kt_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wu = wave >> 1, wv = wave & 1;
  const int u0 = ut * 64, v0 = vt * 64;
  const int cq = tid & 15, pr = tid >> 4;                    // channel quad (4 channels), pixel 0..15 of the thread
  const bool u_ok = u0 + cq * 4 < g.Cu, v_ok = v0 + cq * 4 < g.Cv;
  const float *const Uc = U + (u_ok ? u0 + cq * 4 : 0), *const Vc = V + (v_ok ? v0 + cq * 4 : 0);
  const int st_off = wg9_off(pr, cq >> 1) + 8 * (cq & 1);                  // where the thread's pixel goes in a slot
  const int st_off2 = wg9_off(16 + (pr & 1), cq >> 1) + 8 * (cq & 1);      // threads 0..31: halo pixels 16, 17

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  float4 ru, rv, rv2;
  bool ru_ok, rv_ok, rv2_ok;
  // iteration r of a run over rows [i0, i1) of column (b, jt): dY row r + 1 and X row r + 2
  auto gload = [&](int b, int jt, int r, int i0, int i1) {
    const int ui = r + 1, j = jt * 16 + pr;
    ru_ok = u_ok & (j < g.LW) & (ui >= i0) & (ui < i1);
    ru = *(const float4 *)(Uc + (ru_ok ? ((size_t)(b * g.LH + ui) * g.LW + j) * g.Cu : 0));
    const int iy = r + 2, ix = jt * 16 - 1 + pr;
    const bool row_ok = v_ok & (iy >= 0) & (iy <= i1) & (iy < g.Hv);
    rv_ok = row_ok & ((unsigned)ix < (unsigned)g.Wv);
    rv = *(const float4 *)(Vc + (rv_ok ? ((size_t)(b * g.Hv + iy) * g.Wv + ix) * g.Cv : 0));
    if (tid < 32) {
      const int ix2 = jt * 16 + 15 + (pr & 1);
      rv2_ok = row_ok & (ix2 < g.Wv);
      rv2 = *(const float4 *)(Vc + (rv2_ok ? ((size_t)(b * g.Hv + iy) * g.Wv + ix2) * g.Cv : 0));
    }
  };
  auto put = [&](unsigned char *dst, int plane_bytes, float4 a, bool ok) {
    a.x = ok ? a.x : 0.0f, a.y = ok ? a.y : 0.0f, a.z = ok ? a.z : 0.0f, a.w = ok ? a.w : 0.0f;
    uint2 h, m, l;
    split_bf16x3(a, &h, &m, &l);
    *(uint2 *)dst = h, *(uint2 *)(dst + plane_bytes) = m, *(uint2 *)(dst + 2 * plane_bytes) = l;
  };
  auto sstore = [&](int r) {
    put(ubase + ((r + 1) & 1) * U_BUF + st_off, U_PLANE, ru, ru_ok);
    unsigned char *vs = vbase + ((r + 2) & 3) * V_SLOT;
    put(vs + st_off, V_PLANE, rv, rv_ok);
    if (tid < 32) put(vs + st_off2, V_PLANE, rv2, rv2_ok);
  };
  // transposed fragment reads (see dconv_wgrad_bf16_kernel): lane 4q+p of a 16-lane group supplies row q, columns
  // 4p..4p+3 of its 4 x 16 block; group = (8-pixel half, 16-channel half).  The lane's byte offsets inside a slot
  // are loop constants (two row groups x three pixel offsets t % 3 — the swizzle depends on the row); the slot's
  // own offset is a scalar that changes with the step.  The sum is formed by ONE instruction right before the reads
  // that use it (inline asm: left to itself the compiler forms all 20 sums of a step ahead of time and spills).
  const int grp = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  const int cb = grp & 1, fh = grp >> 1;
  auto lane_off = [&](int row0, int chan0, int rr) {
    return wg9_off(row0 + 8 * fh + 4 * rr + fq, (chan0 + 16 * cb) / 8 + (fp >> 1)) + 8 * (fp & 1);
  };
  const unsigned lds0 = dm_lds_addr(lds);
  unsigned ua[2], va[3][2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    ua[rr] = lds0 + lane_off(0, wu * 32, rr);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) va[dx][rr] = lds0 + 3 * U_PLANE + lane_off(dx, wv * 32, rr);
  }
  auto frag3 = [&](bf16x8 (&out)[3], const unsigned (&la)[2], unsigned slot_off, auto PLANE) {
    constexpr int plane = decltype(PLANE)::value;
    unsigned a0, a1;
    asm volatile("v_add_u32 %0, %2, %3\n\tv_add_u32 %1, %2, %4" : "=&v"(a0), "=v"(a1) : "s"(slot_off), "v"(la[0]), "v"(la[1]));
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      union { s16x4 s[2]; bf16x8 v; } u;
      u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + sp * plane));
      u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a1 + sp * plane));
      out[sp] = u.v;
    }
  };

  int k = kt_lo;
  while (k < kt_hi) {
    const int col = k / g.LH, i0 = k - col * g.LH;
    const int i1 = min(g.LH, i0 + (kt_hi - k));
    const int b = col / njt, jt = col - b * njt;
#pragma unroll 1
    for (int r = i0 - 3; r < i0; ++r) {                       // fill the ring: X rows i0 - 1, i0, i0 + 1 and dY row i0
      gload(b, jt, r, i0, i1);
      sstore(r);
      __syncthreads();
    }
#pragma unroll 1
    for (int r = i0; r < i1; ++r) {
#ifndef W9_PROBE_NOSTAGE
      gload(b, jt, r, i0, i1);                                // travels underneath the products below
#endif
      {
        bf16x8 af[3], bf[2][3];
        frag3(af, ua, (unsigned)((r & 1) * U_BUF), std::integral_constant<int, U_PLANE>());
        frag3(bf[0], va[0], (unsigned)(((r - 1) & 3) * V_SLOT), std::integral_constant<int, V_PLANE>());
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          // the next tap's fragments are on their way while this tap's six products run
          if (t < 8)
            frag3(bf[(t + 1) & 1], va[(t + 1) % 3], (unsigned)(((r - 1 + (t + 1) / 3) & 3) * V_SLOT),
                  std::integral_constant<int, V_PLANE>());
          const bf16x8 (&bt)[3] = bf[t & 1];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bt[0], acc[t], 0, 0, 0);      // smallest terms first
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bt[2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bt[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bt[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bt[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bt[0], acc[t], 0, 0, 0);
#ifndef W9_PROBE_NOSTAGE
          // the rows fetched at the top of the step have had W9_STORE_AT + 1 taps' worth of products to arrive; their
          // split and LDS stores (slots this step does not read) issue underneath the remaining products
          if (t == W9_STORE_AT) sstore(r);
#endif
        }
      }
      __syncthreads();
    }
    k += i1 - i0;
  }
  // nine 32 x 32 tiles per wave, each through LDS into 16-byte stores (see dconv_gemm_kernel's epilogue)
  constexpr int LDC = 36;
  const int lr = lane & 31, lh = lane >> 5;
  float *cs = (float *)lds + wave * 32 * LDC;
  const int ccq = lane & 7, rr = lane >> 3;
  const int vcol = v0 + wv * 32 + ccq * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float *dst = part + ((size_t)blockIdx.y * 9 + t) * g.Cu * g.Cv;
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + lr] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = it * 8 + rr;
      const int u = u0 + wu * 32 + rl;
      if (u < g.Cu && vcol < g.Cv) *(float4 *)(dst + (size_t)u * g.Cv + vcol) = *(const float4 *)(cs + rl * LDC + ccq * 4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- weight gradient of a tall-skinny linear layer: dW (N x K) = dY^T (N x R) . X (R x K) -------------------------
// The shared 1 x 1 layers over grouped rows (pointnet2_modules.py:31-40): R = 65 k ... 885 k rows, N <= 64 outputs,
// K <= 160 inputs — a reduction over rows that moves R * (N + K) * 4 bytes for 2 R N K flops (20-40 flop per byte):
// memory-bound once the products run at the split-arithmetic rate.  The batched split-K BLAS call it replaces pays two
// launches of >= 13 us each whatever the size and runs the large shapes on the fp32 matrix instruction.
// A workgroup streams a contiguous range of rows in steps of 16: both operands are fetched as they lie in memory
// ([row][channel], 16-byte pieces), split once into three bf16 planes in LDS, and read back transposed
// (ds_read_tr16_b64) as in dconv_wgrad9_kernel; the (N / 32) x (K / 32) blocks of 32 x 32 outputs are dealt
// round-robin to the four waves (<= 3 accumulators per wave), six products per block and step.  LDS rows have a
// pitch of 64 or 192 bytes mod 256 — four consecutive rows then start in four different 64-byte quarters of the
// bank row and a transposed read (4 rows x 2 x 32 bytes per half-wave) is conflict-free without a swizzle.
// Rows are requested two steps ahead (one register set in flight, one being split).
constexpr int TW_MAXK = 160, TW_MAXN = 64;

__host__ __device__ inline int tw_pitch(int channels) {          // bytes per LDS row of one plane
  int p = (channels * 2 + 63) / 64 * 64;
  if ((p & 127) == 0) p += 64;
  return p;
}

template <int MAXP>      // output blocks per wave: ceil((N / 32) * (K / 32) / 4)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void tall_wgrad_kernel(const float *__restrict__ U, const float *__restrict__ V, float *__restrict__ part,
                       long long R, int N, int K, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int up = tw_pitch(N), vp = tw_pitch(K);
  const int u_plane = 16 * up, v_plane = 16 * vp;            // one plane of one stage
  const int stage = 3 * (u_plane + v_plane);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long r_lo = (long long)blockIdx.x * rows_per_wg;
  const long long r_hi = min(R, r_lo + rows_per_wg);
  const int n4 = N / 4, q_row = (N + K) / 4;                 // float4 pieces per row: dY first, then X
  const int pieces = 16 * q_row;

  // the thread's (up to four) pieces of a step: row, source offset inside the step, LDS offset inside a stage
  constexpr int PP = (16 * (TW_MAXN + TW_MAXK) / 4 + 255) / 256;      // 4
  int p_row[PP], p_lds[PP];
  long long p_src[PP];
  bool p_isu[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int e = p * 256 + tid;
    const int row = e / q_row, q = e - row * q_row;
    p_row[p] = e < pieces ? row : -1;
    p_isu[p] = q < n4;
    p_src[p] = p_isu[p] ? (long long)row * N + 4 * q : (long long)row * K + 4 * (q - n4);
    p_lds[p] = p_isu[p] ? row * up + 8 * q : 3 * u_plane + row * vp + 8 * (q - n4);
  }
  float4 rg[2][PP];
  auto gload = [&](auto SET, long long r0) {                 // rows r0 .. r0 + 15 (clamped, masked at the store)
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const bool ok = p_row[p] >= 0 && r0 + p_row[p] < r_hi;
      const float *src = p_isu[p] ? U + r0 * N : V + r0 * K;
      rg[S][p] = *(const float4 *)(ok ? src + p_src[p] : (p_isu[p] ? U : V));
    }
  };
  auto sstore = [&](auto SET, long long r0, int buf) {
    constexpr int S = decltype(SET)::value;
    unsigned char *base = lds + buf * stage;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      if (p_row[p] < 0) continue;
      const bool ok = r0 + p_row[p] < r_hi;
      float4 a = rg[S][p];
      a.x = ok ? a.x : 0.0f, a.y = ok ? a.y : 0.0f, a.z = ok ? a.z : 0.0f, a.w = ok ? a.w : 0.0f;
      uint2 h, m, l;
      split_bf16x3(a, &h, &m, &l);
      unsigned char *dst = base + p_lds[p];
      const int plane = p_isu[p] ? u_plane : v_plane;
      *(uint2 *)dst = h, *(uint2 *)(dst + plane) = m, *(uint2 *)(dst + 2 * plane) = l;
    }
  };
  // output blocks of this wave: pair index wave + 4 j -> (n block, k block)
  // (a wave whose j-th pair does not exist repeats the last pair and drops the result: the product block stays free of
  // branches — a conditional block of matrix instructions inside the loop makes the compiler copy accumulators)
  const int nblk = (N + 31) / 32, kblk = (K + 31) / 32, pairs = nblk * kblk;
  f32x16 acc[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  // transposed fragment reads (see dconv_wgrad_bf16_kernel): lane 4q+p of a 16-lane group supplies row q, columns
  // 4p..4p+3 of its 4 x 16 block; group = (8-row half, 16-channel half)
  const int grp = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  const int cb = grp & 1, fh = grp >> 1;
  auto frag = [&](const unsigned char *plane_base, int pitch, int chan0) {
    union { s16x4 s[2]; bf16x8 v; } u;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 8 * fh + 4 * rr + fq;
      u.s[rr] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4 *)(plane_base + row * pitch + (chan0 + 16 * cb) * 2 + 8 * fp));
    }
    return u.v;
  };
  auto products = [&](int buf) {
    const unsigned char *ub = lds + buf * stage, *vb = ub + 3 * u_plane;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const int pi = min(wave + 4 * j, pairs - 1);
      const int nb = pi % nblk, kb = pi / nblk;
      bf16x8 af[3], bf[3];
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
        af[sp] = frag(ub + sp * u_plane, up, nb * 32);
        bf[sp] = frag(vb + sp * v_plane, vp, kb * 32);
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[j], 0, 0, 0);      // smallest terms first
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[j], 0, 0, 0);
    }
  };

  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  const int steps = (int)((r_hi - r_lo + 15) / 16);
  if (steps > 0) {
    // the LDS rows beyond the channels a plane holds are never written: zero the stages once (fragments of the last,
    // partial 32-channel block read them)
    for (int e = tid; e < 2 * stage / 16; e += 256) ((uint4 *)lds)[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    gload(S0(), r_lo);
    if (steps > 1) gload(S1(), r_lo + 16);
    sstore(S0(), r_lo, 0);
    __syncthreads();
    int st = 0;
    // two steps per trip: register set (st & 1) holds step st + 1 at the top of step st
    for (; st + 2 <= steps; st += 2) {
      if (st + 2 < steps) gload(S0(), r_lo + 16LL * (st + 2));
      products(0);
      sstore(S1(), r_lo + 16LL * (st + 1), 1);
      __syncthreads();
      if (st + 3 < steps) gload(S1(), r_lo + 16LL * (st + 3));
      products(1);
      if (st + 2 < steps) sstore(S0(), r_lo + 16LL * (st + 2), 0);
      __syncthreads();
    }
    if (st < steps) products(0);
  }
  // accumulator block j -> part[wg][n][k] (rows n by register / lane half, columns k by lane)
  float *dst = part + (size_t)blockIdx.x * N * K;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int pi = wave + 4 * j;
    if (pi >= pairs) continue;
    const int nb = pi % nblk, kb = pi / nblk;
    const int k = kb * 32 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < N && k < K) dst[(size_t)n * K + k] = acc[j][r];
    }
  }
}

// out[u*su + v*sv + t*st] = scale_u[u] * sum_s part[s][t][u][v]   (v < Cv_out: drops channel padding)
__global__ __launch_bounds__(256) void dconv_wgrad_reduce_kernel(
    const float *__restrict__ part, float *__restrict__ out, const float *__restrict__ scale_u,
    int nsplit, int T, int Cu, int Cv, int Cv_out, long long su, long long sv, long long st,
    int accumulate) {
  const long long total = (long long)T * Cu * Cv;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int v = (int)(e % Cv);
  const long long r = e / Cv;
  const int u = (int)(r % Cu), t = (int)(r / Cu);
  if (v >= Cv_out) return;
  float s = 0.0f;
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {          // eight loads in flight, summed in split order
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = part[(size_t)(k + j) * total + e];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += p[j];
  }
  for (; k < nsplit; ++k) s += part[(size_t)k * total + e];
  if (scale_u) s *= scale_u[u];
  float *o = out + u * su + v * sv + t * st;
  *o = accumulate ? *o + s : s;
}

// dst[s][n][k] = src[n*sn + k*sk + s*st] * scale_n[n] * scale_k[k]   (k >= Ksrc, n >= Nsrc: zero)
__global__ __launch_bounds__(256) void dconv_pack_kernel(const float *__restrict__ src,
                                                         float *__restrict__ dst,
                                                         const float *__restrict__ scale_n,
                                                         const float *__restrict__ scale_k, int S,
                                                         int N, int K, int Nsrc, int Ksrc,
                                                         long long sn, long long sk, long long st) {
  const long long total = (long long)S * N * K;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int k = (int)(e % K);
  const long long r = e / K;
  const int n = (int)(r % N), s = (int)(r / N);
  float v = 0.0f;
  if (k < Ksrc && n < Nsrc) {
    v = src[n * sn + k * sk + s * st];
    if (scale_n) v *= scale_n[n];
    if (scale_k) v *= scale_k[k];
  }
  dst[e] = v;
}

// The `planes` copy of a packed weight for dconv_patch_gl_kernel: the three bf16 planes of dst[s][n][k] in MFMA
// fragment order, [s][k / 16][n / 32][plane][lane = 32 * ((k % 16) / 8) + n % 32][k % 8] — chunk c of this function
// is one lane's 8 values of one (s, 16-channel block, 32-column block): three 16-byte stores.  N is padded to a
// multiple of 32 with zeros; K % 16 == 0.
__device__ __forceinline__ long long dconv_planes_chunks(int S, int N, int K) {
  return (long long)S * (K / 16) * ((N + 31) / 32) * 64;
}
__device__ __forceinline__ void dconv_pack_planes_chunk(long long c, const float *__restrict__ src,
                                                        unsigned char *__restrict__ planes,
                                                        const float *__restrict__ scale_n,
                                                        const float *__restrict__ scale_k, int N, int K, int Nsrc,
                                                        int Ksrc, long long sn, long long sk, long long st) {
  const int NNB = (N + 31) / 32, NCB = K / 16;
  const int lane = (int)(c & 63);
  long long r = c >> 6;
  const int nb = (int)(r % NNB);
  r /= NNB;
  const int cb = (int)(r % NCB), s = (int)(r / NCB);
  const int n = nb * 32 + (lane & 31), k0 = cb * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    float t = 0.0f;
    if (k < Ksrc && n < Nsrc) {
      t = src[n * sn + k * sk + s * st];
      if (scale_n) t *= scale_n[n];
      if (scale_k) t *= scale_k[k];
    }
    v[j] = t;
  }
  uint2 h0, m0, l0, h1, m1, l1;
  split_bf16x3(make_float4(v[0], v[1], v[2], v[3]), &h0, &m0, &l0);
  split_bf16x3(make_float4(v[4], v[5], v[6], v[7]), &h1, &m1, &l1);
  unsigned char *dst = planes + (((size_t)(s * NCB + cb) * NNB + nb) * 3) * 1024 + lane * 16;
  *(uint4 *)dst = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *(uint4 *)(dst + 1024) = make_uint4(m0.x, m0.y, m1.x, m1.y);
  *(uint4 *)(dst + 2048) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

__global__ __launch_bounds__(256) void dconv_pack_planes_kernel(const float *__restrict__ src,
                                                                unsigned char *__restrict__ planes,
                                                                const float *__restrict__ scale_n,
                                                                const float *__restrict__ scale_k, int S, int N,
                                                                int K, int Nsrc, int Ksrc, long long sn,
                                                                long long sk, long long st) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c < dconv_planes_chunks(S, N, K)) dconv_pack_planes_chunk(c, src, planes, scale_n, scale_k, N, K, Nsrc, Ksrc, sn, sk, st);
}

// the same for a table of weights in one launch (blockIdx.y = table row, grid-stride over its elements):
// every packed weight of a network is refreshed by one launch after an optimizer / EMA step.
struct DConvPackDesc {   // 80 bytes; mirrored by dense_conv._DESC (numpy)
  const float *src;
  float *dst;
  const float *scale_n, *scale_k;
  long long sn, sk, st;
  int S, N, K, Nsrc, Ksrc, pad;
};

__global__ __launch_bounds__(256) void dconv_pack_batch_kernel(const DConvPackDesc *__restrict__ table) {
  const DConvPackDesc d = table[blockIdx.y];
  const long long total = (long long)d.S * d.N * d.K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int k = (int)(e % d.K);
    const long long r = e / d.K;
    const int n = (int)(r % d.N), s = (int)(r / d.N);
    float v = 0.0f;
    if (k < d.Ksrc && n < d.Nsrc) {
      v = d.src[n * d.sn + k * d.sk + s * d.st];
      if (d.scale_n) v *= d.scale_n[n];
      if (d.scale_k) v *= d.scale_k[k];
    }
    d.dst[e] = v;
  }
  if (d.pad == 1) {      // the planes copy behind the fp32 block (dm_dconv_planes_offset_bytes)
    unsigned char *planes = (unsigned char *)(d.dst + total);
    const long long chunks = dconv_planes_chunks(d.S, d.N, d.K);
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (long long)gridDim.x * 256)
      dconv_pack_planes_chunk(c, d.src, planes, d.scale_n, d.scale_k, d.N, d.K, d.Nsrc, d.Ksrc, d.sn, d.sk, d.st);
  }
}

// y[row(m)][n] = relu?(bias[n] + sum_s partial[s][m][n])   (split-K epilogue; fixed summation order)
__global__ __launch_bounds__(256) void dconv_splitk_reduce_kernel(const float *__restrict__ partial,
                                                                  const float *__restrict__ bias,
                                                                  float *__restrict__ y,
                                                                  const DConvGeom g, int nsplit) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long long total = (long long)g.M * g.Cout;
  if (e >= total) return;
  const int m = (int)(e / g.Cout), n = (int)(e % g.Cout);     // Cout % 4 == 0 on this path
  float4 s = *(const float4 *)(partial + e);
  for (int k = 1; k < nsplit; ++k) {
    const float4 v = *(const float4 *)(partial + (size_t)k * total + e);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  if (bias) s.x += bias[n], s.y += bias[n + 1], s.z += bias[n + 2], s.w += bias[n + 3];
  size_t row;
  if (g.dense_out) {
    row = (size_t)m;
  } else {
    const int j = m % g.LW, tmp = m / g.LW, i = tmp % g.LH, bb = tmp / g.LH;
    row = ((size_t)bb * g.Hout + g.oy0 + i * g.oys) * g.Wout + g.ox0 + j * g.oxs;
  }
  if (g.residual != nullptr) {
    const float4 r = *(const float4 *)(g.residual + row * g.Cout + n);
    s.x += r.x, s.y += r.y, s.z += r.z, s.w += r.w;
  }
  if (g.relu == 1) s.x = fmaxf(s.x, 0.f), s.y = fmaxf(s.y, 0.f), s.z = fmaxf(s.z, 0.f), s.w = fmaxf(s.w, 0.f);
  *(float4 *)(y + row * g.Cout + n) = s;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool UNI, int LIMIT = 0, int BF = 0>
int launch_gemm(const float *x, const float *w, const float *bias, float *y, DConvGeom g,
                const DConvTaps &tt, hipStream_t st, int m_lo = 0, int m_hi = -1, int nsplit = 1,
                float *partial = nullptr) {
  if (m_hi < 0) m_hi = g.M;
  if (m_hi <= m_lo) return DM_OK;
  g.M = m_hi;
  const int tm = dm_ceil_div(m_hi - m_lo, BM), tn = dm_ceil_div(g.Cout, BN);
  const int blocks = dm_ceil_div(tm, 8) * 8 * tn;
  const int KT = dm_ceil_div(g.Ktot, BF == 1 ? 64 : (BF == 2 ? 16 : 32));
  const int per = dm_ceil_div(KT, nsplit);
  nsplit = dm_ceil_div(KT, per);
  if constexpr (BF != 0) {
    static_assert(UNI, "the bf16 kernel takes whole-tap K-tiles only");
    dconv_gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, LIMIT, BF == 2 ? 3 : 1>
        <<<dim3(blocks, nsplit), WAVES_M * WAVES_N * 64, 0, st>>>(x, w, bias, y, g, tt, tm, tn, m_lo, per,
                                                                 nsplit > 1 ? partial : nullptr);
  } else {
    dconv_gemm_kernel<BM, BN, 32, WAVES_M, WAVES_N, UNI, LIMIT>
        <<<dim3(blocks, nsplit), WAVES_M * WAVES_N * 64, 0, st>>>(x, w, bias, y, g, tt, tm, tn, m_lo, per,
                                                                 nsplit > 1 ? partial : nullptr);
  }
  DM_CHECK_LAUNCH();
  if (nsplit > 1) {
    dconv_splitk_reduce_kernel<<<dm_ceil_div((long long)g.M * g.Cout / 4, 256), 256, 0, st>>>(
        partial, bias, y, g, nsplit);
    DM_CHECK_LAUNCH();
  }
  return DM_OK;
}

// 0: fp32 on the matrix pipe's own fp32 instruction (v_mfma_f32_32x32x2_f32), 1: bf16 multiplicands, fp32
// accumulate (mixed precision), 2: fp32-class through six bf16 products of the three-way split operands
int g_dconv_math = 0;
int g_dconv_planes = 1;   // the patch kernel takes a layer's `planes` weight copy when the caller hands one (dm_dconv_gemm_planes)
int g_dconv_patch = 1;    // math mode 2: 3 x 3 / stride-1 layers on dconv_patch_split_kernel (dm_dconv_set_math(2 + 16) turns it off)
int g_dconv_wgrad9 = 1;      // math mode 2: 3 x 3 / stride-1 weight gradients on dconv_wgrad9_kernel (developer switch: dm_dconv_set_math(2 + 32) turns it off)

// How many ways the reduction of a SMALL problem (fewer 64x64 tiles than half a round) is split.
static int dconv_gemm_splits(const int *q) {
  const long long M = (long long)q[0] * q[7] * q[8];
  const int Cin = q[3], Cout = q[6], T = q[15];
  if ((Cin % 32) != 0 || Cout <= 32 || (Cout & 3)) return 1;
  const long long tiles = (long long)dm_ceil_div(M, 64) * dm_ceil_div(Cout, 64);
  const int KT = dm_ceil_div((long long)T * Cin, 32);
  if (tiles >= 512 || KT < 16) return 1;
  long long s = 1024 / tiles;
  if (s > KT / 8) s = KT / 8;
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int)s;
}

}  // namespace

extern "C" int dm_dconv_pack(const float *src, float *dst, const float *scale_n,
                             const float *scale_k, int S, int N, int K, int Nsrc, int Ksrc,
                             long long sn, long long sk, long long st, dm_stream_t stream) {
  if (S <= 0 || N <= 0 || K <= 0) return DM_OK;
  if (!src || !dst || Ksrc > K || Nsrc > N) return DM_ERR_INVALID_ARG;
  const long long total = (long long)S * N * K;
  dconv_pack_kernel<<<dm_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
      src, dst, scale_n, scale_k, S, N, K, Nsrc, Ksrc, sn, sk, st);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_dconv_pack_batch(const void *table_dev, int n_entries, int blocks_per_entry,
                                   dm_stream_t stream) {
  static_assert(sizeof(DConvPackDesc) == 80, "descriptor layout is part of the ABI");
  if (n_entries <= 0) return DM_OK;
  if (!table_dev || blocks_per_entry <= 0 || n_entries > 65535) return DM_ERR_INVALID_ARG;
  dconv_pack_batch_kernel<<<dim3(blocks_per_entry, n_entries), 256, 0, (hipStream_t)stream>>>(
      (const DConvPackDesc *)table_dev);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// geom_host: 17 ints {B,Hin,Win,Cin, Hout,Wout,Cout, LH,LW, oy0,ox0,oys,oxs, iys,ixs, T, relu};
// taps_host: 3*T shorts {dy[T], dx[T], wslice[T]}.
extern "C" size_t dm_dconv_gemm_workspace_bytes(const int *geom_host) {
  const int ns = dconv_gemm_splits(geom_host);
  if (ns <= 1) return 0;
  return dm_align((size_t)ns * geom_host[0] * geom_host[7] * geom_host[8] * geom_host[6] * sizeof(float));
}

extern "C" int dm_dconv_gemm(const float *x, const float *w_packed, const float *bias, float *y,
                             const int *geom_host, const short *taps_host, void *workspace,
                             size_t workspace_bytes, dm_stream_t stream) {
  return dm_dconv_gemm_residual(x, w_packed, bias, nullptr, y, geom_host, taps_host, workspace,
                                workspace_bytes, stream);
}

extern "C" int dm_dconv_gemm_residual(const float *x, const float *w_packed, const float *bias,
                                      const float *residual, float *y, const int *geom_host,
                                      const short *taps_host, void *workspace, size_t workspace_bytes,
                                      dm_stream_t stream) {
  return dm_dconv_gemm_planes(x, w_packed, nullptr, bias, residual, y, geom_host, taps_host, workspace,
                              workspace_bytes, stream);
}

extern "C" size_t dm_dconv_planes_bytes(int S, int N, int K) {
  if (S <= 0 || N <= 0 || K <= 0 || (K % 16) != 0) return 0;
  // + one tile of slack: a workgroup whose column tile hangs over N reads (and discards) the blocks behind it
  return (size_t)S * (K / 16) * ((N + 31) / 32) * 3072 + 4 * 3072;
}

extern "C" int dm_dconv_pack_planes(const float *src, void *planes, const float *scale_n, const float *scale_k,
                                    int S, int N, int K, int Nsrc, int Ksrc, long long sn, long long sk,
                                    long long st, dm_stream_t stream) {
  if (S <= 0 || N <= 0 || K <= 0) return DM_OK;
  if (!src || !planes || Ksrc > K || Nsrc > N || (K % 16) != 0) return DM_ERR_INVALID_ARG;
  const long long chunks = (long long)S * (K / 16) * ((N + 31) / 32) * 64;
  dconv_pack_planes_kernel<<<dm_ceil_div(chunks, 256), 256, 0, (hipStream_t)stream>>>(
      src, (unsigned char *)planes, scale_n, scale_k, S, N, K, Nsrc, Ksrc, sn, sk, st);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_dconv_gemm_planes(const float *x, const float *w_packed, const void *w_planes,
                                    const float *bias, const float *residual, float *y, const int *geom_host,
                                    const short *taps_host, void *workspace, size_t workspace_bytes,
                                    dm_stream_t stream) {
  if (!x || !w_packed || !y || !geom_host || !taps_host) return DM_ERR_INVALID_ARG;
  DConvGeom g;
  g.residual = residual;
  const int *q = geom_host;
  g.B = q[0], g.Hin = q[1], g.Win = q[2], g.Cin = q[3];
  g.Hout = q[4], g.Wout = q[5], g.Cout = q[6];
  g.LH = q[7], g.LW = q[8];
  g.oy0 = q[9], g.ox0 = q[10], g.oys = q[11], g.oxs = q[12];
  g.iys = q[13], g.ixs = q[14];
  g.T = q[15], g.relu = q[16];
  if (g.T < 1 || g.T > DCONV_MAX_TAPS || (g.Cin & 3) || g.Cin < 4 || g.Cout < 1)
    return DM_ERR_UNSUPPORTED;
  const long long M = (long long)g.B * g.LH * g.LW;
  if (M == 0) return DM_OK;
  int max_slice = 0;
  for (int t = 0; t < g.T; ++t) max_slice = taps_host[2 * g.T + t] > max_slice ? taps_host[2 * g.T + t] : max_slice;
  // the kernel addresses both operands with 32-bit byte offsets
  if (M < 0 || M > 0x7fffffffLL || (long long)g.B * g.Hin * g.Win * g.Cin * 4 >= 0xffffffffLL ||
      (long long)(max_slice + 1) * g.Cout * g.Cin * 4 >= 0xffffffffLL)
    return DM_ERR_INT32_RANGE;
  g.M = (int)M;
  g.Ktot = g.T * g.Cin;
  g.dense_out = (g.oy0 == 0 && g.ox0 == 0 && g.oys == 1 && g.oxs == 1 && g.LH == g.Hout &&
                 g.LW == g.Wout);
  DConvTaps tt;
  for (int t = 0; t < DCONV_MAX_TAPS; ++t) tt.dy[t] = tt.dx[t] = tt.ws[t] = 0;
  for (int t = 0; t < g.T; ++t) {
    tt.dy[t] = taps_host[t];
    tt.dx[t] = taps_host[g.T + t];
    tt.ws[t] = taps_host[2 * g.T + t];
  }
  hipStream_t st = (hipStream_t)stream;
  const bool uni = (g.Cin % 32) == 0;
  if (!uni) return launch_gemm<128, 64, 2, 2, false>(x, w_packed, bias, y, g, tt, st);
  if (g.Cout <= 32) return launch_gemm<128, 32, 4, 1, true>(x, w_packed, bias, y, g, tt, st);
  // Tile choice is a scheduling problem.  128x128 tiles: 2 workgroups per CU, 512 per "round";
  // 64x64 tiles: 4 per CU, 1024 per round, a quarter of the work each (and measured ~10 % more
  // efficient per round: 4 waves per SIMD cover each other's barrier windows).  Whatever is left
  // after the full rounds would occupy a few CUs for a whole round, so it goes to a second launch
  // of 64x64 tiles limited to 1 (2) workgroups per CU, which spreads <= 256 (512) tiles over the
  // chip.  Problems of less than half a 64x64 round split the reduction over workgroups instead
  // (partial sums in the workspace, fixed-order reduce).
  // 3 x 3 (or fewer taps within one pixel), stride 1, same size: the patch kernel (fp32-class split arithmetic)
  // takes the tiles that fill whole rounds of the chip (two workgroups per CU at BN = 128, three at 64); what is
  // left — the last tile rows of the last image, a contiguous range of output pixels — goes to the lattice GEMM's
  // spread-out 64 x 64 tail below, like the remainder of its own 128 x 128 rounds.
  int m_patch = 0;
  if (g_dconv_math == 2 && g_dconv_patch && residual == nullptr && g.dense_out && g.iys == 1 && g.ixs == 1 &&
      g.Hin == g.LH && g.Win == g.LW && g.T <= 9 && g.T >= 4 && (g.Cin % 32) == 0 && (g.Cout % 4) == 0 &&
      g.Cout >= 64) {
    bool near = true;
    for (int t = 0; t < g.T; ++t) near = near && tt.dy[t] >= -1 && tt.dy[t] <= 1 && tt.dx[t] >= -1 && tt.dx[t] <= 1;
    const int tiles_y = dm_ceil_div(g.LH, 8), tiles_x = dm_ceil_div(g.LW, 16);
    const int tm = g.B * tiles_y * tiles_x;
    // column tile: 128 (two workgroups per CU) unless 64-wide tiles (three per CU, half the work each, the A patch
    // staged twice as often) fill their rounds of the chip clearly better
    int bn = g.Cout >= 128 ? 128 : 64;
    if (bn == 128) {
      const long long w128 = (long long)tm * dm_ceil_div(g.Cout, 128), w64 = (long long)tm * dm_ceil_div(g.Cout, 64);
      const double e128 = (double)w128 / ((double)dm_ceil_div(w128, 512) * 512);
      const double e64 = 0.85 * (double)w64 / ((double)dm_ceil_div(w64, 768) * 768);
      if (w128 < 512 && e64 > e128) bn = 64;
    }
    const int tn = dm_ceil_div(g.Cout, bn);
    const int slots = bn == 128 ? 512 : 768;
    const long long wgs = (long long)tm * tn;
    if (near && wgs >= slots / 2) {
      int tm_main = tm;
      const int rem = (int)(wgs % slots);
      if (wgs > slots && rem != 0 && rem <= slots * 3 / 4) {
        // whole rounds only; cut at a tile-row boundary of the LAST image so that the rest is one pixel range
        const int want = (int)((wgs - rem) / tn);
        const int last0 = (g.B - 1) * tiles_y * tiles_x;
        tm_main = want <= last0 ? tm : last0 + (want - last0) / tiles_x * tiles_x;
        if ((tm_main - last0) / tiles_x * 8 >= g.LH) tm_main = tm;
      }
      const int blocks = dm_ceil_div(tm_main, 8) * 8 * tn;
      const int nnb = dm_ceil_div(g.Cout, 32);
      const unsigned char *wpl = (const unsigned char *)w_planes;
      if (wpl != nullptr && g_dconv_planes && bn == 128)
        dconv_patch_gl_kernel<128, 2, 2><<<blocks, 256, 0, st>>>(x, wpl, bias, y, g, tt, tiles_y, tiles_x, tm_main, tn, nnb);
      else if (wpl != nullptr && g_dconv_planes)
        dconv_patch_gl_kernel<64, 4, 1><<<blocks, 256, 0, st>>>(x, wpl, bias, y, g, tt, tiles_y, tiles_x, tm_main, tn, nnb);
      else if (bn == 128)
        dconv_patch_split_kernel<128, 2, 2><<<blocks, 256, 0, st>>>(x, w_packed, bias, y, g, tt, tiles_y, tiles_x, tm_main, tn);
      else
        dconv_patch_split_kernel<64, 4, 1><<<blocks, 256, 0, st>>>(x, w_packed, bias, y, g, tt, tiles_y, tiles_x, tm_main, tn);
      DM_CHECK_LAUNCH();
      if (tm_main == tm) return DM_OK;
      const int rows_done = (tm_main - (g.B - 1) * tiles_y * tiles_x) / tiles_x * 8;
      m_patch = ((g.B - 1) * g.LH + rows_done) * g.LW;
    }
  }
  const int nsplit = dconv_gemm_splits(geom_host);
  const int bf = (g_dconv_math == 1 && (g.Cin % 64) == 0) ? 1 : (g_dconv_math == 2 ? 2 : 0);
#define DM_LG(BM_, BN_, LIM_, ...)                                                  \
  (bf == 1 ? launch_gemm<BM_, BN_, 2, 2, true, LIM_, 1>(x, w_packed, bias, y, g, tt, st, ##__VA_ARGS__) \
   : bf == 2 ? launch_gemm<BM_, BN_, 2, 2, true, LIM_, 2>(x, w_packed, bias, y, g, tt, st, ##__VA_ARGS__) \
             : launch_gemm<BM_, BN_, 2, 2, true, LIM_, 0>(x, w_packed, bias, y, g, tt, st, ##__VA_ARGS__))
  if (nsplit > 1) {
    if (!workspace || workspace_bytes < (size_t)nsplit * g.M * g.Cout * sizeof(float))
      return DM_ERR_WORKSPACE;
    return DM_LG(64, 64, 0, 0, -1, nsplit, (float *)workspace);
  }
  const int tm128 = dm_ceil_div(g.M, 128), tn128 = dm_ceil_div(g.Cout, 128);
  const long long tiles128 = (long long)tm128 * tn128;
  int m_done = m_patch;
  if (m_patch == 0 && tiles128 >= 512 && g.Cout > 64) {          // at least one full round of the big tile
    const int rem = (int)(tiles128 % 512);
    if (rem == 0 || rem > 384) return DM_LG(128, 128, 0);
    const int mt_main = (int)((tiles128 - rem) / tn128) / 8 * 8;
    const int rc = DM_LG(128, 128, 0, 0, mt_main * 128);
    if (rc != DM_OK) return rc;
    m_done = mt_main * 128;
  }
  // the rest (or everything) in 64x64 tiles: full rounds, then a spread-out tail
  const int tn64 = dm_ceil_div(g.Cout, 64);
  const long long tiles64 = (long long)dm_ceil_div(g.M - m_done, 64) * tn64;
  const long long full = tiles64 / 1024 * 1024;
  const int rem64 = (int)(tiles64 - full);
  int m_main = m_done;
  if (full > 0 && rem64 > 0 && rem64 <= 512) m_main = m_done + (int)(full / tn64) / 8 * 8 * 64;
  else if (rem64 == 0 || rem64 > 512) m_main = g.M;
  if (m_main > m_done) {
    const int rc = DM_LG(64, 64, 0, m_done, m_main);
    if (rc != DM_OK) return rc;
  }
  if (m_main >= g.M) return DM_OK;
  const long long tail = (long long)dm_ceil_div(g.M - m_main, 64) * tn64;
  if (tail <= 256) return DM_LG(64, 64, 1, m_main);
  if (tail <= 512) return DM_LG(64, 64, 2, m_main);
  return DM_LG(64, 64, 0, m_main);
#undef DM_LG
}

extern "C" int dm_dconv_set_math(int mode) {
  if (mode == 2 + 16) {       // developer switch: split arithmetic without the patch kernel (A/B)
    g_dconv_math = 2, g_dconv_patch = 0;
    return DM_OK;
  }
  if (mode == 2 + 32) {       // developer switch: split arithmetic with one workgroup set per tap in the weight gradient (A/B)
    g_dconv_math = 2, g_dconv_wgrad9 = 0;
    return DM_OK;
  }
  if (mode < 0 || mode > 2) return DM_ERR_INVALID_ARG;
  g_dconv_math = mode;
  g_dconv_patch = 1;
  g_dconv_wgrad9 = 1;
  return DM_OK;
}

extern "C" int dm_dconv_get_math(void) { return g_dconv_math; }


// the tap-fused kernel's decomposition: 64 x 64 tiles, steps = 16-pixel segments of image rows, split so that the
// workgroups fill about one round of two per CU
static bool dconv_wgrad9_plan(const int *q, const short *taps, int *njt, int *kt_total, int *ns) {
  const int B = q[0], LH = q[1], LW = q[2], Cu = q[3], Cv = q[4], Hv = q[5], Wv = q[6], T = q[9];
  if (T != 9 || q[7] != 1 || q[8] != 1 || Hv != LH || Wv != LW || Cu < 64 || Cv < 64 || LW < 8) return false;
  if (taps)
    for (int t = 0; t < 9; ++t)
      if (taps[t] != t / 3 - 1 || taps[9 + t] != t % 3 - 1) return false;
  *njt = dm_ceil_div(LW, 16);
  const long long kt = (long long)B * LH * *njt;
  if (kt > 0x7fffffffLL) return false;
  *kt_total = (int)kt;
  const long long tiles = (long long)dm_ceil_div(Cu, 64) * dm_ceil_div(Cv, 64);
  long long want = 512 / tiles;
  if (want > kt / 8) want = kt / 8;
  if (want < 1) want = 1;
  *ns = (int)want;
  // short step ranges (small feature maps with many channel tiles) spend their time on the 9-tap partial sums and
  // the ring fill: the per-tap kernel is faster there — except for 64-channel operands, whose per-tap tiles are small
  if (kt / want < 24 && (Cu > 64 || Cv > 64)) return false;
  return true;
}

static int dconv_wgrad_splits(long long M, int T, int Cu, int Cv) {
  // one full round of workgroups: 2 per CU for the 128x128 tile, 4 per CU for the 64x64 tile
  const bool small = Cu <= 64 || Cv <= 64;
  const int tile = small ? 64 : 128, slots = small ? 1024 : 512;
  const long long tiles = (long long)T * dm_ceil_div(Cu, tile) * dm_ceil_div(Cv, tile);
  long long want = slots / tiles;
  const long long max_split = (M + 255) / 256;         // at least 256 pixels per chunk
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  return (int)want;
}

// geom_host: 10 ints {B, LH, LW, Cu, Cv, Hv, Wv, vys, vxs, T}
extern "C" size_t dm_dconv_wgrad_workspace_bytes(const int *geom_host) {
  const int *q = geom_host;
  const long long M = (long long)q[0] * q[1] * q[2];
  int ns = dconv_wgrad_splits(M, q[9], q[3], q[4]);
  int njt, ktt, ns9;
  if (dconv_wgrad9_plan(q, nullptr, &njt, &ktt, &ns9) && ns9 > ns) ns = ns9;      // whichever kernel the mode picks
  return dm_align((size_t)ns * q[9] * q[3] * q[4] * sizeof(float));
}

extern "C" int dm_dconv_wgrad(const float *U, const float *V, float *out, const float *scale_u,
                              const int *geom_host, const short *taps_host, int Cv_out,
                              long long su, long long sv, long long st_, int accumulate,
                              void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  if (!U || !V || !out || !geom_host || !taps_host) return DM_ERR_INVALID_ARG;
  const int *q = geom_host;
  DWgradGeom g;
  g.B = q[0], g.LH = q[1], g.LW = q[2], g.Cu = q[3], g.Cv = q[4], g.Hv = q[5], g.Wv = q[6];
  g.vys = q[7], g.vxs = q[8], g.T = q[9];
  if (g.T < 1 || g.T > DCONV_MAX_TAPS || (g.Cu & 3) || (g.Cv & 3) || Cv_out > g.Cv)
    return DM_ERR_UNSUPPORTED;
  const long long M = (long long)g.B * g.LH * g.LW;
  if (M > 0x7fffffffLL || (long long)g.B * g.Hv * g.Wv * g.Cv > 0x7fffffffLL)
    return DM_ERR_INT32_RANGE;
  g.M = (int)M;
  if (workspace_bytes < dm_dconv_wgrad_workspace_bytes(geom_host) || !workspace)
    return DM_ERR_WORKSPACE;
  int ns = dconv_wgrad_splits(M, g.T, g.Cu, g.Cv);
  g.chunk = (int)(((M + ns - 1) / ns + 31) / 32 * 32);
  hipStream_t st = (hipStream_t)stream;
  float *part = (float *)workspace;
  {
    int njt, ktt, ns9;
    if (g_dconv_math == 2 && g_dconv_wgrad9 && dconv_wgrad9_plan(q, taps_host, &njt, &ktt, &ns9)) {
      const int tu = dm_ceil_div(g.Cu, 64), tv = dm_ceil_div(g.Cv, 64);
      const int chunk9 = dm_ceil_div(ktt, ns9);
      ns = dm_ceil_div(ktt, chunk9);
      dconv_wgrad9_kernel<<<dim3(tu * tv, ns), 256, 0, st>>>(U, V, part, g, tu, tv, njt, ktt, chunk9);
      DM_CHECK_LAUNCH();
      const long long total = (long long)g.T * g.Cu * g.Cv;
      dconv_wgrad_reduce_kernel<<<dm_ceil_div(total, 256), 256, 0, st>>>(
          part, out, scale_u, ns, g.T, g.Cu, g.Cv, Cv_out, su, sv, st_, accumulate);
      DM_CHECK_LAUNCH();
      return DM_OK;
    }
  }
  DConvTaps tt;
  for (int t = 0; t < DCONV_MAX_TAPS; ++t) tt.dy[t] = tt.dx[t] = tt.ws[t] = 0;
  for (int t = 0; t < g.T; ++t) tt.dy[t] = taps_host[t], tt.dx[t] = taps_host[g.T + t];
  if (g.Cu <= 64 || g.Cv <= 64) {
    const int tu = dm_ceil_div(g.Cu, 64), tv = dm_ceil_div(g.Cv, 64);
    dconv_wgrad_kernel<64, 64, 32, 2, 2><<<dim3(g.T * tu * tv, ns), 256, 0, st>>>(U, V, part, g, tt,
                                                                                 tu, tv);
  } else if (g_dconv_math == 1) {      // mixed precision: bf16 multiplicands, fp32 accumulate
    const int tu = dm_ceil_div(g.Cu, 128), tv = dm_ceil_div(g.Cv, 128);
    dconv_wgrad_bf16_kernel<1><<<dim3(g.T * tu * tv, ns), 256, 0, st>>>(U, V, part, g, tt, tu, tv);
  } else if (g_dconv_math == 2) {      // fp32-class: six bf16 products of the three-way split operands
    const int tu = dm_ceil_div(g.Cu, 128), tv = dm_ceil_div(g.Cv, 128);
    dconv_wgrad_bf16_kernel<3><<<dim3(g.T * tu * tv, ns), 256, 0, st>>>(U, V, part, g, tt, tu, tv);
  } else {
    const int tu = dm_ceil_div(g.Cu, 128), tv = dm_ceil_div(g.Cv, 128);
    dconv_wgrad_kernel<128, 128, 32, 2, 2><<<dim3(g.T * tu * tv, ns), 256, 0, st>>>(U, V, part, g,
                                                                                   tt, tu, tv);
  }
  DM_CHECK_LAUNCH();
  const long long total = (long long)g.T * g.Cu * g.Cv;
  dconv_wgrad_reduce_kernel<<<dm_ceil_div(total, 256), 256, 0, st>>>(
      part, out, scale_u, ns, g.T, g.Cu, g.Cv, Cv_out, su, sv, st_, accumulate);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// dW (n, k) = dY^T . X over `rows` rows — the weight gradient of the tall-skinny linears (torch: a batched split-K
// `bmm` + `sum`); fp32-class split arithmetic, bitwise reproducible (fixed row ranges, fixed-order reduce).
static int tall_wgrad_wgs(long long rows) {
  long long w = rows / 256;                      // >= 16 steps of 16 rows per workgroup
  if (w > 1024) w = 1024;
  if (w < 1) w = 1;
  return (int)w;
}

extern "C" int dm_tall_wgrad_supported(int n, int k) {
  return n >= 4 && n <= TW_MAXN && (n & 3) == 0 && k >= 4 && k <= TW_MAXK && (k & 3) == 0;
}

extern "C" size_t dm_tall_wgrad_workspace_bytes(long long rows, int n, int k) {
  if (rows <= 0 || !dm_tall_wgrad_supported(n, k)) return 0;
  return dm_align((size_t)tall_wgrad_wgs(rows) * n * k * sizeof(float));
}

extern "C" int dm_tall_wgrad(const float *dy, const float *x, float *dw, long long rows, int n, int k,
                             int accumulate, void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  if (!dm_tall_wgrad_supported(n, k)) return DM_ERR_UNSUPPORTED;
  if (rows < 0 || rows > 0x7fffffffLL * 8) return DM_ERR_INT32_RANGE;
  if (!dy || !x || !dw) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate) DM_HIP(hipMemsetAsync(dw, 0, (size_t)n * k * sizeof(float), st));
    return DM_OK;
  }
  if (!workspace || workspace_bytes < dm_tall_wgrad_workspace_bytes(rows, n, k)) return DM_ERR_WORKSPACE;
  const int wgs = tall_wgrad_wgs(rows);
  long long per = (rows + wgs - 1) / wgs;
  per = (per + 15) / 16 * 16;
  const int grid = (int)((rows + per - 1) / per);
  const size_t smem = 2 * 3 * 16 * (size_t)(tw_pitch(n) + tw_pitch(k));
  const int pairs = dm_ceil_div(n, 32) * dm_ceil_div(k, 32);
  if (pairs <= 4) tall_wgrad_kernel<1><<<grid, 256, smem, st>>>(dy, x, (float *)workspace, rows, n, k, (int)per);
  else if (pairs <= 8) tall_wgrad_kernel<2><<<grid, 256, smem, st>>>(dy, x, (float *)workspace, rows, n, k, (int)per);
  else tall_wgrad_kernel<3><<<grid, 256, smem, st>>>(dy, x, (float *)workspace, rows, n, k, (int)per);
  DM_CHECK_LAUNCH();
  const long long total = (long long)n * k;
  dconv_wgrad_reduce_kernel<<<dm_ceil_div(total, 256), 256, 0, st>>>((const float *)workspace, dw, nullptr, grid, 1, n, k, k,
                                                                      k, 1, 1, accumulate);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
