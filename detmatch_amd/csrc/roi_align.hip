// RoIAlign (avg pool, aligned / legacy, fixed or adaptive sampling grid) over an FPN pyramid.
//
// Replaces mmcv.ops.RoIAlign (mmcv-full 1.3.16, un-vendored; restated from its published
// algorithm == Detectron2 ROIAlign) as used by SingleRoIExtractor in
// configs/detmatch/001/detmatch/split_0.py:76-80 and processors_2d.py:52-54.  mmcv issues one
// launch per pyramid level on a boolean-masked RoI subset (nonzero() -> host sync, 4 launches
// + index_put per call); here every RoI carries its level and ONE launch serves the pyramid,
// writing rows in RoI order.
//
// Work split: one workgroup per RoI.  The per-RoI sampling geometry (<= 7x7 bins x gh x gw
// samples: 4 corner offsets + 4 bilinear weights) depends only on the RoI, not on the channel,
// so it is computed once into LDS and reused by all C channels; the channel loop then only does
// loads + FMAs.  HBM-bound: algorithmic bytes = R*C*49*4 (out) + the touched feature pixels.
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

constexpr int kMaxLevels = 8;

struct Pyramid {
  const float *feat[kMaxLevels];
  float *grad[kMaxLevels];
  int h[kMaxLevels], w[kMaxLevels];
  float scale[kMaxLevels];
};

struct RoiGeom {
  float start_w, start_h, bin_w, bin_h;
  int gw, gh, level, batch;
};

__device__ inline RoiGeom roi_geom(const float *roi, int level, float scale, int ph, int pw,
                                   int sampling_ratio, int aligned) {
  RoiGeom g;
  g.level = level;
  g.batch = (int)roi[0];
  const float off = aligned ? 0.5f : 0.0f;
  g.start_w = roi[1] * scale - off;
  g.start_h = roi[2] * scale - off;
  const float end_w = roi[3] * scale - off;
  const float end_h = roi[4] * scale - off;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  if (!aligned) {  // legacy: force >= 1x1
    rw = fmaxf(rw, 1.0f);
    rh = fmaxf(rh, 1.0f);
  }
  g.bin_h = rh / (float)ph;
  g.bin_w = rw / (float)pw;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
  return g;
}

// 1-D bilinear tap: position -> (low index, high index, low weight, high weight), valid flag
__device__ inline bool tap(float p, int size, int &lo, int &hi, float &wl, float &wh) {
  if (p < -1.0f || p > (float)size) return false;
  if (p <= 0.0f) p = 0.0f;
  lo = (int)p;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    p = (float)lo;
  } else {
    hi = lo + 1;
  }
  const float l = p - (float)lo;
  wl = 1.0f - l;
  wh = l;
  return true;
}

// Each RoI: y taps for ph*gh sample rows, x taps for pw*gw sample columns (separable).
template <bool BACKWARD>
__global__ __launch_bounds__(256) void roi_align_kernel(Pyramid pyr, const float *__restrict__ rois,
                                                        const int *__restrict__ levels, int C,
                                                        int ph, int pw, int sampling_ratio,
                                                        int aligned, int gmax,
                                                        float *__restrict__ out,
                                                        const float *__restrict__ gout) {
  extern __shared__ unsigned char smem[];
  const int r = blockIdx.x;
  const int level = levels ? levels[r] : 0;
  const int H = pyr.h[level], W = pyr.w[level];
  const RoiGeom g = roi_geom(rois + (size_t)r * 5, level, pyr.scale[level], ph, pw, sampling_ratio,
                             aligned);
  const float inv_count = 1.0f / (float)max(g.gh * g.gw, 1);
  const int bins = ph * pw;
  const size_t plane = (size_t)H * W;
  const float *fbase = BACKWARD ? nullptr : pyr.feat[level] + (size_t)g.batch * C * plane;
  float *gbase = BACKWARD ? pyr.grad[level] + (size_t)g.batch * C * plane : nullptr;
  if (g.gh > gmax || g.gw > gmax) {
    // RoI far larger than its pyramid level was sized for: taps computed on the fly (rare)
    for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
      const int c = o / bins, b = o % bins, by = b / pw, bx = b % pw;
      const size_t oidx = ((size_t)r * C + c) * bins + b;
      const float gv = BACKWARD ? gout[oidx] * inv_count : 0.0f;
      float acc = 0.0f;
      for (int sy = 0; sy < g.gh; ++sy) {
        const float y = g.start_h + (float)by * g.bin_h + ((float)sy + 0.5f) * g.bin_h / (float)g.gh;
        int yl, yh;
        float wyl, wyh;
        if (!tap(y, H, yl, yh, wyl, wyh)) continue;
        for (int sx = 0; sx < g.gw; ++sx) {
          const float x = g.start_w + (float)bx * g.bin_w + ((float)sx + 0.5f) * g.bin_w / (float)g.gw;
          int xl, xh;
          float wxl, wxh;
          if (!tap(x, W, xl, xh, wxl, wxh)) continue;
          if (BACKWARD) {
            float *gp = gbase + (size_t)c * plane;
            unsafeAtomicAdd(gp + (size_t)yl * W + xl, gv * (wyl * wxl));
            unsafeAtomicAdd(gp + (size_t)yl * W + xh, gv * (wyl * wxh));
            unsafeAtomicAdd(gp + (size_t)yh * W + xl, gv * (wyh * wxl));
            unsafeAtomicAdd(gp + (size_t)yh * W + xh, gv * (wyh * wxh));
          } else {
            const float *fp = fbase + (size_t)c * plane;
            acc += (wyl * wxl) * fp[(size_t)yl * W + xl] + (wyl * wxh) * fp[(size_t)yl * W + xh] +
                   (wyh * wxl) * fp[(size_t)yh * W + xl] + (wyh * wxh) * fp[(size_t)yh * W + xh];
          }
        }
      }
      if (!BACKWARD) out[oidx] = acc * inv_count;
    }
    return;
  }
  const int ny = ph * g.gh, nx = pw * g.gw;
  // LDS layout: ylo[ny] yhi[ny] xlo[nx] xhi[nx] (int) | ywl[ny] ywh[ny] xwl[nx] xwh[nx] (float)
  int *ylo = (int *)smem, *yhi = ylo + ny, *xlo = yhi + ny, *xhi = xlo + nx;
  float *ywl = (float *)(xhi + nx), *ywh = ywl + ny, *xwl = ywh + ny, *xwh = xwl + nx;
  for (int i = threadIdx.x; i < ny + nx; i += blockDim.x) {
    if (i < ny) {
      const int b = i / g.gh, s = i % g.gh;
      const float y = g.start_h + (float)b * g.bin_h + ((float)s + 0.5f) * g.bin_h / (float)g.gh;
      int lo = 0, hi = 0;
      float wl = 0.f, wh = 0.f;
      if (!tap(y, H, lo, hi, wl, wh)) lo = -1;
      ylo[i] = lo, yhi[i] = hi, ywl[i] = wl, ywh[i] = wh;
    } else {
      const int j = i - ny, b = j / g.gw, s = j % g.gw;
      const float x = g.start_w + (float)b * g.bin_w + ((float)s + 0.5f) * g.bin_w / (float)g.gw;
      int lo = 0, hi = 0;
      float wl = 0.f, wh = 0.f;
      if (!tap(x, W, lo, hi, wl, wh)) lo = -1;
      xlo[j] = lo, xhi[j] = hi, xwl[j] = wl, xwh[j] = wh;
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
    const int c = o / bins, b = o % bins, by = b / pw, bx = b % pw;
    const size_t oidx = ((size_t)r * C + c) * bins + b;
    if (BACKWARD) {
      const float gv = gout[oidx] * inv_count;
      float *gp = gbase + (size_t)c * plane;
      for (int iy = by * g.gh; iy < (by + 1) * g.gh; ++iy) {
        if (ylo[iy] < 0) continue;
        for (int ix = bx * g.gw; ix < (bx + 1) * g.gw; ++ix) {
          if (xlo[ix] < 0) continue;
          unsafeAtomicAdd(gp + (size_t)ylo[iy] * W + xlo[ix], gv * (ywl[iy] * xwl[ix]));
          unsafeAtomicAdd(gp + (size_t)ylo[iy] * W + xhi[ix], gv * (ywl[iy] * xwh[ix]));
          unsafeAtomicAdd(gp + (size_t)yhi[iy] * W + xlo[ix], gv * (ywh[iy] * xwl[ix]));
          unsafeAtomicAdd(gp + (size_t)yhi[iy] * W + xhi[ix], gv * (ywh[iy] * xwh[ix]));
        }
      }
    } else {
      const float *fp = fbase + (size_t)c * plane;
      float acc = 0.0f;
      for (int iy = by * g.gh; iy < (by + 1) * g.gh; ++iy) {
        if (ylo[iy] < 0) continue;
        const float *r0 = fp + (size_t)ylo[iy] * W, *r1 = fp + (size_t)yhi[iy] * W;
        for (int ix = bx * g.gw; ix < (bx + 1) * g.gw; ++ix) {
          if (xlo[ix] < 0) continue;
          // same association as the published kernel: w1*v1 + w2*v2 + w3*v3 + w4*v4
          const float w1 = ywl[iy] * xwl[ix], w2 = ywl[iy] * xwh[ix];
          const float w3 = ywh[iy] * xwl[ix], w4 = ywh[iy] * xwh[ix];
          acc += w1 * r0[xlo[ix]] + w2 * r0[xhi[ix]] + w3 * r1[xlo[ix]] + w4 * r1[xhi[ix]];
        }
      }
      out[oidx] = acc * inv_count;
    }
  }
}

// ---- backward, separable form, NHWC accumulation -------------------------------------------------
// d feat[y][x][c] = (1/count) * sum_by sum_bx Wy[by][y] * G[by][bx][c] * Wx[bx][x], where Wy / Wx
// collect the bilinear tap weights of every sample of a bin row / column.  One workgroup per RoI:
// the tap tables (7 x patch) and the RoI's output gradient (bins x C) sit in LDS, every pixel of
// the RoI's patch then receives ONE atomic per channel, issued as 256-byte contiguous wave
// instructions into an NHWC buffer — the shape float atomics need to run at the memory-side rate
// (MI355X_MICROARCH.md "Global float atomics"); the per-sample NCHW form above issues
// 4 * samples atomics per bin with 64 lanes in 64 different rows (~17x slower per byte).
constexpr int RA_MAX_SPAN = 160;   // patch rows / columns kept in LDS; larger RoIs -> NCHW path

// Tap tables of one RoI (all 256 threads of the workgroup call this; ends with a barrier):
//   wy[by][yy] / wx[bx][xx]  summed bilinear weights of bin row by / column bx on patch row yy / column xx
//   rng[0..3][span]          per patch row / column: first and last bin with a non-zero weight
//   rng[4*span + 0..63]      per bin row / column: first and last patch row / column it touches
__device__ inline void build_tap_tables(const RoiGeom &g, int H, int W, int y0, int x0, int py, int px,
                                        int ph, int pw, float *wy, float *wx, short *rng) {
  for (int i = threadIdx.x; i < (ph + pw) * RA_MAX_SPAN; i += 256) wy[i] = 0.0f;   // wx follows wy
  __syncthreads();
  short *pix = rng + 4 * RA_MAX_SPAN;      // [yp_lo 16][yp_hi 16][xp_lo 16][xp_hi 16]
  if (threadIdx.x < ph + pw) {             // one thread per bin row / column: sequential, no atomics
    const bool is_y = threadIdx.x < ph;
    const int b = is_y ? threadIdx.x : threadIdx.x - ph;
    const int gn = is_y ? g.gh : g.gw, size = is_y ? H : W, base = is_y ? y0 : x0;
    const float start = is_y ? g.start_h : g.start_w, bin = is_y ? g.bin_h : g.bin_w;
    float *row = (is_y ? wy : wx) + b * RA_MAX_SPAN;
    int first = 32767, last = -1;
    for (int s = 0; s < gn; ++s) {
      const float p = start + (float)b * bin + ((float)s + 0.5f) * bin / (float)gn;
      int lo, hi;
      float wl, wh;
      if (!tap(p, size, lo, hi, wl, wh)) continue;
      row[lo - base] += wl;
      row[hi - base] += wh;
      first = min(first, lo - base);
      last = max(last, hi - base);
    }
    pix[(is_y ? 0 : 32) + b] = (short)first;
    pix[(is_y ? 16 : 48) + b] = (short)last;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < py + px; i += 256) {   // per patch row / column: bins that reach it
    const bool is_y = i < py;
    const int p = is_y ? i : i - py, nb = is_y ? ph : pw;
    const float *tab = is_y ? wy : wx;
    int first = 32767, last = -1;
    for (int b = 0; b < nb; ++b)
      if (tab[b * RA_MAX_SPAN + p] != 0.0f) {
        first = min(first, b);
        last = b;
      }
    rng[(is_y ? 0 : 2 * RA_MAX_SPAN) + p] = (short)first;
    rng[(is_y ? RA_MAX_SPAN : 3 * RA_MAX_SPAN) + p] = (short)last;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void roi_align_bwd_nhwc(Pyramid pyr, const float *__restrict__ rois,
                                                          const int *__restrict__ levels, int C,
                                                          int ph, int pw, int sampling_ratio,
                                                          int aligned,
                                                          const float *__restrict__ gout) {
  extern __shared__ unsigned char smem[];
  const int r = blockIdx.x;
  const int level = levels ? levels[r] : 0;
  const int H = pyr.h[level], W = pyr.w[level];
  const RoiGeom g = roi_geom(rois + (size_t)r * 5, level, pyr.scale[level], ph, pw, sampling_ratio,
                             aligned);
  // pixel span touched by the RoI (taps clamp into [0, size-1])
  const float end_h = g.start_h + g.bin_h * (float)ph, end_w = g.start_w + g.bin_w * (float)pw;
  int y0 = max(0, min(H - 1, (int)floorf(fminf(g.start_h, end_h)))), y1 = max(0, min(H - 1, (int)floorf(fmaxf(g.start_h, end_h)) + 1));
  int x0 = max(0, min(W - 1, (int)floorf(fminf(g.start_w, end_w)))), x1 = max(0, min(W - 1, (int)floorf(fmaxf(g.start_w, end_w)) + 1));
  const int py = y1 - y0 + 1, px = x1 - x0 + 1;
  if (g.gh <= 0 || g.gw <= 0) return;   // empty RoI: no samples
  const int bins = ph * pw;
  const float inv_count = 1.0f / (float)max(g.gh * g.gw, 1);
  float *dst = pyr.grad[level] + (size_t)g.batch * H * W * C;
  if (py > RA_MAX_SPAN || px > RA_MAX_SPAN) {
    // RoI wider than the LDS tap tables: per-sample form (still channel-contiguous atomics)
    for (int c = threadIdx.x; c < C; c += 256) {
      for (int b = 0; b < bins; ++b) {
        const int by = b / pw, bx = b - by * pw;
        const float gv = gout[((size_t)r * C + c) * bins + b] * inv_count;
        for (int sy = 0; sy < g.gh; ++sy) {
          const float y = g.start_h + (float)by * g.bin_h + ((float)sy + 0.5f) * g.bin_h / (float)g.gh;
          int yl, yh;
          float wyl, wyh;
          if (!tap(y, H, yl, yh, wyl, wyh)) continue;
          for (int sx = 0; sx < g.gw; ++sx) {
            const float x = g.start_w + (float)bx * g.bin_w + ((float)sx + 0.5f) * g.bin_w / (float)g.gw;
            int xl, xh;
            float wxl, wxh;
            if (!tap(x, W, xl, xh, wxl, wxh)) continue;
            unsafeAtomicAdd(dst + ((size_t)yl * W + xl) * C + c, gv * (wyl * wxl));
            unsafeAtomicAdd(dst + ((size_t)yl * W + xh) * C + c, gv * (wyl * wxh));
            unsafeAtomicAdd(dst + ((size_t)yh * W + xl) * C + c, gv * (wyh * wxl));
            unsafeAtomicAdd(dst + ((size_t)yh * W + xh) * C + c, gv * (wyh * wxh));
          }
        }
      }
    }
    return;
  }
  float *wy = (float *)smem;                 // [ph][RA_MAX_SPAN]
  float *wx = wy + ph * RA_MAX_SPAN;         // [pw][RA_MAX_SPAN]
  short *rng = (short *)(wx + pw * RA_MAX_SPAN);   // bin / pixel ranges, see build_tap_tables
  float *gt = (float *)(rng + 4 * RA_MAX_SPAN + 4 * 16);   // [ph*pw][C + 1]
  const int CP = C + 1;      // odd row pitch: the transposed stores below (bins run fastest over the lanes) spread over the banks
  build_tap_tables(g, H, W, y0, x0, py, px, ph, pw, wy, wx, rng);
  const short *yb_lo = rng, *yb_hi = rng + RA_MAX_SPAN, *xb_lo = rng + 2 * RA_MAX_SPAN,
              *xb_hi = rng + 3 * RA_MAX_SPAN;
  const float *gsrc = gout + (size_t)r * C * bins;
  for (int e = threadIdx.x; e < C * bins; e += 256) {   // coalesced read, transposed LDS write
    const int c = e / bins, b = e - c * bins;
    gt[b * CP + c] = gsrc[e] * inv_count;   // (pitch C: every lane of a bin run on ONE bank — 58 % of the LDS cycles were conflicts)
  }
  __syncthreads();
  // A wave owns every 4th pixel row, a lane the channels lane, lane+64, lane+128, lane+192 of a
  // 256-channel slab: the range look-ups, the tap weights and the loop control are paid once per four
  // channels, the LDS reads stay conflict free and every atomic instruction covers 256 contiguous bytes.
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int yy = wave; yy < py; yy += 4) {
    const int b0 = yb_lo[yy], b1 = yb_hi[yy];
    if (b0 > b1) continue;
    for (int xx = 0; xx < px; ++xx) {
      const int a0 = xb_lo[xx], a1 = xb_hi[xx];
      if (a0 > a1) continue;
      float *pix = dst + ((size_t)(y0 + yy) * W + (x0 + xx)) * C;
      for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + lane;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int by = b0; by <= b1; ++by) {
          const float a = wy[by * RA_MAX_SPAN + yy];
          for (int bx = a0; bx <= a1; ++bx) {
            const float w = a * wx[bx * RA_MAX_SPAN + xx];
            const float *gp = gt + (by * pw + bx) * CP + c;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c + 64 * k < C) acc[k] += w * gp[64 * k];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + 64 * k < C) unsafeAtomicAdd(pix + c + 64 * k, acc[k]);
      }
    }
  }
}

// ---- forward, separable form, NHWC input ---------------------------------------------------------
// out[by][bx][c] = (1/count) sum_y sum_x Wy[by][y] Wx[bx][x] F[y][x][c]; thread = channel, so every
// feature load is a 256-byte-contiguous wave instruction and a bin reads only the pixels its taps
// touch; the (bins x C) tile is transposed through LDS into the (R, C, ph, pw) output.
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(Pyramid pyr, const float *__restrict__ rois,
                                                          const int *__restrict__ levels, int C,
                                                          int ph, int pw, int sampling_ratio,
                                                          int aligned, float *__restrict__ out) {
  extern __shared__ unsigned char smem[];
  const int r = blockIdx.x;
  const int level = levels ? levels[r] : 0;
  const int H = pyr.h[level], W = pyr.w[level];
  const RoiGeom g = roi_geom(rois + (size_t)r * 5, level, pyr.scale[level], ph, pw, sampling_ratio,
                             aligned);
  const float end_h = g.start_h + g.bin_h * (float)ph, end_w = g.start_w + g.bin_w * (float)pw;
  const int y0 = max(0, min(H - 1, (int)floorf(fminf(g.start_h, end_h)))), y1 = max(0, min(H - 1, (int)floorf(fmaxf(g.start_h, end_h)) + 1));
  const int x0 = max(0, min(W - 1, (int)floorf(fminf(g.start_w, end_w)))), x1 = max(0, min(W - 1, (int)floorf(fmaxf(g.start_w, end_w)) + 1));
  const int py = y1 - y0 + 1, px = x1 - x0 + 1;
  const int bins = ph * pw;
  float *dst = out + (size_t)r * C * bins;
  if (g.gh <= 0 || g.gw <= 0) {            // empty RoI: no samples -> zeros
    for (int e = threadIdx.x; e < C * bins; e += 256) dst[e] = 0.0f;
    return;
  }
  const float inv_count = 1.0f / (float)max(g.gh * g.gw, 1);
  const float *src = pyr.feat[level] + (size_t)g.batch * H * W * C;
  float *wy = (float *)smem;
  float *wx = wy + ph * RA_MAX_SPAN;
  short *rng = (short *)(wx + pw * RA_MAX_SPAN);
  float *ot = (float *)(rng + 4 * RA_MAX_SPAN + 4 * 16);   // [ph*pw][C + 1]
  const bool wide = py > RA_MAX_SPAN || px > RA_MAX_SPAN;
  if (!wide) build_tap_tables(g, H, W, y0, x0, py, px, ph, pw, wy, wx, rng);
  const short *yp_lo = rng + 4 * RA_MAX_SPAN, *yp_hi = yp_lo + 16, *xp_lo = yp_lo + 32, *xp_hi = yp_lo + 48;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!wide && (C & 3) == 0) {
    // a wave owns every 4th bin, a lane 4 consecutive channels: 16-byte loads, 1 KiB per wave-load
    for (int b = wave; b < bins; b += 4) {
      const int by = b / pw, bx = b - by * pw;
      const int ya = yp_lo[by], yb = yp_hi[by], xa = xp_lo[bx], xb = xp_hi[bx];
      for (int cg = lane; cg < C / 4; cg += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xb - xa < 4) {
          // (the usual case — an RoI of ~14 pixels at its level has bins of 2-3 pixels: four columns instead of
          // eight halves the clamped duplicate loads: 299 -> 286 us for 1024 RoIs)
          for (int yy = ya; yy <= yb; yy += 2) {
            const bool two = yy + 1 <= yb;
            const float a0 = wy[by * RA_MAX_SPAN + yy], a1 = two ? wy[by * RA_MAX_SPAN + yy + 1] : 0.f;
            const float4 *row0 = (const float4 *)(src + ((size_t)(y0 + yy) * W + x0) * C) + cg;
            const float4 *row1 = two ? row0 + (size_t)W * (C / 4) : row0;
            float4 f0[4], f1[4];
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int xs = xa + u <= xb ? xa + u : xb;
              f0[u] = row0[(size_t)xs * (C / 4)];
              f1[u] = row1[(size_t)xs * (C / 4)];
              wv[u] = wx[bx * RA_MAX_SPAN + xs];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (xa + u <= xb) {
                const float w = a0 * wv[u];
                acc.x += w * f0[u].x, acc.y += w * f0[u].y, acc.z += w * f0[u].z, acc.w += w * f0[u].w;
              }
            if (two) {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (xa + u <= xb) {
                  const float w = a1 * wv[u];
                  acc.x += w * f1[u].x, acc.y += w * f1[u].y, acc.z += w * f1[u].z, acc.w += w * f1[u].w;
                }
            }
          }
        } else if (xb - xa < 8) {
          // the taps of two pixel rows (up to 16 loads of 16 bytes per lane) are requested before the
          // first is used: one tap per dependent load left the kernel waiting on L2 latency; the sum
          // keeps the row-major tap order
          for (int yy = ya; yy <= yb; yy += 2) {
            const bool two = yy + 1 <= yb;
            const float a0 = wy[by * RA_MAX_SPAN + yy], a1 = two ? wy[by * RA_MAX_SPAN + yy + 1] : 0.f;
            const float4 *row0 = (const float4 *)(src + ((size_t)(y0 + yy) * W + x0) * C) + cg;
            const float4 *row1 = two ? row0 + (size_t)W * (C / 4) : row0;
            float4 f0[8], f1[8];
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int xs = xa + u <= xb ? xa + u : xb;
              f0[u] = row0[(size_t)xs * (C / 4)];
              f1[u] = row1[(size_t)xs * (C / 4)];
              wv[u] = wx[bx * RA_MAX_SPAN + xs];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (xa + u <= xb) {
                const float w = a0 * wv[u];
                acc.x += w * f0[u].x, acc.y += w * f0[u].y, acc.z += w * f0[u].z, acc.w += w * f0[u].w;
              }
            if (two) {
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (xa + u <= xb) {
                  const float w = a1 * wv[u];
                  acc.x += w * f1[u].x, acc.y += w * f1[u].y, acc.z += w * f1[u].z, acc.w += w * f1[u].w;
                }
            }
          }
        } else {
          for (int yy = ya; yy <= yb; ++yy) {
            const float a = wy[by * RA_MAX_SPAN + yy];
            const float4 *row = (const float4 *)(src + ((size_t)(y0 + yy) * W + x0) * C) + cg;
            for (int xx = xa; xx <= xb; ++xx) {
              const float w = a * wx[bx * RA_MAX_SPAN + xx];
              const float4 f = row[(size_t)xx * (C / 4)];
              acc.x += w * f.x, acc.y += w * f.y, acc.z += w * f.z, acc.w += w * f.w;
            }
          }
        }
        // four 4-byte stores per lane at a lane pitch of 4 words: in element order they use 8 of the 32 banks (4-way
        // conflicts); each group of eight lanes starts one element later and the 32 lanes cover every bank
        float *o = ot + b * (C + 1) + 4 * cg;
        const float v[4] = {acc.x * inv_count, acc.y * inv_count, acc.z * inv_count, acc.w * inv_count};
        const int rot = (cg >> 3) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = (j + rot) & 3;
          o[k] = k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
        }
      }
    }
  } else
  for (int c = threadIdx.x; c < C; c += 256) {
    for (int b = 0; b < bins; ++b) {
      const int by = b / pw, bx = b - by * pw;
      float acc = 0.0f;
      if (!wide) {
        for (int yy = yp_lo[by]; yy <= yp_hi[by]; ++yy) {
          const float a = wy[by * RA_MAX_SPAN + yy];
          if (a == 0.0f) continue;
          const float *row = src + ((size_t)(y0 + yy) * W + x0) * C + c;
          for (int xx = xp_lo[bx]; xx <= xp_hi[bx]; ++xx) {
            const float w = a * wx[bx * RA_MAX_SPAN + xx];
            if (w != 0.0f) acc += w * row[(size_t)xx * C];
          }
        }
      } else {                              // RoI wider than the tap tables: taps on the fly
        for (int sy = 0; sy < g.gh; ++sy) {
          const float y = g.start_h + (float)by * g.bin_h + ((float)sy + 0.5f) * g.bin_h / (float)g.gh;
          int yl, yh;
          float wyl, wyh;
          if (!tap(y, H, yl, yh, wyl, wyh)) continue;
          for (int sx = 0; sx < g.gw; ++sx) {
            const float x = g.start_w + (float)bx * g.bin_w + ((float)sx + 0.5f) * g.bin_w / (float)g.gw;
            int xl, xh;
            float wxl, wxh;
            if (!tap(x, W, xl, xh, wxl, wxh)) continue;
            acc += (wyl * wxl) * src[((size_t)yl * W + xl) * C + c] + (wyl * wxh) * src[((size_t)yl * W + xh) * C + c] +
                   (wyh * wxl) * src[((size_t)yh * W + xl) * C + c] + (wyh * wxh) * src[((size_t)yh * W + xh) * C + c];
          }
        }
      }
      ot[b * (C + 1) + c] = acc * inv_count;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * bins; e += 256) {   // (C, bins) order, coalesced
    const int c = e / bins, b = e - c * bins;
    dst[e] = ot[b * (C + 1) + c];
  }
}

// The adaptive sampling grid (ceil(roi / pooled)) is data dependent: LDS is sized for `max_grid`
// samples per bin per axis (caller's estimate; 0 -> 8) and RoIs beyond it take the on-the-fly path.
int launch(bool backward, const float *const *feats, float *const *grads, const int *hs,
           const int *ws, const float *scales, int n_levels, int C, const float *rois,
           const int *levels, int R, int ph, int pw, int sampling_ratio, int aligned, float *out,
           const float *gout, int max_grid, hipStream_t stream) {
  if (n_levels < 1 || n_levels > kMaxLevels || R < 0 || C < 1 || ph < 1 || pw < 1) return DM_ERR_INVALID_ARG;
  if (R == 0) return DM_OK;
  Pyramid p;
  for (int l = 0; l < n_levels; ++l) {
    p.feat[l] = feats ? feats[l] : nullptr;
    p.grad[l] = grads ? grads[l] : nullptr;
    p.h[l] = hs[l], p.w[l] = ws[l], p.scale[l] = scales[l];
  }
  int gmax = sampling_ratio > 0 ? sampling_ratio : (max_grid > 0 ? max_grid : 8);
  while ((size_t)(ph + pw) * gmax * 16 > 48 * 1024 && gmax > 1) gmax /= 2;
  const size_t shm = (size_t)(ph + pw) * gmax * 16;
  if (backward)
    hipLaunchKernelGGL(roi_align_kernel<true>, dim3(R), dim3(256), shm, stream, p, rois, levels, C,
                       ph, pw, sampling_ratio, aligned, gmax, nullptr, gout);
  else
    hipLaunchKernelGGL(roi_align_kernel<false>, dim3(R), dim3(256), shm, stream, p, rois, levels,
                       C, ph, pw, sampling_ratio, aligned, gmax, out, nullptr);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

}  // namespace

extern "C" int dm_roi_align_backward_nhwc(float *const *grads_nhwc_host, const int *heights_host,
                                          const int *widths_host, const float *scales_host,
                                          int n_levels, int channels, const float *rois,
                                          const int *roi_levels, int n_rois, int pooled_h,
                                          int pooled_w, int sampling_ratio, int aligned,
                                          const float *grad_out, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_levels < 1 || n_levels > kMaxLevels || n_rois < 0 || channels < 1 || pooled_h < 1 ||
      pooled_w < 1)
    return DM_ERR_INVALID_ARG;
  if (n_rois == 0) return DM_OK;
  Pyramid p;
  for (int l = 0; l < n_levels; ++l) {
    p.feat[l] = nullptr;
    p.grad[l] = grads_nhwc_host[l];
    p.h[l] = heights_host[l], p.w[l] = widths_host[l], p.scale[l] = scales_host[l];
  }
  if (pooled_h > 16 || pooled_w > 16) return DM_ERR_UNSUPPORTED;
  const size_t shm = (size_t)(pooled_h + pooled_w) * RA_MAX_SPAN * 4 + (4 * RA_MAX_SPAN + 64) * 2 +
                     (size_t)pooled_h * pooled_w * (channels + 1) * 4;
  if (shm > 150 * 1024) return DM_ERR_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)roi_align_bwd_nhwc,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(roi_align_bwd_nhwc, dim3(n_rois), dim3(256), shm, st, p, rois, roi_levels,
                     channels, pooled_h, pooled_w, sampling_ratio, aligned, grad_out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_roi_align_forward_nhwc(const float *const *feats_nhwc_host, const int *heights_host,
                                         const int *widths_host, const float *scales_host,
                                         int n_levels, int channels, const float *rois,
                                         const int *roi_levels, int n_rois, int pooled_h,
                                         int pooled_w, int sampling_ratio, int aligned, float *out,
                                         dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_levels < 1 || n_levels > kMaxLevels || n_rois < 0 || channels < 1 || pooled_h < 1 ||
      pooled_w < 1 || pooled_h > 16 || pooled_w > 16)
    return DM_ERR_INVALID_ARG;
  if (n_rois == 0) return DM_OK;
  Pyramid p;
  for (int l = 0; l < n_levels; ++l) {
    p.feat[l] = feats_nhwc_host[l];
    p.grad[l] = nullptr;
    p.h[l] = heights_host[l], p.w[l] = widths_host[l], p.scale[l] = scales_host[l];
  }
  const size_t shm = (size_t)(pooled_h + pooled_w) * RA_MAX_SPAN * 4 + (4 * RA_MAX_SPAN + 64) * 2 +
                     (size_t)pooled_h * pooled_w * (channels + 1) * 4;
  if (shm > 150 * 1024) return DM_ERR_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    DM_HIP(hipFuncSetAttribute((const void *)roi_align_fwd_nhwc,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(roi_align_fwd_nhwc, dim3(n_rois), dim3(256), shm, st, p, rois, roi_levels,
                     channels, pooled_h, pooled_w, sampling_ratio, aligned, out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_roi_align_forward(const float *const *feats_host, const int *heights_host,
                                    const int *widths_host, const float *scales_host, int n_levels,
                                    int channels, const float *rois, const int *roi_levels,
                                    int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                    int aligned, int max_grid, float *out, dm_stream_t stream) {
  return launch(false, feats_host, nullptr, heights_host, widths_host, scales_host, n_levels,
                channels, rois, roi_levels, n_rois, pooled_h, pooled_w, sampling_ratio, aligned, out,
                nullptr, max_grid, (hipStream_t)stream);
}

extern "C" int dm_roi_align_backward(float *const *grads_host, const int *heights_host,
                                     const int *widths_host, const float *scales_host, int n_levels,
                                     int channels, const float *rois, const int *roi_levels,
                                     int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                     int aligned, int max_grid, const float *grad_out,
                                     dm_stream_t stream) {
  return launch(true, nullptr, grads_host, heights_host, widths_host, scales_host, n_levels, channels,
                rois, roi_levels, n_rois, pooled_h, pooled_w, sampling_ratio, aligned, nullptr,
                grad_out, max_grid, (hipStream_t)stream);
}
