// Element-wise / layout glue of the chained sub-graphs (chain.hip): the operations that sat between this library's
// kernels as ATen launches (ReLU masks, shortcut sums, bias-gradient sums, nearest resize of the FPN top-down path,
// stem max-pool, concatenation copies, evaluation-mode BatchNorm folds, fills) — as entry points of the library, so
// that a whole sub-graph is expressible as one op table.  All HBM-bound streaming kernels: 16-byte accesses,
// grid-stride over >= 1024 workgroups, fixed-order reductions (bitwise reproducible).
#include "dm_common.h"

namespace {

constexpr int kThreads = 256;

inline int grid_for(long long n_vec, int cap = 4096) {
  long long b = (n_vec + kThreads - 1) / kThreads;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ void relu_mask_kernel(const float4 *__restrict__ g, const float4 *__restrict__ y, float4 *__restrict__ out,
                                 long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = g[i];
    const float4 m = y[i];
    a.x = m.x > 0.f ? a.x : 0.f, a.y = m.y > 0.f ? a.y : 0.f, a.z = m.z > 0.f ? a.z : 0.f, a.w = m.w > 0.f ? a.w : 0.f;
    out[i] = a;
  }
}

// out = (a + b) masked by y > 0 (b, y optional)
__global__ void add_mask_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ y,
                                float4 *__restrict__ out, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = a[i];
    if (b != nullptr) {
      const float4 w = b[i];
      v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    if (y != nullptr) {
      const float4 m = y[i];
      v.x = m.x > 0.f ? v.x : 0.f, v.y = m.y > 0.f ? v.y : 0.f, v.z = m.z > 0.f ? v.z : 0.f, v.w = m.w > 0.f ? v.w : 0.f;
    }
    out[i] = v;
  }
}

// column sums of a (rows, C) matrix, two stages: partial[g][c] over a fixed row range per workgroup, then a
// fixed-order sum of the partials.  C % 4 == 0, C <= 1024.
__global__ void colsum_partial_kernel(const float *__restrict__ x, long long rows, int C, float *__restrict__ partial,
                                      int rows_per_block) {
  const int c4 = C / 4;
  const int lanes_per_row = c4;                       // threads that cover one row
  const int rows_at_once = kThreads / lanes_per_row;  // >= 1 (C <= 1024)
  const int col = threadIdx.x % lanes_per_row, sub = threadIdx.x / lanes_per_row;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sub < rows_at_once)
    for (long long r = r0 + sub; r < r1; r += rows_at_once) {
      const float4 v = *(const float4 *)(x + r * C + col * 4);
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
  __shared__ float4 sh[kThreads];
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (sub == 0) {
    for (int s = 1; s < rows_at_once; ++s) {
      const float4 v = sh[s * lanes_per_row + col];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    *(float4 *)(partial + (long long)blockIdx.x * C + col * 4) = acc;
  }
}

// 32 columns per workgroup, 8 lanes per column: lane l sums the partials p = l, l + 8, ... in order, then the eight
// lane sums are added in lane order (a fixed tree: bitwise reproducible)
__global__ void colsum_final_kernel(const float *__restrict__ partial, int n_part, int C, float *__restrict__ out,
                                    int accumulate) {
  __shared__ float sh[8][32];
  const int col = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + col;
  float s = 0.f;
  if (c < C)
    for (int p = lane; p < n_part; p += 8) s += partial[(long long)p * C + c];
  sh[lane][col] = s;
  __syncthreads();
  if (lane == 0 && c < C) {
    float t = sh[0][col];
    for (int l = 1; l < 8; ++l) t += sh[l][col];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// torch's 'nearest': src = min(floor(dst * (float)in / out), in - 1)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ void resize_nearest_kernel(const float *__restrict__ x, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                      float *__restrict__ y) {
  const int c4 = C / 4;
  const long long n = (long long)B * Ho * Wo * c4;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long p = i / c4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho), b = (int)(p / Ho);
    const int hi = nearest_src(ho, sh, Hi), wi = nearest_src(wo, sw, Wi);
    ((float4 *)y)[i] = *(const float4 *)(x + (((long long)b * Hi + hi) * Wi + wi) * C + c * 4);
  }
}

// gx[b,hi,wi,:] (+)= sum of gy over the destination pixels that read (hi, wi); fixed order (rows, then columns)
__global__ void resize_nearest_bwd_kernel(const float *__restrict__ gy, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                          float *__restrict__ gx, int accumulate) {
  const int c4 = C / 4;
  const long long n = (long long)B * Hi * Wi * c4;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long p = i / c4;
    const int wi = (int)(p % Wi);
    p /= Wi;
    const int hi = (int)(p % Hi), b = (int)(p / Hi);
    // candidate destination window: every dst with floor(dst * in/out) == src lies in [src*out/in - 1, (src+1)*out/in + 1]
    int h0 = (int)((long long)hi * Ho / Hi) - 1, h1 = (int)(((long long)hi + 1) * Ho / Hi) + 1;
    int w0 = (int)((long long)wi * Wo / Wi) - 1, w1 = (int)(((long long)wi + 1) * Wo / Wi) + 1;
    if (h0 < 0) h0 = 0;
    if (w0 < 0) w0 = 0;
    if (h1 > Ho - 1) h1 = Ho - 1;
    if (w1 > Wo - 1) w1 = Wo - 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ho = h0; ho <= h1; ++ho) {
      if (nearest_src(ho, sh, Hi) != hi) continue;
      for (int wo = w0; wo <= w1; ++wo) {
        if (nearest_src(wo, sw, Wi) != wi) continue;
        const float4 v = *(const float4 *)(gy + (((long long)b * Ho + ho) * Wo + wo) * C + c * 4);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
    }
    float4 *o = (float4 *)gx + i;
    if (accumulate) {
      const float4 old = *o;
      acc.x += old.x, acc.y += old.y, acc.z += old.z, acc.w += old.w;
    }
    *o = acc;
  }
}

// max_pool2d(kernel k, stride s, padding p) on NHWC (ResNet stem: 3 / 2 / 1; FPN extra level: 1 / 2 / 0)
__global__ void maxpool_kernel(const float *__restrict__ x, int B, int Hi, int Wi, int C, int Ho, int Wo, int k, int s,
                               int pad, float *__restrict__ y) {
  const int c4 = C / 4;
  const long long n = (long long)B * Ho * Wo * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long p = i / c4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho), b = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int a = 0; a < k; ++a) {
      const int hi = ho * s - pad + a;
      if (hi < 0 || hi >= Hi) continue;
      for (int e = 0; e < k; ++e) {
        const int wi = wo * s - pad + e;
        if (wi < 0 || wi >= Wi) continue;
        const float4 v = *(const float4 *)(x + (((long long)b * Hi + hi) * Wi + wi) * C + c * 4);
        // NaN propagates like ATen's max_pool2d: (v > m) || isnan(v)
        m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x, m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
        m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z, m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
      }
    }
    ((float4 *)y)[i] = m;
  }
}

// gradient of the kernel-1 / stride-s pooling (a sub-sampling): gx = 0 except gx[b, ho*s, wo*s] = gy[b, ho, wo]
__global__ void subsample_bwd_kernel(const float *__restrict__ gy, int B, int Hi, int Wi, int C, int Ho, int Wo, int s,
                                     float *__restrict__ gx) {
  const int c4 = C / 4;
  const long long n = (long long)B * Hi * Wi * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long p = i / c4;
    const int wi = (int)(p % Wi);
    p /= Wi;
    const int hi = (int)(p % Hi), b = (int)(p / Hi);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hi % s == 0 && wi % s == 0 && hi / s < Ho && wi / s < Wo)
      v = *(const float4 *)(gy + (((long long)b * Ho + hi / s) * Wo + wi / s) * C + c * 4);
    ((float4 *)gx)[i] = v;
  }
}

// rows x cols (floats, cols % 4 == 0) between two pitched matrices: channel-slice copies of NHWC tensors
__global__ void copy2d_kernel(const float *__restrict__ src, long long src_pitch, float *__restrict__ dst,
                              long long dst_pitch, long long rows, int cols) {
  const int c4 = cols / 4;
  const long long n = rows * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4;
    const int c = (int)(i % c4);
    *(float4 *)(dst + r * dst_pitch + c * 4) = *(const float4 *)(src + r * src_pitch + c * 4);
  }
}

__global__ void copy2d_scalar_kernel(const float *__restrict__ src, long long src_pitch, float *__restrict__ dst,
                                     long long dst_pitch, long long rows, int cols) {
  const long long n = rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int c = (int)(i % cols);
    dst[r * dst_pitch + c] = src[r * src_pitch + c];
  }
}

__global__ void fill_kernel(uint4 *__restrict__ dst, uint4 v, long long n16) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}

struct BnFoldRow {   // one evaluation-mode BatchNorm layer (device pointers)
  const float *gamma, *beta, *mean, *var;
  float *scale, *shift;
  float eps;
  int C;
};

// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale, for every layer of a table in one launch; the
// same operations per element, in the same order, as the tensor formulation (x + eps, rsqrt, * gamma, mean *, beta -)
__global__ void bn_fold_kernel(const BnFoldRow *__restrict__ rows, int n_rows) {
  const int r = blockIdx.y;
  if (r >= n_rows) return;
  const BnFoldRow row = rows[r];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < row.C; c += gridDim.x * blockDim.x) {
    const float s = rsqrtf(row.var[c] + row.eps) * row.gamma[c];   // (torch.rsqrt is the same device function)
    row.scale[c] = s;
    row.shift[c] = row.beta[c] - row.mean[c] * s;
  }
}

struct AxpyRow {   // dst[0..n) += src[0..n)  (n % 4 == 0 not required)
  float *dst;
  const float *src;
  long long n;
};

__global__ void multi_add_kernel(const AxpyRow *__restrict__ rows, int n_rows, int assign) {
  const int r = blockIdx.y;
  if (r >= n_rows) return;
  const AxpyRow row = rows[r];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < row.n; i += (long long)gridDim.x * blockDim.x)
    row.dst[i] = assign ? row.src[i] : row.dst[i] + row.src[i];
}

// HeightCompression (height_compression.py:10-25) in NHWC: active voxel row i with coordinates [b, z, y, x] lands at
// out[b][y][x][c * D + z] (the reference's .dense() -> view(B, C * D, H, W), channels_last memory).  The map is zeroed by
// the caller; the gradient is the matching gather.
__global__ void height_compress_kernel(const float *__restrict__ feat, const int4 *__restrict__ idx, long long n, int C,
                                       int D, int H, int W, float *__restrict__ out) {
  const long long total = n * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i % C);
    const int4 q = idx[r];      // b, z, y, x
    out[((((long long)q.x * H + q.z) * W + q.w) * C + c) * D + q.y] = feat[i];
  }
}

__global__ void height_compress_bwd_kernel(const float *__restrict__ gout, const int4 *__restrict__ idx, long long n, int C,
                                           int D, int H, int W, float *__restrict__ gfeat) {
  const long long total = n * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i % C);
    const int4 q = idx[r];
    gfeat[i] = gout[((((long long)q.x * H + q.z) * W + q.w) * C + c) * D + q.y];
  }
}

}  // namespace

extern "C" int dm_height_compress_forward(const float *features, const int *indices, long long n, int C, int batch, int D,
                                          int H, int W, float *out, dm_stream_t stream) {
  if (n < 0 || C <= 0 || batch <= 0 || D <= 0 || H <= 0 || W <= 0 || !out) return DM_ERR_INVALID_ARG;
  DM_HIP(hipMemsetAsync(out, 0, (size_t)batch * H * W * C * D * sizeof(float), (hipStream_t)stream));
  if (n == 0) return DM_OK;
  if (!features || !indices) return DM_ERR_INVALID_ARG;
  height_compress_kernel<<<grid_for(n * C), kThreads, 0, (hipStream_t)stream>>>(features, (const int4 *)indices, n, C, D, H, W,
                                                                                out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_height_compress_backward(const float *grad_out, const int *indices, long long n, int C, int D, int H, int W,
                                           float *grad_features, dm_stream_t stream) {
  if (n < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  if (!grad_out || !indices || !grad_features) return DM_ERR_INVALID_ARG;
  height_compress_bwd_kernel<<<grid_for(n * C), kThreads, 0, (hipStream_t)stream>>>(grad_out, (const int4 *)indices, n, C, D,
                                                                                    H, W, grad_features);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_relu_mask_f32(const float *grad, const float *y, float *out, long long n, dm_stream_t stream) {
  if (n == 0) return DM_OK;
  if (!grad || !y || !out || n < 0 || n % 4) return DM_ERR_INVALID_ARG;
  relu_mask_kernel<<<grid_for(n / 4), kThreads, 0, (hipStream_t)stream>>>((const float4 *)grad, (const float4 *)y,
                                                                            (float4 *)out, n / 4);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_add_mask_f32(const float *a, const float *b, const float *y, float *out, long long n,
                               dm_stream_t stream) {
  if (n == 0) return DM_OK;
  if (!a || !out || n < 0 || n % 4) return DM_ERR_INVALID_ARG;
  add_mask_kernel<<<grid_for(n / 4), kThreads, 0, (hipStream_t)stream>>>((const float4 *)a, (const float4 *)b,
                                                                           (const float4 *)y, (float4 *)out, n / 4);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

static int colsum_blocks(long long rows, int *rows_per_block) {
  long long per = (rows + 511) / 512;
  if (per < 32) per = 32;
  *rows_per_block = (int)per;
  return (int)((rows + per - 1) / per);
}

extern "C" size_t dm_colsum_workspace_bytes(long long rows, int C) {
  int per;
  return dm_align((size_t)colsum_blocks(rows, &per) * (size_t)C * sizeof(float));
}

extern "C" int dm_colsum_f32(const float *x, long long rows, int C, float *out, int accumulate, void *workspace,
                             size_t workspace_bytes, dm_stream_t stream) {
  if (!x || !out || rows <= 0 || C <= 0 || C % 4 || C > 1024) return DM_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < dm_colsum_workspace_bytes(rows, C)) return DM_ERR_WORKSPACE;
  int per;
  const int nb = colsum_blocks(rows, &per);
  colsum_partial_kernel<<<nb, kThreads, 0, (hipStream_t)stream>>>(x, rows, C, (float *)workspace, per);
  DM_CHECK_LAUNCH();
  colsum_final_kernel<<<dm_ceil_div(C, 32), 256, 0, (hipStream_t)stream>>>((const float *)workspace, nb, C, out,
                                                                           accumulate);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_resize_nearest_nhwc(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, float *y,
                                      dm_stream_t stream) {
  if (!x || !y || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || C % 4) return DM_ERR_INVALID_ARG;
  resize_nearest_kernel<<<grid_for((long long)B * Ho * Wo * (C / 4)), kThreads, 0, (hipStream_t)stream>>>(
      x, B, Hi, Wi, C, Ho, Wo, y);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_resize_nearest_nhwc_backward(const float *grad_y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                               float *grad_x, int accumulate, dm_stream_t stream) {
  if (!grad_y || !grad_x || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || C % 4)
    return DM_ERR_INVALID_ARG;
  resize_nearest_bwd_kernel<<<grid_for((long long)B * Hi * Wi * (C / 4)), kThreads, 0, (hipStream_t)stream>>>(
      grad_y, B, Hi, Wi, C, Ho, Wo, grad_x, accumulate);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_maxpool_nhwc(const float *x, int B, int Hi, int Wi, int C, int k, int stride, int pad, float *y,
                               dm_stream_t stream) {
  if (!x || !y || B <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || C % 4 || k < 1 || stride < 1 || pad < 0 || 2 * pad > k)
    return DM_ERR_INVALID_ARG;
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return DM_ERR_INVALID_ARG;
  maxpool_kernel<<<grid_for((long long)B * Ho * Wo * (C / 4)), kThreads, 0, (hipStream_t)stream>>>(
      x, B, Hi, Wi, C, Ho, Wo, k, stride, pad, y);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_subsample_nhwc_backward(const float *grad_y, int B, int Hi, int Wi, int C, int stride,
                                          float *grad_x, dm_stream_t stream) {
  if (!grad_y || !grad_x || B <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || C % 4 || stride < 1) return DM_ERR_INVALID_ARG;
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  subsample_bwd_kernel<<<grid_for((long long)B * Hi * Wi * (C / 4)), kThreads, 0, (hipStream_t)stream>>>(
      grad_y, B, Hi, Wi, C, Ho, Wo, stride, grad_x);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_copy2d_f32(const float *src, long long src_pitch, float *dst, long long dst_pitch, long long rows,
                             int cols, dm_stream_t stream) {
  if (rows == 0 || cols == 0) return DM_OK;
  if (!src || !dst || rows < 0 || cols < 0 || src_pitch < cols || dst_pitch < cols) return DM_ERR_INVALID_ARG;
  if (cols % 4 || src_pitch % 4 || dst_pitch % 4 || ((uintptr_t)src | (uintptr_t)dst) % 16) {   // e.g. the 18 | 42 | 12 column blocks of the anchor head
    copy2d_scalar_kernel<<<grid_for(rows * cols), kThreads, 0, (hipStream_t)stream>>>(src, src_pitch, dst, dst_pitch,
                                                                                      rows, cols);
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  copy2d_kernel<<<grid_for(rows * (cols / 4)), kThreads, 0, (hipStream_t)stream>>>(src, src_pitch, dst, dst_pitch, rows,
                                                                                   cols);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_fill_bytes(void *dst, int byte_value, size_t nbytes, dm_stream_t stream) {
  if (nbytes == 0) return DM_OK;
  if (!dst) return DM_ERR_INVALID_ARG;
  if (((uintptr_t)dst | nbytes) % 16 == 0) {      // an ordinary kernel on the caller's stream (no runtime blit path)
    const unsigned w = (unsigned)(byte_value & 0xff) * 0x01010101u;
    fill_kernel<<<grid_for((long long)(nbytes / 16)), kThreads, 0, (hipStream_t)stream>>>((uint4 *)dst, make_uint4(w, w, w, w),
                                                                                         (long long)(nbytes / 16));
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  DM_HIP(hipMemsetAsync(dst, byte_value, nbytes, (hipStream_t)stream));
  return DM_OK;
}

extern "C" int dm_bn_fold_batch(const void *table_dev, int n_rows, int max_channels, dm_stream_t stream) {
  if (n_rows == 0) return DM_OK;
  if (!table_dev || n_rows < 0 || max_channels <= 0) return DM_ERR_INVALID_ARG;
  static_assert(sizeof(BnFoldRow) == 56, "dm_bn_fold_batch row layout");
  dim3 grid(dm_ceil_div(max_channels, 256), n_rows);
  bn_fold_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const BnFoldRow *)table_dev, n_rows);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_multi_add_f32(const void *table_dev, int n_rows, long long max_n, int assign, dm_stream_t stream) {
  if (n_rows == 0) return DM_OK;
  if (!table_dev || n_rows < 0 || max_n <= 0) return DM_ERR_INVALID_ARG;
  static_assert(sizeof(AxpyRow) == 24, "dm_multi_add_f32 row layout");
  long long bx = (max_n + kThreads * 4 - 1) / (kThreads * 4);
  if (bx > 256) bx = 256;
  dim3 grid((int)bx, n_rows);
  multi_add_kernel<<<grid, kThreads, 0, (hipStream_t)stream>>>((const AxpyRow *)table_dev, n_rows, assign);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
