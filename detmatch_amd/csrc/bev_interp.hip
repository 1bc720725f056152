// Key-point features from the BEV map: bilinear interpolation of an NHWC feature map at the key points
// (VoxelSetAbstraction.interpolate_from_bev_features + bilinear_interpolate_torch,
// pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:9-40,113-117), forward and the gradient w.r.t.
// the map.  The reference gathers four (K, C) slices per sample with advanced indexing (and
// back-propagates with four sort-based index_put accumulations); here one wave per key point reads the
// four cells once, and the backward pass is a scatter WITHOUT atomics: the first entry that touches a
// cell sums every entry of that cell in entry order (deterministic).
#include "dm_common.h"

namespace {

struct BevGeom {
  int B, H, W, C, K;
  float x0, y0, vx, vy, stride;
};

// one wave per key point; lanes own channels lane*4.. (+256 per round)
__global__ __launch_bounds__(256) void bev_interp_fwd_kernel(const float *__restrict__ im,
                                                             const float *__restrict__ kp, int kp_stride, BevGeom g,
                                                             float *__restrict__ out, int *__restrict__ cells,
                                                             float *__restrict__ weights) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= (long long)g.B * g.K) return;
  const int b = (int)(p / g.K);
  const float *q = kp + p * kp_stride;
  // torch divides a tensor by a Python scalar as a multiplication by the float reciprocal
  const float x = (q[0] - g.x0) * (1.0f / g.vx) * (1.0f / g.stride), y = (q[1] - g.y0) * (1.0f / g.vy) * (1.0f / g.stride);
  long long xf = (long long)floorf(x), yf = (long long)floorf(y);
  const long long xm = g.W - 1, ym = g.H - 1;
  const int x0 = (int)min(max(xf, 0ll), xm), x1 = (int)min(max(xf + 1, 0ll), xm);
  const int y0 = (int)min(max(yf, 0ll), ym), y1 = (int)min(max(yf + 1, 0ll), ym);
  const float wa = ((float)x1 - x) * ((float)y1 - y), wb = ((float)x1 - x) * (y - (float)y0);
  const float wc = (x - (float)x0) * ((float)y1 - y), wd = (x - (float)x0) * (y - (float)y0);
  const int ca = y0 * g.W + x0, cb = y1 * g.W + x0, cc = y0 * g.W + x1, cd = y1 * g.W + x1;
  if (lane == 0) {
    int *cp = cells + p * 4;
    float *wp = weights + p * 4;
    cp[0] = ca, cp[1] = cb, cp[2] = cc, cp[3] = cd;
    wp[0] = wa, wp[1] = wb, wp[2] = wc, wp[3] = wd;
  }
  const float *base = im + (size_t)b * g.H * g.W * g.C;
  for (int c = lane * 4; c < g.C; c += 256) {
    const float4 a = *(const float4 *)(base + (size_t)ca * g.C + c), bb = *(const float4 *)(base + (size_t)cb * g.C + c);
    const float4 cv = *(const float4 *)(base + (size_t)cc * g.C + c), d = *(const float4 *)(base + (size_t)cd * g.C + c);
    float4 o;
    o.x = a.x * wa + bb.x * wb + cv.x * wc + d.x * wd;
    o.y = a.y * wa + bb.y * wb + cv.y * wc + d.y * wd;
    o.z = a.z * wa + bb.z * wb + cv.z * wc + d.z * wd;
    o.w = a.w * wa + bb.w * wb + cv.w * wc + d.w * wd;
    *(float4 *)(out + p * g.C + c) = o;
  }
}

// one wave per entry e = (key point, corner) of a sample; the wave of the FIRST entry of a cell adds up
// all entries of that cell in entry order and writes the cell (grad_im is zero elsewhere)
__global__ __launch_bounds__(256) void bev_interp_bwd_kernel(const float *__restrict__ gout,
                                                             const int *__restrict__ cells,
                                                             const float *__restrict__ weights, BevGeom g,
                                                             float *__restrict__ gim) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int e = blockIdx.x * 4 + wave, E = g.K * 4;
  if (e >= E) return;
  const int *cs = cells + (size_t)b * E;
  const float *ws = weights + (size_t)b * E;
  const int cell = cs[e];
  // any earlier entry on the same cell? (then that entry's wave owns the cell)
  bool dup = false;
  for (int j = lane; j < e; j += 64) dup |= cs[j] == cell;
  if (__any(dup)) return;
  const float *go = gout + (size_t)b * g.K * g.C;
  float *dst = gim + ((size_t)b * g.H * g.W + cell) * g.C;
  float4 acc[4];
  const int rounds = (g.C + 255) / 256;    // C <= 1024
  for (int r = 0; r < rounds; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = e; j0 < E; j0 += 64) {
    const int j = j0 + lane;
    unsigned long long hit = __ballot(j < E && cs[j] == cell);
    while (hit) {
      const int t = __ffsll((long long)hit) - 1;
      hit &= hit - 1;
      const int jj = j0 + t;
      const float w = ws[jj];
      const float *src = go + (size_t)(jj >> 2) * g.C;
      for (int r = 0; r < rounds; ++r) {
        const int c = r * 256 + lane * 4;
        if (c < g.C) {
          const float4 v = *(const float4 *)(src + c);
          acc[r].x += v.x * w, acc[r].y += v.y * w, acc[r].z += v.z * w, acc[r].w += v.w * w;
        }
      }
    }
  }
  for (int r = 0; r < rounds; ++r) {
    const int c = r * 256 + lane * 4;
    if (c < g.C) *(float4 *)(dst + c) = acc[r];
  }
}

int bev_geom(BevGeom &g, int B, int H, int W, int C, int K, const float *geom5) {
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || C < 4 || (C & 3) || C > 1024 || !geom5) return DM_ERR_INVALID_ARG;
  g.B = B, g.H = H, g.W = W, g.C = C, g.K = K;
  g.x0 = geom5[0], g.y0 = geom5[1], g.vx = geom5[2], g.vy = geom5[3], g.stride = geom5[4];
  return DM_OK;
}

}  // namespace

extern "C" int dm_bev_interpolate_forward(const float *bev_nhwc, int batch, int height, int width, int channels,
                                          const float *keypoints, int keypoint_stride, int n_keypoints,
                                          const float *geom5, float *out, int *cells, float *weights,
                                          dm_stream_t stream) {
  BevGeom g;
  int rc = bev_geom(g, batch, height, width, channels, n_keypoints, geom5);
  if (rc != DM_OK) return rc;
  if (!bev_nhwc || !keypoints || keypoint_stride < 2 || !out || !cells || !weights) return DM_ERR_INVALID_ARG;
  bev_interp_fwd_kernel<<<dm_ceil_div((long long)batch * n_keypoints, 4), 256, 0, (hipStream_t)stream>>>(
      bev_nhwc, keypoints, keypoint_stride, g, out, cells, weights);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_bev_interpolate_backward(const float *grad_out, const int *cells, const float *weights, int batch,
                                           int height, int width, int channels, int n_keypoints,
                                           float *grad_bev_nhwc, dm_stream_t stream) {
  const float unit[5] = {0.f, 0.f, 1.f, 1.f, 1.f};
  BevGeom g;
  int rc = bev_geom(g, batch, height, width, channels, n_keypoints, unit);
  if (rc != DM_OK) return rc;
  if (!grad_out || !cells || !weights || !grad_bev_nhwc) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  DM_HIP(hipMemsetAsync(grad_bev_nhwc, 0, (size_t)batch * height * width * channels * sizeof(float), st));
  bev_interp_bwd_kernel<<<dim3(dm_ceil_div((long long)n_keypoints * 4, 4), batch), 256, 0, st>>>(grad_out, cells,
                                                                                               weights, g,
                                                                                               grad_bev_nhwc);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
