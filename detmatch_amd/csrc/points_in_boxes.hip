// points_in_boxes (GPU semantics, MARGIN 1e-5) for gfx950.
// Replaces pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:17-37,313-360.
// cos/sin of each box heading are evaluated once per box per workgroup (LDS), with the
// "double libm rounded to float" rule shared with the CPU oracle.
#include "dm_common.h"

namespace {

constexpr int PIB_MAX_BOXES = 1024;

__global__ __launch_bounds__(256) void points_in_boxes_kernel(int boxes_num, int pts_num,
                                                              const float *__restrict__ boxes,
                                                              const float *__restrict__ pts,
                                                              int *__restrict__ box_idx) {
  __shared__ float sb[PIB_MAX_BOXES * 7];
  __shared__ float2 scs[PIB_MAX_BOXES];
  const int bs = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const float *bx = boxes + (size_t)bs * boxes_num * 7;
  int result = -1;
  bool found = false;
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < pts_num) {
    const float *pp = pts + ((size_t)bs * pts_num + p) * 3;
    x = pp[0];
    y = pp[1];
    z = pp[2];
  }
  for (int b0 = 0; b0 < boxes_num; b0 += PIB_MAX_BOXES) {
    int nb = min(PIB_MAX_BOXES, boxes_num - b0);
    __syncthreads();
    for (int e = threadIdx.x; e < nb * 7; e += 256) sb[e] = bx[(size_t)b0 * 7 + e];
    for (int k = threadIdx.x; k < nb; k += 256) {
      double rz = (double)bx[(size_t)(b0 + k) * 7 + 6];
      scs[k] = make_float2((float)cos(-rz), (float)sin(-rz));  // kernel.cu:18
    }
    __syncthreads();
    if (p < pts_num && !found) {
      for (int k = 0; k < nb; ++k) {
        const float *b = sb + k * 7;
        float cx = b[0], cy = b[1], cz = b[2], dx = b[3], dy = b[4], dz = b[5];
        // kernel.cu:32: fabsf(z - cz) > dz / 2.0 (promoted to double)
        if ((double)fabsf(z - cz) > (double)dz / 2.0) continue;
        float cosa = scs[k].x, sina = scs[k].y;
        float sx = x - cx, sy = y - cy;
        float local_x = sx * cosa + sy * (-sina);
        float local_y = sx * sina + sy * cosa;
        const float MARGIN = 1e-5f;
        bool in = ((double)fabsf(local_x) < (double)dx / 2.0 + (double)MARGIN) &&
                  ((double)fabsf(local_y) < (double)dy / 2.0 + (double)MARGIN);
        if (in) {
          result = b0 + k;
          found = true;
          break;
        }
      }
    }
  }
  if (p < pts_num) box_idx[(size_t)bs * pts_num + p] = result;
}

}  // namespace

extern "C" int dm_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes,
                                  const float *pts, int *box_idx_of_points, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (batch < 0 || boxes_num < 0 || pts_num < 0) return DM_ERR_INVALID_ARG;
  if (batch == 0 || pts_num == 0) return DM_OK;
  if (!pts || !box_idx_of_points || (boxes_num > 0 && !boxes)) return DM_ERR_INVALID_ARG;
  dim3 grid(dm_ceil_div(pts_num, 256), batch);
  points_in_boxes_kernel<<<grid, 256, 0, st>>>(boxes_num, pts_num, boxes, pts, box_idx_of_points);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
