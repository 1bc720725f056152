// Stable sort of short rows of float keys — the score orders of the proposal layers and NMS wrappers
// (pcdet/models/model_utils/model_nms_utils.py:6-25 `scores.sort / topk`, mmcv batched_nms, mmdet RPN
// `_get_bboxes_single`: `torch.sort(scores, descending=True)` on a few hundred to a few thousand candidates).
// torch hands these to rocprim's merge sort: 3-6 launches per call, ~30 calls per DetMatch iteration.
// Here: one launch; one workgroup per row sorts (key, index) pairs in LDS with a bitonic network.
//   key  = the float's bits made monotonic (sign flip), inverted for a descending order; -0.0 counts as +0.0
//   pair = key << 32 | index  — ties keep their original order (what torch.sort(stable=True) guarantees), and the
//          result is a function of the input only.
// Rows of up to 16 384 elements (128 KB of LDS); the padding up to the next power of two sorts behind everything.
// NaN: a positive NaN is the largest key (torch: NaN is greater than any number), a negative NaN the smallest.
#include <hip/hip_runtime.h>

#include "../../include/detmatch_hip.h"
#include "dm_common.h"

namespace {

constexpr int SORT_MAX = 16384;

__device__ __forceinline__ unsigned int sortable(float v, int descending) {
  if (v == 0.0f) v = 0.0f;                       // -0.0 == +0.0
  unsigned int u = __float_as_uint(v);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;    // ascending order of the floats == ascending order of u
  return descending ? ~u : u;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void sort_rows_kernel(const float *__restrict__ keys, int n, long long key_stride,
                                                            int descending, int pow2, long long *__restrict__ idx_out,
                                                            float *__restrict__ keys_out) {
  extern __shared__ unsigned long long sort_lds[];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float *k = keys + (size_t)row * key_stride;
  for (int i = tid; i < pow2; i += THREADS)
    sort_lds[i] = i < n ? ((unsigned long long)sortable(k[i], descending) << 32) | (unsigned int)i : ~0ull;
  __syncthreads();
  for (int size = 2; size <= pow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (pow2 >> 1); t += THREADS) {
        const int lo = 2 * t - (t & (stride - 1));          // index of the lower element of pair t
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;                   // direction of this bitonic block
        const unsigned long long a = sort_lds[lo], b = sort_lds[hi];
        if ((a > b) == up) {
          sort_lds[lo] = b;
          sort_lds[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += THREADS) {
    const unsigned long long v = sort_lds[i];
    const unsigned int src = (unsigned int)(v & 0xFFFFFFFFull);
    idx_out[(size_t)row * n + i] = (long long)src;
    if (keys_out) keys_out[(size_t)row * n + i] = k[src];
  }
}

}  // namespace

extern "C" int dm_sort_rows_max(void) { return SORT_MAX; }

extern "C" int dm_sort_rows_f32(const float *keys, int rows, int n, long long key_stride, int descending,
                                long long *idx_out, float *keys_out, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (rows < 0 || n < 0 || n > SORT_MAX || key_stride < n) return DM_ERR_INVALID_ARG;
  if (rows == 0 || n == 0) return DM_OK;
  if (!keys || !idx_out) return DM_ERR_INVALID_ARG;
  int pow2 = 2;
  while (pow2 < n) pow2 <<= 1;
  const size_t lds = (size_t)pow2 * sizeof(unsigned long long);
  if (pow2 <= 2048) {
    sort_rows_kernel<256><<<rows, 256, lds, st>>>(keys, n, key_stride, descending, pow2, idx_out, keys_out);
  } else {
    static bool attr = false;
    if (!attr) {
      DM_HIP(hipFuncSetAttribute((const void *)sort_rows_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 SORT_MAX * (int)sizeof(unsigned long long)));
      attr = true;
    }
    sort_rows_kernel<1024><<<rows, 1024, lds, st>>>(keys, n, key_stride, descending, pow2, idx_out, keys_out);
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}
