// Shared helpers for libdetmatch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/detmatch_hip.h"

#define DM_MAX_BATCH 64
#define DM_WAVE 64

#define DM_CHECK_LAUNCH()                              \
  do {                                                 \
    if (hipGetLastError() != hipSuccess) return DM_ERR_LAUNCH; \
  } while (0)

#define DM_HIP(expr)                                   \
  do {                                                 \
    if ((expr) != hipSuccess) return DM_ERR_LAUNCH;    \
  } while (0)

static inline int dm_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

static inline size_t dm_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static inline uint32_t dm_pow2_ceil(uint64_t x) {
  uint32_t p = 1;
  while (p < x && p < (1u << 31)) p <<= 1;
  return p;
}

static inline int dm_log2(uint32_t p) {
  int l = 0;
  while ((1u << l) < p) ++l;
  return l;
}

// Bump allocator over a caller-provided workspace.
struct DmArena {
  char *base;
  size_t off;
  size_t cap;
  DmArena(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes) {}
  template <typename T>
  T *take(size_t count) {
    size_t bytes = dm_align(count * sizeof(T));
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= cap; }
};

struct DmBatchOffsets {
  int off[DM_MAX_BATCH + 1];
};

// ---- open-addressing hash of uint32 cell ids -> int32 row ---------------
#define DM_HASH_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ uint32_t dm_hash_slot(uint32_t key, int log2_size) {
  return (key * 2654435761u) >> (32 - log2_size);
}

// returns the slot of `key`, inserting it if absent; *fresh = 1 for the inserter
__device__ __forceinline__ uint32_t dm_hash_insert(uint32_t *keys, int log2_size,
                                                   uint32_t key, int *fresh) {
  uint32_t mask = (1u << log2_size) - 1u;
  uint32_t s = dm_hash_slot(key, log2_size);
  *fresh = 0;
  while (true) {
    uint32_t prev = atomicCAS(&keys[s], DM_HASH_EMPTY, key);
    if (prev == DM_HASH_EMPTY) {
      *fresh = 1;
      return s;
    }
    if (prev == key) return s;
    s = (s + 1) & mask;
  }
}

// returns the value stored for `key` or -1
__device__ __forceinline__ int dm_hash_find(const uint32_t *keys, const int32_t *vals,
                                            int log2_size, uint32_t key) {
  uint32_t mask = (1u << log2_size) - 1u;
  uint32_t s = dm_hash_slot(key, log2_size);
  while (true) {
    uint32_t k = keys[s];
    if (k == key) return vals[s];
    if (k == DM_HASH_EMPTY) return -1;
    s = (s + 1) & mask;
  }
}

__device__ __forceinline__ int dm_lane_id() { return threadIdx.x & 63; }

// ---- optional per-launch HIP-event timing (bench.py roofline leg) ----------
// Disabled (zero overhead) unless dm_profile_enable(1) was called.  Two flavours:
//   dm_prof_begin / dm_prof_end   events recorded on the launch stream immediately before and after
//                                 (a multi-kernel region: the weight gradient);
//   dm_prof_open + hipExtLaunchKernelGGL(..., e0, e1, ...)   the events carry the start / end
//                                 timestamps of THAT dispatch (what rocprofv3's kernel trace reports),
//                                 unaffected by work queued on other streams.
enum { DM_PROF_SPCONV_GG = 0, DM_PROF_SPCONV_WGRAD = 1 };
int dm_prof_begin(hipStream_t st, int kind, int a, int b, int c, int rows, int kvol,
                  const void *table);
void dm_prof_end(int idx, hipStream_t st);
int dm_prof_open(int kind, int a, int b, int c, int rows, int kvol, const void *table, hipEvent_t *e0,
                 hipEvent_t *e1);
