// Sparse-conv rulebook for gfx950: hash-indexed gather tables, no O(batch*volume)
// grid, no per-pair atomics, deterministic slot order.
//
// Replaces the reference's getIndicePair<3> GPU path
//   mmdet3d/ops/spconv/include/spconv/spconv_ops.h:28-141
//   mmdet3d/ops/spconv/include/spconv/indice.cu.h:24-204
//   mmdet3d/ops/spconv/src/indice_cuda.cu:24-135
// (369 MB*B gridOut memset, atomicAdd slot race, torch::_unique on 27*N keys).
//
// Layout in HBM (all int32, row-major):
//   nbr_out (kvol, n_out)   nbr_in (kvol, n_in)   indice_pairs (kvol, 2, n_in)
// Offset-major tables make every wave read/write 256 contiguous bytes per
// wave-instruction (64 consecutive rows of one kernel offset).
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "dm_common.h"

namespace {

struct RbGeom {
  int spatial[3];
  int out_shape[3];
  int ksize[3];
  int stride[3];
  int pad[3];
  int kvol;
  uint32_t in_vol;
  uint32_t out_vol;
};

__device__ __forceinline__ void decode_k(const RbGeom &g, int k, int kk[3]) {
  kk[2] = k % g.ksize[2];
  int t = k / g.ksize[2];
  kk[1] = t % g.ksize[1];
  kk[0] = t / g.ksize[1];
}

__device__ __forceinline__ uint32_t in_key(const RbGeom &g, int b, int z, int y, int x) {
  return (uint32_t)b * g.in_vol + ((uint32_t)z * g.spatial[1] + y) * g.spatial[2] + x;
}
__device__ __forceinline__ uint32_t out_key(const RbGeom &g, int b, int z, int y, int x) {
  return (uint32_t)b * g.out_vol + ((uint32_t)z * g.out_shape[1] + y) * g.out_shape[2] + x;
}

// exclusive prefix of `flag` over a 256-thread block; *total = block sum
__device__ __forceinline__ int block_excl_scan_flag(bool flag, int *total) {
  __shared__ int wave_sum[4];
  unsigned long long m = __ballot(flag);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sum[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = wave_sum[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return base + pre;
}

// `n_dev` (capacity-sized builds, dm_rulebook_*_cap): the row count lives on the device, `n` is then the CAPACITY the
// launch and the table strides are sized for; NULL: `n` is the count.
__device__ __forceinline__ int rb_rows(int n, const int32_t *n_dev) {
  if (!n_dev) return n;
  const int v = *n_dev;
  return v < n ? (v < 0 ? 0 : v) : n;
}

__global__ __launch_bounds__(256) void rb_insert_inputs(const int4 *indices, int n, RbGeom g,
                                                        uint32_t *hkeys, int32_t *hvals,
                                                        int log2_size, const int32_t *n_dev) {
  int i = blockIdx.x * 256 + threadIdx.x;
  n = rb_rows(n, n_dev);
  if (i >= n) return;
  int4 c = indices[i];
  int fresh;
  uint32_t s = dm_hash_insert(hkeys, log2_size, in_key(g, c.x, c.y, c.z, c.w), &fresh);
  hvals[s] = i;  // duplicate coordinates are not a valid input (reference: last writer wins too)
}

// nbr_out[k][o] for a sub-manifold conv: the input at out + (k - ksize/2).
__global__ __launch_bounds__(256) void rb_table_subm(const int4 *indices, int n, RbGeom g,
                                                     const uint32_t *hkeys, const int32_t *hvals,
                                                     int log2_size, int32_t *nbr, int32_t *chunk_cnt,
                                                     int nchunks, const int32_t *n_dev) {
  int o = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int v = -1;
  const int cap = n;                       // table stride
  n = rb_rows(n, n_dev);
  if (o >= n && o < cap) nbr[(size_t)k * cap + o] = -1;      // rows of the capacity nobody owns
  if (o < n) {
    int4 c = indices[o];
    int kk[3];
    decode_k(g, k, kk);
    int z = c.y + kk[0] - g.pad[0], y = c.z + kk[1] - g.pad[1], x = c.w + kk[2] - g.pad[2];
    if (z >= 0 && z < g.spatial[0] && y >= 0 && y < g.spatial[1] && x >= 0 && x < g.spatial[2])
      v = dm_hash_find(hkeys, hvals, log2_size, in_key(g, c.x, z, y, x));
    nbr[(size_t)k * cap + o] = v;
  }
  int total;
  block_excl_scan_flag(v >= 0, &total);
  if (threadIdx.x == 0) chunk_cnt[k * nchunks + blockIdx.x] = total;
}

// candidate outputs of a strided conv: q = (p + pad - k) / stride when divisible
__device__ __forceinline__ bool conv_out_pos(const RbGeom &g, int4 c, const int kk[3], int q[3]) {
  int p[3] = {c.y, c.z, c.w};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int t = p[a] + g.pad[a] - kk[a];
    if (t < 0) return false;
    int qq = t / g.stride[a];
    if (qq * g.stride[a] != t || qq >= g.out_shape[a]) return false;
    q[a] = qq;
  }
  return true;
}

__global__ __launch_bounds__(256) void rb_conv_candidates(const int4 *indices, int n, RbGeom g,
                                                          uint32_t *hkeys, int log2_size,
                                                          uint32_t *uniq, int32_t *counter) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  if (i >= n) return;
  int4 c = indices[i];
  int kk[3], q[3];
  decode_k(g, k, kk);
  if (!conv_out_pos(g, c, kk, q)) return;
  uint32_t key = out_key(g, c.x, q[0], q[1], q[2]);
  int fresh;
  dm_hash_insert(hkeys, log2_size, key, &fresh);
  if (fresh) uniq[atomicAdd(counter, 1)] = key;
}

__global__ __launch_bounds__(256) void rb_assign_out(const uint32_t *sorted, int n_out, RbGeom g,
                                                     const uint32_t *hkeys, int32_t *hvals,
                                                     int log2_size, int4 *out_ids) {
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_out) return;
  uint32_t key = sorted[r];
  uint32_t b = key / g.out_vol, cell = key % g.out_vol;
  int x = cell % g.out_shape[2];
  int t = cell / g.out_shape[2];
  int y = t % g.out_shape[1];
  int z = t / g.out_shape[1];
  out_ids[r] = make_int4((int)b, z, y, x);
  uint32_t mask = (1u << log2_size) - 1u;
  uint32_t s = dm_hash_slot(key, log2_size);
  while (hkeys[s] != key) s = (s + 1) & mask;
  hvals[s] = r;
}

__global__ __launch_bounds__(256) void rb_table_conv_out(const int4 *out_ids, int n_out, RbGeom g,
                                                         const uint32_t *hkeys,
                                                         const int32_t *hvals, int log2_size,
                                                         int32_t *nbr) {
  int o = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int v = -1;
  if (o < n_out) {
    int4 c = out_ids[o];
    int kk[3];
    decode_k(g, k, kk);
    int z = c.y * g.stride[0] - g.pad[0] + kk[0];
    int y = c.z * g.stride[1] - g.pad[1] + kk[1];
    int x = c.w * g.stride[2] - g.pad[2] + kk[2];
    if (z >= 0 && z < g.spatial[0] && y >= 0 && y < g.spatial[1] && x >= 0 && x < g.spatial[2])
      v = dm_hash_find(hkeys, hvals, log2_size, in_key(g, c.x, z, y, x));
    nbr[(size_t)k * n_out + o] = v;
  }
}

// nbr_in[k][i] and, per 256 inputs, how many of them have an output through offset k (the pair lists
// of a strided conv are filled in ascending INPUT row order from this table)
__global__ __launch_bounds__(256) void rb_table_conv_in(const int4 *indices, int n, RbGeom g,
                                                        const uint32_t *hkeys, const int32_t *hvals,
                                                        int log2_size, int32_t *nbr_in,
                                                        int32_t *chunk_cnt, int nchunks) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int v = -1;
  if (i < n) {
    int4 c = indices[i];
    int kk[3], q[3];
    decode_k(g, k, kk);
    if (conv_out_pos(g, c, kk, q))
      v = dm_hash_find(hkeys, hvals, log2_size, out_key(g, c.x, q[0], q[1], q[2]));
    nbr_in[(size_t)k * n + i] = v;
  }
  int total;
  block_excl_scan_flag(v >= 0, &total);
  if (threadIdx.x == 0) chunk_cnt[k * nchunks + blockIdx.x] = total;
}

// per kernel offset: exclusive scan of the chunk counts, total -> indice_num[k]
__global__ __launch_bounds__(256) void rb_scan_chunks(const int32_t *chunk_cnt, int nchunks,
                                                      int32_t *chunk_off, int32_t *indice_num) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  int k = blockIdx.x;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nchunks; base += 256) {
    int i = base + threadIdx.x;
    int v = i < nchunks ? chunk_cnt[k * nchunks + i] : 0;
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int j = 0; j < w; ++j) wbase += wsum[j];
    int carry = carry_s;
    if (i < nchunks) chunk_off[k * nchunks + i] = carry + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) indice_num[k] = carry_s;
}

// Pair lists from a gather table in ONE launch: each block sums the chunk counts before it (its slot
// offset) and all of them (indice_num[k]) — a few hundred int32 per kernel offset — scans its own 256
// rows, writes their pairs and takes its share of the -1 padding of slots [indice_num[k], pair_stride).
// pair_stride == 0: counts only.
__global__ __launch_bounds__(256) void rb_fill_pairs(const int32_t *nbr, int n_rows,
                                                     const int32_t *chunk_cnt, int nchunks,
                                                     int32_t *pairs, int pair_stride,
                                                     int32_t *indice_num, int rows_are_inputs) {
  __shared__ int red[2][4];
  const int k = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int before = 0, all = 0;
  for (int i = threadIdx.x; i < nchunks; i += 256) {
    const int c = chunk_cnt[k * nchunks + i];
    all += c;
    if (i < (int)blockIdx.x) before += c;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d), all += __shfl_xor(all, d);
  if (lane == 0) red[0][w] = before, red[1][w] = all;
  __syncthreads();
  before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  all = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (blockIdx.x == 0 && threadIdx.x == 0) indice_num[k] = all;
  if (pair_stride <= 0) return;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int v = o < n_rows ? nbr[(size_t)k * n_rows + o] : -1;
  int total;
  const int pos = block_excl_scan_flag(v >= 0, &total);
  int32_t *p_in = pairs + ((size_t)k * 2 + 0) * pair_stride, *p_out = pairs + ((size_t)k * 2 + 1) * pair_stride;
  if (v >= 0) {
    p_in[before + pos] = rows_are_inputs ? o : v;
    p_out[before + pos] = rows_are_inputs ? v : o;
  }
  // padding of the unused slots, spread over the blocks of this offset
  for (int sl = all + blockIdx.x * 256 + threadIdx.x; sl < pair_stride; sl += gridDim.x * 256)
    p_in[sl] = -1, p_out[sl] = -1;
}

__global__ __launch_bounds__(256) void rb_pairs_to_table(const int32_t *pairs,
                                                         const int32_t *indice_num,
                                                         int pair_stride, int side, int32_t *table,
                                                         int n_rows) {
  int k = blockIdx.y;
  int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= indice_num[k]) return;
  int in = pairs[((size_t)k * 2 + 0) * pair_stride + s];
  int out = pairs[((size_t)k * 2 + 1) * pair_stride + s];
  if (side) table[(size_t)k * n_rows + out] = in;
  else table[(size_t)k * n_rows + in] = out;
}


// ---- strided conv, occupancy-bitmap path --------------------------------------------------------
// One bit per OUTPUT cell (batch * out_vol bits; 3 MB for KITTI's spconv2 at B = 2) replaces the
// candidate hash + radix sort: the ascending flat cell id the reference orders outputs by
// (indice_cuda.cu:64-70, torch::_unique) IS the rank of a set bit, so
//   mark (atomicOr) -> per-1024-word popcount prefix -> prefix of the block sums -> n_out
// is the whole count phase (memset, mark, scan, block prefix: 4 launches), and in the fill phase (emit, tables,
// pair lists: 3 launches) nbr_in[k][i] is a rank lookup (two prefix reads + one popcount) and
// nbr_out[k][rank] = i its scatter (for one offset k an output has at most one input).  No input
// hash, no sort, no per-pair atomics; same results bit for bit.
// One thread per (input, offset).  Device-scope atomics execute at the memory side of the fabric, so
// they are made few: neighbouring lanes (neighbouring inputs, same offset) mostly hit the same 32-cell
// word, a segmented OR over each run of equal words leaves ONE atomicOr per run, and a word whose bits
// are already seen set is skipped (a stale read only costs a redundant atomic).
__global__ __launch_bounds__(256) void rb_bitmap_mark(const int4 *indices, int n, RbGeom g,
                                                      uint32_t *bits, const int32_t *n_dev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  n = rb_rows(n, n_dev);
  const int k = blockIdx.y;
  const int lane = threadIdx.x & 63;
  uint32_t word = 0xffffffffu, m = 0;
  if (i < n) {
    const int4 c = indices[i];
    int kk[3], q[3];
    decode_k(g, k, kk);
    if (conv_out_pos(g, c, kk, q)) {
      const uint32_t key = out_key(g, c.x, q[0], q[1], q[2]);
      word = key >> 5, m = 1u << (key & 31);
    }
  }
  if (__ballot(m != 0) == 0) return;
  const uint32_t prev = __shfl_up(word, 1), next = __shfl_down(word, 1);
  bool head = lane == 0 || prev != word;
  const bool tail = lane == 63 || next != word;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t mp = __shfl_up(m, d);
    const int hp = __shfl_up((int)head, d);
    if (lane >= d && !head) m |= mp, head = hp != 0;
  }
  if (tail && m != 0 && (bits[word] & m) != m) atomicOr(&bits[word], m);
}

// 1024 words per block: word_pre[w] = set bits in the block's words before w, blk_sum[b] = the block's
// bits (their exclusive prefix blk_off and the total n_out come from one 256-thread rb_scan_chunks block:
// a same-address atomic per block costs more than that launch)
__global__ __launch_bounds__(256) void rb_bitmap_scan(const uint32_t *bits, int32_t *word_pre,
                                                      int32_t *blk_sum) {
  __shared__ int wsum[4];
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 b = ((const uint4 *)bits)[q];
  const int c0 = __popc(b.x), c1 = __popc(b.y), c2 = __popc(b.z), c3 = __popc(b.w);
  const int t = c0 + c1 + c2 + c3;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int u = __shfl_up(incl, d);
    if (lane >= d) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
  for (int j = 0; j < w; ++j) base += wsum[j];
  const int e = base + incl - t;
  ((int4 *)word_pre)[q] = make_int4(e, e + c0, e + c0 + c1, e + c0 + c1 + c2);
  if (threadIdx.x == 255) blk_sum[blockIdx.x] = base + incl;
}

__device__ __forceinline__ int4 out_cell(const RbGeom &g, uint32_t key) {
  uint32_t b = key / g.out_vol, cell = key % g.out_vol;
  int x = cell % g.out_shape[2];
  int t = cell / g.out_shape[2];
  return make_int4((int)b, t / g.out_shape[1], t % g.out_shape[1], x);
}

// One thread per BYTE of the bitmap (a lane walking a whole word of a dense region serialises 32
// dependent stores): every set bit becomes one row of out_ids in ascending cell order — one full decode
// per non-empty byte, then +1 in x with carries per bit.  The blocks launched behind the bitmap's own
// do the -1 fill of nbr_out that rb_bitmap_tables scatters into.
__global__ __launch_bounds__(256) void rb_bitmap_emit(const uint32_t *bits, const int32_t *word_pre,
                                                      const int32_t *blk_off, RbGeom g, int4 *out_ids,
                                                      int n_out, int32_t *nbr_out, size_t nbr_words,
                                                      int nblk_emit) {
  if ((int)blockIdx.x >= nblk_emit) {
    const size_t f = (size_t)(blockIdx.x - nblk_emit) * 256 + threadIdx.x, step = (size_t)(gridDim.x - nblk_emit) * 256;
    if (((uintptr_t)nbr_out & 15) == 0) {
      const int4 neg = make_int4(-1, -1, -1, -1);
      const size_t quads = nbr_words / 4;
      for (size_t i = f; i < quads; i += step) ((int4 *)nbr_out)[i] = neg;
      if (f < (nbr_words & 3)) nbr_out[quads * 4 + f] = -1;
    } else {
      for (size_t i = f; i < nbr_words; i += step) nbr_out[i] = -1;
    }
    return;
  }
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;      // 64 words per block
  const size_t w = t >> 2;
  const int sh = (int)(t & 3) * 8;
  const uint32_t word = bits[w];
  uint32_t b = (word >> sh) & 0xffu;
  if (!b) return;
  int r = blk_off[w >> 10] + word_pre[w] + __popc(word & ((1u << sh) - 1u));
  const int4 c0 = out_cell(g, (uint32_t)(w * 32 + sh));
  while (b) {
    const int bit = __ffs((int)b) - 1;
    b &= b - 1;
    int4 c = c0;
    c.w += bit;
    while (c.w >= g.out_shape[2]) {
      c.w -= g.out_shape[2];
      if (++c.z >= g.out_shape[1]) {
        c.z = 0;
        if (++c.y >= g.out_shape[0]) c.y = 0, ++c.x;
      }
    }
    if (r < n_out) out_ids[r] = c;
    ++r;
  }
}

// nbr_in[k][i] = rank of the output cell, nbr_out[k][rank] = i, and the per-256-input pair counts
__global__ __launch_bounds__(256) void rb_bitmap_tables(const int4 *indices, int n, RbGeom g,
                                                        const uint32_t *bits, const int32_t *word_pre,
                                                        const int32_t *blk_off, int32_t *nbr_in,
                                                        int32_t *nbr_out, int n_out,
                                                        int32_t *chunk_cnt, int nchunks, const int32_t *n_dev) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int v = -1;
  const int cap = n;                       // stride of nbr_in
  n = rb_rows(n, n_dev);
  if (i >= n && i < cap) nbr_in[(size_t)k * cap + i] = -1;
  if (i < n) {
    int4 c = indices[i];
    int kk[3], q[3];
    decode_k(g, k, kk);
    if (conv_out_pos(g, c, kk, q)) {
      const uint32_t key = out_key(g, c.x, q[0], q[1], q[2]);
      const uint32_t w = key >> 5;
      v = blk_off[w >> 10] + word_pre[w] + __popc(bits[w] & ((1u << (key & 31)) - 1u));
      if (v < n_out) nbr_out[(size_t)k * n_out + v] = i;
      else v = -1;                                 // n_out smaller than the count phase reported
    }
    nbr_in[(size_t)k * cap + i] = v;
  }
  int total;
  block_excl_scan_flag(v >= 0, &total);
  if (threadIdx.x == 0) chunk_cnt[k * nchunks + blockIdx.x] = total;
}

struct RbWorkspace {
  uint32_t *ha_keys;
  int32_t *ha_vals;
  int log2_a;
  uint32_t *hb_keys;
  int32_t *hb_vals;
  int log2_b;
  uint32_t *uniq;
  uint32_t *sorted;
  int32_t *counter;
  int32_t *chunk_cnt;
  int32_t *chunk_off;
  void *sort_tmp;
  size_t sort_tmp_bytes;
  size_t total;
};

size_t sort_tmp_bound(size_t n) { return dm_align(n * 4 * 2 + (1u << 20)); }

RbWorkspace carve(void *ws, size_t ws_bytes, int n, int kvol) {
  RbWorkspace w;
  DmArena a(ws, ws_bytes);
  size_t cand = (size_t)n * kvol;
  uint32_t ha = dm_pow2_ceil(2ull * (n > 0 ? n : 1));
  if (ha < 1024) ha = 1024;
  uint32_t hb = dm_pow2_ceil(2ull * (cand > 0 ? cand : 1));
  if (hb < 1024) hb = 1024;
  w.log2_a = dm_log2(ha);
  w.log2_b = dm_log2(hb);
  w.ha_keys = a.take<uint32_t>(ha);
  w.ha_vals = a.take<int32_t>(ha);
  w.hb_keys = a.take<uint32_t>(hb);
  w.hb_vals = a.take<int32_t>(hb);
  w.uniq = a.take<uint32_t>(cand + 1);
  w.sorted = a.take<uint32_t>(cand + 1);
  w.counter = a.take<int32_t>(64);
  size_t nchunks = (cand + 255) / 256 + 1;
  w.chunk_cnt = a.take<int32_t>(nchunks * kvol);
  w.chunk_off = a.take<int32_t>(nchunks * kvol);
  w.sort_tmp_bytes = sort_tmp_bound(cand);
  w.sort_tmp = a.take<char>(w.sort_tmp_bytes);
  w.total = a.off;
  return w;
}

int make_geom(RbGeom *g, int batch, const int *spatial, const int *out_shape, const int *ksize,
              const int *stride, const int *pad) {
  long long iv = 1, ov = 1;
  g->kvol = 1;
  for (int a = 0; a < 3; ++a) {
    g->spatial[a] = spatial[a];
    g->out_shape[a] = out_shape[a];
    g->ksize[a] = ksize[a];
    g->stride[a] = stride[a];
    g->pad[a] = pad[a];
    if (spatial[a] <= 0 || out_shape[a] <= 0 || ksize[a] <= 0 || stride[a] <= 0) return DM_ERR_INVALID_ARG;
    iv *= spatial[a];
    ov *= out_shape[a];
    g->kvol *= ksize[a];
  }
  // same limit as the reference's int32 flat cell id (indice.cu.h:59-60)
  if (iv * batch > 0x7fffffffLL || ov * batch > 0x7fffffffLL) return DM_ERR_INT32_RANGE;
  g->in_vol = (uint32_t)iv;
  g->out_vol = (uint32_t)ov;
  return DM_OK;
}


// 0: bitmap path whenever its arrays fit the workspace region of the hash path's candidate
// structures (dm_rulebook_workspace_bytes is a function of n_in and kvol only); 1: hash + sort always
// (developer A/B switch and the path of tiny inputs on huge grids); 2: bitmap or DM_ERR_WORKSPACE
int g_rb_mode = 0;

struct RbBitmap {
  uint32_t *bits;
  int32_t *word_pre;
  int32_t *blk_sum;
  int32_t *blk_off;
  size_t words;       // multiple of 1024
  int nblk;
  bool fits;
};

RbBitmap carve_bitmap(const RbWorkspace &w, const RbGeom &g, int batch, int n) {
  RbBitmap b;
  const unsigned long long cells = (unsigned long long)g.out_vol * batch;
  b.words = (size_t)((cells + 32767) / 32768) * 1024;
  b.nblk = (int)(b.words / 1024);
  // the region between the input hash and the chunk counters: hb_keys, hb_vals, uniq, sorted
  char *lo = (char *)w.hb_keys, *hi = (char *)w.counter;
  DmArena a(lo, (size_t)(hi - lo));
  b.bits = a.take<uint32_t>(b.words);
  b.word_pre = a.take<int32_t>(b.words);
  b.blk_sum = a.take<int32_t>(b.nblk);
  b.blk_off = a.take<int32_t>(b.nblk);
  b.fits = a.off <= (size_t)(hi - lo);
  (void)n;
  return b;
}

int pairs_from_table(const int32_t *nbr, int n_rows, int kvol, int32_t *chunk_cnt,
                     int32_t *chunk_off, int nchunks, int32_t *pairs, int pair_stride,
                     int32_t *indice_num, hipStream_t st, int rows_are_inputs = 0) {
  (void)chunk_off;
  const bool lists = pairs != nullptr && pair_stride > 0;
  if (lists && n_rows <= 0) DM_HIP(hipMemsetAsync(pairs, 0xff, (size_t)kvol * 2 * pair_stride * sizeof(int32_t), st));
  if (n_rows <= 0 || nchunks <= 0) {
    DM_HIP(hipMemsetAsync(indice_num, 0, kvol * sizeof(int32_t), st));
    return DM_OK;
  }
  rb_fill_pairs<<<dim3(lists ? nchunks : 1, kvol), 256, 0, st>>>(nbr, n_rows, chunk_cnt, nchunks, pairs,
                                                                 lists ? pair_stride : 0, indice_num,
                                                                 rows_are_inputs);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

}  // namespace

extern "C" size_t dm_rulebook_workspace_bytes(int n_in, int kvol) {
  if (n_in < 0 || kvol <= 0) return 0;
  return carve(nullptr, 0, n_in, kvol).total;
}

extern "C" int dm_rulebook_subm(const int32_t *indices, int n, int batch,
                                const int *spatial_shape_host, const int *ksize_host,
                                int32_t *nbr_out, int32_t *indice_pairs, int32_t *indice_num,
                                void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n < 0 || batch <= 0 || !indice_num) return DM_ERR_INVALID_ARG;
  RbGeom g;
  int one[3] = {1, 1, 1};
  int pad[3] = {ksize_host[0] / 2, ksize_host[1] / 2, ksize_host[2] / 2};  // spconv_ops.h:76-79
  int rc = make_geom(&g, batch, spatial_shape_host, spatial_shape_host, ksize_host, one, pad);
  if (rc) return rc;
  if (n == 0) {
    DM_HIP(hipMemsetAsync(indice_num, 0, g.kvol * sizeof(int32_t), st));
    return DM_OK;
  }
  if (!indices || !nbr_out || !workspace) return DM_ERR_INVALID_ARG;
  RbWorkspace w = carve(workspace, workspace_bytes, n, g.kvol);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  DM_HIP(hipMemsetAsync(w.ha_keys, 0xff, sizeof(uint32_t) << w.log2_a, st));
  int nb = dm_ceil_div(n, 256);
  rb_insert_inputs<<<nb, 256, 0, st>>>((const int4 *)indices, n, g, w.ha_keys, w.ha_vals,
                                       w.log2_a, nullptr);
  DM_CHECK_LAUNCH();
  rb_table_subm<<<dim3(nb, g.kvol), 256, 0, st>>>((const int4 *)indices, n, g, w.ha_keys,
                                                  w.ha_vals, w.log2_a, nbr_out, w.chunk_cnt, nb, nullptr);
  DM_CHECK_LAUNCH();
  return pairs_from_table(nbr_out, n, g.kvol, w.chunk_cnt, w.chunk_off, nb, indice_pairs, n,
                          indice_num, st);
}

extern "C" int dm_rulebook_conv_count(const int32_t *indices, int n, int batch,
                                      const int *spatial_shape_host, const int *out_shape_host,
                                      const int *ksize_host, const int *stride_host,
                                      const int *padding_host, int32_t *n_out_dev,
                                      void *workspace, size_t workspace_bytes,
                                      dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n < 0 || batch <= 0 || !n_out_dev) return DM_ERR_INVALID_ARG;
  RbGeom g;
  int rc = make_geom(&g, batch, spatial_shape_host, out_shape_host, ksize_host, stride_host,
                     padding_host);
  if (rc) return rc;
  if (n == 0) {
    DM_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int32_t), st));
    return DM_OK;
  }
  if (!indices || !workspace) return DM_ERR_INVALID_ARG;
  RbWorkspace w = carve(workspace, workspace_bytes, n, g.kvol);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  int nb = dm_ceil_div(n, 256);
  if (g_rb_mode != 1) {
    RbBitmap bm = carve_bitmap(w, g, batch, n);
    if (bm.fits) {
      DM_HIP(hipMemsetAsync(bm.bits, 0, bm.words * sizeof(uint32_t), st));
      rb_bitmap_mark<<<dim3(nb, g.kvol), 256, 0, st>>>((const int4 *)indices, n, g, bm.bits, nullptr);
      DM_CHECK_LAUNCH();
      rb_bitmap_scan<<<bm.nblk, 256, 0, st>>>(bm.bits, bm.word_pre, bm.blk_sum);
      DM_CHECK_LAUNCH();
      rb_scan_chunks<<<1, 256, 0, st>>>(bm.blk_sum, bm.nblk, bm.blk_off, n_out_dev);
      DM_CHECK_LAUNCH();
      return DM_OK;
    }
    if (g_rb_mode == 2) return DM_ERR_WORKSPACE;
  }
  DM_HIP(hipMemsetAsync(w.ha_keys, 0xff, sizeof(uint32_t) << w.log2_a, st));
  DM_HIP(hipMemsetAsync(w.hb_keys, 0xff, sizeof(uint32_t) << w.log2_b, st));
  DM_HIP(hipMemsetAsync(w.counter, 0, sizeof(int32_t), st));
  rb_insert_inputs<<<nb, 256, 0, st>>>((const int4 *)indices, n, g, w.ha_keys, w.ha_vals,
                                       w.log2_a, nullptr);
  DM_CHECK_LAUNCH();
  rb_conv_candidates<<<dim3(nb, g.kvol), 256, 0, st>>>((const int4 *)indices, n, g, w.hb_keys,
                                                       w.log2_b, w.uniq, w.counter);
  DM_CHECK_LAUNCH();
  DM_HIP(hipMemcpyAsync(n_out_dev, w.counter, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return DM_OK;
}

extern "C" int dm_rulebook_conv_fill(const int32_t *indices, int n, int batch,
                                     const int *spatial_shape_host, const int *out_shape_host,
                                     const int *ksize_host, const int *stride_host,
                                     const int *padding_host, int n_out, int32_t *out_ids,
                                     int32_t *nbr_out, int32_t *nbr_in, int32_t *indice_pairs,
                                     int32_t *indice_num, void *workspace,
                                     size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n < 0 || batch <= 0 || n_out < 0 || !indice_num) return DM_ERR_INVALID_ARG;
  RbGeom g;
  int rc = make_geom(&g, batch, spatial_shape_host, out_shape_host, ksize_host, stride_host,
                     padding_host);
  if (rc) return rc;
  if (n == 0 || n_out == 0) {
    DM_HIP(hipMemsetAsync(indice_num, 0, g.kvol * sizeof(int32_t), st));
    if (indice_pairs && n > 0)
      DM_HIP(hipMemsetAsync(indice_pairs, 0xff, (size_t)g.kvol * 2 * n * sizeof(int32_t), st));
    return DM_OK;
  }
  if (!indices || !out_ids || !nbr_out || !nbr_in || !workspace) return DM_ERR_INVALID_ARG;
  if ((size_t)n_out > (size_t)n * g.kvol) return DM_ERR_INVALID_ARG;
  RbWorkspace w = carve(workspace, workspace_bytes, n, g.kvol);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  if (g_rb_mode != 1) {
    RbBitmap bm = carve_bitmap(w, g, batch, n);
    if (bm.fits) {      // same decision as the count phase that filled the bitmap in this workspace
      const int nbo = dm_ceil_div(n_out, 256), nbi = dm_ceil_div(n, 256);
      const size_t nbr_words = (size_t)g.kvol * n_out;
      const int fill_blocks = (int)std::min<size_t>(2048, dm_ceil_div(nbr_words / 4 + 1, 256));
      const int emit_blocks = (int)(bm.words / 64);
      rb_bitmap_emit<<<emit_blocks + fill_blocks, 256, 0, st>>>(bm.bits, bm.word_pre, bm.blk_off, g, (int4 *)out_ids,
                                                                n_out, nbr_out, nbr_words, emit_blocks);
      DM_CHECK_LAUNCH();
      rb_bitmap_tables<<<dim3(nbi, g.kvol), 256, 0, st>>>((const int4 *)indices, n, g, bm.bits, bm.word_pre,
                                                         bm.blk_off, nbr_in, nbr_out, n_out, w.chunk_cnt, nbi, nullptr);
      DM_CHECK_LAUNCH();
      return pairs_from_table(nbr_in, n, g.kvol, w.chunk_cnt, w.chunk_off, nbi, indice_pairs, n, indice_num, st, 1);
    }
    if (g_rb_mode == 2) return DM_ERR_WORKSPACE;
  }
  // ascending flat cell id == the reference GPU output order (torch::_unique)
  int end_bit = 1;
  while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)g.out_vol * batch) ++end_bit;
  size_t need = 0;
  DM_HIP(rocprim::radix_sort_keys(nullptr, need, w.uniq, w.sorted, (size_t)n_out, 0, end_bit, st));
  if (need > w.sort_tmp_bytes) return DM_ERR_WORKSPACE;
  need = w.sort_tmp_bytes;
  DM_HIP(rocprim::radix_sort_keys(w.sort_tmp, need, w.uniq, w.sorted, (size_t)n_out, 0, end_bit,
                                  st));
  int nbo = dm_ceil_div(n_out, 256), nbi = dm_ceil_div(n, 256);
  rb_assign_out<<<nbo, 256, 0, st>>>(w.sorted, n_out, g, w.hb_keys, w.hb_vals, w.log2_b,
                                     (int4 *)out_ids);
  DM_CHECK_LAUNCH();
  rb_table_conv_out<<<dim3(nbo, g.kvol), 256, 0, st>>>((const int4 *)out_ids, n_out, g,
                                                       w.ha_keys, w.ha_vals, w.log2_a, nbr_out);
  DM_CHECK_LAUNCH();
  rb_table_conv_in<<<dim3(nbi, g.kvol), 256, 0, st>>>((const int4 *)indices, n, g, w.hb_keys,
                                                      w.hb_vals, w.log2_b, nbr_in, w.chunk_cnt, nbi);
  DM_CHECK_LAUNCH();
  return pairs_from_table(nbr_in, n, g.kvol, w.chunk_cnt, w.chunk_off, nbi, indice_pairs, n, indice_num, st, 1);
}

// ---- capacity-sized builds: row counts stay on the device -----------------------------------------------------------
// Same kernels, launched over the CAPACITY of the input; the count is read from `n_dev` by the kernels.  Every table has
// the capacity as its row stride (nbr_out (kvol, cap_out), nbr_in / indice_pairs over cap_in); rows beyond the count are
// -1 in the tables and untouched in out_ids.  A strided build writes *n_out_dev; should it exceed cap_out, the tables hold
// the first cap_out outputs only and the caller (who reads the count when it chooses to) rebuilds that layer the
// two-phase way.  Bitmap path only (DM_ERR_WORKSPACE when the bitmap does not fit: the caller falls back).
extern "C" int dm_rulebook_subm_cap(const int32_t *indices, const int32_t *n_dev, int cap, int batch,
                                    const int *spatial_shape_host, const int *ksize_host, int32_t *nbr_out,
                                    int32_t *indice_pairs, int32_t *indice_num, void *workspace,
                                    size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (cap <= 0 || batch <= 0 || !indice_num || !n_dev || !indices || !nbr_out || !workspace) return DM_ERR_INVALID_ARG;
  RbGeom g;
  int one[3] = {1, 1, 1};
  int pad[3] = {ksize_host[0] / 2, ksize_host[1] / 2, ksize_host[2] / 2};
  int rc = make_geom(&g, batch, spatial_shape_host, spatial_shape_host, ksize_host, one, pad);
  if (rc) return rc;
  RbWorkspace w = carve(workspace, workspace_bytes, cap, g.kvol);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  DM_HIP(hipMemsetAsync(w.ha_keys, 0xff, sizeof(uint32_t) << w.log2_a, st));
  int nb = dm_ceil_div(cap, 256);
  rb_insert_inputs<<<nb, 256, 0, st>>>((const int4 *)indices, cap, g, w.ha_keys, w.ha_vals, w.log2_a, n_dev);
  DM_CHECK_LAUNCH();
  rb_table_subm<<<dim3(nb, g.kvol), 256, 0, st>>>((const int4 *)indices, cap, g, w.ha_keys, w.ha_vals, w.log2_a,
                                                  nbr_out, w.chunk_cnt, nb, n_dev);
  DM_CHECK_LAUNCH();
  return pairs_from_table(nbr_out, cap, g.kvol, w.chunk_cnt, w.chunk_off, nb, indice_pairs, cap, indice_num, st);
}

extern "C" int dm_rulebook_conv_cap(const int32_t *indices, const int32_t *n_dev, int cap_in, int batch,
                                    const int *spatial_shape_host, const int *out_shape_host,
                                    const int *ksize_host, const int *stride_host, const int *padding_host,
                                    int cap_out, int32_t *n_out_dev, int32_t *out_ids, int32_t *nbr_out,
                                    int32_t *nbr_in, int32_t *indice_pairs, int32_t *indice_num,
                                    void *workspace, size_t workspace_bytes, dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (cap_in <= 0 || cap_out <= 0 || batch <= 0 || !n_dev || !n_out_dev || !indices || !out_ids || !nbr_out ||
      !nbr_in || !indice_num || !workspace)
    return DM_ERR_INVALID_ARG;
  RbGeom g;
  int rc = make_geom(&g, batch, spatial_shape_host, out_shape_host, ksize_host, stride_host, padding_host);
  if (rc) return rc;
  RbWorkspace w = carve(workspace, workspace_bytes, cap_in, g.kvol);
  if (w.total > workspace_bytes) return DM_ERR_WORKSPACE;
  if (g_rb_mode == 1) return DM_ERR_WORKSPACE;
  RbBitmap bm = carve_bitmap(w, g, batch, cap_in);
  if (!bm.fits) return DM_ERR_WORKSPACE;
  const int nbi = dm_ceil_div(cap_in, 256);
  DM_HIP(hipMemsetAsync(bm.bits, 0, bm.words * sizeof(uint32_t), st));
  rb_bitmap_mark<<<dim3(nbi, g.kvol), 256, 0, st>>>((const int4 *)indices, cap_in, g, bm.bits, n_dev);
  DM_CHECK_LAUNCH();
  rb_bitmap_scan<<<bm.nblk, 256, 0, st>>>(bm.bits, bm.word_pre, bm.blk_sum);
  DM_CHECK_LAUNCH();
  rb_scan_chunks<<<1, 256, 0, st>>>(bm.blk_sum, bm.nblk, bm.blk_off, n_out_dev);
  DM_CHECK_LAUNCH();
  const size_t nbr_words = (size_t)g.kvol * cap_out;
  const int fill_blocks = (int)std::min<size_t>(2048, dm_ceil_div(nbr_words / 4 + 1, 256));
  const int emit_blocks = (int)(bm.words / 64);
  rb_bitmap_emit<<<emit_blocks + fill_blocks, 256, 0, st>>>(bm.bits, bm.word_pre, bm.blk_off, g, (int4 *)out_ids,
                                                            cap_out, nbr_out, nbr_words, emit_blocks);
  DM_CHECK_LAUNCH();
  rb_bitmap_tables<<<dim3(nbi, g.kvol), 256, 0, st>>>((const int4 *)indices, cap_in, g, bm.bits, bm.word_pre,
                                                     bm.blk_off, nbr_in, nbr_out, cap_out, w.chunk_cnt, nbi, n_dev);
  DM_CHECK_LAUNCH();
  return pairs_from_table(nbr_in, cap_in, g.kvol, w.chunk_cnt, w.chunk_off, nbi, indice_pairs, cap_in, indice_num, st, 1);
}

extern "C" int dm_rulebook_set_mode(int mode) {
  if (mode < 0 || mode > 2) return DM_ERR_INVALID_ARG;
  g_rb_mode = mode;
  return DM_OK;
}

extern "C" int dm_pairs_to_table(const int32_t *indice_pairs, const int32_t *indice_num, int kvol,
                                 int pair_stride, int side, int32_t *table, int n_rows,
                                 dm_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kvol <= 0 || n_rows < 0 || pair_stride < 0) return DM_ERR_INVALID_ARG;
  if (n_rows == 0) return DM_OK;
  if (!table || !indice_num) return DM_ERR_INVALID_ARG;
  DM_HIP(hipMemsetAsync(table, 0xff, (size_t)kvol * n_rows * sizeof(int32_t), st));
  if (pair_stride == 0) return DM_OK;
  if (!indice_pairs) return DM_ERR_INVALID_ARG;
  rb_pairs_to_table<<<dim3(dm_ceil_div(pair_stride, 256), kvol), 256, 0, st>>>(
      indice_pairs, indice_num, pair_stride, side, table, n_rows);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
