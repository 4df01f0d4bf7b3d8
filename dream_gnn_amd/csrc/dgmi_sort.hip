// dgmi_sort.hip — the stable LSD radix sort behind the device COO -> CSR build (f1), hand-written for gfx950.
//
// What is sorted is fixed: E records (row, e, col) by the low `bits` bits of `row`, stably (e is the input
// position, so inside a row the edges keep their input order — the in-row summation order of every kernel, as
// DGL's COO -> CSR gives it; augmentation.py:65, data_loader.py:448).  That fixes the shape of the sort:
//   * the column ids travel WITH the records (12 B each) instead of being gathered through the permutation
//     afterwards (10 M random 4-byte loads: 122 us of the 0.45 ms the library-sort build took);
//   * digits are up to 9 bits wide: a config-4 graph (50 k - 100 k rows, 16-17 bits) is TWO passes, not three;
//   * a pass is three launches — tile histograms, one scan, scatter — with 8192-record tiles (one 16-wave
//     workgroup): a wave ranks its records with ballot matches (no atomics: the rank inside a digit is the input
//     order by construction), the tile is put in digit order in LDS and written out in runs (avg 16-32 records per
//     digit and tile), so the stores coalesce;
//   * iota and the id range check ride on the first pass, which reads row / col anyway.
// Out-of-range ids only ever contribute their low bits: every store stays inside the E-record buffers whatever the
// input holds (the range flag tells the caller the result is meaningless).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"
#include "dgmi_tuning.h"

namespace dgmi {
namespace {

constexpr int kSortThreads = 1024;                    // 16 waves; two workgroups per CU (68 KB of LDS each at 9-bit digits)
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kItems = 8;                             // records per thread
constexpr int kTile = kSortThreads * kItems;          // 8192 records per workgroup
constexpr int kMaxDigitBits = 9;
constexpr int kMaxBuckets = 1 << kMaxDigitBits;

struct SortPass {
  int shift, bits;  // digit = (key >> shift) & ((1 << bits) - 1)
  int per_xcd;      // > 0: XCD-aware tile order — block b takes tile (b % 8) * per_xcd + b / 8 (see tile_of)
};

// Which tile a workgroup takes.  A digit's run of tile t is followed, in the output, by the same digit's run of tile t + 1,
// and a run is ~16 records = 64 B at an arbitrary offset: most 128-B lines of the output are shared by the runs of two
// CONSECUTIVE tiles.  Workgroups are dealt round-robin over the 8 XCDs, so with tile = blockIdx.x the two halves of such a
// line are written from two different (non-coherent, write-back) L2s and reach memory as two partial writes.  With
// per_xcd = ceil(n_tiles / 8), XCD x works through the contiguous tile range [x * per_xcd, (x + 1) * per_xcd) in order:
// the neighbouring runs meet in ONE L2 and leave it as whole lines.  Placement is a speed assumption only.
__device__ __forceinline__ int tile_of(const SortPass& ps, int n_tiles) {
  if (ps.per_xcd <= 0) return (int)blockIdx.x;
  const int t = (int)(blockIdx.x & 7u) * ps.per_xcd + (int)(blockIdx.x >> 3);
  return t < n_tiles && (int)(blockIdx.x >> 3) < ps.per_xcd ? t : -1;
}

// tile histogram of one pass: hist[bucket * n_tiles + tile]
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const int32_t* __restrict__ keys, int64_t E, SortPass ps,
                                                                 int n_tiles, int32_t* __restrict__ hist, const int32_t* __restrict__ col,
                                                                 int32_t n_rows, int32_t n_cols, int32_t* __restrict__ flag) {
  __shared__ int cnt[kMaxBuckets];
  const int nb = 1 << ps.bits;
  const int tile = tile_of(ps, n_tiles);
  if (tile < 0) return;  // block-uniform
  for (int b = threadIdx.x; b < nb; b += kSortThreads) cnt[b] = 0;
  __syncthreads();
  const int64_t base = (int64_t)tile * kTile;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = base + (int64_t)k * kSortThreads + threadIdx.x;  // any order: a histogram
    if (i < E) {
      const int32_t key = keys[i];
      if (FIRST && n_rows > 0) bad |= (key < 0) | (key >= n_rows);  // the row-id range check rides on the first read of the
                                                                    // keys; the column ids are checked where they are first read
                                                                    // anyway: the first scatter (80 -> 40 MB for this launch)
      atomicAdd(&cnt[((uint32_t)key >> ps.shift) & (uint32_t)(nb - 1)], 1);
    }
  }
  if (FIRST && bad) *flag = 1;  // benign race: every writer stores the same value
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kSortThreads) hist[(int64_t)b * n_tiles + tile] = cnt[b];
}

// Exclusive scan of hist[bucket][tile] in bucket-major order, in place, as two small launches (one workgroup per
// bucket, contiguous rows of n_tiles counts): totals per bucket, then base-of-bucket + the scan over its tiles.
__device__ __forceinline__ int block_sum(int v, int* part) {  // sum over the workgroup, returned to every thread
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < kSortWaves; ++w) t += part[w];
  return t;
}

__global__ __launch_bounds__(kSortThreads) void sort_bucket_totals_kernel(const int32_t* __restrict__ hist, int n_tiles,
                                                                          int32_t* __restrict__ totals) {
  __shared__ int part[kSortWaves];
  const int32_t* h = hist + (int64_t)blockIdx.x * n_tiles;
  int v = 0;
  for (int i = threadIdx.x; i < n_tiles; i += kSortThreads) v += h[i];
  const int t = block_sum(v, part);
  if (threadIdx.x == 0) totals[blockIdx.x] = t;
}

__global__ __launch_bounds__(kSortThreads) void sort_bucket_scan_kernel(int32_t* __restrict__ hist, int n_tiles,
                                                                        const int32_t* __restrict__ totals) {
  __shared__ int part[kSortWaves];
  __shared__ int wsum[kSortWaves];
  int v = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kSortThreads) v += totals[b];
  int carry = block_sum(v, part);  // records in the buckets before this one
  int32_t* h = hist + (int64_t)blockIdx.x * n_tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c0 = 0; c0 < n_tiles; c0 += kSortThreads) {
    const int i = c0 + threadIdx.x;
    const int x = i < n_tiles ? h[i] : 0;
    int inc = x;  // inclusive scan inside the wave
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(inc, off, 64);
      if (lane >= off) inc += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < kSortWaves; ++w) {
      if (w < wave) before += wsum[w];
      all += wsum[w];
    }
    if (i < n_tiles) h[i] = carry + before + inc - x;
    carry += all;
  }
}

// scatter of one pass.  FIRST: records come from the caller's (row, col), e = position.  LAST: eid / col go to the
// caller's arrays (keys still written: the boundary pass reads them).
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const int32_t* __restrict__ keys_in, const int32_t* __restrict__ eid_in,
                                                                    const int32_t* __restrict__ col_in, int64_t E, SortPass ps, int n_tiles,
                                                                    const int32_t* __restrict__ offs, int32_t* __restrict__ keys_out,
                                                                    int32_t* __restrict__ eid_out, int32_t* __restrict__ col_out,
                                                                    int32_t n_cols, int32_t* __restrict__ flag) {
  extern __shared__ int32_t lds[];
  int32_t* s_buf = lds;                    // [kTile] ONE field of the records in digit order (key, then eid, then col:
                                           // 32 KB instead of 96 KB of staging -> two workgroups per CU, whose phases overlap)
  int32_t* cnt = lds + kTile;              // [kSortWaves][nb] per-wave digit counts, then exclusive over waves
  const int nb = 1 << ps.bits;
  int32_t* dstart = cnt + kSortWaves * nb; // [nb] first LDS slot of a digit in this tile
  int32_t* gbase = dstart + nb;            // [nb] global position of the tile's first record of a digit
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = tile_of(ps, n_tiles);
  if (tile < 0) return;  // block-uniform
  for (int i = threadIdx.x; i < kSortWaves * nb; i += kSortThreads) cnt[i] = 0;
  __syncthreads();
  const int64_t tile0 = (int64_t)tile * kTile;
  const int64_t wave0 = tile0 + (int64_t)wave * (kItems * 64);  // a wave owns 512 CONSECUTIVE records: 8 rounds of 64
  int32_t key[kItems], eid[kItems], col[kItems], rank[kItems];
  uint32_t dig[kItems];
  int32_t* my_cnt = cnt + wave * nb;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = wave0 + k * 64 + lane;
    const bool ok = i < E;
    key[k] = ok ? keys_in[i] : 0;
    eid[k] = (FIRST && eid_in == nullptr) ? (int32_t)i : (ok ? eid_in[i] : 0);
    col[k] = ok ? col_in[i] : 0;
    dig[k] = ((uint32_t)key[k] >> ps.shift) & (uint32_t)(nb - 1);
    if (FIRST && n_cols > 0 && ((col[k] < 0) | (col[k] >= n_cols))) *flag = 1;  // benign race: every writer stores the same value
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const bool ok = wave0 + k * 64 + lane < E;
    // lanes of this round with my digit (ballot per digit bit), restricted to valid lanes
    unsigned long long m = __ballot(ok);
    for (int b = 0; b < ps.bits; ++b) {
      const unsigned long long bal = __ballot((dig[k] >> b) & 1u);
      m &= ((dig[k] >> b) & 1u) ? bal : ~bal;
    }
    const int before = my_cnt[dig[k]];                  // records of my digit in earlier rounds of this wave
    rank[k] = before + __popcll(m & lt);                // ... plus the ones before me in this round: input order
    // (the LDS pipe executes a wave's instructions in order: every lane has read before the leader adds)
    if (ok && (m & lt) == 0ull) my_cnt[dig[k]] = before + __popcll(m);
  }
  __syncthreads();
  // per digit (one thread each: nb <= 512 <= the workgroup): exclusive prefix over the waves (in place), the tile's count,
  // its global base — then the digit's first slot in the tile = exclusive scan of the counts over the digits, done by
  // the whole workgroup (wave scans + the waves' totals).  A single thread walking the 512 counts with dependent LDS
  // reads held the other 1023 threads at the barrier for most of the kernel's time: rounds 1-3, 95 us per 10 M-record pass.
  __shared__ int wtot[kSortWaves];
  int my_count = 0;
  if ((int)threadIdx.x < nb) {
    const int d = threadIdx.x;
    int run = 0;
    for (int w = 0; w < kSortWaves; ++w) {
      const int c = cnt[w * nb + d];
      cnt[w * nb + d] = run;
      run += c;
    }
    my_count = run;
    gbase[d] = offs[(int64_t)d * n_tiles + tile];
  }
  int incl = my_count;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(incl, off, 64);
    if (lane >= off) incl += y;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  if ((int)threadIdx.x < nb) {
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    dstart[threadIdx.x] = before + incl - my_count;
  }
  __syncthreads();
  int slot[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    slot[k] = dstart[dig[k]] + my_cnt[dig[k]] + rank[k];
    if (wave0 + k * 64 + lane < E) s_buf[slot[k]] = key[k];
  }
  __syncthreads();
  const int64_t left = E - tile0;
  const int n_here = left < kTile ? (int)left : kTile;
  int32_t pos[kItems];  // global position of slot threadIdx.x + k * kSortThreads (E <= INT32_MAX)
#pragma unroll
  for (int k = 0; k < kItems; ++k) {  // consecutive slots of one digit -> consecutive addresses
    const int i = threadIdx.x + k * kSortThreads;
    if (i < n_here) {
      const int32_t k2 = s_buf[i];
      const uint32_t d = ((uint32_t)k2 >> ps.shift) & (uint32_t)(nb - 1);
      pos[k] = gbase[d] + (i - dstart[d]);
      keys_out[pos[k]] = k2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if (wave0 + k * 64 + lane < E) s_buf[slot[k]] = eid[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int i = threadIdx.x + k * kSortThreads;
    if (i < n_here) eid_out[pos[k]] = s_buf[i];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if (wave0 + k * 64 + lane < E) s_buf[slot[k]] = col[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int i = threadIdx.x + k * kSortThreads;
    if (i < n_here) col_out[pos[k]] = s_buf[i];
  }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

int radix_sort_passes(int shift0, int bits, SortPassPlan* plan) {
  int passes = (bits + kMaxDigitBits - 1) / kMaxDigitBits;
  if (passes < 1) passes = 1;
  int shift = shift0, left = bits < 1 ? 1 : bits;
  for (int p = 0; p < passes; ++p) {
    const int b = (left + (passes - p) - 1) / (passes - p);
    plan->shift[p] = shift;
    plan->bits[p] = b;
    shift += b;
    left -= b;
  }
  return passes;
}

size_t radix_sort_workspace_bytes(int64_t E, int bits) {
  const int passes = bits <= kMaxDigitBits ? 1 : (bits + kMaxDigitBits - 1) / kMaxDigitBits;
  const int64_t n_tiles = (E + kTile - 1) / kTile;
  size_t b = align_up((size_t)kMaxBuckets * (size_t)(n_tiles > 0 ? n_tiles : 1) * 4, 256) + align_up(kMaxBuckets * 4, 256);  // tile histograms / offsets, bucket totals
  if (passes > 1) b += 2 * 3 * align_up((size_t)E * 4, 256);                                // two record buffers (key, eid, col)
  return b;
}

// (keys_out, eid_out, col_out) <- the records (key[i], eid[i] or i, col[i]) sorted stably by bits [shift0, shift0 + bits)
// of the key.  n_rows > 0 / n_cols > 0: the first pass also checks key in [0, n_rows) / col in [0, n_cols) into *flag.
hipError_t radix_sort_records(const int32_t* key, const int32_t* eid_in, const int32_t* col, int64_t E, int shift0, int bits,
                              int32_t n_rows, int32_t n_cols, int32_t* keys_out, int32_t* eid_out, int32_t* col_out, int32_t* flag,
                              void* workspace, hipStream_t s) {
  if (E <= 0) return hipSuccess;
  SortPassPlan plan;
  const int passes = radix_sort_passes(shift0, bits, &plan);
  const int n_tiles = (int)((E + kTile - 1) / kTile);
  char* ws = static_cast<char*>(workspace);
  int32_t* hist = reinterpret_cast<int32_t*>(ws);
  size_t off = align_up((size_t)kMaxBuckets * (size_t)n_tiles * 4, 256);
  int32_t* totals = reinterpret_cast<int32_t*>(ws + off);
  off += align_up(kMaxBuckets * 4, 256);
  const size_t arr = align_up((size_t)E * 4, 256);
  int32_t* buf[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  if (passes > 1)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j, off += arr) buf[i][j] = reinterpret_cast<int32_t*>(ws + off);
  const int32_t* kin = key;
  const int32_t* ein = eid_in;
  const int32_t* cin = col;
  for (int p = 0; p < passes; ++p) {
    const int per_xcd = tuning().sort_plain_tiles ? 0 : (n_tiles + 7) / 8;
    const SortPass ps{plan.shift[p], plan.bits[p], per_xcd};
    const unsigned n_blocks = per_xcd > 0 ? (unsigned)(8 * per_xcd) : (unsigned)n_tiles;
    const int nb = 1 << ps.bits;
    const bool last = p == passes - 1;
    int32_t* kout = last ? keys_out : buf[p & 1][0];
    int32_t* eout = last ? eid_out : buf[p & 1][1];
    int32_t* cout = last ? col_out : buf[p & 1][2];
    const size_t lds = (size_t)(kTile + kSortWaves * nb + 2 * nb) * sizeof(int32_t);
    if (p == 0) {
      hipLaunchKernelGGL((sort_hist_kernel<true>), dim3(n_blocks), dim3(kSortThreads), 0, s, kin, E, ps, n_tiles, hist, col,
                         n_rows, n_cols, flag);
    } else {
      hipLaunchKernelGGL((sort_hist_kernel<false>), dim3(n_blocks), dim3(kSortThreads), 0, s, kin, E, ps, n_tiles, hist, col,
                         n_rows, n_cols, flag);
    }
    hipLaunchKernelGGL(sort_bucket_totals_kernel, dim3((unsigned)nb), dim3(kSortThreads), 0, s, hist, n_tiles, totals);
    hipLaunchKernelGGL(sort_bucket_scan_kernel, dim3((unsigned)nb), dim3(kSortThreads), 0, s, hist, n_tiles, totals);
    if (p == 0) {
      auto kern = sort_scatter_kernel<true>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(kSortThreads), lds, s, kin, ein, cin, E, ps, n_tiles, hist, kout, eout, cout, n_cols, flag);
    } else {
      auto kern = sort_scatter_kernel<false>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(kSortThreads), lds, s, kin, ein, cin, E, ps, n_tiles, hist, kout, eout, cout, n_cols, flag);
    }
    kin = kout;
    ein = eout;
    cin = cout;
  }
  return hipGetLastError();
}

}  // namespace dgmi
