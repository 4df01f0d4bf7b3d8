// dgmi_compact.hip — (D3 / f1) a layout of the parent graph with the dropped edges REMOVED, made once per
// training step by streaming passes (no sort), for gfx950.
//
// The reference rebuilds its graphs from the kept edges every iteration (train.py:267 ->
// augmentation.py:48-65 `dgl.heterograph` of the kept subset; :114-124 a fresh sparse COO).  The SpMM kernels
// can apply the subset description on the fly (dgmi_keep.h: one hash per edge and the layout's eid stream, per
// product and per column pass), which is the right trade for the reference's own graph sizes — a launch costs
// more than the hashes.  At 10^7 edges it is not: a training step runs every layout 3 times forward and 3 times
// backward (L = 3 layers), and the hash + eid stream + parked gather slots cost each product 19-27 %.  Here the
// layout — any CSR-shaped one: (ptr, indices[, vals], eid), plain or XCD-sliced — is compacted instead:
//
//   K1  flag      bit p = keep(eid[p]) by ballot, kept edges counted per 4096-position tile     reads eid
//   K2  scan      exclusive scan of the tile counts (one workgroup)
//   K3  scatter   kept ids (and values) move to their rank; rank at every 64-position word kept  reads/writes ids
//   K4  pointers  ptr_out[k] = rank(ptr[k])                                                      reads ptr
//
// ~116 MB of streaming traffic for a 10 M-edge layout (HBM-bound integer work), after which the products run the
// plain kernels — no eid stream, no hash, column passes as usual — over 10 % fewer edges, and sum the kept edges
// in exactly the order a CSR rebuilt from them would give.  Stable: kept edges keep their relative order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_keep.h"
#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 16;
constexpr int kTile = kThreads * kPerThread;  // positions per workgroup
constexpr int kWordsPerTile = kTile / 64;     // 64: one word per lane of the wave that ranks them
constexpr int kScanThreads = 1024;
constexpr int kScanPerThread = 8;

struct Workspace {
  uint64_t* bits;       // n_words: bit (p & 63) of word p >> 6 = edge at position p survives
  int32_t* tile_count;  // n_tiles
  int32_t* tile_off;    // n_tiles + 1 (exclusive scan; last = number of survivors)
  int32_t* word_rank;   // n_words: survivors before the word's first position
};

inline int64_t n_tiles_of(int64_t nnz) { return (nnz + kTile - 1) / kTile; }
inline int64_t n_words_of(int64_t nnz) { return n_tiles_of(nnz) * kWordsPerTile; }  // whole tiles: no bounds tests on words

inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

Workspace carve(void* base, int64_t nnz) {
  char* p = static_cast<char*>(base);
  Workspace w;
  w.bits = reinterpret_cast<uint64_t*>(p);
  p += align256((size_t)n_words_of(nnz) * sizeof(uint64_t));
  w.tile_count = reinterpret_cast<int32_t*>(p);
  p += align256((size_t)n_tiles_of(nnz) * sizeof(int32_t));
  w.tile_off = reinterpret_cast<int32_t*>(p);
  p += align256((size_t)(n_tiles_of(nnz) + 1) * sizeof(int32_t));
  w.word_rank = reinterpret_cast<int32_t*>(p);
  return w;
}

// K1: one ballot per 64 positions; wave w of the block owns words w, w + 4, ... of the tile, so a lane's 16 eid loads
// are independent, coalesced 256-B rows.  (4096-position tiles: with 2048 the launch was dispatch-bound — 4883 short-lived
// workgroups; and the first description is read ONCE per wave, not per edge: 25.9 -> 14 us per 10 M edges.)
__global__ __launch_bounds__(kThreads) void compact_flag_kernel(const int32_t* __restrict__ eid, int64_t nnz,
                                                                const KeepSeg* __restrict__ keep, int n_keep,
                                                                uint64_t* __restrict__ bits, int32_t* __restrict__ tile_count) {
  __shared__ int wave_cnt[kThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t e[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int64_t p = base + j * kThreads + threadIdx.x;
    e[j] = p < nnz ? eid[p] : 0;
  }
  const KeepPre first = keep_preload(keep, n_keep);  // n_keep >= 1 (checked by the ABI): wide scalar loads once, not three dependent ones per item
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int64_t p = base + j * kThreads + threadIdx.x;
    const bool k = p < nnz && edge_kept(first, keep, n_keep, (uint32_t)e[j]);
    const unsigned long long m = __ballot(k);
    if (lane == 0) bits[(base >> 6) + j * (kThreads / 64) + wave] = m;
    cnt += __popcll(m);
  }
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// K2: exclusive scan of n counts into n + 1 offsets, one workgroup (8192 counts per sweep: a 10 M-edge layout is
// 2442 tiles — one sweep; 2^31 edges would be 64 sweeps).
__global__ __launch_bounds__(kScanThreads) void compact_scan_kernel(const int32_t* __restrict__ count, int64_t n,
                                                                    int32_t* __restrict__ off) {
  __shared__ int wave_sum[kScanThreads / 64];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += (int64_t)kScanThreads * kScanPerThread) {
    const int64_t i0 = base + (int64_t)threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
      v[j] = i0 + j < n ? count[i0 + j] : 0;
      mine += v[j];
    }
    int incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    int run = before + incl - mine;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
      if (i0 + j < n) off[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (threadIdx.x == kScanThreads - 1) carry_s = run;
    __syncthreads();
  }
  if (threadIdx.x == 0) off[n] = carry_s;
}

// K3: survivors to their rank.  Wave 0 ranks the tile's 64 words first, one per lane (and publishes those ranks for K4).
template <bool HAS_VALS>
__global__ __launch_bounds__(kThreads) void compact_scatter_kernel(const int32_t* __restrict__ indices,
                                                                   const float* __restrict__ vals, int64_t nnz,
                                                                   const uint64_t* __restrict__ bits,
                                                                   const int32_t* __restrict__ tile_off,
                                                                   int32_t* __restrict__ word_rank,
                                                                   int32_t* __restrict__ indices_out,
                                                                   float* __restrict__ vals_out) {
  __shared__ unsigned long long sb[kWordsPerTile];
  __shared__ int sr[kWordsPerTile];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    const int t = lane & (kWordsPerTile - 1);
    const unsigned long long w = bits[(base >> 6) + t];
    int incl = lane < kWordsPerTile ? __popcll(w) : 0;
    const int own = incl;
#pragma unroll
    for (int d = 1; d < kWordsPerTile; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane < kWordsPerTile) {
      const int r = tile_off[blockIdx.x] + incl - own;
      sb[t] = w;
      sr[t] = r;
      word_rank[(base >> 6) + t] = r;
    }
  }
  int32_t id[kPerThread];
  float v[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int64_t p = base + j * kThreads + threadIdx.x;
    id[j] = p < nnz ? indices[p] : 0;
    if (HAS_VALS) v[j] = p < nnz ? vals[p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int word = j * (kThreads / 64) + wave;
    const unsigned long long w = sb[word];  // bits past nnz are zero (K1)
    if ((w >> lane) & 1ull) {
      const int out = sr[word] + __popcll(w & ((1ull << lane) - 1ull));
      indices_out[out] = id[j];
      if (HAS_VALS) vals_out[out] = v[j];
    }
  }
}

// K4: where every row / segment boundary lands.
__global__ __launch_bounds__(kThreads) void compact_ptr_kernel(const int32_t* __restrict__ ptr, int64_t n_ptr, int64_t nnz,
                                                               const uint64_t* __restrict__ bits,
                                                               const int32_t* __restrict__ word_rank,
                                                               const int32_t* __restrict__ tile_off, int64_t n_tiles,
                                                               int32_t* __restrict__ ptr_out) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x; k < n_ptr; k += stride) {
    int64_t q = ptr[k];
    if (q < 0) q = 0;  // a meaningless layout (range flag set at its build) stays in bounds
    int r;
    if (q >= nnz) {
      r = tile_off[n_tiles];
    } else {
      const int64_t w = q >> 6;
      r = word_rank[w] + __popcll(bits[w] & ((1ull << (q & 63)) - 1ull));
    }
    ptr_out[k] = r;
  }
}

__global__ void zero_i32_kernel(int32_t* p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
}

}  // namespace

size_t compact_workspace_bytes(int64_t nnz) {
  const int64_t nt = n_tiles_of(nnz), nw = n_words_of(nnz);
  return align256((size_t)nw * sizeof(uint64_t)) + align256((size_t)nt * sizeof(int32_t)) +
         align256((size_t)(nt + 1) * sizeof(int32_t)) + align256((size_t)nw * sizeof(int32_t)) + 256;
}

hipError_t compact_layout_i32(const int32_t* ptr, int64_t n_ptr, const int32_t* indices, const float* vals, const int32_t* eid,
                              int64_t nnz, const void* keep, int n_keep, int32_t* ptr_out, int32_t* indices_out, float* vals_out,
                              void* workspace, hipStream_t s) {
  if (n_ptr <= 0) return hipSuccess;
  if (nnz == 0) {
    hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)((n_ptr + 255) / 256 < 1024 ? (n_ptr + 255) / 256 : 1024)), dim3(256), 0, s,
                       ptr_out, n_ptr);
    return hipGetLastError();
  }
  const Workspace w = carve(workspace, nnz);
  const int64_t nt = n_tiles_of(nnz);
  hipLaunchKernelGGL(compact_flag_kernel, dim3((unsigned)nt), dim3(kThreads), 0, s, eid, nnz, static_cast<const KeepSeg*>(keep),
                     n_keep, w.bits, w.tile_count);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, w.tile_count, nt, w.tile_off);
  if (vals != nullptr)
    hipLaunchKernelGGL(compact_scatter_kernel<true>, dim3((unsigned)nt), dim3(kThreads), 0, s, indices, vals, nnz, w.bits,
                       w.tile_off, w.word_rank, indices_out, vals_out);
  else
    hipLaunchKernelGGL(compact_scatter_kernel<false>, dim3((unsigned)nt), dim3(kThreads), 0, s, indices, vals, nnz, w.bits,
                       w.tile_off, w.word_rank, indices_out, vals_out);
  int64_t pb = (n_ptr + kThreads - 1) / kThreads;
  if (pb > 4096) pb = 4096;
  hipLaunchKernelGGL(compact_ptr_kernel, dim3((unsigned)pb), dim3(kThreads), 0, s, ptr, n_ptr, nnz, w.bits, w.word_rank, w.tile_off,
                     nt, ptr_out);
  return hipGetLastError();
}

}  // namespace dgmi
