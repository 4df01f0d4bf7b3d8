// dgmi_select.hip — exact-size uniformly random edge subset (gfx950): selection + mask.
//
// The reference drops edges every training iteration by keeping the first
// max(1, int(E*(1-p))) entries of torch.randperm(E) (augmentation.py:48-52, 114-118): a
// uniformly random subset of exactly that size.  A full random permutation is a sort of E keys;
// only the subset is needed.  Here every edge e gets the key (hash32(seed, e), e) — unique by
// construction — and the keep-th smallest key is found by SELECTION, never by sorting:
//   * lists up to 2^16 edges (the label-1 relations and kNN-4 graphs of the reference's datasets): one workgroup
//     runs a 4-pass most-significant-byte radix select on its own (256-bin LDS histogram per pass, then the ties
//     at the threshold hash) — one launch, a few microseconds;
//   * longer lists: the hash is uniform, so the threshold lies within 8 sigma of keep / E * 2^32.
//     ONE pass counts the hashes below that window and histograms the window (4096 bins, a few
//     edges per bin), a second pass lists the edges of the bin that holds the keep-th key, and one
//     thread orders those few (hash, id) pairs — two hashing passes instead of five.  If the window
//     ever misses (probability < 1e-14) or the bin overflows its list, the workgroup select above
//     takes over for that list: the result is always exact.
// The result is the 8-word description of dgmi_keep.h (seed, threshold hash, tie cut): the SpMM
// kernels evaluate it per edge while they walk their own layout; a 0/1 mask over the COO order is
// one more pass, on demand.  Pure integer work, nothing is read but the state; deterministic in
// (seed, E, keep), restated bit for bit by the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_keep.h"
#include "dgmi_kernels.h"
#include "dgmi_tuning.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;
constexpr int kSelectThreads = 1024;      // the workgroup select
constexpr int kTieCap = 512;
// Shorter lists: one workgroup selects on its own (one launch).  Round 2 drew this line at 2^20 to save launches, but
// a 465 k-edge list (every real dataset's label-0 relation) keeps ONE workgroup busy for 226 us while the other 255 CUs
// idle — 0.45 ms of an lrssl-shaped training step (two selections per step, rocprofv3).  From 2^16 edges the window
// passes (5 small launches spread over the chip) take ~20 us instead.
constexpr int64_t kWindowMinE = 1 << 16;
constexpr int kBins = 4096;
constexpr int kCollCap = 1024;

enum : uint32_t { kDone = 0, kWindow = 1, kFallback = 2 };

struct SelectState {
  uint32_t mode;
  uint32_t lo;                 // window [lo, lo + kBins * binw) of the hash space
  uint32_t binw;
  uint32_t n_coll;
  unsigned long long below;    // hashes < lo
  int64_t remaining;           // rank (1-based) of the keep-th key inside the chosen bin
  uint32_t bin_lo, bin_hi;     // hashes of the chosen bin: [bin_lo, bin_hi]
  uint32_t hist[kBins];
  uint2 coll[kCollCap];        // (hash, edge id) of the chosen bin
};

// Up to kMaxKeepSegs subsets are selected by ONE series of launches (a training step drops edges on
// 4 relations and 4 similarity graphs at once — train.py:267): blockIdx.y says which subset a block
// works on.
struct BatchParams {
  int64_t E[kMaxKeepSegs];
  int64_t keep[kMaxKeepSegs];
  uint64_t seed[kMaxKeepSegs];
  const uint64_t* seed_dev;  // not NULL: the seeds are read from device memory (seed[] unused) — a captured HIP graph
                             // then draws a new subset on every replay without a host value baked into its launches
  uint32_t e_offset[kMaxKeepSegs];
  int64_t window_min;  // lists of at least this many edges take the window passes (kWindowMinE; DGMI_SELECT_WINDOW_MIN for A/B)
  int narrow;  // tests only (DGMI_SELECT_NARROW_WINDOW=1): a window of ~2 edges, so that it misses and the take-over path runs
};

__device__ __forceinline__ uint64_t seed_of(const BatchParams& p, int i) {
  return p.seed_dev != nullptr ? p.seed_dev[i] : p.seed[i];
}

__device__ __forceinline__ void write_desc(KeepSeg* out, int64_t E, uint64_t seed, uint32_t e_offset, uint32_t thr,
                                           int32_t tie_cut) {
  KeepSeg sg;
  sg.e_begin = e_offset;
  sg.e_end = e_offset + (uint32_t)E;
  sg.seed_lo = (uint32_t)seed;
  sg.seed_hi = (uint32_t)(seed >> 32);
  sg.thr = thr;
  sg.tie_cut = tie_cut;
  sg.flags = sg.reserved1 = 0;
  *out = sg;
}

// One workgroup: 4-pass MSB radix select of the keep-th smallest key, then the ties at that hash.
// keep == 0 gives (thr 0, tie_cut -1): nothing kept.
__device__ void block_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset, KeepSeg* out) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ties[kTieCap];
  __shared__ uint32_t n_ties, sh_prefix, sh_mask;
  __shared__ int64_t sh_rem;
  const int t = threadIdx.x;
  if (t == 0) {
    sh_prefix = 0;
    sh_mask = 0;
    sh_rem = keep;
    n_ties = 0;
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const uint32_t prefix = sh_prefix, pmask = sh_mask;
    for (int64_t e = t; e < E; e += kSelectThreads) {
      const uint32_t h = edge_hash(seed, (uint64_t)e);
      if ((h & pmask) == prefix) atomicAdd(&hist[(h >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t < 64) {
      // The bin holding the `rem`-th smallest candidate becomes the next prefix byte: the first bin whose running count
      // reaches rem (bin 255 if none does).  Wave 0 finds it with a scan over 64 groups of 4 bins — one thread walking
      // 256 dependent LDS reads took ~10 us per pass, 43 us per selection of a 3 k-edge list (rocprofv3 of a training step).
      const int64_t rem = sh_rem;
      const uint32_t c0 = hist[4 * t], c1 = hist[4 * t + 1], c2 = hist[4 * t + 2], c3 = hist[4 * t + 3];
      const int64_t own = (int64_t)c0 + c1 + c2 + c3;
      int64_t inc = own;
      for (int off = 1; off < 64; off <<= 1) {
        const int64_t y = __shfl_up(inc, off, 64);
        if (t >= off) inc += y;
      }
      const unsigned long long reached = __ballot(inc >= rem);
      const int owner = reached ? __ffsll((long long)reached) - 1 : 63;
      if (t == owner) {
        int64_t before = inc - own;
        uint32_t b = 4u * (uint32_t)t;
        if (before + c0 < rem) {
          before += c0;
          ++b;
          if (before + c1 < rem) {
            before += c1;
            ++b;
            if (before + c2 < rem) {
              before += c2;
              ++b;
            }
          }
        }
        sh_rem = rem - before;
        sh_prefix = prefix | (b << shift);
        sh_mask = pmask | (255u << shift);
      }
    }
    __syncthreads();
  }
  const uint32_t thr = sh_prefix;
  for (int64_t e = t; e < E; e += kSelectThreads)
    if (edge_hash(seed, (uint64_t)e) == thr) {
      const uint32_t slot = atomicAdd(&n_ties, 1u);
      if (slot < (uint32_t)kTieCap) ties[slot] = (uint32_t)e;
    }
  __syncthreads();
  if (t != 0) return;
  int32_t tie_cut = -1;
  if (n_ties > (uint32_t)kTieCap) {
    // More ties than the list holds (expected count is E / 2^32 < 1, so this is a degenerate
    // hash / seed): stay exact with a sequential scan in edge-id order instead of mis-counting.
    int64_t left = sh_rem;
    for (int64_t e = 0; e < E && left > 0; ++e)
      if (edge_hash(seed, (uint64_t)e) == thr) {
        tie_cut = (int32_t)e;
        --left;
      }
  } else {
    const uint32_t n = n_ties;
    for (uint32_t i = 1; i < n; ++i) {  // insertion sort by edge id
      const uint32_t v = ties[i];
      uint32_t j = i;
      for (; j > 0 && ties[j - 1] > v; --j) ties[j] = ties[j - 1];
      ties[j] = v;
    }
    const int64_t take = sh_rem < (int64_t)n ? sh_rem : (int64_t)n;
    if (take > 0) tie_cut = (int32_t)ties[take - 1];
  }
  write_desc(out, E, seed, e_offset, thr, tie_cut);
}

// launch 1: short lists (and keep == 0 / keep >= E) are selected outright; long ones get their window
__global__ __launch_bounds__(kSelectThreads) void select_or_init_kernel(BatchParams p, SelectState* states, KeepSeg* descs) {
  const int i = blockIdx.y;
  const int64_t E = p.E[i], keep = p.keep[i];
  SelectState* st = states + i;
  if (E < p.window_min || keep <= 0 || keep >= E) {
    if (threadIdx.x == 0) st->mode = kDone;
    if (keep <= 0) {  // nothing kept: what the radix select arrives at for rank 0, without the passes
      if (threadIdx.x == 0) write_desc(descs + i, E, seed_of(p, i), p.e_offset[i], 0u, -1);
      return;
    }
    block_select(E, keep, seed_of(p, i), p.e_offset[i], descs + i);
    return;
  }
  for (int b = threadIdx.x; b < kBins; b += kSelectThreads) st->hist[b] = 0;
  if (threadIdx.x == 0) {
    // expected threshold keep / E * 2^32; the count of hashes below a fixed value has sigma <= sqrt(E) / 2:
    // the window spans +- (8 sigma + 64) edges of hash space
    const double scale = 4294967296.0 / (double)E;
    const double centre = (double)keep * scale;
    const double half = p.narrow ? scale : (4.0 * sqrt((double)E) + 64.0) * scale;
    const double lo_d = centre - half < 0.0 ? 0.0 : centre - half;
    const double hi_d = centre + half > 4294967296.0 ? 4294967296.0 : centre + half;
    const uint64_t lo = (uint64_t)lo_d, hi = (uint64_t)hi_d;
    st->lo = (uint32_t)lo;
    st->binw = (uint32_t)((hi - lo + kBins - 1) / kBins) + 1u;
    st->below = 0;
    st->n_coll = 0;
    st->mode = kWindow;
  }
}

// launch 2: hashes below the window are counted, hashes inside it are histogrammed
__global__ __launch_bounds__(kBlock) void window_hist_kernel(BatchParams p, SelectState* states) {
  SelectState* st = states + blockIdx.y;
  if (st->mode != kWindow) return;
  const int64_t E = p.E[blockIdx.y];
  const uint64_t seed = seed_of(p, blockIdx.y);
  const uint32_t lo = st->lo, binw = st->binw;
  const uint64_t hi = (uint64_t)lo + (uint64_t)binw * kBins;  // exclusive
  unsigned long long below = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t h = edge_hash(seed, (uint64_t)e);
    if (h < lo)
      ++below;
    else if ((uint64_t)h < hi)
      atomicAdd(&st->hist[(h - lo) / binw], 1u);  // ~8 sqrt(E) of the E edges land here
  }
  for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
  __shared__ unsigned long long part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = below;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    if (s) atomicAdd(&st->below, s);
  }
}

// launch 3 (one workgroup per list): the bin that holds the keep-th key
__global__ __launch_bounds__(kBlock) void window_pick_kernel(BatchParams p, SelectState* states) {
  SelectState* st = states + blockIdx.y;
  if (st->mode != kWindow) return;
  __shared__ unsigned long long part[kBlock];
  constexpr int kPer = kBins / kBlock;  // 16 bins per thread
  unsigned long long s = 0;
  for (int j = 0; j < kPer; ++j) s += st->hist[threadIdx.x * kPer + j];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  // wave 0: the first thread-group (of kPer bins) whose running count reaches rem — a scan over 64 lanes x 4 groups
  // instead of one thread walking 2 x 256 dependent LDS reads (12 us)
  const int lane = threadIdx.x;
  const int64_t rem = p.keep[blockIdx.y] - (int64_t)st->below;  // rank of the key among hashes >= lo
  unsigned long long g[kBlock / 64];
  unsigned long long own = 0;
  for (int j = 0; j < kBlock / 64; ++j) {
    g[j] = part[lane * (kBlock / 64) + j];
    own += g[j];
  }
  unsigned long long inc = own;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long y = __shfl_up(inc, off, 64);
    if (lane >= off) inc += y;
  }
  const unsigned long long total = __shfl(inc, 63, 64);
  if (rem <= 0 || (unsigned long long)rem > total) {  // the window missed the threshold
    if (lane == 0) st->mode = kFallback;
    return;
  }
  const unsigned long long reached = __ballot(inc >= (unsigned long long)rem);
  if (lane != __ffsll((long long)reached) - 1) return;  // rem <= total: some lane reaches it
  unsigned long long before = inc - own;
  int t = lane * (kBlock / 64);
  for (int j = 0; j < kBlock / 64 - 1 && before + g[j] < (unsigned long long)rem; ++j, ++t) before += g[j];
  int b = t * kPer;
  for (; b < t * kPer + kPer - 1 && before + st->hist[b] < (unsigned long long)rem; ++b) before += st->hist[b];
  st->remaining = rem - (int64_t)before;
  const uint64_t blo = (uint64_t)st->lo + (uint64_t)b * st->binw;
  st->bin_lo = (uint32_t)blo;
  const uint64_t bhi = blo + st->binw - 1;
  st->bin_hi = bhi > 0xffffffffull ? 0xffffffffu : (uint32_t)bhi;
}

// launch 4: the (hash, id) pairs of that bin
__global__ __launch_bounds__(kBlock) void window_collect_kernel(BatchParams p, SelectState* states) {
  SelectState* st = states + blockIdx.y;
  if (st->mode != kWindow) return;
  const int64_t E = p.E[blockIdx.y];
  const uint64_t seed = seed_of(p, blockIdx.y);
  const uint32_t blo = st->bin_lo, bhi = st->bin_hi;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t h = edge_hash(seed, (uint64_t)e);
    if (h >= blo && h <= bhi) {
      const uint32_t slot = atomicAdd(&st->n_coll, 1u);
      if (slot < (uint32_t)kCollCap) st->coll[slot] = make_uint2(h, (uint32_t)e);
    }
  }
}

// launch 5 (one workgroup per list): order the bin's few keys, take the `remaining`-th; a list whose window
// missed or whose bin overflowed is selected by the workgroup on its own
__global__ __launch_bounds__(kSelectThreads) void window_finalize_kernel(BatchParams p, SelectState* states, KeepSeg* descs) {
  const int i = blockIdx.y;
  SelectState* st = states + i;
  const uint32_t mode = st->mode;  // uniform over the block
  if (mode == kDone) return;
  if (mode == kWindow && st->n_coll <= (uint32_t)kCollCap) {
    if (threadIdx.x != 0) return;
    const uint32_t n = st->n_coll;
    for (uint32_t a = 1; a < n; ++a) {  // insertion sort by (hash, id); n is a handful
      const uint2 v = st->coll[a];
      uint32_t j = a;
      for (; j > 0 && (st->coll[j - 1].x > v.x || (st->coll[j - 1].x == v.x && st->coll[j - 1].y > v.y)); --j)
        st->coll[j] = st->coll[j - 1];
      st->coll[j] = v;
    }
    const uint2 key = st->coll[st->remaining - 1];  // 1 <= remaining <= n by the bin's count
    write_desc(descs + i, p.E[i], seed_of(p, i), p.e_offset[i], key.x, (int32_t)key.y);
    return;
  }
  block_select(p.E[i], p.keep[i], seed_of(p, i), p.e_offset[i], descs + i);
}

__global__ __launch_bounds__(kBlock) void keep_mask_kernel(const KeepSeg* __restrict__ tab, int n_seg, int64_t E,
                                                           float* __restrict__ mask) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if (n_seg <= 0) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) mask[e] = 1.f;
    return;
  }
  const KeepPre first = keep_preload(tab, n_seg);
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride)
    mask[e] = edge_kept(first, tab, n_seg, (uint32_t)e) ? 1.f : 0.f;
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

}  // namespace

size_t random_subset_workspace_bytes() { return sizeof(SelectState) * kMaxKeepSegs + sizeof(KeepSeg); }

hipError_t random_subset_select_batch(int n, const int64_t* E, const int64_t* keep, const uint64_t* seed,
                                      const uint32_t* e_offset, void* descs, void* workspace, hipStream_t s,
                                      const uint64_t* seed_dev) {
  if (n <= 0) return hipSuccess;
  BatchParams p = {};
  const Tuning& tune = tuning();
  p.window_min = tune.select_window_min > 0 ? tune.select_window_min : kWindowMinE;
  int64_t e_max = 0;  // of the lists that take the window passes
  for (int i = 0; i < n; ++i) {
    p.E[i] = E[i];
    p.keep[i] = keep[i];
    p.seed[i] = seed != nullptr ? seed[i] : 0;
    p.e_offset[i] = e_offset ? e_offset[i] : 0u;
    if (E[i] >= p.window_min && keep[i] > 0 && keep[i] < E[i] && E[i] > e_max) e_max = E[i];
  }
  p.seed_dev = seed_dev;
  p.narrow = tune.select_narrow_window != 0;
  SelectState* st = static_cast<SelectState*>(workspace);
  KeepSeg* out = static_cast<KeepSeg*>(descs);
  const dim3 one(1, (unsigned)n);
  hipLaunchKernelGGL(select_or_init_kernel, one, dim3(kSelectThreads), 0, s, p, st, out);
  if (e_max > 0) {
    // >= 16 edges per thread: every block ends in ONE atomic on its list's `below` counter, and 1816 blocks of a
    // 465 k-edge list queued 30 us of them on that one address (rocprofv3 of a training step)
    int64_t wb = (e_max + (int64_t)kBlock * 16 - 1) / ((int64_t)kBlock * 16);
    if (wb < 1) wb = 1;
    if (wb > 2048) wb = 2048;
    const dim3 wide((unsigned)wb, (unsigned)n);
    hipLaunchKernelGGL(window_hist_kernel, wide, dim3(kBlock), 0, s, p, st);
    hipLaunchKernelGGL(window_pick_kernel, one, dim3(kBlock), 0, s, p, st);
    hipLaunchKernelGGL(window_collect_kernel, wide, dim3(kBlock), 0, s, p, st);
    hipLaunchKernelGGL(window_finalize_kernel, one, dim3(kSelectThreads), 0, s, p, st, out);
  }
  return hipGetLastError();
}

hipError_t random_subset_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset, void* seg_out,
                                void* workspace, hipStream_t s) {
  return random_subset_select_batch(1, &E, &keep, &seed, &e_offset, seg_out, workspace, s, nullptr);
}

hipError_t keep_mask_f32(const void* table, int n_seg, int64_t E, float* mask, hipStream_t s) {
  if (E == 0) return hipSuccess;
  hipLaunchKernelGGL(keep_mask_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, static_cast<const KeepSeg*>(table), n_seg,
                     E, mask);
  return hipGetLastError();
}

hipError_t random_subset_mask_f32(int64_t E, int64_t keep, uint64_t seed, float* mask, void* workspace,
                                  hipStream_t s) {
  if (E == 0) return hipSuccess;
  KeepSeg* seg = reinterpret_cast<KeepSeg*>(static_cast<char*>(workspace) + sizeof(SelectState) * kMaxKeepSegs);
  hipError_t err = random_subset_select(E, keep, seed, 0u, seg, workspace, s);
  if (err != hipSuccess) return err;
  return keep_mask_f32(seg, 1, E, mask, s);
}

}  // namespace dgmi
