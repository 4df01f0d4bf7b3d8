// dgmi_select.hip — exact-size uniformly random edge subset as a 0/1 mask (gfx950).
//
// The reference drops edges every training iteration by keeping the first
// max(1, int(E*(1-p))) entries of torch.randperm(E) (augmentation.py:48-52, 114-118): a
// uniformly random subset of exactly that size.  A full random permutation is a sort of E keys;
// only the subset is needed.  Here every edge e gets the key (hash32(seed, e), e) — unique by
// construction — and the keep-th smallest key is found by a 4-pass most-significant-byte radix
// SELECT (256-bin histogram of the candidates per pass, no data movement), then one pass writes
// mask[e] = key(e) <= threshold.  Pure integer work, nothing is read but the 4 KiB state; the
// result is a deterministic function of (seed, E, keep), restated bit for bit by the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;
constexpr int kTieCap = 512;

struct SelectState {
  uint32_t prefix;      // threshold hash bits decided so far (high bytes first)
  uint32_t prefix_mask; // which bits of `prefix` are decided
  int64_t remaining;    // rank (1-based) of the threshold among the current candidates
  uint32_t hist[256];
  uint32_t n_ties;
  uint32_t ties[kTieCap];
};

__device__ __forceinline__ uint32_t edge_hash(uint64_t seed, uint64_t e) {
  uint64_t z = seed + (e + 1) * 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

__global__ void init_state_kernel(SelectState* st, int64_t keep) {
  const int t = threadIdx.x;
  if (t == 0) {
    st->prefix = 0;
    st->prefix_mask = 0;
    st->remaining = keep;
    st->n_ties = 0;
  }
  st->hist[t] = 0;
}

__global__ __launch_bounds__(kBlock) void hist_kernel(int64_t E, uint64_t seed, SelectState* st, int shift) {
  __shared__ uint32_t local[256];
  local[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix, pmask = st->prefix_mask;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t h = edge_hash(seed, (uint64_t)e);
    if ((h & pmask) == prefix) atomicAdd(&local[(h >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (local[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], local[threadIdx.x]);
}

// one thread: the bin holding the `remaining`-th smallest candidate becomes the next prefix byte
__global__ void pick_kernel(SelectState* st, int shift) {
  if (threadIdx.x != 0) return;
  int64_t rem = st->remaining, before = 0;
  uint32_t b = 0;
  for (; b < 255u; ++b) {
    if (before + st->hist[b] >= rem) break;
    before += st->hist[b];
  }
  st->remaining = rem - before;
  st->prefix |= b << shift;
  st->prefix_mask |= 255u << shift;
  for (int i = 0; i < 256; ++i) st->hist[i] = 0;
}

__global__ __launch_bounds__(kBlock) void mask_kernel(int64_t E, uint64_t seed, SelectState* st,
                                                      float* __restrict__ mask) {
  const uint32_t thr = st->prefix;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t h = edge_hash(seed, (uint64_t)e);
    mask[e] = h < thr ? 1.f : 0.f;
    if (h == thr) {  // ~E / 2^32 edges: decided by edge id in ties_kernel
      const uint32_t slot = atomicAdd(&st->n_ties, 1u);
      if (slot < (uint32_t)kTieCap) st->ties[slot] = (uint32_t)e;
    }
  }
}

// one thread: among the edges whose hash equals the threshold keep the `remaining` smallest ids
__global__ void ties_kernel(int64_t E, uint64_t seed, SelectState* st, float* __restrict__ mask) {
  if (threadIdx.x != 0) return;
  if (st->n_ties > (uint32_t)kTieCap) {
    // More ties than the list holds (expected count is E / 2^32 < 1, so this is a degenerate
    // hash / seed): stay exact with a sequential scan in edge-id order instead of mis-counting.
    const uint32_t thr = st->prefix;
    int64_t left = st->remaining;
    for (int64_t e = 0; e < E && left > 0; ++e)
      if (edge_hash(seed, (uint64_t)e) == thr) {
        mask[e] = 1.f;
        --left;
      }
    return;
  }
  const uint32_t n = st->n_ties;
  for (uint32_t i = 1; i < n; ++i) {  // insertion sort by edge id
    const uint32_t v = st->ties[i];
    uint32_t j = i;
    for (; j > 0 && st->ties[j - 1] > v; --j) st->ties[j] = st->ties[j - 1];
    st->ties[j] = v;
  }
  const int64_t take = st->remaining < (int64_t)n ? st->remaining : (int64_t)n;
  for (int64_t i = 0; i < take; ++i) mask[st->ties[i]] = 1.f;
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

}  // namespace

size_t random_subset_workspace_bytes() { return sizeof(SelectState); }

hipError_t random_subset_mask_f32(int64_t E, int64_t keep, uint64_t seed, float* mask, void* workspace,
                                  hipStream_t s) {
  if (E == 0) return hipSuccess;
  SelectState* st = static_cast<SelectState*>(workspace);
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, s, st, keep);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(hist_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, E, seed, st, shift);
    hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(64), 0, s, st, shift);
  }
  hipLaunchKernelGGL(mask_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, E, seed, st, mask);
  hipLaunchKernelGGL(ties_kernel, dim3(1), dim3(64), 0, s, E, seed, st, mask);
  return hipGetLastError();
}

}  // namespace dgmi
