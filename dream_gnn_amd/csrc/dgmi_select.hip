// dgmi_select.hip — exact-size uniformly random edge subset (gfx950): selection + mask.
//
// The reference drops edges every training iteration by keeping the first
// max(1, int(E*(1-p))) entries of torch.randperm(E) (augmentation.py:48-52, 114-118): a
// uniformly random subset of exactly that size.  A full random permutation is a sort of E keys;
// only the subset is needed.  Here every edge e gets the key (hash32(seed, e), e) — unique by
// construction — and the keep-th smallest key is found by a 4-pass most-significant-byte radix
// SELECT (256-bin histogram of the candidates per pass, no data movement) plus one pass that
// lists the edges whose hash equals the threshold (ties, ordered by id).  The result is the 8-word
// description of dgmi_keep.h (seed, threshold hash, tie cut): the SpMM kernels evaluate it per edge
// while they walk their own layout; a 0/1 mask over the COO order is one more pass, on demand.
// Pure integer work, nothing is read but the 4 KiB state; deterministic in (seed, E, keep),
// restated bit for bit by the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_keep.h"
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

// Up to kMaxKeepSegs subsets are selected by ONE series of launches (a training step drops edges on
// 4 relations and 4 similarity graphs at once — train.py:267 — and at dataset scale the launches,
// not the hashing, are the cost): blockIdx.y says which subset a block works on.
struct BatchParams {
  int64_t E[kMaxKeepSegs];
  int64_t keep[kMaxKeepSegs];
  uint64_t seed[kMaxKeepSegs];
  uint32_t e_offset[kMaxKeepSegs];
};

__global__ void init_state_kernel(SelectState* states, BatchParams p) {
  SelectState* st = states + blockIdx.y;
  const int64_t keep = p.keep[blockIdx.y];
  const int t = threadIdx.x;
  if (t == 0) {
    st->prefix = 0;
    st->prefix_mask = 0;
    st->remaining = keep;
    st->n_ties = 0;
  }
  st->hist[t] = 0;
}

__global__ __launch_bounds__(kBlock) void hist_kernel(BatchParams p, SelectState* states, int shift) {
  const int64_t E = p.E[blockIdx.y];
  const uint64_t seed = p.seed[blockIdx.y];
  SelectState* st = states + blockIdx.y;
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
__global__ void pick_kernel(SelectState* states, int shift) {
  SelectState* st = states + blockIdx.y;
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

// lists the edges whose hash equals the threshold (~E / 2^32 of them)
__global__ __launch_bounds__(kBlock) void ties_collect_kernel(BatchParams p, SelectState* states) {
  const int64_t E = p.E[blockIdx.y];
  const uint64_t seed = p.seed[blockIdx.y];
  SelectState* st = states + blockIdx.y;
  const uint32_t thr = st->prefix;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    if (edge_hash(seed, (uint64_t)e) == thr) {
      const uint32_t slot = atomicAdd(&st->n_ties, 1u);
      if (slot < (uint32_t)kTieCap) st->ties[slot] = (uint32_t)e;
    }
  }
}

// one thread: among the ties keep the `remaining` smallest ids -> tie_cut; write the description
__global__ void finalize_kernel(BatchParams p, SelectState* states, KeepSeg* descs) {
  const int64_t E = p.E[blockIdx.y];
  const uint64_t seed = p.seed[blockIdx.y];
  const uint32_t e_offset = p.e_offset[blockIdx.y];
  SelectState* st = states + blockIdx.y;
  KeepSeg* out = descs + blockIdx.y;
  if (threadIdx.x != 0) return;
  const uint32_t thr = st->prefix;
  int32_t tie_cut = -1;
  if (st->n_ties > (uint32_t)kTieCap) {
    // More ties than the list holds (expected count is E / 2^32 < 1, so this is a degenerate
    // hash / seed): stay exact with a sequential scan in edge-id order instead of mis-counting.
    int64_t left = st->remaining;
    for (int64_t e = 0; e < E && left > 0; ++e)
      if (edge_hash(seed, (uint64_t)e) == thr) {
        tie_cut = (int32_t)e;
        --left;
      }
  } else {
    const uint32_t n = st->n_ties;
    for (uint32_t i = 1; i < n; ++i) {  // insertion sort by edge id
      const uint32_t v = st->ties[i];
      uint32_t j = i;
      for (; j > 0 && st->ties[j - 1] > v; --j) st->ties[j] = st->ties[j - 1];
      st->ties[j] = v;
    }
    const int64_t take = st->remaining < (int64_t)n ? st->remaining : (int64_t)n;
    if (take > 0) tie_cut = (int32_t)st->ties[take - 1];
  }
  KeepSeg sg;
  sg.e_begin = e_offset;
  sg.e_end = e_offset + (uint32_t)E;
  sg.seed_lo = (uint32_t)seed;
  sg.seed_hi = (uint32_t)(seed >> 32);
  sg.thr = thr;
  sg.tie_cut = tie_cut;
  sg.reserved0 = sg.reserved1 = 0;
  *out = sg;
}

__global__ __launch_bounds__(kBlock) void keep_mask_kernel(const KeepSeg* __restrict__ tab, int n_seg, int64_t E,
                                                           float* __restrict__ mask) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride)
    mask[e] = edge_kept(tab, n_seg, (uint32_t)e) ? 1.f : 0.f;
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
                                      const uint32_t* e_offset, void* descs, void* workspace, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  BatchParams p = {};
  int64_t e_max = 0;
  for (int i = 0; i < n; ++i) {
    p.E[i] = E[i];
    p.keep[i] = keep[i];
    p.seed[i] = seed[i];
    p.e_offset[i] = e_offset ? e_offset[i] : 0u;
    if (E[i] > e_max) e_max = E[i];
  }
  SelectState* st = static_cast<SelectState*>(workspace);
  const dim3 wide(grid_for(e_max), (unsigned)n), one(1, (unsigned)n);
  hipLaunchKernelGGL(init_state_kernel, one, dim3(256), 0, s, st, p);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(hist_kernel, wide, dim3(kBlock), 0, s, p, st, shift);
    hipLaunchKernelGGL(pick_kernel, one, dim3(64), 0, s, st, shift);
  }
  hipLaunchKernelGGL(ties_collect_kernel, wide, dim3(kBlock), 0, s, p, st);
  hipLaunchKernelGGL(finalize_kernel, one, dim3(64), 0, s, p, st, static_cast<KeepSeg*>(descs));
  return hipGetLastError();
}

hipError_t random_subset_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset, void* seg_out,
                                void* workspace, hipStream_t s) {
  return random_subset_select_batch(1, &E, &keep, &seed, &e_offset, seg_out, workspace, s);
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
