// dgmi_keep.h — "which edges survive edge dropout" as a FUNCTION of the edge id, shared by the
// subset selection (dgmi_select.hip) and the SpMM kernels that apply it on the fly.
//
// The reference drops edges every training iteration (train.py:267): per edge type it keeps the
// first max(1, int(E*(1-p))) entries of torch.randperm(E) (augmentation.py:48-52, 114-118) — a
// uniformly random subset of exactly that size — and builds a new graph from them.  Here the
// subset of one edge list is described by four numbers instead of a list or a mask:
//     keep(e)  <=>  hash32(seed, e) < thr  ||  (hash32(seed, e) == thr  &&  e <= tie_cut)
// (thr = the keep-th smallest key's hash, found by radix select; tie_cut orders equal hashes by
// edge id).  A kernel that walks ANY layout of the parent graph (CSR, transposed CSR, XCD-sliced)
// evaluates keep(eid[p]) for the edges it is about to gather — nothing is re-sorted, no COO-order
// mask is carried into each layout, and dropped edges are skipped rather than multiplied by zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dgmi {

// One entry per edge list that was dropped independently (a relation-fused CSR concatenates
// several): edges [e_begin, e_end) of the layout's eid space belong to it.  8 x 32 bit.
struct KeepSeg {
  uint32_t e_begin, e_end;
  uint32_t seed_lo, seed_hi;
  uint32_t thr;
  int32_t tie_cut;  // local edge id; -1: no tie kept
  uint32_t flags;   // bit 0 (kKeepInvert): the segment keeps exactly the edges the description DROPS — the complement
                    // form of a near-complete relation subtracts its dropped edges (graph.py fused_relations_complement)
  uint32_t reserved1;
};
static_assert(sizeof(KeepSeg) == 32, "KeepSeg is 8 words in the C ABI");

constexpr int kMaxKeepSegs = 8;
constexpr uint32_t kKeepInvert = 1u;
constexpr uint32_t kDroppedBit = 0x80000000u;  // flag carried in the sign bit of a source id

__device__ __forceinline__ uint32_t edge_hash(uint64_t seed, uint64_t e) {
  uint64_t z = seed + (e + 1) * 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// Edges outside every segment are kept; an edge covered by several segments (a dropout applied to an
// already dropped view) survives only if EVERY covering segment keeps it — the intersection, as a
// chain of independent dropouts composes.
__device__ __forceinline__ bool edge_kept(const KeepSeg* __restrict__ tab, int n_seg, uint32_t e) {
  for (int k = 0; k < n_seg; ++k) {
    const KeepSeg sg = tab[k];
    if (e >= sg.e_begin && e < sg.e_end) {
      const uint32_t local = e - sg.e_begin;
      const uint32_t h = edge_hash(((uint64_t)sg.seed_hi << 32) | sg.seed_lo, local);
      const bool in_subset = h < sg.thr || (h == sg.thr && (int32_t)local <= sg.tie_cut);
      if (in_subset == ((sg.flags & kKeepInvert) != 0u)) return false;
    }
  }
  return true;
}

}  // namespace dgmi
