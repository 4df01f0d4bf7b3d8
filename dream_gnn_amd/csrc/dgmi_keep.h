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

// hash32(seed, e): two 32-bit avalanche rounds (the murmur3 finaliser, then the "lowbias32" mixer), keyed with one half
// of the 64-bit seed each.  For a fixed seed it is a BIJECTION of the 32-bit edge id, so the E keys of a list are distinct
// and the threshold's tie rule is only ever exercised by the tests.  Four 32-bit multiplies: rounds 2-3 used the splitmix64
// finaliser (eleven, quarter-rate on this chip), which made every pass that hashes — the selection's two window passes, the
// compaction's flag pass, the on-the-fly KEEP kernels — ALU-bound (compaction flag pass 25 -> 12 us per 10 M edges).
__device__ __forceinline__ uint32_t edge_hash(uint64_t seed, uint64_t e) {
  uint32_t x = (uint32_t)e ^ (uint32_t)seed;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  x += (uint32_t)(seed >> 32);
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// Does segment `sg` let edge e take part?  (Edges it does not cover: yes.)
__device__ __forceinline__ bool seg_lets(const KeepSeg& sg, uint32_t e) {
  if (e >= sg.e_begin && e < sg.e_end) {
    const uint32_t local = e - sg.e_begin;
    const uint32_t h = edge_hash(((uint64_t)sg.seed_hi << 32) | sg.seed_lo, local);
    const bool in_subset = h < sg.thr || (h == sg.thr && (int32_t)local <= sg.tie_cut);
    if (in_subset == ((sg.flags & kKeepInvert) != 0u)) return false;
  }
  return true;
}

// Edges outside every segment are kept; an edge covered by several segments (a dropout applied to an
// already dropped view) survives only if EVERY covering segment keeps it — the intersection, as a
// chain of independent dropouts composes.
__device__ __forceinline__ bool edge_kept(const KeepSeg* __restrict__ tab, int n_seg, uint32_t e) {
  for (int k = 0; k < n_seg; ++k) {
    const KeepSeg sg = tab[k];
    if (!seg_lets(sg, e)) return false;
  }
  return true;
}

// The same with the first TWO descriptions already in registers (loaded once per wave before its edge loop — wave-uniform,
// so they live in SGPRs): almost every product has one description, the relation-fused aggregates of the real datasets two
// (one per rating), and reading the table per edge cost three DEPENDENT scalar loads per description (e_begin, then e_end,
// then the rest: the compiler sinks the loads into the range tests), each behind a full s_waitcnt.
struct KeepPre {
  KeepSeg a, b;
};

__device__ __forceinline__ KeepPre keep_preload(const KeepSeg* __restrict__ tab, int n_seg) {
  KeepPre p;
  p.a = tab[0];                      // n_seg >= 1 wherever this is called
  p.b = tab[n_seg > 1 ? 1 : 0];      // a copy of the first when there is no second: never consulted
  return p;
}

__device__ __forceinline__ bool edge_kept(const KeepPre& pre, const KeepSeg* __restrict__ tab, int n_seg, uint32_t e) {
  if (!seg_lets(pre.a, e)) return false;
  if (n_seg > 1 && !seg_lets(pre.b, e)) return false;
  for (int k = 2; k < n_seg; ++k) {
    const KeepSeg sg = tab[k];
    if (!seg_lets(sg, e)) return false;
  }
  return true;
}

}  // namespace dgmi
