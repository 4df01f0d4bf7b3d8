// dgmi_segment.h — the wave-level row-segment gather shared by the SpMM kernels
// (dgmi_spmm.hip: one wave per row / per plan item; dgmi_sliced.hip: XCD-local slices).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_keep.h"
#include "dgmi_kernels.h"

namespace dgmi {
namespace {

// Epilogue of the kernel that writes Y: activation, then the (dropout) keep mask.
__device__ __forceinline__ float epilogue1(const Epilogue& ep, float v, int64_t row, int col) {
  if (ep.act == 1) v = v > 0.f ? v : v * ep.slope;
  if (ep.mask != nullptr) v *= ep.mask[row * ep.ldm + col] * ep.mask_scale;
  return v;
}

__device__ __forceinline__ float4 epilogue4(const Epilogue& ep, float4 v, int64_t row, int col) {
  if (ep.act == 1) {
    v.x = v.x > 0.f ? v.x : v.x * ep.slope;
    v.y = v.y > 0.f ? v.y : v.y * ep.slope;
    v.z = v.z > 0.f ? v.z : v.z * ep.slope;
    v.w = v.w > 0.f ? v.w : v.w * ep.slope;
  }
  if (ep.mask != nullptr) {
    const float4 m = *reinterpret_cast<const float4*>(ep.mask + row * ep.ldm + col);
    v.x *= m.x * ep.mask_scale;
    v.y *= m.y * ep.mask_scale;
    v.z *= m.z * ep.mask_scale;
    v.w *= m.w * ep.mask_scale;
  }
  return v;
}

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 8;

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

__device__ __forceinline__ void tree_sum(float4 (&v)[kUnroll], int n_live) {
  (void)n_live;
#pragma unroll
  for (int span = 1; span < kUnroll; span <<= 1) {
#pragma unroll
    for (int u = 0; u + span < kUnroll; u += 2 * span) {
      v[u].x += v[u + span].x;
      v[u].y += v[u + span].y;
      v[u].z += v[u + span].z;
      v[u].w += v[u + span].w;
    }
  }
}

// One batch of up to 64 edges whose ids/weights sit one-per-lane in
// (my_idx, my_w).  FULL: all 64 are valid, no predication in the loop.
// KEEP: an id with kDroppedBit set is an edge removed by edge dropout — its gather is redirected to
// the row the same lanes fetched last (an L1 hit; a fixed row such as row 0 would funnel 10 % of all
// gathers into one L2 channel: measured +50 % on a 10 M-edge product) and its contribution replaced
// by zeros (a select, not 0 * x: Inf / NaN in a dropped edge's source row must not leak into the sum).
template <int LPR, bool WEIGHTED, bool FULL, bool KEEP>
__device__ __forceinline__ void gather_batch(const float* __restrict__ Xc, int64_t ldx,
                                             int my_idx, float my_w, int n, int sub,
                                             float4& acc) {
  constexpr int EPI = kWave / LPR;  // edges per wave-instruction
  constexpr int STEPS = kWave / EPI;
  static_assert(STEPS % kUnroll == 0, "unroll must divide steps");
#pragma unroll 1
  for (int s = 0; s < STEPS; s += kUnroll) {
    if (!FULL && s * EPI >= n) break;
    float4 v[kUnroll];
    float w[kUnroll];
    int last = -1;  // KEEP: the row these lanes gathered last
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int e = (s + u) * EPI + sub;
      int idx = __shfl(my_idx, e, kWave);
      if (WEIGHTED) w[u] = __shfl(my_w, e, kWave);
      const bool dropped = KEEP && idx < 0;
      if (KEEP) {
        idx = dropped && last >= 0 ? last : idx & 0x7fffffff;
        last = idx;
      }
      // Lanes past the end of a tail batch carry the first id of the batch
      // (a valid row) and are zeroed below, so no load leaves the matrix.
      v[u] = ld4(Xc + (int64_t)idx * ldx);
      if ((!FULL && e >= n) || dropped) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (WEIGHTED) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u].x *= w[u];
        v[u].y *= w[u];
        v[u].z *= w[u];
        v[u].w *= w[u];
      }
    }
    // Balanced tree over the 8 rows, then one add into the running sum: same
    // add count as a chain, 8-way ILP, and the long dependent chain (and its
    // rounding growth on 10^4..10^6-edge rows) shrinks 8-fold.
    tree_sum(v, kUnroll);
    acc.x += v[0].x;
    acc.y += v[0].y;
    acc.z += v[0].z;
    acc.w += v[0].w;
  }
}

// Edge dropout on the fly, wave-per-segment kernels: the ids of one 64-lane batch come with the dropped ones flagged
// in the sign bit.  The survivors are moved, in order, to lanes [0, cnt) with ONE ds_permute per register (a full
// permutation: survivor -> its rank, dropped lane -> cnt + its rank among the dropped), so the gather loop only ever
// sees live edges: a dropped edge costs its id fetch and one hash, not a load slot — which is what makes segments
// that drop MOST of their edges (the inverted segment of the complement form, dgmi_keep.h) cheap — and the kept
// edges are summed in exactly the order a CSR rebuilt from them would give (the reference's construction).
// Returns cnt; ids come back unflagged (every lane holds a valid row id).
__device__ __forceinline__ int compact_live(int& idx, float& w, bool weighted, int n, int lane) {
  const bool live = lane < n && idx >= 0;
  const unsigned long long m = __ballot(live);
  const int cnt = __popcll(m);
  if (cnt < n) {
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    const int target = (live ? rank : cnt + (lane - rank)) << 2;
    idx = __builtin_amdgcn_ds_permute(target, idx);
    if (weighted) w = __int_as_float(__builtin_amdgcn_ds_permute(target, __float_as_int(w)));
  }
  idx &= 0x7fffffff;
  return cnt;
}

// Sum of edges [start, end) of one row over this lane's 4 columns; the result
// is complete (all 64/LPR lane groups combined) in every lane.
// id of edge q as the gathers will use it: the source id, with (KEEP) kDroppedBit set when
// keep(eid[q]) says the edge was dropped.
// `first` = the first two descriptions, loaded once by the caller before its edge loop (dgmi_keep.h: per-edge table reads were three
// dependent scalar loads each).
template <bool KEEP>
__device__ __forceinline__ int fetch_id(const int32_t* __restrict__ indices, const int32_t* __restrict__ eid,
                                        const KeepPre& first, const KeepSeg* __restrict__ keep, int n_keep, int q) {
  int idx = indices[q];
  if (KEEP && !edge_kept(first, keep, n_keep, (uint32_t)eid[q])) idx |= (int)kDroppedBit;
  return idx;
}

template <bool KEEP>
__device__ __forceinline__ KeepPre first_seg(const KeepSeg* __restrict__ keep, int n_keep) {
  if (KEEP) return keep_preload(keep, n_keep);  // KEEP instantiations are only launched with n_keep >= 1
  const KeepSeg none{0u, 0u, 0u, 0u, 0u, -1, 0u, 0u};
  return KeepPre{none, none};
}

template <int LPR, bool HAS_VALS, bool HAS_SS, bool KEEP>
__device__ __forceinline__ float4 segment_vec4(const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals,
                                               const float* __restrict__ src_scale,
                                               const int32_t* __restrict__ eid,
                                               const KeepSeg* __restrict__ keep, int n_keep,
                                               const float* __restrict__ Xc, int64_t ldx,
                                               int start, int end, int lane) {
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  const int sub = lane / LPR;
  const KeepPre first = first_seg<KEEP>(keep, n_keep);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // software pipeline: ids (and weights) of batch b+1 are requested before the
  // row loads of batch b are issued.
  int nxt_idx = 0;
  float nxt_w = 0.f;
  if (start < end) {
    const int p = start + lane;
    const int q = p < end ? p : start;
    nxt_idx = fetch_id<KEEP>(indices, eid, first, keep, n_keep, q);
    if (WEIGHTED) {
      float w = HAS_VALS ? vals[q] : 1.f;
      if (HAS_SS) w *= src_scale[KEEP ? nxt_idx & 0x7fffffff : nxt_idx];
      nxt_w = w;
    }
  }
  for (int base = start; base < end; base += kWave) {
    int my_idx = nxt_idx;
    float my_w = nxt_w;
    int n = end - base;
    const int nb = base + kWave;
    if (nb < end) {
      const int p = nb + lane;
      const int q = p < end ? p : nb;
      nxt_idx = fetch_id<KEEP>(indices, eid, first, keep, n_keep, q);
      if (WEIGHTED) {
        float w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) w *= src_scale[KEEP ? nxt_idx & 0x7fffffff : nxt_idx];
        nxt_w = w;
      }
    }
    if (KEEP) {
      n = compact_live(my_idx, my_w, WEIGHTED, n < kWave ? n : kWave, lane);
      if (n == 0) continue;  // wave-uniform
    }
    if (n >= kWave)
      gather_batch<LPR, WEIGHTED, true, false>(Xc, ldx, my_idx, my_w, kWave, sub, acc);
    else
      gather_batch<LPR, WEIGHTED, false, false>(Xc, ldx, my_idx, my_w, n, sub, acc);
  }
  // combine the 64/LPR partial rows (fixed order -> deterministic)
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, kWave);
    acc.y += __shfl_xor(acc.y, off, kWave);
    acc.z += __shfl_xor(acc.z, off, kWave);
    acc.w += __shfl_xor(acc.w, off, kWave);
  }
  return acc;
}

// Same for one column per lane (any F, any alignment).
template <bool HAS_VALS, bool HAS_SS, bool KEEP>
__device__ __forceinline__ float segment_dword(const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals,
                                               const float* __restrict__ src_scale,
                                               const int32_t* __restrict__ eid,
                                               const KeepSeg* __restrict__ keep, int n_keep,
                                               const float* __restrict__ Xc, int64_t ldx,
                                               int start, int end, int lane) {
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  const KeepPre first = first_seg<KEEP>(keep, n_keep);
  float acc = 0.f;
  for (int base = start; base < end; base += kWave) {
    int n = min(kWave, end - base);
    const int q = lane < n ? base + lane : base;
    int my_idx = fetch_id<KEEP>(indices, eid, first, keep, n_keep, q);
    float my_w = 0.f;
    if (WEIGHTED) {
      my_w = HAS_VALS ? vals[q] : 1.f;
      if (HAS_SS) my_w *= src_scale[KEEP ? my_idx & 0x7fffffff : my_idx];
    }
    if (KEEP) n = compact_live(my_idx, my_w, WEIGHTED, n, lane);  // survivors in lanes [0, n), every lane a valid id
#pragma unroll 1
    for (int s = 0; s < n; s += kUnroll) {
      float v[kUnroll], w[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int e = s + u;  // < 64 because n <= 64 and 64 % kUnroll == 0
        const int idx = __shfl(my_idx, e, kWave);
        if (WEIGHTED) w[u] = __shfl(my_w, e, kWave);
        v[u] = Xc[(int64_t)idx * ldx];
        if (e >= n) v[u] = 0.f;
      }
      if (WEIGHTED) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] *= w[u];
      }
#pragma unroll
      for (int span = 1; span < kUnroll; span <<= 1) {
#pragma unroll
        for (int u = 0; u + span < kUnroll; u += 2 * span) v[u] += v[u + span];
      }
      acc += v[0];
    }
  }
  return acc;
}

}  // namespace
}  // namespace dgmi
