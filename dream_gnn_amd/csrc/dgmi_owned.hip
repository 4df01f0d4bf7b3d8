// dgmi_owned.hip — row-owned, slice-swept CSR SpMM for gfx950 (MI355X): no partial planes.
//
// The XCD-local kernel (dgmi_sliced.hip) pins each XCD to ONE eighth of the feature table, so
// every destination row is produced in 8 pieces that a second kernel has to add: 2 x 8 x N_dst x 4F
// bytes of plane scratch (36 % of the fabric traffic of a config-4 product) and a 61 us reduce
// kernel (17 % of the product).  Here the roles are swapped: a lane group OWNS a few destination
// rows for the whole launch and sweeps the source slices one after the other,
//
//      for slice s = 0 .. S-1:   for each owned row r:   acc[r] += sum of X[src] over r's edges in s
//
// with acc[] in LDS (the rows a group owns are indexed dynamically; registers cannot be).  All
// groups start at slice 0 together and advance at the same average rate, so at any moment the
// gathers of an XCD fall into one or two slices of X — sized to fit its 4 MiB L2 — and each XCD
// reads the table from the fabric once per sweep (8 x |X| in total) instead of once per
// destination row.  Y is written exactly once, straight from LDS with dst_scale applied.
// Nothing is exchanged between workgroups: co-residency and lock-step are speed assumptions,
// never correctness ones, and the in-row summation order (slice by slice, input order inside a
// slice) is a function of the layout alone -> bitwise reproducible.
//
// Layout (csr_owned_from_coo_i32): edges sorted, stably, by key = (group(row) * S + slice(src)) * rmax
// + local_row, so the edges of a group form ONE contiguous run ordered slice-major.  Each edge is a
// 32-bit word: source id in the low 27 bits, local row (< 32) in the high 5 — the kernel needs no
// per-(row, slice) pointer array, only seg_ptr[G * S + 1] (first edge of each (group, slice)).
//
// Geometry: the grid is exactly (#CUs x m) workgroups of 4 waves, all co-resident (the dynamic LDS
// request is padded to 160 KiB / m so that no CU takes an (m+1)-th); a wave holds 64 / LPR lane
// groups; rows are dealt evenly over rounds x #CUs x m x groups-per-block groups.  When
// N_dst x 4F exceeds what 256 x 160 KiB of LDS can hold (config 4's 100 k-row direction) every
// group does several rounds, each a full sweep over its next few rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "dgmi_kernels.h"
#include "dgmi_segment.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;
constexpr int kRowShift = 27;                       // word = src | local_row << 27
constexpr uint32_t kSrcMask = (1u << kRowShift) - 1u;
constexpr int kMaxLocalRows = 32;
// 4-wave blocks per CU = waves per SIMD: the kernel needs ~85 VGPRs (8 gathers of 16 B in flight per
// lane plus the run state), which admits 5
constexpr int kOwnedBlocksPerCu = 5;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int bits_for(int64_t n) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < n) ++b;
  return b;
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

int cu_count() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
    cached = n;
  else
    cached = 256;  // MI355X
  return cached;
}

__device__ __forceinline__ void group_of_row(int32_t r, const OwnedGeom& gm, int32_t& g, int32_t& lrow) {
  // the first `extra` groups own rows_lo + 1 rows, the others rows_lo
  const int32_t big = gm.extra * (gm.rows_lo + 1);
  if (r < big) {
    g = r / (gm.rows_lo + 1);
    lrow = r - g * (gm.rows_lo + 1);
  } else {
    const int32_t q = (r - big) / gm.rows_lo;  // rows_lo >= 1 whenever a row lies beyond `big`
    g = gm.extra + q;
    lrow = (r - big) - q * gm.rows_lo;
  }
}

__global__ __launch_bounds__(kBlock) void owned_key_kernel(const int32_t* __restrict__ row,
                                                           const int32_t* __restrict__ col, int64_t E,
                                                           int32_t n_rows, int32_t n_cols, OwnedGeom gm,
                                                           int32_t* __restrict__ key,
                                                           int32_t* __restrict__ tmp_eid,
                                                           int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    tmp_eid[e] = (int32_t)e;
    int32_t r = row[e], c = col[e];
    const bool oob = (r < 0) | (r >= n_rows) | (c < 0) | (c >= n_cols);
    bad |= oob;
    if (oob) r = c = 0;  // keep the key in range; the flag reports it
    int32_t g, lrow;
    group_of_row(r, gm, g, lrow);
    int32_t sl = c / gm.slice_width;
    if (sl >= gm.n_slices) sl = gm.n_slices - 1;
    key[e] = (g * gm.n_slices + sl) * gm.rmax + lrow;
  }
  if (bad) *flag = 1;
}

// seg_ptr[g * S + s] = first edge of (group g, slice s) from the sorted keys (fills empty
// segments too; seg_ptr[G * S] = E) and the packed edge words.
__global__ __launch_bounds__(kBlock) void owned_finish_kernel(const int32_t* __restrict__ sorted_key,
                                                              const int32_t* __restrict__ eid,
                                                              const int32_t* __restrict__ col, int64_t E,
                                                              int32_t n_cols, OwnedGeom gm,
                                                              int32_t* __restrict__ seg_ptr,
                                                              uint32_t* __restrict__ words) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int32_t n_seg = gm.n_groups * gm.n_slices;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p <= E; p += stride) {
    int32_t prev = p > 0 ? sorted_key[p - 1] / gm.rmax : -1;
    int32_t cur = p < E ? sorted_key[p] / gm.rmax : n_seg;
    prev = max(-1, min(prev, n_seg));
    cur = max(-1, min(cur, n_seg));
    for (int32_t g = prev + 1; g <= cur; ++g) seg_ptr[g] = (int32_t)p;
    if (p < E) {
      int32_t c = col[eid[p]];
      if (c < 0 || c >= n_cols) c = 0;  // flagged by owned_key_kernel; keep the gather in bounds
      words[p] = (uint32_t)c | ((uint32_t)(sorted_key[p] % gm.rmax) << kRowShift);
    }
  }
}

// ---------------------------------------------------------------------------------------------
constexpr int kPaceMaxSpins = 400;  // x (s_sleep + one L2 round trip) ~ 0.5 ms: a bound, never reached in step
constexpr int kPaceSleep = 48;      // x 64 clocks ~ 1.3 us between polls: thousands of waves poll one line
constexpr int kPaceStride = 32;     // uint32 per counter: one 128-B line each (pollers of slice s must not queue
                                    // in front of the adds to slice s + 1)

// PACED: the groups of an XCD label (blockIdx % 8) keep each other within `lag` slices: a group
// signals every slice it finishes on progress[label][round][slice] and, before entering slice s,
// waits until `need` groups of its label have finished slice s - lag.  Leaders are throttled, nobody
// waits for the slowest (the minimum-progress group can always proceed), every spin is bounded, and
// a group that ever times out stops pacing — placement and co-residency change speed only.
template <int LPR, bool HAS_VALS, bool HAS_SS, bool PACED>
__global__ __launch_bounds__(kBlock, kOwnedBlocksPerCu) void spmm_owned_kernel(
    const int32_t* __restrict__ seg_ptr, const uint32_t* __restrict__ words,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, const float* __restrict__ dst_scale, float* __restrict__ Y,
    int64_t ldy, int F, OwnedGeom gm, uint32_t* __restrict__ progress, int pace_lag, uint32_t pace_need) {
  extern __shared__ float4 owned_acc[];  // [groups per block][rmax][LPR] float4
  constexpr int NG = kWave / LPR;        // lane groups per wave
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  constexpr unsigned kAll = (1u << kUnroll) - 1u;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR, glane = lane % LPR, gbase = grp * LPR;
  const int gib = wave * NG + grp;  // group within the block
  float4* acc_rows = owned_acc + (size_t)gib * gm.rmax * LPR + glane;  // local row r at [r * LPR]
  int col = glane * 4;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;  // keep the loads in bounds; result discarded
  const float* Xc = X + col;
  const int S = gm.n_slices;
  uint32_t* my_progress = PACED ? progress + (size_t)(blockIdx.x % 8u) * gm.rounds * S * kPaceStride : nullptr;
  bool pacing = PACED;

  for (int round = 0; round < gm.rounds; ++round) {
    const int g = round * gm.groups_per_round + (int)blockIdx.x * (kWavesPerBlock * NG) + gib;
    const int row_start = g * gm.rows_lo + min(g, gm.extra);
    const int cnt = gm.rows_lo + (g < gm.extra ? 1 : 0);
    const int32_t* sp = seg_ptr + (int64_t)g * S;
    const int e_begin = sp[0], e_end = sp[S];
    for (int r = 0; r < gm.rmax; ++r) acc_rows[r * LPR] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur = 0;  // local row being accumulated in registers
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    // pacing state: lane k of the group holds the end of slice k (S <= LPR when PACED)
    int my_slice_end = 0, cur_slice = 0, slice_end = 0;
    if (PACED) {
      my_slice_end = sp[glane + 1 < S ? glane + 1 : S];
      slice_end = __shfl(my_slice_end, gbase, kWave);
    }
    // edge words (and weights) are fetched one batch ahead of the gathers that use them
    uint32_t nxt_word = 0;
    float nxt_w = 0.f;
    if (e_begin < e_end) {
      const int q = e_begin + glane < e_end ? e_begin + glane : e_end - 1;
      nxt_word = words[q];
      if (WEIGHTED) {
        nxt_w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) nxt_w *= src_scale[nxt_word & kSrcMask];
      }
    }
    int signalled = 0, waited = -1;  // wave-uniform: slices this wave has reported / waited for
    int base = e_begin;
    for (;;) {
      // The loop is wave-uniform (a group that has run out of edges idles through its partner's
      // batches): pacing decisions are taken by the whole wave, never by one group while the other
      // is masked off — a group spinning inside divergent code would hold its partner back, and
      // the partner is exactly who the spinning group may be waiting for.
      const bool active = base < e_end;
      if (!__any(active)) break;
      if (PACED) {
        if (active)
          while (base >= slice_end && cur_slice < S - 1) {  // this batch starts in a later slice
            ++cur_slice;
            slice_end = __shfl(my_slice_end, gbase + cur_slice, kWave);
          }
        int pos = active ? cur_slice : S;  // the wave is where its slowest unfinished group is
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) pos = min(pos, __shfl_xor(pos, off, kWave));
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (pos > signalled) {
          if (lane == 0)
            for (int sl = signalled; sl < pos; ++sl)
              __hip_atomic_fetch_add(my_progress + (round * S + sl) * kPaceStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          signalled = pos;
        }
        const int target = pos - pace_lag;
        if (pacing && target > waited) {
          const uint32_t* p = my_progress + (round * S + target) * kPaceStride;
          int spins = 0;
          while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pace_need) {
            if (++spins > kPaceMaxSpins) {
              pacing = false;
              break;
            }
            __builtin_amdgcn_s_sleep(kPaceSleep);
          }
          waited = target;
        }
      }
      if (!active) continue;
      const int n = min(LPR, e_end - base);
      const uint32_t my_word = nxt_word;
      const float my_w = nxt_w;
      if (base + LPR < e_end) {
        const int nb = base + LPR;
        const int q = nb + glane < e_end ? nb + glane : e_end - 1;
        nxt_word = words[q];
        if (WEIGHTED) {
          nxt_w = HAS_VALS ? vals[q] : 1.f;
          if (HAS_SS) nxt_w *= src_scale[nxt_word & kSrcMask];
        }
      }
      const int my_src = (int)(my_word & kSrcMask);
      const int my_row = (int)(my_word >> kRowShift);
      for (int j = 0; j < n; j += kUnroll) {
        float4 v[kUnroll];
        float w[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int e = j + u < n ? j + u : n - 1;  // the tail repeats a valid edge; ignored below
          const int idx = __shfl(my_src, gbase + e, kWave);
          if (WEIGHTED) w[u] = __shfl(my_w, gbase + e, kWave);
          v[u] = ld4(Xc + (int64_t)idx * ldx);
        }
        // fast path (group-uniform): 8 real edges, all of the row being accumulated
        const unsigned long long same = __ballot(my_row == cur);
        const unsigned bits = (unsigned)(same >> (gbase + j)) & kAll;
        if (j + kUnroll <= n && bits == kAll) {
          if (WEIGHTED) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
              v[u].x *= w[u];
              v[u].y *= w[u];
              v[u].z *= w[u];
              v[u].w *= w[u];
            }
          }
          tree_sum(v, kUnroll);
          acc.x += v[0].x;
          acc.y += v[0].y;
          acc.z += v[0].z;
          acc.w += v[0].w;
          continue;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          if (j + u < n) {  // group-uniform
            const int r = __shfl(my_row, gbase + j + u, kWave);
            if (r != cur) {  // the run moved on to another owned row (or the next slice): bank the sum
              float4 t = acc_rows[cur * LPR];
              t.x += acc.x;
              t.y += acc.y;
              t.z += acc.z;
              t.w += acc.w;
              acc_rows[cur * LPR] = t;
              acc = make_float4(0.f, 0.f, 0.f, 0.f);
              cur = r;
            }
            if (WEIGHTED) {
              acc.x = fmaf(w[u], v[u].x, acc.x);
              acc.y = fmaf(w[u], v[u].y, acc.y);
              acc.z = fmaf(w[u], v[u].z, acc.z);
              acc.w = fmaf(w[u], v[u].w, acc.w);
            } else {
              acc.x += v[u].x;
              acc.y += v[u].y;
              acc.z += v[u].z;
              acc.w += v[u].w;
            }
          }
        }
      }
      base += LPR;
    }
    {
      float4 t = acc_rows[cur * LPR];
      t.x += acc.x;
      t.y += acc.y;
      t.z += acc.z;
      t.w += acc.w;
      acc_rows[cur * LPR] = t;
    }
    if (PACED) {  // every wave reports every slice exactly once per round, with or without edges
      if (lane == 0)
        for (int sl = signalled; sl < S; ++sl)
          __hip_atomic_fetch_add(my_progress + (round * S + sl) * kPaceStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int rr = 0; rr < cnt; ++rr) {
      float4 t = acc_rows[rr * LPR];
      if (dst_scale != nullptr) {
        const float d = dst_scale[row_start + rr];
        t.x *= d;
        t.y *= d;
        t.z *= d;
        t.w *= d;
      }
      if (col_ok) *reinterpret_cast<float4*>(Y + (int64_t)(row_start + rr) * ldy + col) = t;
    }
  }
}

template <int LPR>
hipError_t launch_owned(const OwnedArgs& a, hipStream_t s) {
  dim3 grid((unsigned)a.geom.blocks), block(kBlock);
  const size_t lds = (size_t)a.geom.lds_bytes;
  // pacing needs one slice boundary per lane of a group and a co-resident grid of >= 8 blocks
  const bool paced = a.progress != nullptr && a.geom.n_slices > 1 && a.geom.n_slices <= LPR && a.geom.blocks % 8 == 0;
  int lag = 1;
  uint32_t need = 0;
  if (paced) {
    const char* e1 = getenv("DGMI_OWNED_PACE_LAG");  // tuning aids
    const char* e2 = getenv("DGMI_OWNED_PACE_PCT");
    lag = e1 != nullptr && atoi(e1) > 0 ? atoi(e1) : 1;
    const int pct = e2 != nullptr && atoi(e2) > 0 && atoi(e2) <= 100 ? atoi(e2) : 75;
    need = (uint32_t)((int64_t)(a.geom.groups_per_round / 8 / (kWave / LPR)) * pct / 100);  // waves per label
    hipError_t err = hipMemsetAsync(a.progress, 0, owned_progress_bytes(a.geom), s);
    if (err != hipSuccess) return err;
  }
  const int key = (a.vals ? 2 : 0) | (a.src_scale ? 1 : 0);
#define DGMI_LAUNCH(V, S, P)                                                                               \
  hipLaunchKernelGGL((spmm_owned_kernel<LPR, V, S, P>), grid, block, lds, s, a.seg_ptr, a.words, a.vals, a.X, \
                     a.ldx, a.src_scale, a.dst_scale, a.Y, a.ldy, (int)a.F, a.geom, a.progress, lag, need)
#define DGMI_LAUNCH_P(V, S) \
  if (paced) DGMI_LAUNCH(V, S, true); else DGMI_LAUNCH(V, S, false)
  switch (key) {
    case 0: DGMI_LAUNCH_P(false, false); break;
    case 1: DGMI_LAUNCH_P(false, true); break;
    case 2: DGMI_LAUNCH_P(true, false); break;
    default: DGMI_LAUNCH_P(true, true); break;
  }
#undef DGMI_LAUNCH_P
#undef DGMI_LAUNCH
  return hipGetLastError();
}

int lpr_for(int64_t F) {
  const int64_t f4 = (F + 3) / 4;
  for (int lpr : {8, 16, 32, 64})
    if (f4 <= lpr) return lpr;
  return 0;
}

}  // namespace

bool owned_geometry(int64_t n_rows, int64_t n_cols, int64_t F, int blocks_per_cu, int n_slices,
                    OwnedGeom* out) {
  const int lpr = lpr_for(F);
  if (lpr == 0 || F % 4 != 0 || n_rows <= 0 || n_cols <= 0 || n_cols > (int64_t)kSrcMask) return false;
  int m = blocks_per_cu;
  if (m <= 0) {
    const char* env = getenv("DGMI_OWNED_BLOCKS_PER_CU");  // tuning aid
    m = env != nullptr && atoi(env) > 0 ? atoi(env) : kOwnedBlocksPerCu;
  }
  if (m > kOwnedBlocksPerCu) m = kOwnedBlocksPerCu;  // what the register budget admits
  const int gpb = kWavesPerBlock * (kWave / lpr);
  const int64_t lds_alloc = ((int64_t)160 * 1024 / m) / 256 * 256;  // m of these fill a CU's LDS, m + 1 do not
  if (lds_alloc > 64 * 1024) return false;                            // stay inside the default dynamic-LDS limit
  const int64_t row_slot = (int64_t)gpb * lpr * 16;                   // one local row across the block
  int64_t rmax_lds = lds_alloc / row_slot;
  if (rmax_lds > kMaxLocalRows) rmax_lds = kMaxLocalRows;
  if (rmax_lds < 1) return false;
  const int64_t gpr = (int64_t)cu_count() * m * gpb;
  const int64_t rounds = (n_rows + gpr * rmax_lds - 1) / (gpr * rmax_lds);
  const int64_t G = rounds * gpr;
  int64_t S = n_slices;
  if (S <= 0) {
    const char* env = getenv("DGMI_OWNED_SLICE_KB");  // tuning aid
    const int64_t target = (env != nullptr && atoll(env) > 0 ? atoll(env) : 3200) * 1024;
    S = (n_cols * F * 4 + target - 1) / target;
  }
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  if (S > n_cols) S = n_cols;
  OwnedGeom g;
  g.n_groups = (int32_t)G;
  g.n_slices = (int32_t)S;
  g.rows_lo = (int32_t)(n_rows / G);
  g.extra = (int32_t)(n_rows % G);
  g.rmax = g.rows_lo + (g.extra ? 1 : 0);
  g.slice_width = (int32_t)((n_cols + S - 1) / S);
  g.groups_per_round = (int32_t)gpr;
  g.rounds = (int32_t)rounds;
  g.lanes_per_row = lpr;
  g.blocks = (int32_t)((int64_t)cu_count() * m);
  const int64_t need = (int64_t)gpb * g.rmax * lpr * 16;
  g.lds_bytes = (int32_t)(need > lds_alloc ? need : lds_alloc);
  g.reserved = 0;
  if (G * S * (int64_t)g.rmax >= INT32_MAX || G >= INT32_MAX) return false;
  *out = g;
  return true;
}

size_t owned_progress_bytes(const OwnedGeom& gm) {
  return (size_t)8 * (size_t)gm.rounds * (size_t)gm.n_slices * kPaceStride * sizeof(uint32_t);
}

hipError_t csr_owned_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E, int64_t n_rows,
                                  int64_t n_cols, const OwnedGeom& gm, int32_t* seg_ptr, uint32_t* words,
                                  int32_t* eid, void* workspace, size_t* workspace_bytes, hipStream_t s) {
  const int64_t n_keys = (int64_t)gm.n_groups * gm.n_slices * gm.rmax;
  const int end_bit = bits_for(n_keys);
  size_t sort_bytes = 0;
  if (E > 0) {
    hipError_t err = rocprim::radix_sort_pairs(
        nullptr, sort_bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
        static_cast<const int32_t*>(nullptr), eid, (size_t)E, 0u, (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  const size_t off_flag = 0;
  const size_t off_keys_in = 256;
  const size_t off_keys = off_keys_in + align_up((size_t)E * 4, 256);
  const size_t off_iota = off_keys + align_up((size_t)E * 4, 256);
  const size_t off_sort = off_iota + align_up((size_t)E * 4, 256);
  const size_t total = off_sort + align_up(sort_bytes, 256);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + off_flag);
  int32_t* keys_in = reinterpret_cast<int32_t*>(ws + off_keys_in);
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + off_keys);
  int32_t* tmp_eid = reinterpret_cast<int32_t*>(ws + off_iota);
  void* sort_tmp = ws + off_sort;
  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  if (E > 0) {
    hipLaunchKernelGGL(owned_key_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, row, col, E, (int32_t)n_rows,
                       (int32_t)n_cols, gm, keys_in, tmp_eid, flag);
    err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, reinterpret_cast<const uint32_t*>(keys_in),
                                    reinterpret_cast<uint32_t*>(keys_out), static_cast<const int32_t*>(tmp_eid),
                                    eid, (size_t)E, 0u, (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(owned_finish_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, keys_out, eid, col, E,
                     (int32_t)n_cols, gm, seg_ptr, words);
  return hipGetLastError();
}

hipError_t spmm_owned_f32(const OwnedArgs& a, hipStream_t s) {
  if (a.n_dst == 0 || a.F == 0) return hipSuccess;
  switch (a.geom.lanes_per_row) {
    case 8: return launch_owned<8>(a, s);
    case 16: return launch_owned<16>(a, s);
    case 32: return launch_owned<32>(a, s);
    case 64: return launch_owned<64>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dgmi
