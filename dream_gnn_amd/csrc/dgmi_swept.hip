// dgmi_swept.hip — row-owned, slice-swept CSR SpMM for gfx950 (MI355X): no partial planes.
//
// The XCD-local kernel (dgmi_sliced.hip) pins each XCD to one eighth of the feature table, so every
// destination row is produced in 8 pieces that a second kernel adds: 2 x 8 x N_dst x 4F bytes of plane
// scratch and a 61 us reduce kernel per config-4 product.  Here a lane group OWNS a few destination rows
// for the whole launch (running sums in LDS) and sweeps the source slices of X one after the other; Y is
// written once.  The XCD's 4 MiB L2 only helps if the workgroups that share it gather from the SAME slice
// at the same time.  Round 2 measured that free-running groups drift apart (L2 hit rate 52-60 %, below the
// pinned kernel's) and that per-wave progress counters cost more than they buy.  This version synchronises
// per WORKGROUP, not per wave, and softly: a workgroup may enter slice phase p only when every workgroup
// with its XCD label (blockIdx % 8) has finished phase p - 1 - lag.  With lag = 1 and slices of half an L2
// nobody waits in the steady state (the condition was met a whole phase ago, and the counter is read one id
// batch ahead of its use); at most two slices are live per XCD.  The barrier is ADVISORY: every spin is
// bounded, and nothing but speed depends on it or on where a workgroup runs.
//
// Layout (built by the caller, see ops.SweptCSR): the grid has TG lane groups (grid x waves x 64/LPR);
// destination row `row` belongs to round q = row / (TG*R), group g = (row % (TG*R)) / R, local row
// lr = row % R.  Edges are sorted, stably, by ((g*Q + q)*S + slice(src))*R + lr, so a group's whole sweep
// is ONE contiguous run: round 0 slice 0..S-1, round 1 slice 0..S-1, ...  Each edge is one 32-bit word,
// source id in the low 27 bits and lr in the high 5; seg[g*PH + p] (PH = Q*S, TG*PH + 1 entries) is the
// first edge of phase p of group g.  In-row summation order: slice by slice, input order inside a slice —
// a function of the layout alone, hence bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dgmi_kernels.h"
#include "dgmi_segment.h"

namespace dgmi {
namespace {

constexpr int kRowShift = 27;  // word = src | lr << 27
constexpr uint32_t kSrcMask = (1u << kRowShift) - 1u;
constexpr int kSpinBoundTicks = 3000;  // 30 us of the 100 MHz real-time counter: a stuck wait falls through

struct SweptGeom {
  int S, Q, R, lag;
  int sync_stride;  // unsigned words per label in `sync`
};

template <int LPR, bool HAS_VALS, bool HAS_SS>
__global__ __launch_bounds__(1024) void spmm_swept_kernel(
    const int32_t* __restrict__ seg, const uint32_t* __restrict__ words, const float* __restrict__ vals,
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ src_scale,
    const float* __restrict__ dst_scale, float* __restrict__ Y, int64_t ldy, int64_t n_dst, int F, SweptGeom gm,
    unsigned* __restrict__ sync, Epilogue ep) {
  constexpr int G = kWave / LPR;
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  extern __shared__ float lds[];
  const int W = (int)(blockDim.x >> 6);
  const int S = gm.S, R = gm.R, PH = gm.S * gm.Q, lag = gm.lag;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR, glane = lane % LPR, gbase = grp * LPR;
  const int64_t TG = (int64_t)gridDim.x * W * G;
  const int64_t g = ((int64_t)blockIdx.x * W + wave) * G + grp;
  const int label = (int)(blockIdx.x & 7u);
  const unsigned nb = (gridDim.x - (unsigned)label + 7u) >> 3;  // workgroups carrying this label
  unsigned* my_sync = sync + (size_t)label * gm.sync_stride;

  // LDS: float4 acc[W*G][R][LPR] (a lane owns its 16 bytes of every row of its group: plain ds_read_b128 /
  // ds_write_b128 read-modify-write — LDS float atomics cost the CU's one LDS pipe ~100 cycles each: 2.3x the
  // whole product), then int arrive[PH] (waves of this workgroup done with a phase), int verified
  float4* my_acc = reinterpret_cast<float4*>(lds) + (size_t)(wave * G + grp) * R * LPR + glane;
  int* arrive = reinterpret_cast<int*>(lds + (size_t)W * G * R * 4 * LPR);
  int* verified_lds = arrive + PH;
  for (int i = threadIdx.x; i <= PH; i += blockDim.x) arrive[i] = 0;  // arrive[PH] is `verified`
  for (int i = 0; i < R; ++i) my_acc[i * LPR] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const int col = glane * 4;
  const bool col_ok = col < F;
  const float* Xc = X + (col_ok ? col : 0);
  const int32_t* sg = seg + g * PH;
  const int e_begin = sg[0], e_end = sg[PH];
  int pe0 = sg[1];
  int pe1 = sg[PH >= 2 ? 2 : PH];
  int ph = 0;

  // Y rows of round q <- LDS sums; LDS re-zeroed for the next round
  auto flush_round = [&](int q) {
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((int64_t)q * TG + g) * R + r;
      float4 v = my_acc[r * LPR];
      my_acc[r * LPR] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n_dst && col_ok) {
        if (dst_scale != nullptr) {
          const float d = dst_scale[row];
          v.x *= d;
          v.y *= d;
          v.z *= d;
          v.w *= d;
        }
        *reinterpret_cast<float4*>(Y + row * ldy + col) = epilogue4(ep, v, row, col);
      }
    }
  };
  auto emit = [&](int r, const float4& acc) {
    float4 t = my_acc[r * LPR];
    t.x += acc.x;
    t.y += acc.y;
    t.z += acc.z;
    t.w += acc.w;
    my_acc[r * LPR] = t;
  };

  uint32_t nxt_word = 0;
  float nxt_w = 0.f;
  if (e_begin < e_end) {
    const int q = e_begin + glane < e_end ? e_begin + glane : e_begin;
    nxt_word = words[q];
    if (WEIGHTED) {
      nxt_w = HAS_VALS ? vals[q] : 1.f;
      if (HAS_SS) nxt_w *= src_scale[nxt_word & kSrcMask];
    }
  }
  int cur_r = (int)(__shfl(nxt_word, gbase, kWave) >> kRowShift);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // wave-level barrier state (uniform)
  int signalled = 0;    // phases this wave has reported complete
  int verified = -1;    // highest phase known complete by every workgroup of the label
  unsigned pre_val = 0; // counter value requested one batch ago ...
  int pre_target = -1;  // ... for this phase

  for (int base = e_begin;; base += LPR) {
    const bool active = base < e_end;
    if (!__any(active)) break;
    // ---- phase bookkeeping, once per id batch, wave-uniform ----
    {
      int lo = PH, hi = -1;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int a = __builtin_amdgcn_readlane(active ? 1 : 0, k * LPR);
        const int p = __builtin_amdgcn_readlane(ph, k * LPR);
        if (a) {
          lo = p < lo ? p : lo;
          hi = p > hi ? p : hi;
        }
      }
      while (signalled < lo) {  // every group of this wave is past these phases
        if (lane == 0) {
          const int old = __hip_atomic_fetch_add(&arrive[signalled], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old == W - 1) __hip_atomic_fetch_add(&my_sync[signalled], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++signalled;
      }
      if (lag >= 0) {
        const int need = hi - 1 - lag;  // must be complete before gathers of phase `hi` are issued
        if (need > verified && lo > need) {  // (lo <= need: a group of THIS wave is the straggler — never wait for oneself)
          bool ok = pre_target == need && (unsigned)__builtin_amdgcn_readlane((int)pre_val, 0) >= nb;
          if (!ok && lane == 0) {
            if (__hip_atomic_load(verified_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= need) {
              const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
              while (__hip_atomic_load(&my_sync[need], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nb &&
                     __builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)kSpinBoundTicks)
                __builtin_amdgcn_s_sleep(4);
            }
          }
          if (lane == 0) __hip_atomic_fetch_max(verified_lds, need + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          verified = need;
        }
        // request the counter the NEXT phase entry will need; consumed one batch from now, when it has long arrived
        const int want = hi - lag;
        if (want >= 0 && want > verified && want < PH) {
          pre_target = want;
          if (lane == 0) pre_val = __hip_atomic_load(&my_sync[want], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          pre_target = -1;
        }
      }
    }
    if (!active) continue;
    const int n = min(LPR, e_end - base);
    const uint32_t my_word = nxt_word;
    const float my_w = nxt_w;
    if (base + LPR < e_end) {
      const int nbq = base + LPR;
      const int q = nbq + glane < e_end ? nbq + glane : nbq;
      nxt_word = words[q];
      if (WEIGHTED) {
        nxt_w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) nxt_w *= src_scale[nxt_word & kSrcMask];
      }
    }
    // row-change bits of this batch: bit i <=> edge i starts another local row than edge i-1 (than cur_r for i = 0)
    const int my_lr = (int)(my_word >> kRowShift);
    int prev_lr = __shfl(my_lr, lane - 1, kWave);
    if (glane == 0) prev_lr = cur_r;
    const unsigned long long ball = __ballot(glane < n && my_lr != prev_lr);
    const unsigned long long chg = (ball >> gbase) & (LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull));
    for (int j = 0; j < n; j += kUnroll) {
      float4 v[kUnroll];
      float w[kUnroll];
      int lr[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int e = j + u;  // < LPR; past n: the batch's first id again (a valid row), never accumulated
        const uint32_t wd = __shfl(my_word, gbase + (e < n ? e : 0), kWave);
        if (WEIGHTED) w[u] = __shfl(my_w, gbase + e, kWave);
        lr[u] = (int)(wd >> kRowShift);
        v[u] = ld4(Xc + (int64_t)(wd & kSrcMask) * ldx);
      }
      const bool fast = j + kUnroll <= n && ((chg >> j) & 0xffull) == 0ull && base + j + kUnroll <= pe0;
      if (fast) {
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
        const int p = base + j + u;
        if (j + u < n) {  // group-uniform
          const bool crossing = p >= pe0;
          if (crossing || ((chg >> (j + u)) & 1ull)) {
            emit(cur_r, acc);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          if (crossing) {
            do {
              ++ph;
              pe0 = pe1;
              pe1 = sg[ph + 2 < PH ? ph + 2 : PH];
              if (ph % S == 0) flush_round(ph / S - 1);
            } while (p >= pe0);
          }
          cur_r = lr[u];
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
  }
  // the last open row, then every round not flushed yet (groups without edges write their zero rows here)
  emit(cur_r, acc);
  while (ph < PH) {
    ++ph;
    if (ph % S == 0) flush_round(ph / S - 1);
  }
  while (signalled < PH) {
    if (lane == 0) {
      const int old = __hip_atomic_fetch_add(&arrive[signalled], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old == W - 1) __hip_atomic_fetch_add(&my_sync[signalled], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ++signalled;
  }
}

template <int LPR>
hipError_t launch_swept(const SweptArgs& a, hipStream_t s) {
  constexpr int G = kWave / LPR;
  const int PH = a.S * a.Q;
  const size_t lds = (size_t)a.waves * G * a.R * 4 * LPR * sizeof(float) + (size_t)(PH + 1) * sizeof(int);
  SweptGeom gm{a.S, a.Q, a.R, a.lag, a.sync_stride};
  dim3 grid((unsigned)a.grid), block((unsigned)(a.waves * kWave));
  const int key = (a.vals ? 2 : 0) | (a.src_scale ? 1 : 0);
#define DGMI_SWEPT(V, SS)                                                                                              \
  do {                                                                                                                \
    auto kern = spmm_swept_kernel<LPR, V, SS>;                                                                        \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds);                                                                     \
    if (e != hipSuccess) return e;                                                                                    \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a.seg, a.words, a.vals, a.X, a.ldx, a.src_scale, a.dst_scale, a.Y,  \
                       a.ldy, a.n_dst, (int)a.F, gm, a.sync, a.ep);                                                   \
  } while (0)
  switch (key) {
    case 0: DGMI_SWEPT(false, false); break;
    case 1: DGMI_SWEPT(false, true); break;
    case 2: DGMI_SWEPT(true, false); break;
    default: DGMI_SWEPT(true, true); break;
  }
#undef DGMI_SWEPT
  return hipGetLastError();
}

}  // namespace

size_t swept_lds_bytes(int64_t F, int waves, int R, int PH) {
  const int lpr = swept_lpr(F);
  return (size_t)waves * (kWave / lpr) * R * 4 * lpr * sizeof(float) + (size_t)(PH + 1) * sizeof(int);
}

hipError_t spmm_swept_f32(const SweptArgs& a, hipStream_t s) {
  if (a.n_dst == 0 || a.F == 0) return hipSuccess;
  hipError_t err = hipMemsetAsync(a.sync, 0, (size_t)8 * a.sync_stride * sizeof(unsigned), s);
  if (err != hipSuccess) return err;
  switch (swept_lpr(a.F)) {
    case 8: return launch_swept<8>(a, s);
    case 16: return launch_swept<16>(a, s);
    case 32: return launch_swept<32>(a, s);
    default: return launch_swept<64>(a, s);
  }
}

}  // namespace dgmi
