// dgmi_spmm.hip — CSR SpMM for gfx950 (MI355X), fp32, int32 ids.
//
// Replaces DGL's gspmm('copy_lhs','sum') behind update_all(copy_u, sum)
// (reference layers.py:229-232) and ATen's sparse addmm behind th.spmm
// (reference layers.py:312), with the reference's two diagonal scalings
// (layers.py:224-225 `feat * dropout(cj)`, layers.py:234 `rst * ci`) fused in.
//
// Shape of the work: an HBM / Infinity-Cache bound row gather.  Per edge the
// kernel moves one source row (4*F bytes) and 4 (+4) bytes of index (value);
// arithmetic intensity is 0.25-0.5 flop/B, so there is nothing for MFMA here.
// Design for CDNA4:
//   * one 64-lane wave owns one destination row (segmented reduce = the wave's
//     own registers; no atomics, fixed summation order -> bitwise reproducible);
//   * a source row is read with 16-B loads: LPR = F/4 lanes cover a row, so a
//     wave-instruction carries 64/LPR whole rows (F=128: 2 rows x 512 B = 1 KiB,
//     the widest coalesced access the hardware has);
//   * the row's column ids (and values) are read 64 at a time, one per lane,
//     coalesced, one batch ahead of use, and handed to the row loads by
//     ds_bpermute; 8 row loads are kept in flight per wave (8 KiB at F=128,
//     x 16-32 waves per CU) to cover Infinity-Cache / HBM latency;
//   * the 64/LPR partial sums are combined with xor-shuffles and written with
//     one coalesced 16-B-per-lane store, scaled by dst_scale.
// Feature widths that are not a multiple of 4 (layer-0's 341, layers.py:55-57)
// or rows that are not 16-B aligned take the dword kernel below.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"

namespace dgmi {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 8;

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

// One batch of up to 64 edges whose ids/weights sit one-per-lane in
// (my_idx, my_w).  FULL: all 64 are valid, no predication in the loop.
template <int LPR, bool WEIGHTED, bool FULL>
__device__ __forceinline__ void gather_batch(const float* __restrict__ Xc, int64_t ldx,
                                             int my_idx, float my_w, int n, int sub,
                                             float4& acc) {
  constexpr int EPI = kWave / LPR;  // edges per wave-instruction
  constexpr int STEPS = kWave / EPI;
  static_assert(STEPS % kUnroll == 0 || STEPS < kUnroll, "unroll must divide steps");
  constexpr int U = STEPS < kUnroll ? STEPS : kUnroll;
#pragma unroll 1
  for (int s = 0; s < STEPS; s += U) {
    if (!FULL && s * EPI >= n) break;
    float4 v[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = (s + u) * EPI + sub;
      const int idx = __shfl(my_idx, e, kWave);
      if (WEIGHTED) w[u] = __shfl(my_w, e, kWave);
      // Lanes past the end of a tail batch carry idx = first id of the batch
      // (a valid row) and are zeroed below, so no load leaves the matrix.
      v[u] = ld4(Xc + (int64_t)idx * ldx);
      if (!FULL && e >= n) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Sum the U rows as a balanced tree and add the result to the running sum: the same
    // number of adds as a chain, U-way ILP, and the long chain shrinks U-fold (rounding
    // error of a 60k-edge row stays inside 1e-5 relative).
    if (WEIGHTED) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u].x *= w[u];
        v[u].y *= w[u];
        v[u].z *= w[u];
        v[u].w *= w[u];
      }
    }
#pragma unroll
    for (int span = 1; span < U; span <<= 1) {
#pragma unroll
      for (int u = 0; u + span < U; u += 2 * span) {
        v[u].x += v[u + span].x;
        v[u].y += v[u + span].y;
        v[u].z += v[u + span].z;
        v[u].w += v[u + span].w;
      }
    }
    acc.x += v[0].x;
    acc.y += v[0].y;
    acc.z += v[0].z;
    acc.w += v[0].w;
  }
}

// grid.x = ceil(n_dst / 4), grid.y = ceil(F / (4*LPR)); block = 256 (4 waves).
template <int LPR, bool HAS_VALS, bool HAS_SS, bool HAS_DS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_csr_vec4_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, const float* __restrict__ dst_scale,
    float* __restrict__ Y, int64_t ldy, int64_t n_dst, int F) {
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (row >= n_dst) return;  // wave-uniform
  const int sub = lane / LPR;
  int col = ((int)blockIdx.y * LPR + (lane % LPR)) * 4;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;  // keep the loads in bounds; result discarded
  const float* Xc = X + col;

  const int start = indptr[row];
  const int end = indptr[row + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  // software pipeline: ids (and weights) of batch b+1 are requested before the
  // row loads of batch b are issued.
  int nxt_idx = 0;
  float nxt_w = 0.f;
  if (start < end) {
    const int p = start + lane;
    const int q = p < end ? p : start;
    nxt_idx = indices[q];
    if (WEIGHTED) {
      float w = HAS_VALS ? vals[q] : 1.f;
      if (HAS_SS) w *= src_scale[nxt_idx];
      nxt_w = w;
    }
  }
  for (int base = start; base < end; base += kWave) {
    const int my_idx = nxt_idx;
    const float my_w = nxt_w;
    const int n = end - base;
    const int nb = base + kWave;
    if (nb < end) {
      const int p = nb + lane;
      const int q = p < end ? p : nb;
      nxt_idx = indices[q];
      if (WEIGHTED) {
        float w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) w *= src_scale[nxt_idx];
        nxt_w = w;
      }
    }
    if (n >= kWave)
      gather_batch<LPR, WEIGHTED, true>(Xc, ldx, my_idx, my_w, kWave, sub, acc);
    else
      gather_batch<LPR, WEIGHTED, false>(Xc, ldx, my_idx, my_w, n, sub, acc);
  }

  // combine the 64/LPR partial rows (fixed order -> deterministic)
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, kWave);
    acc.y += __shfl_xor(acc.y, off, kWave);
    acc.z += __shfl_xor(acc.z, off, kWave);
    acc.w += __shfl_xor(acc.w, off, kWave);
  }
  if (sub == 0 && col_ok) {
    if (HAS_DS) {
      const float d = dst_scale[row];
      acc.x *= d;
      acc.y *= d;
      acc.z *= d;
      acc.w *= d;
    }
    *reinterpret_cast<float4*>(Y + row * ldy + col) = acc;
  }
}

// Any F, any alignment: lanes across 64 consecutive columns, one dword each;
// grid.y = ceil(F/64).  Used for F % 4 != 0 (341) and unaligned views.
template <bool HAS_VALS, bool HAS_SS, bool HAS_DS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_csr_dword_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, const float* __restrict__ dst_scale,
    float* __restrict__ Y, int64_t ldy, int64_t n_dst, int F) {
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (row >= n_dst) return;
  int col = (int)blockIdx.y * kWave + lane;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;
  const float* Xc = X + col;
  const int start = indptr[row];
  const int end = indptr[row + 1];
  float acc = 0.f;
  for (int base = start; base < end; base += kWave) {
    const int n = min(kWave, end - base);
    int my_idx = 0;
    float my_w = 0.f;
    {
      const int q = lane < n ? base + lane : base;
      my_idx = indices[q];
      if (WEIGHTED) {
        float w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) w *= src_scale[my_idx];
        my_w = w;
      }
    }
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
  if (col_ok) {
    if (HAS_DS) acc *= dst_scale[row];
    Y[row * ldy + col] = acc;
  }
}

template <int LPR>
hipError_t launch_vec4(const SpmmArgs& a, hipStream_t s) {
  dim3 grid((unsigned)((a.n_dst + kWavesPerBlock - 1) / kWavesPerBlock),
            (unsigned)((a.F + 4 * LPR - 1) / (4 * LPR)));
  dim3 block(kWave * kWavesPerBlock);
  const int key = (a.vals ? 4 : 0) | (a.src_scale ? 2 : 0) | (a.dst_scale ? 1 : 0);
#define DGMI_LAUNCH(V, S, D)                                                         \
  hipLaunchKernelGGL((spmm_csr_vec4_kernel<LPR, V, S, D>), grid, block, 0, s,         \
                     a.indptr, a.indices, a.vals, a.X, a.ldx, a.src_scale,            \
                     a.dst_scale, a.Y, a.ldy, a.n_dst, (int)a.F)
  switch (key) {
    case 0: DGMI_LAUNCH(false, false, false); break;
    case 1: DGMI_LAUNCH(false, false, true); break;
    case 2: DGMI_LAUNCH(false, true, false); break;
    case 3: DGMI_LAUNCH(false, true, true); break;
    case 4: DGMI_LAUNCH(true, false, false); break;
    case 5: DGMI_LAUNCH(true, false, true); break;
    case 6: DGMI_LAUNCH(true, true, false); break;
    default: DGMI_LAUNCH(true, true, true); break;
  }
#undef DGMI_LAUNCH
  return hipGetLastError();
}

hipError_t launch_dword(const SpmmArgs& a, hipStream_t s) {
  dim3 grid((unsigned)((a.n_dst + kWavesPerBlock - 1) / kWavesPerBlock),
            (unsigned)((a.F + kWave - 1) / kWave));
  dim3 block(kWave * kWavesPerBlock);
  const int key = (a.vals ? 4 : 0) | (a.src_scale ? 2 : 0) | (a.dst_scale ? 1 : 0);
#define DGMI_LAUNCH(V, S, D)                                                         \
  hipLaunchKernelGGL((spmm_csr_dword_kernel<V, S, D>), grid, block, 0, s, a.indptr,   \
                     a.indices, a.vals, a.X, a.ldx, a.src_scale, a.dst_scale, a.Y,    \
                     a.ldy, a.n_dst, (int)a.F)
  switch (key) {
    case 0: DGMI_LAUNCH(false, false, false); break;
    case 1: DGMI_LAUNCH(false, false, true); break;
    case 2: DGMI_LAUNCH(false, true, false); break;
    case 3: DGMI_LAUNCH(false, true, true); break;
    case 4: DGMI_LAUNCH(true, false, false); break;
    case 5: DGMI_LAUNCH(true, false, true); break;
    case 6: DGMI_LAUNCH(true, true, false); break;
    default: DGMI_LAUNCH(true, true, true); break;
  }
#undef DGMI_LAUNCH
  return hipGetLastError();
}

}  // namespace

hipError_t spmm_csr_f32(const SpmmArgs& a, hipStream_t s) {
  if (a.n_dst == 0 || a.F == 0) return hipSuccess;
  const bool aligned = (a.F % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(a.Y) & 15) == 0);
  if (!aligned) return launch_dword(a, s);
  if (a.F <= 32) return launch_vec4<8>(a, s);
  if (a.F <= 64) return launch_vec4<16>(a, s);
  if (a.F <= 128) return launch_vec4<32>(a, s);
  return launch_vec4<64>(a, s);
}

}  // namespace dgmi
