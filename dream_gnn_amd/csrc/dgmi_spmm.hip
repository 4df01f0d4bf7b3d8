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
//   * one 64-lane wave owns one *segment* of one destination row (segmented
//     reduce = the wave's own registers; no atomics, fixed summation order ->
//     bitwise reproducible).  Unplanned launch: segment = whole row.  Planned
//     launch (dgmi_plan.hip): rows longer than `chunk` edges are cut into
//     chunks so that a power-law degree distribution or a small dense graph
//     still fills the chip; chunk partials are summed in chunk order by a
//     second kernel;
//   * a source row is read with 16-B loads: LPR lanes cover (a tile of) a row,
//     so a wave-instruction carries 64/LPR whole rows (F=128: 2 rows x 512 B =
//     1 KiB, the widest coalesced access the hardware has);
//   * the segment's column ids (and values) are read 64 at a time, one per
//     lane, coalesced, one batch ahead of use, and handed to the row loads by
//     ds_bpermute; 8 row loads are kept in flight per wave (8 KiB at F=128,
//     x 16-32 waves per CU) to cover Infinity-Cache / HBM latency;
//   * the 64/LPR partial sums are combined with xor-shuffles and written with
//     one coalesced 16-B-per-lane store, scaled by dst_scale.
// Feature widths that are not a multiple of 4 (layer-0's 341, layers.py:55-57)
// or rows that are not 16-B aligned take the dword kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"
#include "dgmi_segment.h"

namespace dgmi {

namespace {

struct Segment {
  int64_t row;
  int start, end, slot;  // slot < 0: final result -> Y[row]; else partial -> P[slot]
};

// PLANNED = false: wave w handles row w whole.  PLANNED = true: wave w handles
// item w of the plan (dgmi_plan.hip): {row, start, end, slot}.
template <bool PLANNED>
__device__ __forceinline__ bool fetch_segment(const int32_t* __restrict__ indptr,
                                              const int32_t* __restrict__ plan, int64_t n_dst,
                                              Segment& sg) {
  const int wave = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (!PLANNED) {
    if (w >= n_dst) return false;
    sg.row = w;
    sg.start = indptr[w];
    sg.end = indptr[w + 1];
    sg.slot = -1;
    return true;
  }
  if (w >= plan[kPlanNumItems]) return false;
  const int4 it = reinterpret_cast<const int4*>(plan + kPlanHeaderWords)[w];
  sg.row = it.x;
  sg.start = it.y;
  sg.end = it.z;
  sg.slot = it.w;
  return true;
}

// grid.x = ceil(#segments / 4), grid.y = ceil(F / (4*LPR)); block = 256 (4 waves).
template <int LPR, bool HAS_VALS, bool HAS_SS, bool KEEP, bool PLANNED>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_csr_vec4_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, const float* __restrict__ dst_scale,
    float* __restrict__ Y, int64_t ldy, int64_t n_dst, int F,
    const int32_t* __restrict__ plan, float* __restrict__ P, int64_t ldp,
    const int32_t* __restrict__ eid, const KeepSeg* __restrict__ keep, int n_keep, Epilogue ep) {
  Segment sg;
  if (!fetch_segment<PLANNED>(indptr, plan, n_dst, sg)) return;  // wave-uniform
  const int lane = threadIdx.x & (kWave - 1);
  int col = ((int)blockIdx.y * LPR + (lane % LPR)) * 4;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;  // keep the loads in bounds; result discarded
  float4 acc = segment_vec4<LPR, HAS_VALS, HAS_SS, KEEP>(indices, vals, src_scale, eid, keep, n_keep, X + col, ldx,
                                                         sg.start, sg.end, lane);
  if (lane < LPR && col_ok) {
    if (PLANNED && sg.slot >= 0) {
      *reinterpret_cast<float4*>(P + (int64_t)sg.slot * ldp + col) = acc;
    } else {
      if (dst_scale != nullptr) {
        const float d = dst_scale[sg.row];
        acc.x *= d;
        acc.y *= d;
        acc.z *= d;
        acc.w *= d;
      }
      *reinterpret_cast<float4*>(Y + sg.row * ldy + col) = epilogue4(ep, acc, sg.row, col);
    }
  }
}

// Any F, any alignment: lanes across 64 consecutive columns, one dword each;
// grid.y = ceil(F/64).
template <bool HAS_VALS, bool HAS_SS, bool KEEP, bool PLANNED>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_csr_dword_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, const float* __restrict__ dst_scale,
    float* __restrict__ Y, int64_t ldy, int64_t n_dst, int F,
    const int32_t* __restrict__ plan, float* __restrict__ P, int64_t ldp,
    const int32_t* __restrict__ eid, const KeepSeg* __restrict__ keep, int n_keep, Epilogue ep) {
  Segment sg;
  if (!fetch_segment<PLANNED>(indptr, plan, n_dst, sg)) return;
  const int lane = threadIdx.x & (kWave - 1);
  int col = (int)blockIdx.y * kWave + lane;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;
  float acc = segment_dword<HAS_VALS, HAS_SS, KEEP>(indices, vals, src_scale, eid, keep, n_keep, X + col, ldx,
                                                    sg.start, sg.end, lane);
  if (col_ok) {
    if (PLANNED && sg.slot >= 0) {
      P[(int64_t)sg.slot * ldp + col] = acc;
    } else {
      if (dst_scale != nullptr) acc *= dst_scale[sg.row];
      Y[sg.row * ldy + col] = epilogue1(ep, acc, sg.row, col);
    }
  }
}

// Second pass of a planned launch: Y[row] = dst_scale[row] * sum_k P[slot0 + k], chunks
// added in chunk order (tree over groups of 8, then a chain) -> deterministic.
// One wave per (long row, 64-column tile).
template <bool HAS_DS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_reduce_partials_kernel(
    const int32_t* __restrict__ plan, const float* __restrict__ P, int64_t ldp,
    const float* __restrict__ dst_scale, float* __restrict__ Y, int64_t ldy, int F,
    int64_t items_cap, Epilogue ep) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & (kWave - 1);
  const int col = (int)blockIdx.y * kWave + lane;
  if (col >= F) return;
  const int64_t n_long = plan[kPlanNumLong];
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  // grid-stride over the long rows: the host sizes the grid from an upper bound (it never
  // reads the plan back), so with no long row every wave leaves here at once.
  for (int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave; w < n_long; w += stride) {
    const int4 lr = reinterpret_cast<const int4*>(plan + kPlanHeaderWords)[items_cap + w];
    const int64_t row = lr.x;
    const int slot0 = lr.y, n = lr.z;
    const float* p = P + (int64_t)slot0 * ldp + col;
    float acc = 0.f;
    int k = 0;
    for (; k + kUnroll <= n; k += kUnroll) {
      float v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) v[u] = p[(int64_t)(k + u) * ldp];
#pragma unroll
      for (int span = 1; span < kUnroll; span <<= 1) {
#pragma unroll
        for (int u = 0; u + span < kUnroll; u += 2 * span) v[u] += v[u + span];
      }
      acc += v[0];
    }
    for (; k < n; ++k) acc += p[(int64_t)k * ldp];
    if (HAS_DS) acc *= dst_scale[row];
    Y[row * ldy + col] = epilogue1(ep, acc, row, col);
  }
}

inline unsigned blocks_for(int64_t waves) {
  return (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
}

template <int LPR, bool PLANNED>
hipError_t launch_vec4(const SpmmArgs& a, int64_t segments, hipStream_t s) {
  dim3 grid(blocks_for(segments), (unsigned)((a.F + 4 * LPR - 1) / (4 * LPR)));
  dim3 block(kWave * kWavesPerBlock);
  const int key = (a.vals ? 4 : 0) | (a.src_scale ? 2 : 0) | (a.n_keep > 0 ? 1 : 0);
#define DGMI_LAUNCH(V, S, K)                                                              \
  hipLaunchKernelGGL((spmm_csr_vec4_kernel<LPR, V, S, K, PLANNED>), grid, block, 0, s,     \
                     a.indptr, a.indices, a.vals, a.X, a.ldx, a.src_scale, a.dst_scale,    \
                     a.Y, a.ldy, a.n_dst, (int)a.F, a.plan, a.partials, a.ldp, a.eid,      \
                     static_cast<const KeepSeg*>(a.keep), a.n_keep, a.ep)
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

template <bool PLANNED>
hipError_t launch_dword(const SpmmArgs& a, int64_t segments, hipStream_t s) {
  dim3 grid(blocks_for(segments), (unsigned)((a.F + kWave - 1) / kWave));
  dim3 block(kWave * kWavesPerBlock);
  const int key = (a.vals ? 4 : 0) | (a.src_scale ? 2 : 0) | (a.n_keep > 0 ? 1 : 0);
#define DGMI_LAUNCH(V, S, K)                                                              \
  hipLaunchKernelGGL((spmm_csr_dword_kernel<V, S, K, PLANNED>), grid, block, 0, s,         \
                     a.indptr, a.indices, a.vals, a.X, a.ldx, a.src_scale, a.dst_scale,    \
                     a.Y, a.ldy, a.n_dst, (int)a.F, a.plan, a.partials, a.ldp, a.eid,      \
                     static_cast<const KeepSeg*>(a.keep), a.n_keep, a.ep)
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

template <bool PLANNED>
hipError_t dispatch(const SpmmArgs& a, int64_t segments, hipStream_t s) {
  const bool aligned = (a.F % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(a.Y) & 15) == 0) &&
                       (!PLANNED || ((a.ldp % 4 == 0) &&
                                     ((reinterpret_cast<uintptr_t>(a.partials) & 15) == 0))) &&
                       (a.ep.mask == nullptr ||
                        ((a.ep.ldm % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.ep.mask) & 15) == 0)));
  if (!aligned) return launch_dword<PLANNED>(a, segments, s);
  switch (pick_lpr(a.F)) {
    case 8: return launch_vec4<8, PLANNED>(a, segments, s);
    case 16: return launch_vec4<16, PLANNED>(a, segments, s);
    case 32: return launch_vec4<32, PLANNED>(a, segments, s);
    default: return launch_vec4<64, PLANNED>(a, segments, s);
  }
}

}  // namespace

hipError_t spmm_csr_f32(const SpmmArgs& a, hipStream_t s) {
  if (a.n_dst == 0 || a.F == 0) return hipSuccess;
  if (a.plan == nullptr) return dispatch<false>(a, a.n_dst, s);

  const int64_t items_cap = plan_items_cap(a.n_dst, a.nnz, a.chunk);
  const int64_t long_cap = plan_long_cap(a.nnz, a.chunk);
  hipError_t err = dispatch<true>(a, items_cap, s);
  if (err != hipSuccess || long_cap == 0) return err;
  const unsigned rblocks = blocks_for(long_cap) < 1024u ? blocks_for(long_cap) : 1024u;
  dim3 grid(rblocks, (unsigned)((a.F + kWave - 1) / kWave));
  dim3 block(kWave * kWavesPerBlock);
  if (a.dst_scale)
    hipLaunchKernelGGL(spmm_reduce_partials_kernel<true>, grid, block, 0, s, a.plan, a.partials,
                       a.ldp, a.dst_scale, a.Y, a.ldy, (int)a.F, items_cap, a.ep);
  else
    hipLaunchKernelGGL(spmm_reduce_partials_kernel<false>, grid, block, 0, s, a.plan, a.partials,
                       a.ldp, a.dst_scale, a.Y, a.ldy, (int)a.F, items_cap, a.ep);
  return hipGetLastError();
}

}  // namespace dgmi
