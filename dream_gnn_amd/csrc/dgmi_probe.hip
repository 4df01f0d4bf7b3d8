// dgmi_probe.hip — measurement probes (bench.py): what the memory hierarchy gives the SpMM's
// access SHAPE when nothing else is in the way.
//
// The §8(d) byte model counts every per-edge row read, and those reads are served by L2 and the
// Infinity Cache, not by HBM — so the roof that actually bounds the gather kernels is the rate at
// which a CU can pull whole feature rows by index out of the cache level the table lives in.
// `probe_row_gather` reproduces exactly the product kernels' access: LPR lanes x 16 B cover one
// row, 64 / LPR rows per wave-instruction, 8 gathers in flight per lane, sums kept in registers —
// but with row ids computed from a hash (no index stream, no row bookkeeping, no output beyond
// one row per wave).  Run over a table that fits one L2 it measures the L2 indexed-row ceiling;
// over the product's own table, what an unsliced gather gets from the Infinity Cache / HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"
#include "dgmi_segment.h"

namespace dgmi {
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// Each LPR-lane group gathers `per_group` pseudo-random rows of `table` (n_rows x F, ld = F) and
// writes their sum to out[group].  `window`: rows are drawn from [base, base + window) where base
// depends on blockIdx % 8 when `per_xcd` — every XCD then works on its own window of the table
// (the L2-local regime of the XCD-sliced kernel).
template <int LPR>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void probe_row_gather_kernel(
    const float* __restrict__ table, int64_t n_rows, int F, int64_t groups, int64_t per_group, int64_t window,
    int per_xcd, float* __restrict__ out) {
  constexpr int NG = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR, glane = lane % LPR;
  const int64_t gid = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * NG + grp;
  if (gid >= groups) return;  // group-uniform
  int col = glane * 4;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;
  int64_t base = 0;
  if (per_xcd) base = (int64_t)(blockIdx.x % 8u) * window;
  if (base + window > n_rows) base = n_rows - window;
  const float* T = table + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t state = mix32((uint32_t)gid * 2654435761u + 12345u);
  for (int64_t i = 0; i < per_group; i += kUnroll) {
    float4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      state = mix32(state + 0x9e3779b9u);
      const int64_t r = base + (int64_t)(state % (uint32_t)window);
      v[u] = ld4(T + r * F);
    }
    tree_sum(v, kUnroll);
    acc.x += v[0].x;
    acc.y += v[0].y;
    acc.z += v[0].z;
    acc.w += v[0].w;
  }
  if (col_ok) *reinterpret_cast<float4*>(out + gid * F + col) = acc;
}

}  // namespace

hipError_t probe_row_gather(const float* table, int64_t n_rows, int64_t F, int64_t groups, int64_t per_group,
                            int64_t window, int per_xcd, float* out, hipStream_t s) {
  const int64_t f4 = F / 4;
  int lpr = 64;
  for (int c : {8, 16, 32, 64})
    if (f4 <= c) {
      lpr = c;
      break;
    }
  const int64_t gpb = (int64_t)kWavesPerBlock * (kWave / lpr);
  dim3 grid((unsigned)((groups + gpb - 1) / gpb)), block(kWave * kWavesPerBlock);
#define DGMI_PROBE(L)                                                                                      \
  hipLaunchKernelGGL((probe_row_gather_kernel<L>), grid, block, 0, s, table, n_rows, (int)F, groups, per_group, \
                     window, per_xcd, out)
  switch (lpr) {
    case 8: DGMI_PROBE(8); break;
    case 16: DGMI_PROBE(16); break;
    case 32: DGMI_PROBE(32); break;
    default: DGMI_PROBE(64); break;
  }
#undef DGMI_PROBE
  return hipGetLastError();
}

}  // namespace dgmi
