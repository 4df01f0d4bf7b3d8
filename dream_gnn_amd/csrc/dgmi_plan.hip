// dgmi_plan.hip — nnz-balanced launch plan for the CSR SpMM (gfx950).
//
// A wave-per-row SpMM runs as long as its longest row.  The reference's graphs come in
// three regimes: regular kNN rows (64-128 edges), near-complete bipartite
// slices (600-700 edges per row, only ~700 rows: too few waves to fill 256 CUs), and
// power-law synthetic graphs (median 10^1, maximum 10^5..10^6).  The plan cuts every row
// into chunks of at most `chunk` edges, one wave each:
//     items[i]     = {row, start, end, slot}      slot = -1 when the row is a single chunk
//     long_rows[j] = {row, slot0, nchunks, 0}     rows whose chunk partials are summed, in
//                                                 chunk order, by the reduce pass
// Built on the device from indptr alone, without a host round-trip: per-row chunk counts ->
// two exclusive scans (rocPRIM) -> one fill pass.  Integer / index work, HBM-trivial
// (12 bytes per row in, 16 bytes per item out).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// counts[r] = (#chunks of row r) | (#chunks if the row is long, else 0) << 32
__global__ __launch_bounds__(kBlock) void count_chunks_kernel(const int32_t* __restrict__ indptr,
                                                              int64_t n_rows, int64_t nnz, int32_t chunk,
                                                              uint64_t* __restrict__ counts,
                                                              uint32_t* __restrict__ is_long) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += stride) {
    // a well-formed indptr has 0 <= deg <= nnz; anything else (an indptr built from unchecked,
    // out-of-range ids) is clamped so that the plan never outgrows the caps the host sized it by
    int64_t deg = (int64_t)indptr[r + 1] - (int64_t)indptr[r];
    deg = deg < 0 ? 0 : (deg > nnz ? nnz : deg);
    const uint32_t c = deg <= chunk ? 1u : (uint32_t)((deg + chunk - 1) / chunk);
    const bool lng = c > 1u;
    counts[r] = (uint64_t)c | ((uint64_t)(lng ? c : 0u) << 32);
    is_long[r] = lng ? 1u : 0u;
  }
}

__global__ __launch_bounds__(kBlock) void fill_plan_kernel(
    const int32_t* __restrict__ indptr, int64_t n_rows, int32_t chunk,
    const uint64_t* __restrict__ counts, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ long_offs, int64_t items_cap, int64_t long_cap,
    int64_t slots_cap, int32_t* __restrict__ plan) {
  int4* items = reinterpret_cast<int4*>(plan + kPlanHeaderWords);
  int4* longs = items + items_cap;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += stride) {
    const int32_t start = indptr[r], end = indptr[r + 1];
    const uint32_t c = (uint32_t)(counts[r] & 0xffffffffu);
    const uint32_t item0 = (uint32_t)(offs[r] & 0xffffffffu);
    const uint32_t slot0 = (uint32_t)(offs[r] >> 32);
    // Every store is bounded by the caps (they hold for any well-formed indptr; a malformed one
    // — see count_chunks_kernel — loses items instead of writing past the plan buffer).
    if (c == 1u) {
      if ((int64_t)item0 < items_cap) items[item0] = make_int4((int)r, start, max(start, end), -1);
    } else {
      for (uint32_t k = 0; k < c && (int64_t)item0 + k < items_cap && (int64_t)slot0 + k < slots_cap; ++k) {
        const int32_t s = start + (int32_t)k * chunk;
        const int32_t e = min(s + chunk, end);
        items[item0 + k] = make_int4((int)r, s, e, (int)(slot0 + k));
      }
      if ((int64_t)long_offs[r] < long_cap) longs[long_offs[r]] = make_int4((int)r, (int)slot0, (int)c, 0);
    }
    if (r == n_rows - 1) {
      plan[kPlanNumItems] = (int32_t)min((int64_t)item0 + c, items_cap);
      plan[kPlanNumLong] = (int32_t)min((int64_t)long_offs[r] + (c > 1u ? 1 : 0), long_cap);
      plan[kPlanNumSlots] = (int32_t)(slot0 + (c > 1u ? c : 0u));
      plan[kPlanChunk] = chunk;
    }
  }
}

}  // namespace

hipError_t spmm_plan_build(const int32_t* indptr, int64_t n_rows, int64_t nnz, int64_t chunk,
                           int32_t* plan, void* workspace, size_t* workspace_bytes, hipStream_t s) {
  size_t scan64 = 0, scan32 = 0;
  if (n_rows > 0) {
    hipError_t err = rocprim::exclusive_scan(nullptr, scan64, static_cast<uint64_t*>(nullptr),
                                             static_cast<uint64_t*>(nullptr), (uint64_t)0,
                                             (size_t)n_rows, rocprim::plus<uint64_t>(), s);
    if (err != hipSuccess) return err;
    err = rocprim::exclusive_scan(nullptr, scan32, static_cast<uint32_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), (uint32_t)0, (size_t)n_rows,
                                  rocprim::plus<uint32_t>(), s);
    if (err != hipSuccess) return err;
  }
  const size_t rows = (size_t)(n_rows > 0 ? n_rows : 1);
  const size_t off_counts = 0;
  const size_t off_offs = off_counts + align_up(rows * 8, 256);
  const size_t off_long = off_offs + align_up(rows * 8, 256);
  const size_t off_loffs = off_long + align_up(rows * 4, 256);
  const size_t off_tmp = off_loffs + align_up(rows * 4, 256);
  const size_t total = off_tmp + align_up(scan64 > scan32 ? scan64 : scan32, 256);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;

  hipError_t err = hipMemsetAsync(plan, 0, sizeof(int32_t) * kPlanHeaderWords, s);
  if (err != hipSuccess || n_rows == 0) return err;
  char* ws = static_cast<char*>(workspace);
  uint64_t* counts = reinterpret_cast<uint64_t*>(ws + off_counts);
  uint64_t* offs = reinterpret_cast<uint64_t*>(ws + off_offs);
  uint32_t* is_long = reinterpret_cast<uint32_t*>(ws + off_long);
  uint32_t* long_offs = reinterpret_cast<uint32_t*>(ws + off_loffs);
  void* tmp = ws + off_tmp;

  hipLaunchKernelGGL(count_chunks_kernel, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, indptr,
                     n_rows, nnz, (int32_t)chunk, counts, is_long);
  err = rocprim::exclusive_scan(tmp, scan64, counts, offs, (uint64_t)0, (size_t)n_rows,
                                rocprim::plus<uint64_t>(), s);
  if (err != hipSuccess) return err;
  err = rocprim::exclusive_scan(tmp, scan32, is_long, long_offs, (uint32_t)0, (size_t)n_rows,
                                rocprim::plus<uint32_t>(), s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(fill_plan_kernel, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, indptr, n_rows,
                     (int32_t)chunk, counts, offs, long_offs, plan_items_cap(n_rows, nnz, chunk),
                     plan_long_cap(nnz, chunk), plan_slots_cap(nnz, chunk), plan);
  return hipGetLastError();
}

}  // namespace dgmi
