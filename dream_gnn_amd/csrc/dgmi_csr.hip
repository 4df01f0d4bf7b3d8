// dgmi_csr.hip — device-side stable COO -> CSR for gfx950.
//
// Replaces the host/DGL graph build the reference pays every training step
// (augmentation.py:65 builds a new dgl.heterograph from a randperm prefix;
// DGL then sorts its edges by destination lazily) and the coalesce->CSR inside
// th.spmm (layers.py:312) for the shuffled COO of augmentation.py:117-124.
//
// Pipeline (all on `stream`, no host sync, no atomics -> bit-exact & reproducible):
//   1. iota + range check : tmp_eid[e] = e ; flag |= row[e] outside [0, n_rows) or col[e] outside [0, n_cols)
//   2. stable LSD radix sort of (row, e) pairs on the low ceil(log2 n_rows) bits
//      — rocPRIM's device radix sort (a plain library primitive; stability is
//      what makes eid the original order inside a row)
//   3. row boundaries     : thread p in [0, E] compares sorted_row[p-1] / [p] and
//      writes indptr[r] = p for every r in (prev, cur] — also fills empty rows
//   4. gather             : indices[p] = col[eid[p]]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void iota_check_kernel(const int32_t* __restrict__ row,
                                                            const int32_t* __restrict__ col,
                                                            int64_t E, int32_t n_rows, int32_t n_cols,
                                                            int32_t* __restrict__ tmp_eid,
                                                            int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
    tmp_eid[e] = (int32_t)e;
    const int32_t r = row[e];
    bad |= (r < 0) | (r >= n_rows);
    if (n_cols > 0) {
      const int32_t c = col[e];
      bad |= (c < 0) | (c >= n_cols);
    }
  }
  if (bad) *flag = 1;  // benign race: every writer stores the same value
}

__global__ __launch_bounds__(kBlock) void boundaries_gather_kernel(
    const int32_t* __restrict__ sorted_row, const int32_t* __restrict__ eid,
    const int32_t* __restrict__ col, int64_t E, int32_t n_rows,
    int32_t* __restrict__ indptr, int32_t* __restrict__ indices) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p <= E; p += stride) {
    // rows in (prev, cur] start at p.  Out-of-range ids (flagged in step 1) are
    // clamped so no store leaves indptr[0..n_rows].
    int32_t prev = p > 0 ? sorted_row[p - 1] : -1;
    int32_t cur = p < E ? sorted_row[p] : n_rows;
    prev = max(-1, min(prev, n_rows));
    cur = max(-1, min(cur, n_rows));
    for (int32_t r = prev + 1; r <= cur; ++r) indptr[r] = (int32_t)p;
    if (p < E) indices[p] = col[eid[p]];
  }
}

__global__ __launch_bounds__(kBlock) void gather_f32_kernel(const float* __restrict__ in,
                                                            const int32_t* __restrict__ perm,
                                                            int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride)
    out[p] = in[perm[p]];
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  return (unsigned)b;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int bits_for(int64_t n_rows) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < n_rows) ++b;
  return b;
}

}  // namespace

hipError_t csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                            int64_t n_rows, int64_t n_cols, int32_t* indptr, int32_t* indices,
                            int32_t* eid, void* workspace, size_t* workspace_bytes,
                            hipStream_t s) {
  const int end_bit = bits_for(n_rows);
  size_t sort_bytes = 0;
  if (E > 0) {
    hipError_t err = rocprim::radix_sort_pairs(
        nullptr, sort_bytes, reinterpret_cast<const uint32_t*>(row),
        static_cast<uint32_t*>(nullptr), static_cast<const int32_t*>(nullptr), eid,
        (size_t)E, 0u, (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  const size_t off_flag = 0;
  const size_t off_keys = 256;
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
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + off_keys);
  int32_t* tmp_eid = reinterpret_cast<int32_t*>(ws + off_iota);
  void* sort_tmp = ws + off_sort;

  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  if (E > 0) {
    hipLaunchKernelGGL(iota_check_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, row, col, E,
                       (int32_t)n_rows, (int32_t)n_cols, tmp_eid, flag);
    err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, reinterpret_cast<const uint32_t*>(row),
                                    reinterpret_cast<uint32_t*>(keys_out),
                                    static_cast<const int32_t*>(tmp_eid), eid, (size_t)E, 0u,
                                    (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(boundaries_gather_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s,
                     keys_out, eid, col, E, (int32_t)n_rows, indptr, indices);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Source-sliced CSR for the XCD-local SpMM (dgmi_sliced.hip): the source id range is cut into
// `n_slices` contiguous slices of `slice_width` ids and the edges are sorted, stably, by
// key = slice(col) * n_rows + row.  segptr[s * n_rows + r] .. segptr[s * n_rows + r + 1] are the
// edges of destination row r whose source lies in slice s: n_slices column-blocked CSRs stacked
// slice-major, built by the same sort -> boundaries -> gather pipeline on the composite key.
namespace {
__global__ __launch_bounds__(kBlock) void slice_key_kernel(const int32_t* __restrict__ row,
                                                           const int32_t* __restrict__ col, int64_t E,
                                                           int32_t n_rows, int32_t n_cols,
                                                           int32_t n_slices, int32_t slice_width,
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
    if (oob) r = c = 0;  // keep the key inside [0, n_slices * n_rows); the flag reports it
    int32_t sl = c / slice_width;
    if (sl >= n_slices) sl = n_slices - 1;
    key[e] = sl * n_rows + r;
  }
  if (bad) *flag = 1;
}
}  // namespace

hipError_t csr_sliced_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E, int64_t n_rows,
                                   int64_t n_cols, int64_t n_slices, int64_t slice_width,
                                   int32_t* segptr, int32_t* indices, int32_t* eid, void* workspace,
                                   size_t* workspace_bytes, hipStream_t s) {
  const int64_t n_keys = n_rows * n_slices;
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
    hipLaunchKernelGGL(slice_key_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, row, col, E,
                       (int32_t)n_rows, (int32_t)n_cols, (int32_t)n_slices, (int32_t)slice_width,
                       keys_in, tmp_eid, flag);
    err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, reinterpret_cast<const uint32_t*>(keys_in),
                                    reinterpret_cast<uint32_t*>(keys_out),
                                    static_cast<const int32_t*>(tmp_eid), eid, (size_t)E, 0u,
                                    (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(boundaries_gather_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, keys_out,
                     eid, col, E, (int32_t)n_keys, segptr, indices);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The same sliced layout derived from an existing CSR.  The CSR is already sorted by row (stably),
// so a STABLE partition of its positions by slice(col) leaves exactly the order (slice, row, input
// order): one radix pass over ceil(log2 n_slices) bits instead of the 2-3 passes of a full sort on
// the composite key, and the index / edge-id gathers read near-sequentially (the positions of a slice
// are increasing).  Bit-identical to csr_sliced_from_coo_i32 on the same edge list.
namespace {
__global__ __launch_bounds__(kBlock) void slice_of_position_kernel(const int32_t* __restrict__ indices, int64_t E,
                                                                   int32_t n_cols, int32_t n_slices, int32_t slice_width,
                                                                   uint32_t* __restrict__ key, int32_t* __restrict__ pos,
                                                                   int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  bool bad = false;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < E; p += stride) {
    pos[p] = (int32_t)p;
    int32_t c = indices[p];
    const bool oob = (c < 0) | (c >= n_cols);
    bad |= oob;
    if (oob) c = 0;
    int32_t sl = c / slice_width;
    if (sl >= n_slices) sl = n_slices - 1;
    key[p] = (uint32_t)sl;
  }
  if (bad) *flag = 1;
}

// i-th entry of the sliced order: CSR position pos[i]; its row by binary search in indptr
__global__ __launch_bounds__(kBlock) void sliced_from_csr_gather_kernel(
    const uint32_t* __restrict__ slice_sorted, const int32_t* __restrict__ pos, const int32_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const int32_t* __restrict__ eid, int64_t E, int32_t n_rows,
    int32_t* __restrict__ full_key, int32_t* __restrict__ s_indices, int32_t* __restrict__ s_eid) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < E; i += stride) {
    const int32_t p = pos[i];
    int32_t lo = 0, hi = n_rows;  // first j in (0, n_rows] with indptr[j] > p; the row is j - 1
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if (indptr[mid + 1] > p)
        hi = mid;
      else
        lo = mid + 1;
    }
    full_key[i] = (int32_t)slice_sorted[i] * n_rows + lo;
    s_indices[i] = indices[p];
    s_eid[i] = eid[p];
  }
}

__global__ __launch_bounds__(kBlock) void boundaries_kernel(const int32_t* __restrict__ sorted_key, int64_t E, int32_t n_keys,
                                                            int32_t* __restrict__ ptr) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p <= E; p += stride) {
    int32_t prev = p > 0 ? sorted_key[p - 1] : -1;
    int32_t cur = p < E ? sorted_key[p] : n_keys;
    prev = max(-1, min(prev, n_keys));
    cur = max(-1, min(cur, n_keys));
    for (int32_t r = prev + 1; r <= cur; ++r) ptr[r] = (int32_t)p;
  }
}
}  // namespace

hipError_t csr_sliced_from_csr_i32(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t E,
                                   int64_t n_rows, int64_t n_cols, int64_t n_slices, int64_t slice_width, int32_t* segptr,
                                   int32_t* s_indices, int32_t* s_eid, void* workspace, size_t* workspace_bytes,
                                   hipStream_t s) {
  const int end_bit = bits_for(n_slices);
  size_t sort_bytes = 0;
  if (E > 0) {
    hipError_t err = rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<const uint32_t*>(nullptr),
                                               static_cast<uint32_t*>(nullptr), static_cast<const int32_t*>(nullptr),
                                               s_eid, (size_t)E, 0u, (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
  }
  const size_t off_flag = 0;
  const size_t off_key_in = 256;  // slice of each CSR position; reused for the composite key after the sort
  const size_t off_key_out = off_key_in + align_up((size_t)E * 4, 256);
  const size_t off_pos_in = off_key_out + align_up((size_t)E * 4, 256);
  const size_t off_pos_out = off_pos_in + align_up((size_t)E * 4, 256);
  const size_t off_sort = off_pos_out + align_up((size_t)E * 4, 256);
  const size_t total = off_sort + align_up(sort_bytes, 256);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + off_flag);
  uint32_t* key_in = reinterpret_cast<uint32_t*>(ws + off_key_in);
  uint32_t* key_out = reinterpret_cast<uint32_t*>(ws + off_key_out);
  int32_t* pos_in = reinterpret_cast<int32_t*>(ws + off_pos_in);
  int32_t* pos_out = reinterpret_cast<int32_t*>(ws + off_pos_out);
  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  const int32_t n_keys = (int32_t)(n_rows * n_slices);
  if (E > 0) {
    hipLaunchKernelGGL(slice_of_position_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, indices, E, (int32_t)n_cols,
                       (int32_t)n_slices, (int32_t)slice_width, key_in, pos_in, flag);
    err = rocprim::radix_sort_pairs(ws + off_sort, sort_bytes, key_in, key_out, pos_in, pos_out, (size_t)E, 0u,
                                    (unsigned)end_bit, s);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(sliced_from_csr_gather_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, key_out, pos_out, indptr, indices,
                       eid, E, (int32_t)n_rows, reinterpret_cast<int32_t*>(key_in), s_indices, s_eid);
  }
  hipLaunchKernelGGL(boundaries_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, reinterpret_cast<const int32_t*>(key_in), E,
                     n_keys, segptr);
  return hipGetLastError();
}

hipError_t gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                      hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, in, perm, n, out);
  return hipGetLastError();
}

}  // namespace dgmi
