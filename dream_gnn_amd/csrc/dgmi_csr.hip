// dgmi_csr.hip — device-side stable COO -> CSR for gfx950.
//
// Replaces the host/DGL graph build the reference pays every training step
// (augmentation.py:65 builds a new dgl.heterograph from a randperm prefix;
// DGL then sorts its edges by destination lazily) and the coalesce->CSR inside
// th.spmm (layers.py:312) for the shuffled COO of augmentation.py:117-124.
//
// Pipeline (all on `stream`, no host sync, no atomics -> bit-exact & reproducible):
//   1. stable LSD radix sort of the records (row, e, col) on the low ceil(log2 n_rows) bits — the hand-written
//      sort of dgmi_sort.hip (<= 9-bit digits, the id range check and the iota of e folded into its first pass,
//      the column carried as payload so that no gather pass follows); stability is what makes eid the original
//      order inside a row
//   2. row boundaries : thread p in [0, E] compares sorted_row[p-1] / [p] and writes indptr[r] = p for every r in
//      (prev, cur] — also fills empty rows
// The source-sliced layouts use the same two steps on the key slice * n_rows + row (from COO), or ONE sort pass on
// the slice bits of an existing CSR, whose positions are already in (row, input) order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cstring>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;

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

// ptr[k] = first sorted position whose key is >= k, for k in [0, n_keys]: row boundaries (and every empty row in
// between).  split_bits > 0: the sorted key is (slice << split_bits) | row and stands for slice * n_rows + row.
// A long run of empty keys (edges confined to a few rows of a large range, an empty trailing slice) is filled by the
// whole wave, not by the one lane that found it: 6 M empty keys cost one lane 115 ms.
constexpr int kLongGap = 64;
__global__ __launch_bounds__(kBlock) void boundaries_kernel(const int32_t* __restrict__ sorted_key, int64_t E, int32_t n_rows,
                                                            int split_bits, int32_t n_keys, int32_t* __restrict__ ptr) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int lane = threadIdx.x & 63;
  const int32_t mask = split_bits > 0 ? (1 << split_bits) - 1 : -1;
  for (int64_t p0 = (int64_t)blockIdx.x * kBlock + (threadIdx.x - lane); p0 <= E; p0 += stride) {  // wave-uniform
    const int64_t p = p0 + lane;
    int32_t prev = 0, cur = -1;  // lanes past E: nothing to fill
    if (p <= E) {
      prev = -1;
      cur = n_keys;
      if (p > 0) {
        const int32_t k = sorted_key[p - 1];
        prev = split_bits > 0 ? (k >> split_bits) * n_rows + (k & mask) : k;
      }
      if (p < E) {
        const int32_t k = sorted_key[p];
        cur = split_bits > 0 ? (k >> split_bits) * n_rows + (k & mask) : k;
      }
      // out-of-range ids (flagged) are clamped so that no store leaves ptr[0..n_keys]
      prev = max(-1, min(prev, n_keys));
      cur = max(-1, min(cur, n_keys));
    }
    const bool is_long = cur - prev > kLongGap;
    if (!is_long)
      for (int32_t r = prev + 1; r <= cur; ++r) ptr[r] = (int32_t)p;
    uint64_t todo = __ballot(is_long);
    while (todo) {
      const int l = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int32_t pv = __shfl(prev, l), cu = __shfl(cur, l), pp = __shfl((int32_t)p, l);
      for (int32_t r = pv + 1 + lane; r <= cu; r += 64) ptr[r] = pp;
    }
  }
}

}  // namespace

hipError_t csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                            int64_t n_rows, int64_t n_cols, int32_t* indptr, int32_t* indices,
                            int32_t* eid, void* workspace, size_t* workspace_bytes,
                            hipStream_t s) {
  const int end_bit = bits_for(n_rows);
  // hand-written sort: records (row, e, col) in up to 9-bit digits, id range check folded into the first pass
  // (dgmi_sort.hip); the boundary pass only reads the sorted rows — the columns arrive sorted
  const size_t off_flag = 0, off_keys = 256;
  const size_t off_sort = off_keys + align_up((size_t)E * 4, 256);
  const size_t total = off_sort + radix_sort_workspace_bytes(E, end_bit);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + off_flag);
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + off_keys);
  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  err = radix_sort_records(row, nullptr, col, E, 0, end_bit, (int32_t)n_rows, (int32_t)n_cols, keys_out, eid, indices, flag,
                           ws + off_sort, s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(boundaries_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, keys_out, E, (int32_t)n_rows, 0, (int32_t)n_rows, indptr);
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
                                                           int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += stride) {
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
  const size_t off_flag = 0;
  const size_t off_keys_in = 256;
  const size_t off_keys = off_keys_in + align_up((size_t)E * 4, 256);
  const size_t off_sort = off_keys + align_up((size_t)E * 4, 256);
  const size_t total = off_sort + radix_sort_workspace_bytes(E, end_bit);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + off_flag);
  int32_t* keys_in = reinterpret_cast<int32_t*>(ws + off_keys_in);
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + off_keys);
  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  if (E > 0) {
    hipLaunchKernelGGL(slice_key_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, row, col, E,
                       (int32_t)n_rows, (int32_t)n_cols, (int32_t)n_slices, (int32_t)slice_width, keys_in, flag);
    err = radix_sort_records(keys_in, nullptr, col, E, 0, end_bit, 0, 0, keys_out, eid, indices, flag, ws + off_sort, s);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(boundaries_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, keys_out, E, (int32_t)n_keys, 0, (int32_t)n_keys, segptr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The same sliced layout derived from an existing CSR.  The CSR is already sorted by row (stably),
// so a STABLE partition of its positions by slice(col) leaves exactly the order (slice, row, input
// order): one radix pass over ceil(log2 n_slices) bits instead of the 2-3 passes of a full sort on
// the composite key, and the index / edge-id gathers read near-sequentially (the positions of a slice
// are increasing).  Bit-identical to csr_sliced_from_coo_i32 on the same edge list.
namespace {
// key of CSR position p: (slice of its column << row_bits) | its row.  A thread takes kPosPerThread CONSECUTIVE positions:
// ONE binary search in indptr for the first of them (17 dependent, L2-resident loads), then it walks — the row only moves
// on where a row ends inside the chunk.  (Until round 4 every position ran its own search: 70-110 us of the 190 us this
// builder took on a 10 M-edge CSR.)
// PPT = 1 (short lists: every position its own search — more threads in flight, 0.038 against 0.060 ms at 8 k edges) / 8.
template <int kPosPerThread>
__global__ __launch_bounds__(kBlock) void position_key_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                              int64_t E, int32_t n_rows, int32_t n_cols, int32_t n_slices,
                                                              int32_t slice_width, int row_bits, int32_t* __restrict__ key,
                                                              int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * kBlock * kPosPerThread;
  bool bad = false;
  for (int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPosPerThread; p0 < E; p0 += stride) {
    int32_t lo = 0, hi = n_rows;  // first j with indptr[j + 1] > p0: the row of position p0
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if (indptr[mid + 1] > (int32_t)p0)
        hi = mid;
      else
        lo = mid + 1;
    }
    if (lo >= n_rows) lo = n_rows - 1;  // an indptr that does not cover E positions (garbage in): stay in range
    int32_t row_end = indptr[lo + 1];
    const int n = E - p0 < kPosPerThread ? (int)(E - p0) : kPosPerThread;
    for (int i = 0; i < n; ++i) {
      const int32_t p = (int32_t)p0 + i;
      // the next row, or a few empty ones, are stepped over; a long run of empty rows (sparse graphs: 70 000 rows, 8 191
      // edges) is searched, not walked
      for (int step = 0; p >= row_end && lo < n_rows - 1; ++step) {
        if (step == 4) {
          int32_t hi2 = n_rows;
          ++lo;
          while (lo < hi2) {
            const int32_t mid = (lo + hi2) >> 1;
            if (indptr[mid + 1] > p)
              hi2 = mid;
            else
              lo = mid + 1;
          }
          if (lo >= n_rows) lo = n_rows - 1;
          row_end = indptr[lo + 1];
          break;
        }
        row_end = indptr[++lo + 1];
      }
      int32_t c = indices[p];
      const bool oob = (c < 0) | (c >= n_cols);
      bad |= oob;
      if (oob) c = 0;
      int32_t sl = c / slice_width;
      if (sl >= n_slices) sl = n_slices - 1;
      key[p] = (sl << row_bits) | lo;
    }
  }
  if (bad) *flag = 1;
}
}  // namespace

hipError_t csr_sliced_from_csr_i32(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t E,
                                   int64_t n_rows, int64_t n_cols, int64_t n_slices, int64_t slice_width, int32_t* segptr,
                                   int32_t* s_indices, int32_t* s_eid, void* workspace, size_t* workspace_bytes,
                                   hipStream_t s) {
  const int slice_bits = bits_for(n_slices), row_bits = bits_for(n_rows);  // n_rows * n_slices < 2^31 (checked by the ABI)
  const size_t off_flag = 0;
  const size_t off_key_in = 256;
  const size_t off_key_out = off_key_in + align_up((size_t)E * 4, 256);
  const size_t off_sort = off_key_out + align_up((size_t)E * 4, 256);
  const size_t total = off_sort + radix_sort_workspace_bytes(E, slice_bits);
  if (workspace == nullptr) {
    *workspace_bytes = total;
    return hipSuccess;
  }
  if (*workspace_bytes < total) return hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + off_flag);
  int32_t* key_in = reinterpret_cast<int32_t*>(ws + off_key_in);
  int32_t* key_out = reinterpret_cast<int32_t*>(ws + off_key_out);
  hipError_t err = hipMemsetAsync(flag, 0, 256, s);
  if (err != hipSuccess) return err;
  const int32_t n_keys = (int32_t)(n_rows * n_slices);
  if (E > 0) {
    if (E >= (1 << 18))
      hipLaunchKernelGGL(position_key_kernel<8>, dim3(grid_for((E + 7) / 8)), dim3(kBlock), 0, s, indptr, indices, E, (int32_t)n_rows,
                         (int32_t)n_cols, (int32_t)n_slices, (int32_t)slice_width, row_bits, key_in, flag);
    else
      hipLaunchKernelGGL(position_key_kernel<1>, dim3(grid_for(E)), dim3(kBlock), 0, s, indptr, indices, E, (int32_t)n_rows,
                         (int32_t)n_cols, (int32_t)n_slices, (int32_t)slice_width, row_bits, key_in, flag);
    // ONE pass on the slice bits: the CSR is already in (row, input) order, the records carry eid and column along
    err = radix_sort_records(key_in, eid, indices, E, row_bits, slice_bits, 0, 0, key_out, s_eid, s_indices, flag, ws + off_sort, s);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(boundaries_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, s, key_out, E, (int32_t)n_rows, row_bits, n_keys,
                     segptr);
  return hipGetLastError();
}

hipError_t gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                      hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, in, perm, n, out);
  return hipGetLastError();
}

}  // namespace dgmi
