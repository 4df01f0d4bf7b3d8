// dgmi_kernels.h — internal launch interface between the C ABI (dgmi_api.hip)
// and the kernel translation units.  Not installed; the public contract is
// include/dgmi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dgmi {

// ---- SpMM plan (dgmi_plan.hip): int32 words on the device --------------------------------
//   [0, kPlanHeaderWords)                      header
//   then items_cap x int4 {row, start, end, slot}   one per wave of the main launch;
//                                                  slot < 0 -> the wave owns the whole row
//   then long_cap  x int4 {row, slot0, nchunks, 0}  rows cut into > 1 chunk (reduce pass)
constexpr int kPlanHeaderWords = 16;
constexpr int kPlanNumItems = 0;
constexpr int kPlanNumLong = 1;
constexpr int kPlanNumSlots = 2;
constexpr int kPlanChunk = 3;

// Upper bounds the host can compute without reading the degree distribution back:
// sum_r max(1, ceil(deg_r / chunk)) <= n_rows + nnz / chunk;  rows with deg > chunk number
// at most nnz / (chunk + 1);  their chunks at most nnz / chunk + #long rows.
inline int64_t plan_items_cap(int64_t n_rows, int64_t nnz, int64_t chunk) {
  return n_rows + nnz / chunk;
}
inline int64_t plan_long_cap(int64_t nnz, int64_t chunk) { return nnz / (chunk + 1); }
inline int64_t plan_slots_cap(int64_t nnz, int64_t chunk) {
  return nnz / chunk + plan_long_cap(nnz, chunk);
}
inline size_t plan_bytes(int64_t n_rows, int64_t nnz, int64_t chunk) {
  return sizeof(int32_t) *
         (size_t)(kPlanHeaderWords + 4 * (plan_items_cap(n_rows, nnz, chunk) + plan_long_cap(nnz, chunk)));
}

// Lanes-per-row for a 16-B-aligned width: the widest tile whose last tile is
// still >= 85 % used (F=344 -> 32 lanes x 3 tiles, not 64 x 2 at 67 %).
inline int pick_lpr(int64_t F) {
  const int64_t f4 = (F + 3) / 4;
  int best = 8;
  double best_util = 0.0;
  for (int lpr : {64, 32, 16, 8}) {
    const int64_t tiles = (f4 + lpr - 1) / lpr;
    const double util = (double)f4 / (double)(tiles * lpr);
    if (util >= 0.85) return lpr;
    if (util > best_util + 1e-9) {
      best_util = util;
      best = lpr;
    }
  }
  return best;
}

// Output epilogue fused into the kernel that writes Y (f3: GCMCLayer's `dropout(agg_act(...))`,
// reference layers.py:134-138):  Y = mask * mask_scale * act(dst_scale * sum)
struct Epilogue {
  int act;            // 0: none; 1: leaky-relu with `slope` (slope 0 = relu), as torch: v > 0 ? v : v * slope
  float slope;
  const float* mask;  // nullable: (n_dst, F) keep mask with leading dimension ldm (a multiple of 4 on the 16-B paths)
  int64_t ldm;
  float mask_scale;   // 1 / (1 - p)
};

struct SpmmArgs {
  const int32_t* indptr;
  const int32_t* indices;
  const float* vals;       // nullable
  const float* X;
  int64_t ldx;
  const float* src_scale;  // nullable
  const float* dst_scale;  // nullable
  float* Y;
  int64_t ldy;
  int64_t n_dst;
  int64_t n_src;
  int64_t F;
  // planned launch only (plan == nullptr: one wave per row)
  const int32_t* plan;
  int64_t nnz;
  int64_t chunk;
  float* partials;
  int64_t ldp;
  // edge dropout applied on the fly (dgmi_keep.h): n_keep > 0 -> edge p survives iff
  // keep(eid[p]) under the n_keep descriptions in `keep` (device, 8 words each)
  const int32_t* eid;
  const void* keep;
  int n_keep;
  Epilogue ep;
};

// Y = diag(dst_scale) A diag(src_scale) X   (dgmi_spmm.hip)
hipError_t spmm_csr_f32(const SpmmArgs& a, hipStream_t s);

// Builds the plan for `indptr` (dgmi_plan.hip).  workspace == nullptr: size query only.
hipError_t spmm_plan_build(const int32_t* indptr, int64_t n_rows, int64_t nnz, int64_t chunk,
                           int32_t* plan, void* workspace, size_t* workspace_bytes, hipStream_t s);

// Stable COO->CSR (dgmi_csr.hip).  workspace == nullptr: size query only.
hipError_t csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                            int64_t n_rows, int64_t n_cols, int32_t* indptr, int32_t* indices,
                            int32_t* eid, void* workspace, size_t* workspace_bytes,
                            hipStream_t s);

// Stable LSD radix sort of the records (key[i], eid[i] or i, col[i]) by bits [shift0, shift0 + bits) of the key
// (dgmi_sort.hip): digits of up to 9 bits; n_rows / n_cols > 0: the id range check rides on the first pass.
// Workspace: radix_sort_workspace_bytes(E, bits).
struct SortPassPlan {
  int shift[8], bits[8];
};
int radix_sort_passes(int shift0, int bits, SortPassPlan* plan);
size_t radix_sort_workspace_bytes(int64_t E, int bits);
hipError_t radix_sort_records(const int32_t* key, const int32_t* eid_in, const int32_t* col, int64_t E, int shift0, int bits,
                              int32_t n_rows, int32_t n_cols, int32_t* keys_out, int32_t* eid_out, int32_t* col_out, int32_t* flag,
                              void* workspace, hipStream_t s);

// Source-sliced CSR (dgmi_csr.hip) and the XCD-local SpMM over it (dgmi_sliced.hip).
hipError_t csr_sliced_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E, int64_t n_rows,
                                   int64_t n_cols, int64_t n_slices, int64_t slice_width,
                                   int32_t* segptr, int32_t* indices, int32_t* eid, void* workspace,
                                   size_t* workspace_bytes, hipStream_t s);

// the same layout from an existing CSR: one partition pass instead of a full sort (dgmi_csr.hip)
hipError_t csr_sliced_from_csr_i32(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t E,
                                   int64_t n_rows, int64_t n_cols, int64_t n_slices, int64_t slice_width, int32_t* segptr,
                                   int32_t* s_indices, int32_t* s_eid, void* workspace, size_t* workspace_bytes,
                                   hipStream_t s);

// Multiplicities carried in the id words of an XCD-sliced layout (SlicedArgs::id_mult): bits 28..30 hold m - 1, bit 31 stays
// the dropped flag of edge dropout on the fly, ids are < 2^28.
constexpr int kMultShift = 28;
constexpr int kMultMax = 7;                  // m - 1 <= 7
constexpr uint32_t kMultIdMask = 0x0fffffffu;

struct SlicedArgs {
  const int32_t* segptr;   // n_slices * n_dst + 1
  const int32_t* indices;
  const float* vals;       // nullable, sliced order
  const float* X;
  int64_t ldx;
  const float* src_scale;  // nullable
  const float* dst_scale;  // nullable
  float* Y;
  int64_t ldy;
  int64_t n_dst, n_src, F;
  int64_t n_slices;
  float* planes;           // n_slices * min(n_dst, chunk_rows) * ldp floats of scratch
  int64_t ldp;
  int64_t chunk_rows;      // destination rows per launch pair (planes stay Infinity-Cache resident)
  const int32_t* eid;      // edge dropout on the fly, as SpmmArgs
  const void* keep;
  int n_keep;
  Epilogue ep;
  bool full_width;         // never split the columns into half-width passes (see spmm_sliced_f32)
  bool id_mult;            // vals == nullptr and the ids carry integer multiplicities (kMultShift)
};

// per row of a CSR with positive values: scale s and per-edge m in 1..8 with vals[p] = m * s to within rel_tol, or
// *fail = 1 when some row has no such form (dgmi_edge.hip); mult[p] = m - 1
hipError_t row_multiplicity_f32(const int32_t* indptr, const float* vals, int64_t n_rows, float rel_tol, float* row_scale,
                                int32_t* mult, int32_t* fail, hipStream_t s);
hipError_t spmm_sliced_f32(const SlicedArgs& a, hipStream_t s);

// out[e] = cat(A[src[e]], B[dst[e]])   (dgmi_edge.hip)
hipError_t gather_concat_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                             int64_t lda, int64_t Fa, const float* B, int64_t ldb, int64_t Fb,
                             float* out, int64_t ldo, hipStream_t s);

// out[e] = A[src[e]] + B[dst[e]] (+ bias)   (dgmi_edge.hip)
hipError_t gather_add_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A, int64_t lda,
                          const float* B, int64_t ldb, const float* bias, int64_t F, float* out,
                          int64_t ldo, int act, hipStream_t s);  // act 1: relu applied to the sum

// g = dY * act'(Y) * mask * mask_scale over n contiguous elements: backward of the fused epilogue (dgmi_edge.hip)
hipError_t epilogue_backward_f32(const float* dY, const float* Y, const float* mask, int64_t n, int act, float slope,
                                 float mask_scale, float* out, hipStream_t s);

// (f3 complement form) weighted column sums of a strided row set, and their gradient added back (dgmi_edge.hip)
hipError_t weighted_colsum_f32(const float* A, int64_t lda, const float* coef, int64_t ldc, int64_t n, int64_t W, int B,
                               float* out, int64_t ldo, hipStream_t s);
hipError_t rank_add_f32(float* G, int64_t ldg, const float* coef, int64_t ldc, const float* gs, int64_t lds, int64_t n,
                        int64_t W, int B, hipStream_t s);

// out = diag(scale) X, one streaming pass (dgmi_edge.hip)
hipError_t scale_rows_f32(const float* X, int64_t ldx, const float* scale, int64_t n, int64_t F, float* out, int64_t ldo,
                          hipStream_t s);

// mask[e] = 1 for a uniformly random subset of exactly `keep` of the E edges (dgmi_select.hip)
size_t random_subset_workspace_bytes();
// the 8-word description (dgmi_keep.h) of a uniformly random subset of exactly `keep` of E edges
hipError_t random_subset_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset, void* seg_out,
                                void* workspace, hipStream_t s);
// the same for up to 8 edge lists in one series of launches (host arrays of length n; descs: n x 8 words)
// seed_dev != NULL: n seeds in DEVICE memory, read by the kernels (seed may be NULL then)
hipError_t random_subset_select_batch(int n, const int64_t* E, const int64_t* keep, const uint64_t* seed,
                                      const uint32_t* e_offset, void* descs, void* workspace, hipStream_t s,
                                      const uint64_t* seed_dev = nullptr);
// mask[e] = keep(e) under `n_seg` descriptions, e in [0, E)
hipError_t keep_mask_f32(const void* table, int n_seg, int64_t E, float* mask, hipStream_t s);
hipError_t random_subset_mask_f32(int64_t E, int64_t keep, uint64_t seed, float* mask, void* workspace,
                                  hipStream_t s);

hipError_t gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                      hipStream_t s);

// a CSR-shaped layout (ptr[n_ptr], indices[, vals], eid over nnz positions) with the edges dropped under `keep`
// removed, stable (dgmi_compact.hip): ptr_out[n_ptr], indices_out / vals_out sized nnz (the survivors fill a prefix)
size_t compact_workspace_bytes(int64_t nnz);
hipError_t compact_layout_i32(const int32_t* ptr, int64_t n_ptr, const int32_t* indices, const float* vals, const int32_t* eid,
                              int64_t nnz, const void* keep, int n_keep, int32_t* ptr_out, int32_t* indices_out, float* vals_out,
                              void* workspace, hipStream_t s);

// (f4) neighbours of the k largest cosine similarities per row of a normalised (N, D) matrix (dgmi_knn.hip)
bool knn_supported(int64_t N, int64_t D, int64_t k);
size_t knn_workspace_bytes(int64_t N, int64_t D, int k);  // small N: partial lists of the candidate splits; large N: the screen's buffers
// large N (dgmi_knn_screen.hip): bf16-MFMA screen + exact fp32 rescoring; flagged tiles recomputed by the fp32 kernel
constexpr int kKnnTileMaxK = 16;    // fp32 tile kernel: per-lane lists in LDS
constexpr int kKnnScreenMaxK = 64;  // bf16 screen: a wave's lanes hold the top-k
int64_t knn_screen_min_rows();
bool knn_screen_supported(int64_t N, int64_t D, int64_t k);
size_t knn_screen_workspace_bytes(int64_t N, int64_t D, int k);
hipError_t knn_cosine_topk_screened(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr, void* workspace,
                                    hipStream_t s);
hipError_t knn_cosine_topk_exact_tiles(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr,
                                       const int32_t* flags, hipStream_t s);
hipError_t knn_cosine_topk_f32(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr, void* workspace,
                               hipStream_t s);

// Measurement probe (dgmi_probe.hip): hash-indexed whole-row gathers in the product kernels' shape.
hipError_t probe_row_gather(const float* table, int64_t n_rows, int64_t F, int64_t groups, int64_t per_group,
                            int64_t window, int per_xcd, float* out, hipStream_t s);

}  // namespace dgmi
