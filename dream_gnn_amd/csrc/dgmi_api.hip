// dgmi_api.hip — the extern "C" boundary declared in include/dgmi.h.
// Host-side validation only; the kernels live in dgmi_spmm.hip / dgmi_csr.hip.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "dgmi.h"
#include "dgmi_keep.h"
#include "dgmi_kernels.h"
#include "dgmi_tuning.h"

namespace {

inline hipStream_t as_stream(dgmi_stream_t s) { return static_cast<hipStream_t>(s); }

inline int from_hip(hipError_t e) { return e == hipSuccess ? DGMI_OK : DGMI_ERR_LAUNCH; }

long long env_ll(const char* name, long long dflt) {
  const char* e = getenv(name);
  return e != nullptr && e[0] != '\0' ? atoll(e) : dflt;
}

}  // namespace

namespace dgmi {

// The environment is read here, once (thread-safe static initialisation), never on a launch path.
Tuning& tuning() {
  static Tuning t = [] {
    Tuning v;
    v.sliced_rows = (int)env_ll("DGMI_SLICED_ROWS", 0);
    v.sliced_touch_lead = (int)env_ll("DGMI_SLICED_PF", -1);
    v.sliced_lpr = (int)env_ll("DGMI_SLICED_LPR", 0);
    v.sliced_no_off32 = getenv("DGMI_NO_OFF32") != nullptr ? 1 : 0;
    v.sliced_chunk_rows = env_ll("DGMI_SLICED_CHUNK_ROWS", 0);
    v.select_window_min = env_ll("DGMI_SELECT_WINDOW_MIN", 0);
    v.select_narrow_window = env_ll("DGMI_SELECT_NARROW_WINDOW", 0) != 0 ? 1 : 0;
    v.sort_plain_tiles = env_ll("DGMI_SORT_PLAIN_TILES", 0) != 0 ? 1 : 0;
    v.knn_screen_first = env_ll("DGMI_KNN_SCREEN_V1", 0) != 0 ? 1 : 0;
    v.knn_pool_chunks = env_ll("DGMI_KNN_POOL_CHUNKS", 0);
    return v;
  }();
  return t;
}

}  // namespace dgmi

extern "C" {

DGMI_API int dgmi_set_tuning(const char* name, int64_t value) {
  if (name == nullptr) return DGMI_ERR_INVALID_ARG;
  dgmi::Tuning& t = dgmi::tuning();
  if (strcmp(name, "sliced_rows") == 0) t.sliced_rows = (int)value;
  else if (strcmp(name, "sliced_touch_lead") == 0) t.sliced_touch_lead = (int)value;
  else if (strcmp(name, "sliced_lpr") == 0) t.sliced_lpr = (int)value;
  else if (strcmp(name, "sliced_no_off32") == 0) t.sliced_no_off32 = value != 0;
  else if (strcmp(name, "sliced_chunk_rows") == 0) t.sliced_chunk_rows = value;
  else if (strcmp(name, "select_window_min") == 0) t.select_window_min = value;
  else if (strcmp(name, "select_narrow_window") == 0) t.select_narrow_window = value != 0;
  else if (strcmp(name, "sort_plain_tiles") == 0) t.sort_plain_tiles = value != 0;
  else if (strcmp(name, "knn_screen_first") == 0) t.knn_screen_first = value != 0;
  else if (strcmp(name, "knn_pool_chunks") == 0) t.knn_pool_chunks = value;
  else return DGMI_ERR_INVALID_ARG;
  return DGMI_OK;
}

DGMI_API int dgmi_abi_version(void) { return DGMI_ABI_VERSION; }

DGMI_API const char* dgmi_status_string(int status) {
  switch (status) {
    case DGMI_OK: return "ok";
    case DGMI_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or ld < F)";
    case DGMI_ERR_TOO_LARGE: return "size does not fit the int32 id space";
    case DGMI_ERR_WORKSPACE: return "workspace missing or too small";
    case DGMI_ERR_LAUNCH: return "HIP launch failed";
    case DGMI_ERR_NO_DEVICE: return "no gfx950 device visible";
    default: return "unknown dgmi status";
  }
}

DGMI_API int dgmi_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

DGMI_API int dgmi_csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E, int64_t n_rows,
                          int64_t n_cols, int32_t* indptr, int32_t* indices, int32_t* eid, void* workspace,
                          size_t* workspace_bytes, dgmi_stream_t stream) {
  if (E < 0 || n_rows < 0 || workspace_bytes == nullptr) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || n_rows >= INT32_MAX || n_cols >= INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (workspace != nullptr) {
    if (indptr == nullptr) return DGMI_ERR_INVALID_ARG;
    if (E > 0 && (row == nullptr || col == nullptr || indices == nullptr || eid == nullptr))
      return DGMI_ERR_INVALID_ARG;
  }
  size_t need = 0;
  hipError_t err = dgmi::csr_from_coo_i32(row, col, E, n_rows, n_cols, indptr, indices, eid, nullptr,
                                          &need, as_stream(stream));
  if (err != hipSuccess) return DGMI_ERR_LAUNCH;
  if (workspace == nullptr) {
    *workspace_bytes = need;
    return DGMI_OK;
  }
  if (*workspace_bytes < need) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::csr_from_coo_i32(row, col, E, n_rows, n_cols, indptr, indices, eid, workspace,
                                         workspace_bytes, as_stream(stream)));
}

static bool epilogue_ok(int32_t act, const float* out_mask, int64_t ld_mask, int64_t F) {
  return (act == 0 || act == 1) && (out_mask == nullptr || ld_mask >= F);
}

static bool keep_args_ok(const int32_t* eid, const uint32_t* keep, int32_t n_keep) {
  if (n_keep < 0 || n_keep > dgmi::kMaxKeepSegs) return false;
  return n_keep == 0 || (eid != nullptr && keep != nullptr);
}

DGMI_API int dgmi_spmm_csr_f32(const int32_t* indptr, const int32_t* indices, const float* vals,
                      const int32_t* eid, const uint32_t* keep, int32_t n_keep,
                      const float* X, int64_t ldx, const float* src_scale,
                      const float* dst_scale, float* Y, int64_t ldy, int64_t n_dst,
                      int64_t n_src, int64_t F, int32_t act, float act_slope, const float* out_mask,
                      int64_t ld_mask, float out_mask_scale, dgmi_stream_t stream) {
  if (n_dst < 0 || n_src < 0 || F < 0 || !keep_args_ok(eid, keep, n_keep) || !epilogue_ok(act, out_mask, ld_mask, F))
    return DGMI_ERR_INVALID_ARG;
  if (n_dst >= INT32_MAX || n_src >= INT32_MAX || F > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n_dst == 0 || F == 0) return DGMI_OK;
  if (indptr == nullptr || Y == nullptr) return DGMI_ERR_INVALID_ARG;
  if (ldx < F || ldy < F) return DGMI_ERR_INVALID_ARG;
  // `indices` (and `vals`) are dereferenced only for rows with edges, so an edgeless
  // graph may pass NULL for them; X may be NULL only when there is no source node.
  if (X == nullptr && n_src > 0) return DGMI_ERR_INVALID_ARG;
  if (static_cast<const void*>(X) == static_cast<const void*>(Y)) return DGMI_ERR_INVALID_ARG;
  dgmi::SpmmArgs a{indptr, indices, vals, X, ldx, src_scale, dst_scale, Y, ldy, n_dst, n_src, F,
                   nullptr, 0, 0, nullptr, 0, eid, keep, n_keep, {act, act_slope, out_mask, ld_mask, out_mask_scale}};
  return from_hip(dgmi::spmm_csr_f32(a, as_stream(stream)));
}

DGMI_API int32_t dgmi_spmm_default_chunk(int64_t n_rows, int64_t nnz) {
  // Aim for >= 4 waves on each of the 1024 SIMDs, never cut below one id batch (64) and
  // never above 512 edges (a 512-edge chunk at F=128 is 256 KiB of gathers: long enough
  // to amortise the wave's prologue, short enough to balance).
  (void)n_rows;
  int64_t c = (nnz / 4096 + 63) / 64 * 64;
  if (c < 64) c = 64;
  if (c > 512) c = 512;
  return (int32_t)c;
}

static bool chunk_ok(int32_t chunk) { return chunk >= 16 && chunk <= 65536; }

DGMI_API size_t dgmi_spmm_plan_bytes(int64_t n_rows, int64_t nnz, int32_t chunk) {
  if (n_rows < 0 || nnz < 0 || !chunk_ok(chunk)) return 0;
  return dgmi::plan_bytes(n_rows, nnz, chunk);
}

DGMI_API size_t dgmi_spmm_partials_bytes(int64_t nnz, int32_t chunk, int64_t F) {
  if (nnz < 0 || F < 0 || !chunk_ok(chunk)) return 0;
  const size_t ldp = (size_t)((F + 3) / 4 * 4);
  const size_t b = (size_t)dgmi::plan_slots_cap(nnz, chunk) * ldp * sizeof(float);
  return b < 16 ? 16 : b;
}

DGMI_API int dgmi_spmm_plan_build(const int32_t* indptr, int64_t n_rows, int64_t nnz, int32_t chunk,
                                  void* plan, size_t plan_bytes, void* workspace,
                                  size_t* workspace_bytes, dgmi_stream_t stream) {
  if (n_rows < 0 || nnz < 0 || !chunk_ok(chunk) || workspace_bytes == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (n_rows >= INT32_MAX || nnz > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  size_t need = 0;
  if (dgmi::spmm_plan_build(indptr, n_rows, nnz, chunk, nullptr, nullptr, &need, as_stream(stream)) !=
      hipSuccess)
    return DGMI_ERR_LAUNCH;
  if (workspace == nullptr) {
    *workspace_bytes = need;
    return DGMI_OK;
  }
  if (plan == nullptr || (indptr == nullptr && n_rows > 0)) return DGMI_ERR_INVALID_ARG;
  if (*workspace_bytes < need || plan_bytes < dgmi::plan_bytes(n_rows, nnz, chunk))
    return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::spmm_plan_build(indptr, n_rows, nnz, chunk, static_cast<int32_t*>(plan),
                                        workspace, workspace_bytes, as_stream(stream)));
}

DGMI_API int dgmi_spmm_csr_planned_f32(const int32_t* indptr, const int32_t* indices,
                                       const float* vals, const int32_t* eid, const uint32_t* keep,
                                       int32_t n_keep, const float* X, int64_t ldx,
                                       const float* src_scale, const float* dst_scale, float* Y,
                                       int64_t ldy, int64_t n_dst, int64_t n_src, int64_t F,
                                       int64_t nnz, int32_t chunk, const void* plan, void* partials,
                                       size_t partials_bytes, int32_t act, float act_slope,
                                       const float* out_mask, int64_t ld_mask, float out_mask_scale,
                                       dgmi_stream_t stream) {
  if (n_dst < 0 || n_src < 0 || F < 0 || nnz < 0 || !chunk_ok(chunk) || !keep_args_ok(eid, keep, n_keep) ||
      !epilogue_ok(act, out_mask, ld_mask, F))
    return DGMI_ERR_INVALID_ARG;
  if (n_dst >= INT32_MAX || n_src >= INT32_MAX || F > INT32_MAX || nnz > INT32_MAX)
    return DGMI_ERR_TOO_LARGE;
  if (n_dst == 0 || F == 0) return DGMI_OK;
  if (indptr == nullptr || Y == nullptr || plan == nullptr || partials == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (ldx < F || ldy < F) return DGMI_ERR_INVALID_ARG;
  if (X == nullptr && n_src > 0) return DGMI_ERR_INVALID_ARG;
  if (static_cast<const void*>(X) == static_cast<const void*>(Y)) return DGMI_ERR_INVALID_ARG;
  if (partials_bytes < dgmi_spmm_partials_bytes(nnz, chunk, F)) return DGMI_ERR_WORKSPACE;
  dgmi::SpmmArgs a{indptr, indices, vals, X, ldx, src_scale, dst_scale, Y, ldy, n_dst, n_src, F,
                   static_cast<const int32_t*>(plan), nnz, chunk, static_cast<float*>(partials),
                   (F + 3) / 4 * 4, eid, keep, n_keep, {act, act_slope, out_mask, ld_mask, out_mask_scale}};
  return from_hip(dgmi::spmm_csr_f32(a, as_stream(stream)));
}

static int64_t slice_width_for(int64_t n_cols, int32_t n_slices) {
  const int64_t w = (n_cols + n_slices - 1) / n_slices;
  return w > 0 ? w : 1;
}

DGMI_API int dgmi_csr_sliced_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E, int64_t n_rows,
                                          int64_t n_cols, int32_t n_slices, int32_t* segptr,
                                          int32_t* indices, int32_t* eid, void* workspace,
                                          size_t* workspace_bytes, dgmi_stream_t stream) {
  if (E < 0 || n_rows < 0 || n_cols < 0 || n_slices < 1 || n_slices > 64 || workspace_bytes == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || n_cols >= INT32_MAX || n_rows * (int64_t)n_slices >= INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (workspace != nullptr) {
    if (segptr == nullptr) return DGMI_ERR_INVALID_ARG;
    if (E > 0 && (row == nullptr || col == nullptr || indices == nullptr || eid == nullptr))
      return DGMI_ERR_INVALID_ARG;
  }
  const int64_t width = slice_width_for(n_cols, n_slices);
  size_t need = 0;
  if (dgmi::csr_sliced_from_coo_i32(row, col, E, n_rows, n_cols, n_slices, width, segptr, indices, eid, nullptr,
                                    &need, as_stream(stream)) != hipSuccess)
    return DGMI_ERR_LAUNCH;
  if (workspace == nullptr) {
    *workspace_bytes = need;
    return DGMI_OK;
  }
  if (*workspace_bytes < need) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::csr_sliced_from_coo_i32(row, col, E, n_rows, n_cols, n_slices, width, segptr, indices,
                                                eid, workspace, workspace_bytes, as_stream(stream)));
}

DGMI_API int dgmi_csr_sliced_from_csr_i32(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t E,
                                          int64_t n_rows, int64_t n_cols, int32_t n_slices, int32_t* segptr,
                                          int32_t* s_indices, int32_t* s_eid, void* workspace, size_t* workspace_bytes,
                                          dgmi_stream_t stream) {
  if (E < 0 || n_rows < 0 || n_cols < 0 || n_slices < 1 || n_slices > 64 || workspace_bytes == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || n_cols >= INT32_MAX || n_rows * (int64_t)n_slices >= INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (workspace != nullptr) {
    if (segptr == nullptr) return DGMI_ERR_INVALID_ARG;
    if (E > 0 && (indptr == nullptr || indices == nullptr || eid == nullptr || s_indices == nullptr || s_eid == nullptr))
      return DGMI_ERR_INVALID_ARG;
  }
  const int64_t width = slice_width_for(n_cols, n_slices);
  size_t need = 0;
  if (dgmi::csr_sliced_from_csr_i32(indptr, indices, eid, E, n_rows, n_cols, n_slices, width, segptr, s_indices, s_eid,
                                    nullptr, &need, as_stream(stream)) != hipSuccess)
    return DGMI_ERR_LAUNCH;
  if (workspace == nullptr) {
    *workspace_bytes = need;
    return DGMI_OK;
  }
  if (*workspace_bytes < need) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::csr_sliced_from_csr_i32(indptr, indices, eid, E, n_rows, n_cols, n_slices, width, segptr, s_indices,
                                                s_eid, workspace, workspace_bytes, as_stream(stream)));
}

// Destination rows per launch pair: the n_slices partial planes of one chunk stay <= 32 MiB.
static int64_t sliced_chunk_rows(int64_t n_dst, int32_t n_slices, int64_t F) {
  const int64_t row_bytes = (int64_t)n_slices * ((F + 3) / 4 * 4) * (int64_t)sizeof(float);
  int64_t rows = ((int64_t)4096 << 20) / (row_bytes > 0 ? row_bytes : 1);  // chunking off by default (measured slower)
  rows = rows / 1024 * 1024;
  if (rows < 1024) rows = 1024;
  if (dgmi::tuning().sliced_chunk_rows > 0) rows = dgmi::tuning().sliced_chunk_rows;  // tuning aid (dgmi_set_tuning)
  return rows < n_dst ? rows : n_dst;
}

DGMI_API size_t dgmi_spmm_sliced_planes_bytes(int64_t n_dst, int32_t n_slices, int64_t F) {
  if (n_dst < 0 || n_slices < 1 || F < 0) return 0;
  const size_t b = (size_t)sliced_chunk_rows(n_dst, n_slices, F) * (size_t)n_slices *
                   (size_t)((F + 3) / 4 * 4) * sizeof(float);
  return b < 16 ? 16 : b;
}

DGMI_API int dgmi_spmm_sliced_f32(const int32_t* segptr, const int32_t* indices, const float* vals,
                                  const int32_t* eid, const uint32_t* keep, int32_t n_keep,
                                  const float* X, int64_t ldx, const float* src_scale,
                                  const float* dst_scale, float* Y, int64_t ldy, int64_t n_dst,
                                  int64_t n_src, int64_t F, int32_t n_slices, int32_t column_passes,
                                  int32_t id_multiplicity, void* planes,
                                  size_t planes_bytes, int32_t act, float act_slope, const float* out_mask,
                                  int64_t ld_mask, float out_mask_scale, dgmi_stream_t stream) {
  if (n_dst < 0 || n_src < 0 || F < 0 || n_slices < 1 || n_slices > 64 || !keep_args_ok(eid, keep, n_keep) ||
      !epilogue_ok(act, out_mask, ld_mask, F) || column_passes < 0 || column_passes > 1 || id_multiplicity < 0 ||
      id_multiplicity > 1)
    return DGMI_ERR_INVALID_ARG;
  if (id_multiplicity == 1 && (vals != nullptr || n_src > (int64_t)dgmi::kMultIdMask + 1)) return DGMI_ERR_INVALID_ARG;
  if (out_mask != nullptr && (ld_mask % 4 != 0 || (reinterpret_cast<uintptr_t>(out_mask) & 15))) return DGMI_ERR_INVALID_ARG;
  if (n_dst >= INT32_MAX || n_src >= INT32_MAX || F > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n_dst == 0 || F == 0) return DGMI_OK;
  if (segptr == nullptr || Y == nullptr || planes == nullptr) return DGMI_ERR_INVALID_ARG;
  if (X == nullptr && n_src > 0) return DGMI_ERR_INVALID_ARG;
  if (ldx < F || ldy < F || F % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0) return DGMI_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
      (reinterpret_cast<uintptr_t>(planes) & 15))
    return DGMI_ERR_INVALID_ARG;
  if (static_cast<const void*>(X) == static_cast<const void*>(Y)) return DGMI_ERR_INVALID_ARG;
  if (planes_bytes < dgmi_spmm_sliced_planes_bytes(n_dst, n_slices, F)) return DGMI_ERR_WORKSPACE;
  dgmi::SlicedArgs a{segptr, indices, vals, X, ldx, src_scale, dst_scale, Y, ldy, n_dst, n_src, F, n_slices,
                     static_cast<float*>(planes), F, sliced_chunk_rows(n_dst, n_slices, F), eid, keep, n_keep,
                     {act, act_slope, out_mask, ld_mask, out_mask_scale}, column_passes == 1, id_multiplicity == 1};
  return from_hip(dgmi::spmm_sliced_f32(a, as_stream(stream)));
}

DGMI_API int dgmi_gather_concat_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                                    int64_t lda, int64_t Fa, const float* B, int64_t ldb, int64_t Fb,
                                    float* out, int64_t ldo, dgmi_stream_t stream) {
  if (E < 0 || Fa < 0 || Fb < 0) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || Fa + Fb > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (E == 0 || Fa + Fb == 0) return DGMI_OK;
  if (src == nullptr || dst == nullptr || out == nullptr) return DGMI_ERR_INVALID_ARG;
  if ((Fa > 0 && A == nullptr) || (Fb > 0 && B == nullptr)) return DGMI_ERR_INVALID_ARG;
  if (lda < Fa || ldb < Fb || ldo < Fa + Fb) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::gather_concat_f32(src, dst, E, A, lda, Fa, B, ldb, Fb, out, ldo, as_stream(stream)));
}

DGMI_API int dgmi_gather_add_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                                 int64_t lda, const float* B, int64_t ldb, const float* bias, int64_t F,
                                 float* out, int64_t ldo, int32_t act, dgmi_stream_t stream) {
  if (E < 0 || F < 0 || (act != 0 && act != 1)) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || F > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (E == 0 || F == 0) return DGMI_OK;
  if (src == nullptr || dst == nullptr || A == nullptr || B == nullptr || out == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (lda < F || ldb < F || ldo < F) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::gather_add_f32(src, dst, E, A, lda, B, ldb, bias, F, out, ldo, act, as_stream(stream)));
}

DGMI_API int dgmi_epilogue_backward_f32(const float* dY, const float* Y, const float* mask, int64_t n, int32_t act,
                                        float act_slope, float mask_scale, float* out, dgmi_stream_t stream) {
  if (n < 0 || act < 0 || act > 2 || (act == 2 && mask != nullptr)) return DGMI_ERR_INVALID_ARG;
  if (n == 0) return DGMI_OK;
  if (dY == nullptr || out == nullptr || (act != 0 && Y == nullptr)) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::epilogue_backward_f32(dY, Y, mask, n, act, act_slope, mask_scale, out, as_stream(stream)));
}

DGMI_API int dgmi_scale_rows_f32(const float* X, int64_t ldx, const float* scale, int64_t n, int64_t F, float* out, int64_t ldo,
                                 dgmi_stream_t stream) {
  if (n < 0 || F < 0 || ldx < F || ldo < F) return DGMI_ERR_INVALID_ARG;
  if (F > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n == 0 || F == 0) return DGMI_OK;
  if (X == nullptr || scale == nullptr || out == nullptr) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::scale_rows_f32(X, ldx, scale, n, F, out, ldo, as_stream(stream)));
}

DGMI_API int dgmi_weighted_colsum_f32(const float* A, int64_t lda, const float* coef, int64_t ldc, int64_t n, int64_t W,
                                      int32_t B, float* out, int64_t ldo, dgmi_stream_t stream) {
  if (n < 0 || W < 0 || B < 0 || B > 64 || lda < W || ldo < W || ldc < n) return DGMI_ERR_INVALID_ARG;
  if (W > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (W == 0 || B == 0) return DGMI_OK;
  if (out == nullptr || coef == nullptr || (A == nullptr && n > 0)) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::weighted_colsum_f32(A, lda, coef, ldc, n, W, B, out, ldo, as_stream(stream)));
}

DGMI_API int dgmi_rank_add_f32(float* G, int64_t ldg, const float* coef, int64_t ldc, const float* gs, int64_t lds, int64_t n,
                               int64_t W, int32_t B, dgmi_stream_t stream) {
  if (n < 0 || W < 0 || B < 0 || B > 64 || ldg < W || lds < W || ldc < n) return DGMI_ERR_INVALID_ARG;
  if (W > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n == 0 || W == 0 || B == 0) return DGMI_OK;
  if (G == nullptr || coef == nullptr || gs == nullptr) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::rank_add_f32(G, ldg, coef, ldc, gs, lds, n, W, B, as_stream(stream)));
}

DGMI_API size_t dgmi_random_subset_workspace_bytes(void) { return dgmi::random_subset_workspace_bytes(); }

DGMI_API int dgmi_random_subset_mask_f32(int64_t E, int64_t keep, uint64_t seed, float* mask, void* workspace,
                                         size_t workspace_bytes, dgmi_stream_t stream) {
  if (E < 0 || keep < 0 || keep > E) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (E == 0) return DGMI_OK;
  if (mask == nullptr || workspace == nullptr) return DGMI_ERR_INVALID_ARG;
  if (workspace_bytes < dgmi::random_subset_workspace_bytes()) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::random_subset_mask_f32(E, keep, seed, mask, workspace, as_stream(stream)));
}

DGMI_API int dgmi_random_subset_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset, uint32_t* desc,
                                      void* workspace, size_t workspace_bytes, dgmi_stream_t stream) {
  if (E < 0 || keep < 0 || keep > E || desc == nullptr) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX || (uint64_t)e_offset + (uint64_t)E > (uint64_t)INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (workspace == nullptr) return DGMI_ERR_INVALID_ARG;
  if (workspace_bytes < dgmi::random_subset_workspace_bytes()) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::random_subset_select(E, keep, seed, e_offset, desc, workspace, as_stream(stream)));
}

DGMI_API int dgmi_random_subset_select_batch(int32_t n, const int64_t* E, const int64_t* keep, const uint64_t* seed,
                                            const uint32_t* e_offset, uint32_t* descs, void* workspace,
                                            size_t workspace_bytes, dgmi_stream_t stream) {
  if (n < 0 || n > dgmi::kMaxKeepSegs) return DGMI_ERR_INVALID_ARG;
  if (n == 0) return DGMI_OK;
  if (E == nullptr || keep == nullptr || seed == nullptr || descs == nullptr || workspace == nullptr)
    return DGMI_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i) {
    if (E[i] < 0 || keep[i] < 0 || keep[i] > E[i]) return DGMI_ERR_INVALID_ARG;
    if (E[i] > INT32_MAX || (uint64_t)(e_offset ? e_offset[i] : 0u) + (uint64_t)E[i] > (uint64_t)INT32_MAX)
      return DGMI_ERR_TOO_LARGE;
  }
  if (workspace_bytes < dgmi::random_subset_workspace_bytes()) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::random_subset_select_batch(n, E, keep, seed, e_offset, descs, workspace, as_stream(stream)));
}

DGMI_API int dgmi_random_subset_select_batch_dseed(int32_t n, const int64_t* E, const int64_t* keep, const uint64_t* seed_dev,
                                                  const uint32_t* e_offset, uint32_t* descs, void* workspace,
                                                  size_t workspace_bytes, dgmi_stream_t stream) {
  if (n < 0 || n > dgmi::kMaxKeepSegs) return DGMI_ERR_INVALID_ARG;
  if (n == 0) return DGMI_OK;
  if (E == nullptr || keep == nullptr || seed_dev == nullptr || descs == nullptr || workspace == nullptr)
    return DGMI_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(seed_dev) & 7) return DGMI_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i) {
    if (E[i] < 0 || keep[i] < 0 || keep[i] > E[i]) return DGMI_ERR_INVALID_ARG;
    if (E[i] > INT32_MAX || (uint64_t)(e_offset ? e_offset[i] : 0u) + (uint64_t)E[i] > (uint64_t)INT32_MAX)
      return DGMI_ERR_TOO_LARGE;
  }
  if (workspace_bytes < dgmi::random_subset_workspace_bytes()) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::random_subset_select_batch(n, E, keep, nullptr, e_offset, descs, workspace, as_stream(stream), seed_dev));
}

DGMI_API int dgmi_keep_mask_f32(const uint32_t* keep, int32_t n_keep, int64_t E, float* mask, dgmi_stream_t stream) {
  if (E < 0 || n_keep < 0 || n_keep > dgmi::kMaxKeepSegs || (n_keep > 0 && keep == nullptr)) return DGMI_ERR_INVALID_ARG;
  if (E > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (E == 0) return DGMI_OK;
  if (mask == nullptr) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::keep_mask_f32(keep, n_keep, E, mask, as_stream(stream)));
}

DGMI_API int dgmi_row_multiplicity_f32(const int32_t* indptr, const float* vals, int64_t n_rows, int64_t nnz, float rel_tol,
                                       float* row_scale, int32_t* mult, int32_t* fail, dgmi_stream_t stream) {
  if (n_rows < 0 || nnz < 0 || !(rel_tol >= 0.f) || fail == nullptr) return DGMI_ERR_INVALID_ARG;
  if (n_rows >= INT32_MAX || nnz > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n_rows > 0 && (indptr == nullptr || row_scale == nullptr)) return DGMI_ERR_INVALID_ARG;
  if (nnz > 0 && (vals == nullptr || mult == nullptr)) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::row_multiplicity_f32(indptr, vals, n_rows, rel_tol, row_scale, mult, fail, as_stream(stream)));
}

DGMI_API size_t dgmi_compact_layout_workspace_bytes(int64_t nnz) {
  return nnz < 0 ? 0 : dgmi::compact_workspace_bytes(nnz);
}

DGMI_API int dgmi_compact_layout_i32(const int32_t* ptr, int64_t n_ptr, const int32_t* indices, const float* vals,
                                     const int32_t* eid, int64_t nnz, const uint32_t* keep, int32_t n_keep, int32_t* ptr_out,
                                     int32_t* indices_out, float* vals_out, void* workspace, size_t workspace_bytes,
                                     dgmi_stream_t stream) {
  if (n_ptr < 0 || nnz < 0 || n_keep < 1 || n_keep > dgmi::kMaxKeepSegs || keep == nullptr)  // nothing dropped: nothing to compact
    return DGMI_ERR_INVALID_ARG;
  if (nnz > INT32_MAX || n_ptr > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (n_ptr == 0) return DGMI_OK;
  if (ptr == nullptr || ptr_out == nullptr) return DGMI_ERR_INVALID_ARG;
  if (nnz > 0 && (indices == nullptr || eid == nullptr || indices_out == nullptr || workspace == nullptr)) return DGMI_ERR_INVALID_ARG;
  if ((vals == nullptr) != (vals_out == nullptr)) return DGMI_ERR_INVALID_ARG;
  if (nnz > 0 && workspace_bytes < dgmi::compact_workspace_bytes(nnz)) return DGMI_ERR_WORKSPACE;
  if (nnz > 0 && (reinterpret_cast<uintptr_t>(workspace) & 7)) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::compact_layout_i32(ptr, n_ptr, indices, vals, eid, nnz, keep, n_keep, ptr_out, indices_out, vals_out,
                                           workspace, as_stream(stream)));
}

DGMI_API int dgmi_gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                    dgmi_stream_t stream) {
  if (n < 0) return DGMI_ERR_INVALID_ARG;
  if (n == 0) return DGMI_OK;
  if (in == nullptr || perm == nullptr || out == nullptr) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::gather_f32(in, perm, n, out, as_stream(stream)));
}

DGMI_API int dgmi_knn_cosine_supported(int64_t N, int64_t D, int64_t k) { return dgmi::knn_supported(N, D, k) ? 1 : 0; }

DGMI_API size_t dgmi_knn_cosine_workspace_bytes(int64_t N, int64_t D, int32_t k) {
  return dgmi::knn_supported(N, D, k) ? dgmi::knn_workspace_bytes(N, D, k) : 0;
}

DGMI_API int dgmi_knn_cosine_topk_f32(const float* Xn, int64_t ld, int64_t N, int64_t D, int32_t k, int32_t* nbr,
                                      void* workspace, size_t workspace_bytes, dgmi_stream_t stream) {
  if (N < 0 || D < 0 || k < 0 || ld < D) return DGMI_ERR_INVALID_ARG;
  if (N > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (N == 0 || k == 0) return DGMI_OK;
  if (Xn == nullptr || nbr == nullptr || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(Xn) & 15)) return DGMI_ERR_INVALID_ARG;
  if (!dgmi::knn_supported(N, D, k)) return DGMI_ERR_INVALID_ARG;
  const size_t need = dgmi::knn_workspace_bytes(N, D, k);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) return DGMI_ERR_WORKSPACE;
  return from_hip(dgmi::knn_cosine_topk_f32(Xn, ld, N, D, k, nbr, workspace, as_stream(stream)));
}

DGMI_API int dgmi_probe_row_gather_f32(const float* table, int64_t n_rows, int64_t F, int64_t groups,
                                       int64_t per_group, int64_t window, int32_t per_xcd, float* out,
                                       dgmi_stream_t stream) {
  if (table == nullptr || out == nullptr || n_rows < 1 || groups < 1 || per_group < 1) return DGMI_ERR_INVALID_ARG;
  if (F < 4 || F % 4 != 0 || F > 256 || window < 1 || window > n_rows) return DGMI_ERR_INVALID_ARG;
  if (window > UINT32_MAX || groups > INT32_MAX) return DGMI_ERR_TOO_LARGE;
  if (per_xcd && window * 8 > n_rows) return DGMI_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(table) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return DGMI_ERR_INVALID_ARG;
  return from_hip(dgmi::probe_row_gather(table, n_rows, F, groups, per_group, window, per_xcd, out, as_stream(stream)));
}

}  // extern "C"
