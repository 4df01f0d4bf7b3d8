/*
 * dgmi.h — C ABI of libdgmi.so: the MI355X (gfx950) message-passing primitives
 * that stand in for the two third-party kernel call sites of DREAM-GNN's
 * layers.py.
 *
 * The reference has no FFI of its own: control leaves its Python at
 *   (B-i)  layers.py:229-232  graph.update_all(fn.copy_u('h','m'), fn.sum('m','h'))
 *                             -> DGL gspmm('copy_lhs','sum') over the in-edge CSR
 *   (B-ii) layers.py:312      th.spmm(adj, support)
 *                             -> ATen sparse-COO addmm
 *   (f1)   data_loader.py:448 / augmentation.py:65  dgl.heterograph(...) + DGL's lazy
 *                             COO->CSR build (stable by destination row)
 *   (f2)   layers.py:364,378  graph.apply_edges(udf_u_mul_e)  (cat(h_src, h_dst) per edge)
 *                             -> DGL Python-UDF path (two index_selects + cat)
 * Every entry point below names the call site it replaces.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; a stream is an opaque
 *     `void*` holding a hipStream_t (NULL = the null stream).
 *   - every pointer is a DEVICE pointer on the current HIP device unless the
 *     parameter says "host".  All buffers are caller-owned; the library keeps
 *     no state between calls and allocates nothing.  (One exception, for the
 *     measurement tools: the launch-parameter overrides of dgmi_set_tuning, which
 *     are also the only thing the library ever reads from the environment — once,
 *     not per launch.)
 *   - calls are asynchronous on `stream`, re-entrant, and never synchronise.
 *   - return value: DGMI_OK (0) or a negative dgmi_status; nothing throws,
 *     nothing aborts.  Shapes are validated on the host before any launch;
 *     index VALUES are range-checked only by dgmi_csr_from_coo_i32 (flag word
 *     in the workspace, see below), not per SpMM launch.
 *   - ids are int32 (the reference casts its graphs with .int(): train.py:199,
 *     evaluation.py:33); features / scales / values are fp32, row-major.
 */
#ifndef DGMI_H_
#define DGMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGMI_ABI_VERSION 20

/* exported-symbol marker (the library is built with -fvisibility=hidden) */
#if defined(__GNUC__)
#define DGMI_API __attribute__((visibility("default")))
#else
#define DGMI_API
#endif

typedef void* dgmi_stream_t; /* hipStream_t */

typedef enum dgmi_status {
  DGMI_OK = 0,
  DGMI_ERR_INVALID_ARG = -1,    /* null pointer, negative size, ld < F ...        */
  DGMI_ERR_TOO_LARGE = -2,      /* a count does not fit the int32 id space        */
  DGMI_ERR_WORKSPACE = -3,      /* workspace missing or too small                 */
  DGMI_ERR_LAUNCH = -4,         /* hipGetLastError() != hipSuccess after a launch */
  DGMI_ERR_NO_DEVICE = -5       /* no gfx950 device visible                       */
} dgmi_status;

/* ABI version of the loaded library (== DGMI_ABI_VERSION it was built with). */
DGMI_API int dgmi_abi_version(void);

/* Static, NUL-terminated description of a dgmi_status. */
DGMI_API const char* dgmi_status_string(int status);

/* 1 if a HIP device is visible and its arch is gfx950, else 0. Host-only query. */
DGMI_API int dgmi_device_ok(void);

/* -------------------------------------------------------------------------
 * (f1) COO -> CSR, stable by row.
 * Replaces DGL's COO->CSR behind dgl.heterograph (data_loader.py:448,
 * augmentation.py:65) and the coalesce->CSR inside th.spmm (layers.py:312).
 *
 *   indptr[r]            = #edges with row < r                (n_rows + 1 ints)
 *   eid[p]               = original position of the p-th edge in row-major,
 *                          ties (same row) kept in input order (stable)
 *   indices[p]           = col[eid[p]]
 * Duplicate (row, col) pairs are kept (multigraph semantics, as DGL and an
 * uncoalesced torch COO both have).
 *
 * Workspace protocol: call with workspace == NULL to get the required size in
 * *workspace_bytes (nothing is launched); then call again with a device
 * buffer of at least that size.  The first int32 of the workspace is an error
 * flag the launch sets to 1 if any row id is outside [0, n_rows) or — when n_cols > 0 —
 * any column id is outside [0, n_cols): read it back after synchronising the stream if the
 * ids are untrusted (n_cols <= 0 skips the column check).
 * E == 0 is valid (indptr is zero-filled).
 */
DGMI_API int dgmi_csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                          int64_t n_rows, int64_t n_cols, int32_t* indptr, int32_t* indices,
                          int32_t* eid, void* workspace, size_t* workspace_bytes,
                          dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (B-i, B-ii) CSR SpMM with optional diagonal scalings:
 *
 *   Y[v, :] = dst_scale[v] * sum_{p in [indptr[v], indptr[v+1])}
 *                 vals[p] * src_scale[indices[p]] * X[indices[p], :]
 *
 * vals == NULL      -> 1   : DGL copy_u -> sum          (layers.py:229-232)
 * vals != NULL             : DGL u_mul_e -> sum == th.spmm(adj, support)
 *                                                       (layers.py:312)
 * src_scale != NULL        : fuses `feat * dropout(cj)` (layers.py:224-225)
 * dst_scale != NULL        : fuses `rst * ci`           (layers.py:234)
 * Rows with no edges produce zeros (DGL's sum reducer; torch.spmm likewise).
 * X is (n_src, F) with leading dimension ldx >= F (elements); Y is (n_dst, F)
 * with ldy >= F.  Y must not alias X.  `indices`/`vals` are read only for rows
 * that have edges (an edgeless graph may pass NULL).  fp32 accumulate in a fixed order: the
 * result is bitwise reproducible from run to run (no atomics).
 * The backward of the op is the same call on the transposed CSR with the two
 * scales swapped: dX = diag(src_scale) A^T diag(dst_scale) dY.
 *
 * Edge dropout on the fly (all three SpMM entry points): n_keep > 0 restricts the sum to the
 * edges that survive the reference's per-iteration edge dropout (train.py:267,
 * augmentation.py:13-124) WITHOUT rebuilding the graph: position p takes part iff
 * keep(eid[p]), where eid[p] is the edge's position in the caller's COO order (as produced by
 * the layout builders) and `keep` holds n_keep (<= 8) eight-word descriptions written by
 * dgmi_random_subset_select (one per independently dropped edge list; several when a layout
 * concatenates relations).  Dropped edges are skipped — their source rows are not read into the
 * sum, so Inf / NaN there do not leak (unlike a 0/1 value mask).  n_keep == 0: eid, keep unused.
 *
 * Output epilogue (all three SpMM entry points), fused into the kernel that writes Y —
 * GCMCLayer's `dropout(agg_act(...))` on the aggregated messages (layers.py:134-138):
 *   Y[v, c] = out_mask[v * ld_mask + c] * out_mask_scale * act(dst_scale[v] * sum)
 * act: 0 = none, 1 = leaky-relu with act_slope (v > 0 ? v : v * slope; slope 0 = relu);
 * out_mask (nullable): the dropout keep mask, 0/1 floats, out_mask_scale = 1 / (1 - p).
 */
DGMI_API int dgmi_spmm_csr_f32(const int32_t* indptr, const int32_t* indices,
                      const float* vals, const int32_t* eid, const uint32_t* keep, int32_t n_keep,
                      const float* X, int64_t ldx,
                      const float* src_scale, const float* dst_scale, float* Y,
                      int64_t ldy, int64_t n_dst, int64_t n_src, int64_t F,
                      int32_t act, float act_slope, const float* out_mask, int64_t ld_mask,
                      float out_mask_scale, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * Planned SpMM: the same product with an nnz-balanced launch.
 *
 * dgmi_spmm_csr_f32 gives one wave to each destination row, which is right for regular
 * degrees (kNN graphs, the uniform synthetic bipartite graph) and wrong for power-law degree
 * distributions (one 10^6-edge row = one wave) or for the reference's near-complete
 * bipartite slices (~700 rows of ~680 edges: too few waves for 256 CUs).  A *plan* cuts rows
 * into chunks of at most `chunk` edges, one wave each; partial sums of multi-chunk rows go to
 * a caller-provided `partials` buffer and are added in chunk order by a second kernel, so the
 * result stays bitwise reproducible.  The plan depends only on indptr and `chunk`; build it
 * once per CSR (it is rebuilt with the CSR after edge dropout) and reuse it for every F.
 *
 *   chunk    = dgmi_spmm_default_chunk(n_rows, nnz)          (or any value in [16, 65536])
 *   plan     = device buffer of dgmi_spmm_plan_bytes(n_rows, nnz, chunk) bytes
 *   dgmi_spmm_plan_build(...)                                 workspace protocol as above
 *   partials = device buffer of dgmi_spmm_partials_bytes(nnz, chunk, F) bytes (16-B aligned);
 *              scratch, may be shared by calls that are ordered on one stream
 *   dgmi_spmm_csr_planned_f32(...)                            same semantics as dgmi_spmm_csr_f32
 * Sizes are upper bounds computed from (n_rows, nnz, chunk) alone; nothing is read back.
 */
DGMI_API int32_t dgmi_spmm_default_chunk(int64_t n_rows, int64_t nnz);
DGMI_API size_t dgmi_spmm_plan_bytes(int64_t n_rows, int64_t nnz, int32_t chunk);
DGMI_API size_t dgmi_spmm_partials_bytes(int64_t nnz, int32_t chunk, int64_t F);
DGMI_API int dgmi_spmm_plan_build(const int32_t* indptr, int64_t n_rows, int64_t nnz,
                                  int32_t chunk, void* plan, size_t plan_bytes, void* workspace,
                                  size_t* workspace_bytes, dgmi_stream_t stream);
DGMI_API int dgmi_spmm_csr_planned_f32(const int32_t* indptr, const int32_t* indices,
                                       const float* vals, const int32_t* eid, const uint32_t* keep,
                                       int32_t n_keep, const float* X, int64_t ldx,
                                       const float* src_scale, const float* dst_scale, float* Y,
                                       int64_t ldy, int64_t n_dst, int64_t n_src, int64_t F,
                                       int64_t nnz, int32_t chunk, const void* plan,
                                       void* partials, size_t partials_bytes,
                                       int32_t act, float act_slope, const float* out_mask,
                                       int64_t ld_mask, float out_mask_scale, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * Gather of per-edge values through a permutation: out[p] = in[perm[p]].
 * Used to carry th.spmm's COO values (utils.py:24) into CSR order with the
 * eid array of dgmi_csr_from_coo_i32.
 */
DGMI_API int dgmi_gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                    dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * XCD-local SpMM: the same product over a source-sliced CSR.
 *
 * MI355X's 8 XCDs each have a private 4 MiB L2 and workgroups are dealt round-robin over
 * them.  When the feature table X is a few L2s large (n_src * 4F up to ~64 MB) the gather is
 * 2-3x faster if every workgroup of XCD s only touches slice s of X.  The sliced CSR stores
 * the edges sorted by (slice(src), row): segptr[s * n_rows + r] .. [s * n_rows + r + 1] are row
 * r's edges with source in [s * slice_width, (s+1) * slice_width).  `eid` maps sliced positions
 * to the caller's edge order (for per-edge values).  The in-row summation order is slice by
 * slice, input order inside a slice: deterministic, but not the plain CSR's order.
 *
 *   dgmi_csr_sliced_from_coo_i32   workspace protocol and error flag as dgmi_csr_from_coo_i32
 *                                  (n_cols is required here; n_slices in [1, 64])
 *   planes = device scratch of dgmi_spmm_sliced_planes_bytes(n_dst, n_slices, F) bytes
 *   dgmi_spmm_sliced_f32           requires F % 4 == 0, ldx % 4 == 0, ldy % 4 == 0 and 16-B
 *                                  aligned X / Y / planes (returns DGMI_ERR_INVALID_ARG otherwise).
 *                                  column_passes: 0 = automatic — when an XCD's slice of X
 *                                  (n_src / n_slices rows) exceeds its 4 MiB L2 the columns are swept
 *                                  in two half-width passes (half the footprint each; +4-10 % on
 *                                  regular graphs); 1 = always one full-width pass (graphs whose time
 *                                  is set by a few very long rows: every pass repeats their chain)
 *                                  id_multiplicity: 0 = `indices` are plain source ids; 1 (vals must be NULL, n_src <=
 *                                  2^28) = bits 28..30 of every id word hold m - 1 for an integer edge multiplicity
 *                                  m in 1..8 and the edge contributes m * X[id & 0x0fffffff]: the reference's
 *                                  adjacencies are D^-1 (A + A^T + I) (data_loader.py:297-308, utils.py:11-17) — value =
 *                                  row scale x multiplicity; with the scale passed as dst_scale (forward) or folded
 *                                  into src_scale (transpose) the 4 B / edge value stream disappears
 *                                  (dgmi_row_multiplicity_f32 finds that form or says there is none).
 */
DGMI_API int dgmi_csr_sliced_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                                          int64_t n_rows, int64_t n_cols, int32_t n_slices,
                                          int32_t* segptr, int32_t* indices, int32_t* eid,
                                          void* workspace, size_t* workspace_bytes,
                                          dgmi_stream_t stream);
/* The same layout from a CSR that dgmi_csr_from_coo_i32 built (indptr, indices, eid of E entries): one
 * stable partition pass by source slice instead of a full sort — the CSR is already in (row, input)
 * order.  Bit-identical output; same workspace protocol and error flag (an out-of-range source id). */
DGMI_API int dgmi_csr_sliced_from_csr_i32(const int32_t* indptr, const int32_t* indices, const int32_t* eid,
                                          int64_t E, int64_t n_rows, int64_t n_cols, int32_t n_slices,
                                          int32_t* segptr, int32_t* sliced_indices, int32_t* sliced_eid,
                                          void* workspace, size_t* workspace_bytes, dgmi_stream_t stream);
DGMI_API size_t dgmi_spmm_sliced_planes_bytes(int64_t n_dst, int32_t n_slices, int64_t F);
DGMI_API int dgmi_spmm_sliced_f32(const int32_t* segptr, const int32_t* indices, const float* vals,
                                  const int32_t* eid, const uint32_t* keep, int32_t n_keep,
                                  const float* X, int64_t ldx, const float* src_scale,
                                  const float* dst_scale, float* Y, int64_t ldy, int64_t n_dst,
                                  int64_t n_src, int64_t F, int32_t n_slices, int32_t column_passes,
                                  int32_t id_multiplicity, void* planes, size_t planes_bytes, int32_t act, float act_slope,
                                  const float* out_mask, int64_t ld_mask, float out_mask_scale,
                                  dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (f2) Per-edge gather-concat: out[e, 0:Fa] = A[src[e], :], out[e, Fa:Fa+Fb] = B[dst[e], :].
 * Replaces graph.apply_edges(udf_u_mul_e) of the MLP decoder (layers.py:364,378-379:
 * th.cat([edges.src['h'], edges.dst['h']], 1)), which DGL runs as two index_selects and a
 * cat.  Pure copies: the result is bit-identical to the reference's.  A is (n_a, Fa) with
 * leading dimension lda, B is (n_b, Fb) with ldb, out is (E, Fa+Fb) with ldo >= Fa+Fb.
 * The backward (copy_e -> sum onto the source / destination nodes) is dgmi_spmm_csr_f32 on
 * the CSR whose `indices` are edge ids (dgmi_csr_from_coo_i32(src, iota, ...)) with
 * X = d_out (+ column offset), ldx = ldo.
 */
DGMI_API int dgmi_gather_concat_f32(const int32_t* src, const int32_t* dst, int64_t E,
                                    const float* A, int64_t lda, int64_t Fa, const float* B,
                                    int64_t ldb, int64_t Fb, float* out, int64_t ldo,
                                    dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (f2) Per-edge gather-add: out[e, :] = A[src[e], :] + B[dst[e], :] (+ bias[:]).
 * The decoder's first Linear applied to cat(h_src, h_dst) (layers.py:364-367) is linear in the two
 * halves: lin1(cat(a, b)) = a W_a^T + b W_b^T + bias.  Projecting the NODE tables first
 * (two N x F x H GEMMs) and adding the projected rows per edge avoids materialising the E x 2F
 * matrix and the E x 2F x H GEMM altogether.  Same bytes-per-edge shape as the gather-concat:
 * bound by the streaming write of E*4*F bytes.  bias may be NULL.  act = 1: out = relu(...) — the
 * decoder's next operation (layers.py:366) in the pass that writes the E x F result; 0: none.
 */
DGMI_API int dgmi_gather_add_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                                 int64_t lda, const float* B, int64_t ldb, const float* bias,
                                 int64_t F, float* out, int64_t ldo, int32_t act, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * Backward of the fused output epilogue over n contiguous elements:
 *   out[i] = dY[i] * (act == 1 ? (Y[i] > 0 ? 1 : act_slope) : 1) * (mask ? mask[i] * mask_scale : 1)
 * where Y is the epilogue's output (layers.py:134-138 backward: dropout then activation).
 *   act == 2 (mask must be NULL): Y = dropout(relu(z)) as the decoder applies it after each Linear
 *   (layers.py:366-369): out[i] = Y[i] > 0 ? dY[i] * mask_scale : 0 — Y is positive exactly where z was positive
 *   AND the element was kept, so the backward of both needs neither z nor the dropout mask, and is one pass
 *   instead of two.
 */
DGMI_API int dgmi_epilogue_backward_f32(const float* dY, const float* Y, const float* mask, int64_t n,
                                        int32_t act, float act_slope, float mask_scale, float* out,
                                        dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * out[u, :] = scale[u] * X[u, :]: diag(src_scale) X as one streaming pass.  Every SpMM entry point accepts
 * src_scale and applies it per edge; for the XCD-local form on tables of tens of MB that per-edge 4-byte gather (one
 * more cache-line request beside the row's four, repeated per column pass) costs 8-18 % of the product — pre-scale
 * with this call and pass src_scale = NULL instead (what dream_gnn_amd.ops.SlicedCSR does from 8 MB).  `out` may be X.
 */
DGMI_API int dgmi_scale_rows_f32(const float* X, int64_t ldx, const float* scale, int64_t n, int64_t F, float* out,
                                 int64_t ldo, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (f3) Complement form of a near-complete relation slice.  The reference's encoder graph holds
 * every train pair, both labels (data_loader.py:146-150,170), so its label-0 slice covers ~89 % of the cells and
 *     A_0 H = 1 colsum(H)^T - C H,    C = the cells NOT in A_0 (8x fewer),
 * which the SpMM entry points evaluate as an ordinary signed-unit-valued CSR with one extra source row per block
 * holding the column sum (layers.py:220-234 is what this replaces; dream_gnn_amd/graph.py builds the CSR).  Two
 * helpers keep that extra row out of library GEMMs (M = 1 there: 34 us against 3 us):
 *   dgmi_weighted_colsum_f32   out[b, c] = sum_u coef[b * ldc + u] * A[u * lda + c]     b < B <= 64, c < W
 *   dgmi_rank_add_f32          G[u * ldg + c] += sum_b coef[b * ldc + u] * gs[b * lds + c]   (its backward)
 * Fixed summation order: bitwise reproducible.
 */
DGMI_API int dgmi_weighted_colsum_f32(const float* A, int64_t lda, const float* coef, int64_t ldc, int64_t n,
                                      int64_t W, int32_t B, float* out, int64_t ldo, dgmi_stream_t stream);
DGMI_API int dgmi_rank_add_f32(float* G, int64_t ldg, const float* coef, int64_t ldc, const float* gs, int64_t lds,
                               int64_t n, int64_t W, int32_t B, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (D3) Edge dropout selection: a uniformly random subset of exactly `keep` of E edges — what
 * `perm = randperm(E); keep = perm[:num_keep]` (augmentation.py:48-52, 114-118) selects,
 * without materialising the permutation: per-edge keys (hash32(seed, e), e) and a SELECTION of the
 * keep-th smallest key (lists below 2^16 edges: a radix select by one workgroup; longer lists: two
 * passes over a window of the hash space around keep / E * 2^32, exact in every case).
 * Deterministic in (seed, E, keep).  workspace: dgmi_random_subset_workspace_bytes() bytes.
 * 0 <= keep <= E.
 *
 *   dgmi_random_subset_select   writes the subset's 8-word description to desc[0..7] (device):
 *                               {e_begin = e_offset, e_end = e_offset + E, seed_lo, seed_hi,
 *                               threshold hash, tie cut, flags = 0, 0}; edge e_offset + i is kept iff
 *                               hash32(seed, i) < thr || (hash32(seed, i) == thr && i <= tie cut).
 *                               This is what the SpMM entry points take as `keep`.  A caller may set
 *                               bit 0 of word 6 (flags) to INVERT the description: the edges it drops are
 *                               the ones that take part (the complement form of a near-complete relation
 *                               subtracts its dropped edges).  An edge covered by several descriptions
 *                               takes part only if every one of them lets it (dropouts compose).
 *   dgmi_random_subset_select_batch  the same for n <= 8 edge lists with ONE series of launches (the
 *                               E / keep / seed / e_offset arrays are HOST arrays of length n, e_offset
 *                               may be NULL; descs: n x 8 words on the device): a training step selects
 *                               8 subsets at once (train.py:267) and at dataset scale launches dominate.
 *   dgmi_random_subset_select_batch_dseed  the same with the n seeds read from DEVICE memory (8-byte aligned) by
 *                               the kernels: no host value is baked into the launches, so a training step captured
 *                               as a HIP graph draws new subsets on every replay (the seeds are refreshed by a
 *                               device-side RNG inside the same graph).
 *   dgmi_keep_mask_f32          mask[e] = 1.0f / 0.0f for e in [0, E) under n_keep descriptions.
 *   dgmi_random_subset_mask_f32 both in one call (e_offset = 0).
 */
DGMI_API size_t dgmi_random_subset_workspace_bytes(void);
DGMI_API int dgmi_random_subset_select(int64_t E, int64_t keep, uint64_t seed, uint32_t e_offset,
                                       uint32_t* desc, void* workspace, size_t workspace_bytes,
                                       dgmi_stream_t stream);
DGMI_API int dgmi_random_subset_select_batch(int32_t n, const int64_t* E /* host */, const int64_t* keep /* host */,
                                             const uint64_t* seed /* host */, const uint32_t* e_offset /* host */,
                                             uint32_t* descs, void* workspace, size_t workspace_bytes,
                                             dgmi_stream_t stream);
DGMI_API int dgmi_random_subset_select_batch_dseed(int32_t n, const int64_t* E /* host */, const int64_t* keep /* host */,
                                                   const uint64_t* seed_dev /* device */, const uint32_t* e_offset /* host */,
                                                   uint32_t* descs, void* workspace, size_t workspace_bytes,
                                                   dgmi_stream_t stream);
DGMI_API int dgmi_keep_mask_f32(const uint32_t* keep, int32_t n_keep, int64_t E, float* mask,
                                dgmi_stream_t stream);
DGMI_API int dgmi_random_subset_mask_f32(int64_t E, int64_t keep, uint64_t seed, float* mask,
                                         void* workspace, size_t workspace_bytes,
                                         dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (D2) Row scale x multiplicity form of a weighted adjacency, found once when its layouts are built.  For a CSR with
 * positive values (CSR order): row_scale[r] = s_r and mult[p] = m_p - 1 with vals[p] = m_p * s_r to within rel_tol
 * (relative; the fp32 rounding of m / rowsum: 2.4e-7 = 2 ulp), m_p in 1..8 — what `normalize(adj + adj.T + I)` of a 0/1
 * kNN matrix produces (data_loader.py:297-308, utils.py:11-17).  *fail (device int32) is set to 1 if some row has no
 * such form — arbitrary-valued matrices keep their value stream.  One wave per row.
 */
DGMI_API int dgmi_row_multiplicity_f32(const int32_t* indptr, const float* vals, int64_t n_rows, int64_t nnz, float rel_tol,
                                       float* row_scale, int32_t* mult, int32_t* fail, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (D3 / f1) The per-iteration graph rebuild of the reference (train.py:267 -> augmentation.py:48-65: a new
 * heterograph from the kept edges; :114-124: a new sparse COO) for one layout, without a sort: a CSR-shaped layout
 * of the PARENT graph — `ptr` (n_ptr entries: indptr of a CSR, or segptr of an XCD-sliced CSR), `indices`, optional
 * `vals`, `eid`, nnz positions — is copied with the edges dropped under the n_keep subset descriptions removed
 * (position p survives iff keep(eid[p]), as in the SpMM entry points).  Stable: survivors keep their order, so the
 * result is bit-identical to the same layout built from the kept edge list.  ptr_out has n_ptr entries;
 * indices_out / vals_out are sized nnz and the ptr_out[n_ptr - 1] survivors fill their prefix (nothing is read
 * back).  Four streaming launches, ~12 B per edge (+ 8 with values); the products that follow run the plain
 * kernels over the compacted layout (n_keep = 0): no eid stream, no hash per edge and pass — at 10 M edges an
 * edge-dropped product costs 19-27 % more than the un-dropped one on the fly, and less than it after compaction.
 * workspace: dgmi_compact_layout_workspace_bytes(nnz) bytes, 8-B aligned.  vals and vals_out: both or neither.
 * 1 <= n_keep <= 8 (a layout nothing was dropped from needs no copy).
 */
DGMI_API size_t dgmi_compact_layout_workspace_bytes(int64_t nnz);
DGMI_API int dgmi_compact_layout_i32(const int32_t* ptr, int64_t n_ptr, const int32_t* indices, const float* vals,
                                     const int32_t* eid, int64_t nnz, const uint32_t* keep, int32_t n_keep,
                                     int32_t* ptr_out, int32_t* indices_out, float* vals_out, void* workspace,
                                     size_t workspace_bytes, dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * (f4) Cosine-similarity kNN: nbr[i, 0..k) = the k rows j with the largest <Xn[i], Xn[j]> (self
 * included, as the reference's argpartition includes it), in descending order; Xn is (N, D) fp32 with
 * L2-normalised rows and leading dimension ld.  Replaces the dense N x N similarity matrix plus
 * `np.argpartition(-sim, k)[:, :k]` of data_loader.py:332-341 / :293: similarity tiles are computed
 * with fp32 MFMA (exact f32 fma chains) and reduced to a running top-k on chip; nothing of size
 * N x N is written.  From N = 1536 rows the candidates are first screened on the bf16 matrix cores
 * with a rounding bound that provably keeps every member of the fp32 top-k, then rescored in fp32
 * (rows must be unit vectors or zero for that bound).  Ties at the k-th value are broken arbitrarily
 * (upstream too).
 * dgmi_knn_cosine_supported: 1 if the shape fits the kernels (D % 8 == 0, k <= N, k <= 16 — or k <= 64 when
 * N >= 1536 and D <= 1024, where the bf16 screen runs and a wave's lanes hold the top-k —, the 32-query
 * tile + lists within LDS: D <= 1024 for k <= 4, D <= 896 for k = 16); callers fall back otherwise.
 * workspace: dgmi_knn_cosine_workspace_bytes(N, D, k) bytes of device scratch (small N: partial lists of the
 * candidate splits; N >= 1536: the bf16 copy + 4-16 KiB of screened candidates per row).
 */
DGMI_API int dgmi_knn_cosine_supported(int64_t N, int64_t D, int64_t k);
DGMI_API size_t dgmi_knn_cosine_workspace_bytes(int64_t N, int64_t D, int32_t k);
DGMI_API int dgmi_knn_cosine_topk_f32(const float* Xn, int64_t ld, int64_t N, int64_t D, int32_t k,
                                      int32_t* nbr, void* workspace, size_t workspace_bytes,
                                      dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * Measurement probe (bench.py; not part of the product path): `groups` lane groups each gather
 * `per_group` pseudo-random whole rows of `table` (n_rows x F fp32, contiguous) in exactly the
 * SpMM kernels' access shape — F/4 lanes x 16 B per row, 8 gathers in flight per lane, sums in
 * registers — with hash-generated row ids, and write one F-wide row each to out[groups x F].
 * Rows are drawn from a window of `window` rows; per_xcd != 0 gives workgroup b the window
 * number b % 8 (each XCD gathers from its own L2-sized part of the table).  Bytes gathered =
 * groups * per_group * 4F.  Requires F % 4 == 0, F <= 256, 1 <= window <= n_rows.
 */
DGMI_API int dgmi_probe_row_gather_f32(const float* table, int64_t n_rows, int64_t F, int64_t groups,
                                       int64_t per_group, int64_t window, int32_t per_xcd, float* out,
                                       dgmi_stream_t stream);

/* -------------------------------------------------------------------------
 * Launch-parameter overrides for the measurement tools (the scripts under tools/; not part of the product path).  The built-in
 * choices are measured ones; a tool that wants to A/B one inside a single process sets it here.
 * Names: "sliced_rows", "sliced_touch_lead" (-1 = built-in, 0 = no touch-ahead), "sliced_lpr", "sliced_no_off32",
 * "sliced_chunk_rows", "select_window_min", "select_narrow_window", "sort_plain_tiles", "knn_screen_first", "knn_pool_chunks"
 * (change the latter only between a workspace-size query and nothing: the size depends on it); 0 restores the built-in choice (-1 for
 * sliced_touch_lead).  The same values are read from the environment variables DGMI_SLICED_ROWS, DGMI_SLICED_PF,
 * DGMI_SLICED_LPR, DGMI_NO_OFF32, DGMI_SLICED_CHUNK_ROWS, DGMI_SELECT_WINDOW_MIN, DGMI_SELECT_NARROW_WINDOW, DGMI_SORT_PLAIN_TILES,
 * DGMI_KNN_SCREEN_V1, DGMI_KNN_POOL_CHUNKS ONCE, when
 * the library first needs them; no launch reads the environment.  Not thread-safe against concurrent launches.
 */
DGMI_API int dgmi_set_tuning(const char* name /* host */, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* DGMI_H_ */
