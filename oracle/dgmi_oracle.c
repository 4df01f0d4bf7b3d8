/*
 * dgmi_oracle.c — CPU restatement of DREAM-GNN's message-passing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * check of __graft_entry__.py and bench.py's `cpu_baseline` leg may load it,
 * and only as the checker / the timed CPU baseline.  The product path
 * (dream_gnn_amd/) never imports, links or falls back to anything here.
 *
 * Parity status
 *   - th.spmm (reference layers.py:312) is pinned: tests/golden holds outputs of
 *     the installed torch's th.spmm and of the reference's own utils.normalize /
 *     utils.sparse_mx_to_torch_sparse_tensor (utils.py:11-27), imported in place.
 *   - DGL's copy_u->sum (reference layers.py:229-232) is "PARITY UNPINNED": DGL is
 *     absent from requirements.txt, not vendored and not installable here, and the
 *     reference holds no test or golden vector for it.  What is restated is DGL's
 *     published semantics: h_dst[v] = sum over in-edges (u->v) of h_src[u],
 *     multigraph (duplicates count), zero for rows without in-edges, CPU kernel
 *     = row-parallel loop over the in-edge CSR in CSR order.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Stable COO -> CSR (counting sort by row).
 * Follows: DGL's COOToCSR behind dgl.heterograph (data_loader.py:448,
 * augmentation.py:65) — edges of one destination keep their input order —
 * and equals numpy argsort(row, kind='stable').
 * Returns 0, or -1 if a row id is outside [0, n_rows).
 */
int oracle_csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                            int64_t n_rows, int32_t* indptr, int32_t* indices,
                            int32_t* eid) {
  memset(indptr, 0, (size_t)(n_rows + 1) * sizeof(int32_t));
  for (int64_t e = 0; e < E; ++e) {
    if (row[e] < 0 || row[e] >= n_rows) return -1;
    indptr[row[e] + 1]++;
  }
  for (int64_t r = 0; r < n_rows; ++r) indptr[r + 1] += indptr[r];
  int32_t* cursor = (int32_t*)malloc((size_t)(n_rows > 0 ? n_rows : 1) * sizeof(int32_t));
  if (!cursor) return -2;
  memcpy(cursor, indptr, (size_t)n_rows * sizeof(int32_t));
  for (int64_t e = 0; e < E; ++e) {
    int32_t p = cursor[row[e]]++;
    indices[p] = col[e];
    eid[p] = (int32_t)e;
  }
  free(cursor);
  return 0;
}

/* ------------------------------------------------------------------------
 * Y[v,:] = dst_scale[v] * sum_{p in row v} vals[p] * src_scale[idx[p]] * X[idx[p],:]
 *
 * vals == NULL  : update_all(copy_u, sum)      layers.py:229-232
 * vals != NULL  : th.spmm(adj, support)        layers.py:312
 * src_scale     : feat * dropout(cj)           layers.py:224-225
 * dst_scale     : rst * ci                     layers.py:234
 * The product order mirrors the reference: the source row is scaled first
 * ((X*src_scale) as layers.py:225 does before the SpMM), the edge value
 * multiplies that, the sum runs over the row's edges in CSR order in fp32,
 * and the destination scale is applied last (layers.py:234).
 * `threads` <= 0 means "all OpenMP threads"; rows are distributed statically,
 * which is what DGL's CPU SpMMSumCsr does (parallel_for over destination rows).
 */
void oracle_spmm_csr_f32(const int32_t* indptr, const int32_t* indices,
                         const float* vals, const float* X, int64_t ldx,
                         const float* src_scale, const float* dst_scale, float* Y,
                         int64_t ldy, int64_t n_dst, int64_t F, int threads) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
  for (int64_t v = 0; v < n_dst; ++v) {
    float* y = Y + v * ldy;
    for (int64_t f = 0; f < F; ++f) y[f] = 0.0f;
    for (int32_t p = indptr[v]; p < indptr[v + 1]; ++p) {
      const int32_t u = indices[p];
      const float* x = X + (int64_t)u * ldx;
      if (vals == NULL && src_scale == NULL) {
        for (int64_t f = 0; f < F; ++f) y[f] += x[f];
      } else if (vals == NULL) {
        const float s = src_scale[u];
        for (int64_t f = 0; f < F; ++f) y[f] += x[f] * s;
      } else if (src_scale == NULL) {
        const float w = vals[p];
        for (int64_t f = 0; f < F; ++f) y[f] += w * x[f];
      } else {
        const float w = vals[p], s = src_scale[u];
        for (int64_t f = 0; f < F; ++f) y[f] += w * (x[f] * s);
      }
    }
    if (dst_scale != NULL) {
      const float d = dst_scale[v];
      for (int64_t f = 0; f < F; ++f) y[f] *= d;
    }
  }
}

/* Same definition with every product and the sum carried in double; used to
 * bound the fp32 rounding of both the oracle above and the HIP kernel. */
void oracle_spmm_csr_f64(const int32_t* indptr, const int32_t* indices,
                         const float* vals, const float* X, int64_t ldx,
                         const float* src_scale, const float* dst_scale, double* Y,
                         int64_t ldy, int64_t n_dst, int64_t F, int threads) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
  for (int64_t v = 0; v < n_dst; ++v) {
    double* y = Y + v * ldy;
    for (int64_t f = 0; f < F; ++f) y[f] = 0.0;
    for (int32_t p = indptr[v]; p < indptr[v + 1]; ++p) {
      const int32_t u = indices[p];
      const float* x = X + (int64_t)u * ldx;
      double w = vals ? (double)vals[p] : 1.0;
      if (src_scale) w *= (double)src_scale[u];
      for (int64_t f = 0; f < F; ++f) y[f] += w * (double)x[f];
    }
    if (dst_scale != NULL) {
      const double d = (double)dst_scale[v];
      for (int64_t f = 0; f < F; ++f) y[f] *= d;
    }
  }
}

/* Sum over |terms| in double: the scale against which an fp32 summation
 * error is measured (|err| <= eps * n * sum|terms| for any order). */
void oracle_spmm_csr_abs_f64(const int32_t* indptr, const int32_t* indices,
                             const float* vals, const float* X, int64_t ldx,
                             const float* src_scale, const float* dst_scale,
                             double* Y, int64_t ldy, int64_t n_dst, int64_t F) {
#pragma omp parallel for schedule(static) /* rows are independent; a bound, so any thread count gives the same values */
  for (int64_t v = 0; v < n_dst; ++v) {
    double* y = Y + v * ldy;
    for (int64_t f = 0; f < F; ++f) y[f] = 0.0;
    for (int32_t p = indptr[v]; p < indptr[v + 1]; ++p) {
      const int32_t u = indices[p];
      const float* x = X + (int64_t)u * ldx;
      double w = vals ? (double)vals[p] : 1.0;
      if (src_scale) w *= (double)src_scale[u];
      if (w < 0) w = -w;
      for (int64_t f = 0; f < F; ++f) y[f] += w * (x[f] < 0 ? -(double)x[f] : (double)x[f]);
    }
    if (dst_scale != NULL) {
      double d = (double)dst_scale[v];
      if (d < 0) d = -d;
      for (int64_t f = 0; f < F; ++f) y[f] *= d;
    }
  }
}

/* ------------------------------------------------------------------------
 * th.spmm on an UNCOALESCED COO (layers.py:312 after augmentation.py:117-124
 * hands it a shuffled, possibly duplicated edge list): every stored entry
 * contributes, in storage order.  Y is fully overwritten.
 */
void oracle_spmm_coo_f32(const int64_t* row, const int64_t* col, const float* vals,
                         int64_t nnz, const float* X, int64_t ldx, float* Y,
                         int64_t ldy, int64_t n_rows, int64_t F) {
  for (int64_t v = 0; v < n_rows; ++v)
    for (int64_t f = 0; f < F; ++f) Y[v * ldy + f] = 0.0f;
  for (int64_t e = 0; e < nnz; ++e) {
    float* y = Y + row[e] * ldy;
    const float* x = X + col[e] * ldx;
    const float w = vals[e];
    for (int64_t f = 0; f < F; ++f) y[f] += w * x[f];
  }
}

/* out[p] = in[perm[p]] — edge values into CSR order. */
void oracle_gather_f32(const float* in, const int32_t* perm, int64_t n, float* out) {
  for (int64_t p = 0; p < n; ++p) out[p] = in[perm[p]];
}

/* ------------------------------------------------------------------------
 * Decoder edge gather: out[e] = cat(A[src[e]], B[dst[e]]) — the reference's
 * udf_u_mul_e under graph.apply_edges (layers.py:364,378-379).
 */
void oracle_gather_concat_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                              int64_t lda, int64_t Fa, const float* B, int64_t ldb, int64_t Fb,
                              float* out, int64_t ldo) {
  for (int64_t e = 0; e < E; ++e) {
    memcpy(out + e * ldo, A + (int64_t)src[e] * lda, (size_t)Fa * sizeof(float));
    memcpy(out + e * ldo + Fa, B + (int64_t)dst[e] * ldb, (size_t)Fb * sizeof(float));
  }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
