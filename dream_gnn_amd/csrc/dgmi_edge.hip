// dgmi_edge.hip — per-edge gather-concat for the MLP decoder (gfx950).
//
// Replaces graph.apply_edges(udf_u_mul_e) (reference layers.py:364,378-379): for every
// decoder edge e, out[e] = cat(H_drug[src[e]], H_dis[dst[e]]).  The node tables are small
// (N x F, L2 / Infinity-Cache resident) and the output is E x (Fa+Fb) fp32, so the kernel is
// bound by the streaming HBM write: one 16-B store per lane, (Fa+Fb)/4 lanes per edge, whole
// output rows written contiguously (F=128+128: one edge = one 1-KiB wave store).
// Algorithmic bytes per edge: 4*(Fa+Fb) written + 4*(Fa+Fb) gathered + 8 of ids.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kBlock = 256;

// 16-B path: Fa, Fb, lda, ldb, ldo multiples of 4 and 16-B aligned bases.  Thread t of the
// grid handles float4 slot (t % W4) of edge (t / W4), W4 = (Fa+Fb)/4, grid-strided.
__global__ __launch_bounds__(kBlock) void gather_concat_vec4_kernel(
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
    const float* __restrict__ A, int64_t lda, int Fa4, const float* __restrict__ B, int64_t ldb,
    int W4, float* __restrict__ out, int64_t ldo) {
  const int64_t total = E * W4;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / W4;
    const int c = (int)(t - e * W4);
    float4 v;
    if (c < Fa4)
      v = *reinterpret_cast<const float4*>(A + (int64_t)src[e] * lda + 4 * c);
    else
      v = *reinterpret_cast<const float4*>(B + (int64_t)dst[e] * ldb + 4 * (c - Fa4));
    *reinterpret_cast<float4*>(out + e * ldo + 4 * c) = v;
  }
}

// Same, for W4 dividing the block size (every power-of-two width up to 1024 floats, e.g. the
// model's 128+128): the lane's column and edge offset are fixed, so the loop carries no division.
__global__ __launch_bounds__(kBlock) void gather_concat_vec4_pow2_kernel(
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
    const float* __restrict__ A, int64_t lda, int Fa4, const float* __restrict__ B, int64_t ldb,
    int W4, float* __restrict__ out, int64_t ldo) {
  const int c = threadIdx.x % W4;
  const int epb = kBlock / W4;  // edges per block iteration
  const bool from_a = c < Fa4;
  const int32_t* __restrict__ ids = from_a ? src : dst;
  const float* __restrict__ tab = from_a ? A + 4 * c : B + 4 * (c - Fa4);
  const int64_t ld = from_a ? lda : ldb;
  float* __restrict__ o = out + 4 * c;
  const int64_t stride = (int64_t)gridDim.x * epb;
  int64_t e = (int64_t)blockIdx.x * epb + threadIdx.x / W4;
  // two edges in flight per lane
  for (; e + stride < E; e += 2 * stride) {
    const int32_t i0 = ids[e], i1 = ids[e + stride];
    const float4 v0 = *reinterpret_cast<const float4*>(tab + (int64_t)i0 * ld);
    const float4 v1 = *reinterpret_cast<const float4*>(tab + (int64_t)i1 * ld);
    *reinterpret_cast<float4*>(o + e * ldo) = v0;
    *reinterpret_cast<float4*>(o + (e + stride) * ldo) = v1;
  }
  if (e < E) *reinterpret_cast<float4*>(o + e * ldo) = *reinterpret_cast<const float4*>(tab + (int64_t)ids[e] * ld);
}

__global__ __launch_bounds__(kBlock) void gather_concat_dword_kernel(
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
    const float* __restrict__ A, int64_t lda, int Fa, const float* __restrict__ B, int64_t ldb,
    int W, float* __restrict__ out, int64_t ldo) {
  const int64_t total = E * W;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / W;
    const int c = (int)(t - e * W);
    out[e * ldo + c] = c < Fa ? A[(int64_t)src[e] * lda + c] : B[(int64_t)dst[e] * ldb + (c - Fa)];
  }
}

// out[e] = A[src[e]] + B[dst[e]] (+ bias): VEC = 4 (16-B path) or 1; thread t handles element
// group (t % W) of edge (t / W), W = F / VEC, grid-strided.
template <int VEC, bool HAS_BIAS>
__global__ __launch_bounds__(kBlock) void gather_add_kernel(
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    const float* __restrict__ bias, int W, float* __restrict__ out, int64_t ldo, int act) {
  const int64_t total = E * W;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / W;
    const int c = (int)(t - e * W) * VEC;
    const float* a = A + (int64_t)src[e] * lda + c;
    const float* b = B + (int64_t)dst[e] * ldb + c;
    float* o = out + e * ldo + c;
    if (VEC == 4) {
      float4 x = *reinterpret_cast<const float4*>(a);
      const float4 y = *reinterpret_cast<const float4*>(b);
      x.x += y.x;
      x.y += y.y;
      x.z += y.z;
      x.w += y.w;
      if (HAS_BIAS) {
        const float4 z = *reinterpret_cast<const float4*>(bias + c);
        x.x += z.x;
        x.y += z.y;
        x.z += z.z;
        x.w += z.w;
      }
      if (act == 1) {  // relu in the pass that writes the E x F result (the decoder's next op, layers.py:366)
        x.x = fmaxf(x.x, 0.f);
        x.y = fmaxf(x.y, 0.f);
        x.z = fmaxf(x.z, 0.f);
        x.w = fmaxf(x.w, 0.f);
      }
      *reinterpret_cast<float4*>(o) = x;
    } else {
      float x = a[0] + b[0];
      if (HAS_BIAS) x += bias[c];
      if (act == 1) x = fmaxf(x, 0.f);
      o[0] = x;
    }
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;  // 256 CUs x 8 blocks x 4: enough stores in flight, grid-stride the rest
  return (unsigned)b;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

hipError_t gather_concat_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A,
                             int64_t lda, int64_t Fa, const float* B, int64_t ldb, int64_t Fb,
                             float* out, int64_t ldo, hipStream_t s) {
  if (E == 0 || Fa + Fb == 0) return hipSuccess;
  const bool vec = (Fa % 4 == 0) && (Fb % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                   (ldo % 4 == 0) && al16(A) && al16(B) && al16(out);
  if (vec) {
    const int W4 = (int)((Fa + Fb) / 4);
    if (W4 <= kBlock && kBlock % W4 == 0) {
      const int64_t blocks = (E * W4 + kBlock - 1) / kBlock;
      hipLaunchKernelGGL(gather_concat_vec4_pow2_kernel, dim3(grid_for(blocks * kBlock / 2)),
                         dim3(kBlock), 0, s, src, dst, E, A, lda, (int)(Fa / 4), B, ldb, W4, out, ldo);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(gather_concat_vec4_kernel, dim3(grid_for(E * W4)), dim3(kBlock), 0, s, src,
                       dst, E, A, lda, (int)(Fa / 4), B, ldb, W4, out, ldo);
  } else {
    const int W = (int)(Fa + Fb);
    hipLaunchKernelGGL(gather_concat_dword_kernel, dim3(grid_for(E * W)), dim3(kBlock), 0, s, src,
                       dst, E, A, lda, (int)Fa, B, ldb, W, out, ldo);
  }
  return hipGetLastError();
}

hipError_t gather_add_f32(const int32_t* src, const int32_t* dst, int64_t E, const float* A, int64_t lda,
                          const float* B, int64_t ldb, const float* bias, int64_t F, float* out,
                          int64_t ldo, int act, hipStream_t s) {
  if (E == 0 || F == 0) return hipSuccess;
  const bool vec = (F % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (ldo % 4 == 0) && al16(A) &&
                   al16(B) && al16(out) && (bias == nullptr || al16(bias));
  if (vec) {
    const int W = (int)(F / 4);
    if (bias)
      hipLaunchKernelGGL((gather_add_kernel<4, true>), dim3(grid_for(E * W)), dim3(kBlock), 0, s, src, dst, E,
                         A, lda, B, ldb, bias, W, out, ldo, act);
    else
      hipLaunchKernelGGL((gather_add_kernel<4, false>), dim3(grid_for(E * W)), dim3(kBlock), 0, s, src, dst, E,
                         A, lda, B, ldb, bias, W, out, ldo, act);
  } else {
    const int W = (int)F;
    if (bias)
      hipLaunchKernelGGL((gather_add_kernel<1, true>), dim3(grid_for(E * W)), dim3(kBlock), 0, s, src, dst, E,
                         A, lda, B, ldb, bias, W, out, ldo, act);
    else
      hipLaunchKernelGGL((gather_add_kernel<1, false>), dim3(grid_for(E * W)), dim3(kBlock), 0, s, src, dst, E,
                         A, lda, B, ldb, bias, W, out, ldo, act);
  }
  return hipGetLastError();
}

// Backward of the fused output epilogue (dgmi_kernels.h Epilogue):  g = dY * act'(Y) * mask * mask_scale.
// Y is the epilogue's OUTPUT: where the mask kept an element its sign is the pre-activation's sign (leaky
// slope > 0; for relu a zero output has zero derivative either way), where the mask dropped it the factor is 0.
namespace {
__global__ __launch_bounds__(256) void epilogue_backward_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                const float* __restrict__ mask, int64_t n, int act,
                                                                float slope, float mask_scale, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float g = dY[i];
    if (act == 2) {  // Y = dropout(relu(z)): Y > 0 exactly where z > 0 and the element was kept; no mask tensor needed
      out[i] = Y[i] > 0.f ? g * mask_scale : 0.f;
      continue;
    }
    if (act == 1) g = Y[i] > 0.f ? g : g * slope;
    if (mask != nullptr) g *= mask[i] * mask_scale;
    out[i] = g;
  }
}
}  // namespace

hipError_t epilogue_backward_f32(const float* dY, const float* Y, const float* mask, int64_t n, int act, float slope,
                                 float mask_scale, float* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(epilogue_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dY, Y, mask, n, act, slope,
                     mask_scale, out);
  return hipGetLastError();
}

// (f3, complement form) out[b, c] = sum_u coef[b, u] * A[u * lda + c] — the column sums of the scaled transformed
// features of the near-complete relation, one per block (graph.py fused_relations_complement).  A library
// GEMM with M = 1..8 takes 34 us for this on MI355X; it is a 1 MB streaming read.  One workgroup per 64 columns and
// block, lanes across the columns (coalesced 256-B reads), fixed summation order -> bitwise reproducible.
namespace {
constexpr int kColsumWaves = 16;
__global__ __launch_bounds__(64 * kColsumWaves) void weighted_colsum_kernel(const float* __restrict__ A, int64_t lda,
                                                                            const float* __restrict__ coef, int64_t ldc,
                                                                            int64_t n, int W, float* __restrict__ out,
                                                                            int64_t ldo) {
  // 16 waves per workgroup, wave w takes rows w, w + 16, ...; 8 independent row loads in flight per lane (the sum is
  // latency-bound: 6 workgroups read 1 MB), partial sums added in a fixed order: per lane over its rows, then the 8
  // accumulators, then the 16 waves through LDS
  __shared__ float part[kColsumWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (int)blockIdx.x * 64 + lane;
  const float* cf = coef + (int64_t)blockIdx.y * ldc;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < W) {
    int64_t u = wave;
    for (; u + 7 * kColsumWaves < n; u += 8 * kColsumWaves) {
      float v[8], w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = A[(u + k * kColsumWaves) * lda + c];
        w[k] = cf[u + k * kColsumWaves];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(w[k], v[k], acc[k]);
    }
    for (int k = 0; u < n; u += kColsumWaves, ++k) acc[k] = fmaf(cf[u], A[u * lda + c], acc[k]);
  }
  part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < W) {
    float t = part[0][lane];
#pragma unroll
    for (int k = 1; k < kColsumWaves; ++k) t += part[k][lane];
    out[(int64_t)blockIdx.y * ldo + c] = t;
  }
}

// G[u * ldg + c] += sum_b coef[b, u] * gs[b * lds + c]: the column sums' gradient handed back to every source row
__global__ __launch_bounds__(256) void rank_add_kernel(float* __restrict__ G, int64_t ldg, const float* __restrict__ coef,
                                                       int64_t ldc, const float* __restrict__ gs, int64_t lds, int64_t n,
                                                       int W, int B) {
  const int64_t total = n * W, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t u = i / W;
    const int c = (int)(i - u * W);
    float add = 0.f;
    for (int b = 0; b < B; ++b) add = fmaf(coef[(int64_t)b * ldc + u], gs[(int64_t)b * lds + c], add);
    G[u * ldg + c] += add;
  }
}
}  // namespace

hipError_t weighted_colsum_f32(const float* A, int64_t lda, const float* coef, int64_t ldc, int64_t n, int64_t W, int B,
                               float* out, int64_t ldo, hipStream_t s) {
  if (W == 0 || B == 0) return hipSuccess;
  hipLaunchKernelGGL(weighted_colsum_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)B), dim3(64 * kColsumWaves), 0, s, A, lda, coef, ldc,
                     n, (int)W, out, ldo);
  return hipGetLastError();
}

hipError_t rank_add_f32(float* G, int64_t ldg, const float* coef, int64_t ldc, const float* gs, int64_t lds, int64_t n,
                        int64_t W, int B, hipStream_t s) {
  if (n == 0 || W == 0 || B == 0) return hipSuccess;
  int64_t blocks = (n * W + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rank_add_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, ldg, coef, ldc, gs, lds, n, (int)W, B);
  return hipGetLastError();
}

// out[u, :] = scale[u] * X[u, :] — diag(src_scale) X as one streaming pass ahead of an XCD-local product: inside the
// gather kernels the scale is a random 4-byte load per edge and column pass, 8-18 % of a config-4 product.
namespace {
template <int VEC>
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ scale,
                                                         int64_t n, int W, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n * W, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t u = i / W;
    const int c = (int)(i - u * W) * VEC;
    const float sc = scale[u];
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4*>(X + u * ldx + c);
      v.x *= sc;
      v.y *= sc;
      v.z *= sc;
      v.w *= sc;
      *reinterpret_cast<float4*>(out + u * ldo + c) = v;
    } else {
      out[u * ldo + c] = X[u * ldx + c] * sc;
    }
  }
}
}  // namespace

hipError_t scale_rows_f32(const float* X, int64_t ldx, const float* scale, int64_t n, int64_t F, float* out, int64_t ldo,
                          hipStream_t s) {
  if (n == 0 || F == 0) return hipSuccess;
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && al16(X) && al16(out);
  const int64_t W = vec ? F / 4 : F;
  int64_t blocks = (n * W + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec)
    hipLaunchKernelGGL((scale_rows_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, X, ldx, scale, n, (int)W, out, ldo);
  else
    hipLaunchKernelGGL((scale_rows_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, X, ldx, scale, n, (int)W, out, ldo);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Row scale x multiplicity form of a weighted adjacency (one-time, at layout build).  The reference's adjacencies are
// D^-1 (A + A^T + I) with A a 0/1 kNN matrix (data_loader.py:297-308 `adj + adj.T`, utils.py:11-17 `normalize`): every
// value of row i is m / rowsum_i with m a small integer.  One wave per row: s = (smallest value) / c for the first
// c in 1..8 under which every value of the row is m * s, m in 1..8, to within rel_tol (fp32 rounding of m / rowsum);
// a row with no such form raises *fail and the caller keeps the value stream.
namespace {
__global__ __launch_bounds__(256) void row_multiplicity_kernel(const int32_t* __restrict__ indptr, const float* __restrict__ vals,
                                                               int64_t n_rows, float rel_tol, float* __restrict__ row_scale,
                                                               int32_t* __restrict__ mult, int32_t* __restrict__ fail) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int b = indptr[row], e = indptr[row + 1];
  if (e <= b) {
    if (lane == 0) row_scale[row] = 0.f;
    return;
  }
  float vmin = INFINITY;
  for (int p = b + lane; p < e; p += 64) vmin = fminf(vmin, vals[p]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off, 64));
  float s = 0.f;
  bool found = false;
  if (vmin > 0.f && vmin < INFINITY) {
    for (int c = 1; c <= 8 && !found; ++c) {
      const float cand = vmin / (float)c;
      bool ok = true;
      for (int p = b + lane; p < e; p += 64) {
        const float v = vals[p];
        const float m = rintf(v / cand);
        ok = ok && m >= 1.f && m <= 8.f && fabsf(v - m * cand) <= rel_tol * v;
      }
      if (__all(ok)) {  // wave-uniform
        found = true;
        s = cand;
      }
    }
  }
  if (!found) {
    if (lane == 0) {
      row_scale[row] = 0.f;
      atomicOr(fail, 1);
    }
    for (int p = b + lane; p < e; p += 64) mult[p] = 0;
    return;
  }
  if (lane == 0) row_scale[row] = s;
  for (int p = b + lane; p < e; p += 64) mult[p] = (int)rintf(vals[p] / s) - 1;
}
}  // namespace

hipError_t row_multiplicity_f32(const int32_t* indptr, const float* vals, int64_t n_rows, float rel_tol, float* row_scale,
                                int32_t* mult, int32_t* fail, hipStream_t s) {
  hipError_t err = hipMemsetAsync(fail, 0, sizeof(int32_t), s);
  if (err != hipSuccess) return err;
  if (n_rows == 0) return hipSuccess;
  hipLaunchKernelGGL(row_multiplicity_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, indptr, vals, n_rows, rel_tol,
                     row_scale, mult, fail);
  return hipGetLastError();
}

}  // namespace dgmi
