// dgmi_knn.hip — (f4) cosine-similarity kNN: fused fp32-MFMA similarity tiles + running top-k (gfx950).
//
// Replaces `np.dot(normalized, normalized.T)` followed by `np.argpartition(-sim, k)[:, :k]`
// (reference data_loader.py:332-341 -> :293): the N x N similarity matrix is never written.
// Here a dense tile DOES materialise (north_star: "MFMA only if a dense-block tile actually
// materialises"): a 32 x 32 block of dot products over D = 768 is 384 `v_mfma_f32_32x32x2_f32`
// (f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain — no reduced precision).
//
// One workgroup = 32 query rows, kept whole in LDS (rows padded by 4 floats: conflict-free 16-B
// reads).  Its 4 waves take the candidate tiles (32 rows each) round-robin; a wave reads its
// candidate fragments straight from global memory (16 B per lane, the same 128-B lines re-used by
// consecutive k steps from L1) and the query fragments from LDS, accumulates the 32 x 32 tile in 16
// registers, then offers each lane's 16 results (ONE query per lane: column = lane & 31, rows =
// candidates) to that lane's private top-k list in LDS (sorted, k <= 16; an insertion is rare:
// ~k ln(n / k) per list).  At the end one thread per query merges the 8 lists that saw its row.
// All workgroups sweep the candidates in the same order, so concurrently running ones share
// candidate tiles through L2 / Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

constexpr int kQ = 32;          // queries per workgroup = MFMA tile width
constexpr int kC = 32;          // candidates per tile
constexpr int kThreads = 256;   // 4 waves
constexpr int kPad = 4;         // floats added to a query row in LDS
constexpr int kMaxK = 16;
constexpr int kAhead = 8;        // candidate fragments in flight per lane

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Descending insertion of (s, id) into a lane's sorted list of `k` entries.
__device__ __forceinline__ void topk_insert(float* __restrict__ val, int32_t* __restrict__ idx, int k, float s, int32_t id) {
  int p = k - 1;
  while (p > 0 && val[p - 1] < s) {
    val[p] = val[p - 1];
    idx[p] = idx[p - 1];
    --p;
  }
  val[p] = s;
  idx[p] = id;
}

// Xn: (N, D) fp32, rows L2-normalised, leading dimension ld; D % 8 == 0.  nbr: (N, k) int32.
// gridDim.y > 1: the candidate tiles are split over blockIdx.y and each workgroup leaves its queries' k best
// (value, id) of ITS candidates in part_val / part_idx [split][N][k] for knn_merge_kernel — small N
// would otherwise occupy N / 32 of the 256 CUs.
__global__ __launch_bounds__(kThreads) void knn_cosine_topk_kernel(const float* __restrict__ Xn, int64_t ld, int N, int D,
                                                                   int k, int32_t* __restrict__ nbr,
                                                                   float* __restrict__ part_val,
                                                                   int32_t* __restrict__ part_idx,
                                                                   const int32_t* __restrict__ only_flagged) {
  extern __shared__ float knn_lds[];
  const int stride = D + kPad;
  float* q_lds = knn_lds;                                             // [kQ][D + kPad]
  float* list_val = knn_lds + kQ * stride;                            // [kThreads][k]
  int32_t* list_idx = reinterpret_cast<int32_t*>(list_val + kThreads * k);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int q0 = (int)blockIdx.x * kQ;
  // exact recomputation behind the bf16 screen (dgmi_knn_screen.hip): only tiles holding a flagged query run
  if (only_flagged != nullptr && !__syncthreads_or(tid < kQ && q0 + tid < N && only_flagged[q0 + tid] != 0)) return;

  // queries -> LDS (rows past the end repeat the last row; their results are never written)
  for (int i = tid; i < kQ * (D / 4); i += kThreads) {
    const int r = i / (D / 4), c4 = i - r * (D / 4);
    const int row = q0 + r < N ? q0 + r : N - 1;
    *reinterpret_cast<float4*>(q_lds + r * stride + 4 * c4) = *reinterpret_cast<const float4*>(Xn + (int64_t)row * ld + 4 * c4);
  }
  float* my_val = list_val + tid * k;
  int32_t* my_idx = list_idx + tid * k;
  for (int i = 0; i < k; ++i) {
    my_val[i] = -INFINITY;
    my_idx[i] = -1;
  }
  __syncthreads();

  // this lane's operands: B = query `col`, k offset 4 * half inside every 8-wide k step
  const float* qb = q_lds + col * stride + 4 * half;
  const int n_tiles = (N + kC - 1) / kC;
  const int per_split = (n_tiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t_begin = (int)blockIdx.y * per_split;
  const int t_end = t_begin + per_split < n_tiles ? t_begin + per_split : n_tiles;
  for (int t = t_begin + wave; t < t_end; t += kThreads / 64) {
    const int c0 = t * kC;
    const int crow = c0 + col < N ? c0 + col : N - 1;  // A = candidate row `col` of the tile
    const float* ca = Xn + (int64_t)crow * ld + 4 * half;
    floatx16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // 8 k per step: lanes of half 0 supply k .. k+3, half 1 supply k+4 .. k+7 (A and B alike).
    // One wave per SIMD (the query tile fills the LDS), so nothing else hides the latency of the
    // candidate loads: a ring of kAhead fragments keeps kAhead steps (4 MFMAs = 256 clocks each)
    // of loads in flight.
    float4 ring[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) ring[i] = *reinterpret_cast<const float4*>(ca + (8 * i < D ? 8 * i : 0));
    for (int kk = 0; kk < D; kk += 8 * kAhead) {
#pragma unroll
      for (int i = 0; i < kAhead; ++i) {
        if (kk + 8 * i < D) {  // wave-uniform
          const float4 a = ring[i];
          const int nxt = kk + 8 * (i + kAhead);
          if (nxt < D) ring[i] = *reinterpret_cast<const float4*>(ca + nxt);
          const float4 b = *reinterpret_cast<const float4*>(qb + kk + 8 * i);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
      }
    }
    // acc[v] = <candidate c0 + 8 (v >> 2) + 4 half + (v & 3), query q0 + col>
    float thr = my_val[k - 1];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int cand = c0 + 8 * (v >> 2) + 4 * half + (v & 3);
      const float s = acc[v];
      if (cand < N && s > thr) {
        topk_insert(my_val, my_idx, k, s, cand);
        thr = my_val[k - 1];
      }
    }
  }
  __syncthreads();

  // merge: query j's results sit in the lists of threads {64 w + j, 64 w + 32 + j}, w = 0..3
  if (tid < kQ && q0 + tid < N) {
    int pos[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pos[i] = 0;
    for (int out = 0; out < k; ++out) {
      float best = -INFINITY;
      int which = -1, at = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int owner = 64 * (i >> 1) + 32 * (i & 1) + tid;
        if (pos[i] < k) {
          const float s = list_val[owner * k + pos[i]];
          if (which < 0 || s > best) {
            best = s;
            which = i;
            at = owner * k + pos[i];
          }
        }
      }
      if (gridDim.y == 1) {
        nbr[(int64_t)(q0 + tid) * k + out] = list_idx[at];  // k <= N: the 8 lists hold at least k real entries
      } else {
        const int64_t o = ((int64_t)blockIdx.y * N + (q0 + tid)) * k + out;
        part_val[o] = which >= 0 ? best : -INFINITY;
        part_idx[o] = which >= 0 ? list_idx[at] : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i == which) ++pos[i];
    }
  }
}

// nbr[q] = the k best of the `splits` sorted partial lists of query q
__global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
                                                        int N, int k, int splits, int32_t* __restrict__ nbr) {
  const int q = (int)blockIdx.x * 256 + threadIdx.x;
  if (q >= N) return;
  float last = INFINITY;
  int32_t last_split = -1, last_pos = -1;
  for (int out = 0; out < k; ++out) {
    // the next entry in (value desc, split asc, position asc) order after the one emitted last
    float best = -INFINITY;
    int bs = -1, bp = -1;
    for (int sp = 0; sp < splits; ++sp) {
      const float* v = part_val + ((int64_t)sp * N + q) * k;
      for (int p = 0; p < k; ++p) {
        const float s = v[p];
        const bool after = s < last || (s == last && (sp > last_split || (sp == last_split && p > last_pos)));
        if (after && (bs < 0 || s > best)) {
          best = s;
          bs = sp;
          bp = p;
        }
        if (s < best) break;  // sorted descending: nothing better further down this list
      }
    }
    nbr[(int64_t)q * k + out] = part_idx[((int64_t)bs * N + q) * k + bp];
    last = best;
    last_split = bs;
    last_pos = bp;
  }
}

}  // namespace

size_t knn_lds_bytes(int64_t D, int k) { return (size_t)(kQ * (D + kPad) + kThreads * k * 2) * 4; }

bool knn_supported(int64_t N, int64_t D, int64_t k) {
  if (N < 1 || D < 8 || D % 8 != 0 || k < 1 || k > N) return false;
  if (k <= kMaxK) return knn_lds_bytes(D, (int)k) <= 160 * 1024 - 1024;  // the tile kernel (alone, or behind the screen)
  // 16 < k <= 64: only through the bf16 screen (this kernel's lists would not fit the LDS); its take-over is
  // knn_exact_rows_kernel
  return k <= kKnnScreenMaxK && D <= 1024 && N >= knn_screen_min_rows();
}

// candidate splits: enough workgroups for ~2 per CU when N / 32 alone gives fewer, each wave >= 1 tile
int knn_splits(int64_t N) {
  const int64_t q_tiles = (N + kQ - 1) / kQ, c_tiles = (N + kC - 1) / kC;
  int64_t s = (512 + q_tiles - 1) / q_tiles;
  const int64_t cap = (c_tiles + 3) / 4;
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

size_t knn_workspace_bytes(int64_t N, int64_t D, int k) {
  if (knn_screen_supported(N, D, k)) return knn_screen_workspace_bytes(N, D, k);
  const int s = knn_splits(N);
  return s == 1 ? 0 : (size_t)s * (size_t)N * (size_t)k * 8;
}

hipError_t knn_cosine_topk_exact_tiles(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr,
                                       const int32_t* flags, hipStream_t s) {
  const size_t lds = knn_lds_bytes(D, k);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_cosine_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(knn_cosine_topk_kernel, dim3((unsigned)((N + kQ - 1) / kQ), 1), dim3(kThreads), lds, s, Xn, ld, (int)N,
                     (int)D, k, nbr, (float*)nullptr, (int32_t*)nullptr, flags);
  return hipGetLastError();
}

hipError_t knn_cosine_topk_f32(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr, void* workspace,
                               hipStream_t s) {
  if (knn_screen_supported(N, D, k)) return knn_cosine_topk_screened(Xn, ld, N, D, k, nbr, workspace, s);
  // the tile kernel keeps per-lane lists of k entries in LDS: never past kKnnTileMaxK, never past the LDS
  if (k > kKnnTileMaxK || knn_lds_bytes(D, k) > 160 * 1024 - 1024) return hipErrorInvalidValue;
  const size_t lds = knn_lds_bytes(D, k);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_cosine_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  const int splits = knn_splits(N);
  float* part_val = static_cast<float*>(workspace);
  int32_t* part_idx = reinterpret_cast<int32_t*>(part_val + (size_t)splits * N * k);
  const unsigned blocks = (unsigned)((N + kQ - 1) / kQ);
  hipLaunchKernelGGL(knn_cosine_topk_kernel, dim3(blocks, (unsigned)splits), dim3(kThreads), lds, s, Xn, ld, (int)N, (int)D, k,
                     nbr, part_val, part_idx, (const int32_t*)nullptr);
  if (splits > 1)
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, part_val, part_idx, (int)N, k,
                       splits, nbr);
  return hipGetLastError();
}

}  // namespace dgmi
