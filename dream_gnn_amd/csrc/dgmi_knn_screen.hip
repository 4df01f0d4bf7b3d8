// dgmi_knn_screen.hip — (f4, at scale) cosine kNN for large N: bf16-MFMA screen + exact fp32 rescoring (gfx950).
//
// The fp32 kernel of dgmi_knn.hip runs on `v_mfma_f32_32x32x2_f32`, 1/16 of the bf16 matrix rate.  At
// scale (reference data_loader.py:312-344 on N >> 10^4 rows) the neighbour SEARCH is done on the bf16
// matrix cores, and the answer is still the fp32 one — by a bound, not by hope:
//
//   rows are L2-normalised, bf16 rounding is |d| <= 2^-9 per element, bf16 x bf16 products are exact in
//   the fp32 accumulator, so  |approx(q, c) - <q, c>| <= (2^-8 + 2^-18) |q| |c| + fp32 summation error
//   <= kScreenEps.  Let tau = the k-th largest approx score of query q.  k candidates have an exact
//   score >= tau - eps, so every member of the exact top-k has exact >= tau - eps, hence
//   approx >= tau - 2 eps.  Any lower bound tau' <= tau only enlarges that set.
//
// Pipeline (all on one stream, no host round trip):
//   1. knn_to_bf16_kernel        Xn -> bf16 copy, rows / columns zero-padded to 128 / 64
//   2. knn_screen_kernel<SAMPLE> every 8th candidate tile: each lane keeps the FOUR largest approx scores of the
//                                candidates it sees for its query (registers; 32 / 64 disjoint subsets per query)
//      knn_tau_kernel            tau'(q) = k-th largest of those 128 / 256 scores of distinct candidates: k
//                                candidates reach it, so it is a lower bound of tau (k <= 64)
//   3. knn_screen_kernel<EMIT>   every pair with approx >= tau' - 2 eps is appended to the query's buffer
//                                (one global counter per query; order irrelevant).  The similarity matrix is
//                                symmetric, so only tile pairs (i, j >= i) are multiplied: a score is offered
//                                to its column's query (the lane's) and, off the diagonal, to its row's query
//                                too (thresholds of the candidate tile's rows in LDS) — half the MFMA work
//   4. knn_rescore_kernel        one wave per query: tau from the buffer, exact fp32 dot products of the
//                                entries >= tau - 2 eps, top-k by (score desc, id asc); a query whose
//                                buffer overflowed is flagged
//   5. knn_cosine_topk_kernel    (dgmi_knn.hip) recomputes the 32-query tiles holding a flagged query
//                                exactly; every other workgroup leaves at once
//
// Screen kernel: workgroup = 128 queries x 128 candidates per step, K chunks of 64 bf16 staged through
// LDS (two buffers, 16-B slots XOR-swizzled by (row >> 1) & 7: `ds_read_b128` and the staging writes
// are conflict-free on the 64-bank array), 4 waves as 2 x 2, each wave 2 x 2 `v_mfma_f32_32x32x16_bf16`
// tiles (A = candidates, B = queries: a lane's 16 results per tile belong to ONE query).  Block ids are
// laid out per XCD in super-tiles of 8 query tiles x 8 candidate splits: the 64 workgroups one XCD runs
// together share 8 query tiles (1.5 MB at D = 768, resident in its 4 MiB L2) and walk 8 candidate
// ranges in step, so a candidate tile is fetched from HBM once per 8 query tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dgmi_kernels.h"

namespace dgmi {
namespace {

// Tile shapes.  BIG = false: 128 queries x 128 candidates per step, 4 waves as 2 (q) x 2 (c), two workgroups
// per CU.  BIG = true (large N): 256 x 256, 8 waves as 2 (q) x 4 (c), each wave 4 x 2 MFMA tiles — half the
// L2 -> LDS bytes per flop (the 128 x 128 shape staged 270 GB at N = 100 000: 12 ms at the 22 TB/s L2 rate,
// against 12.7 ms of MFMA + LDS-read time, and the two overlapped poorly).
template <bool BIG>
struct Shape {
  static constexpr int kWC = BIG ? 4 : 2;              // waves along the candidates
  static constexpr int kQS = BIG ? 4 : 2;              // 32-query sub-tiles per wave
  static constexpr int kTile = BIG ? 256 : 128;        // queries per workgroup = candidates per step
  static constexpr int kThreads = 64 * 2 * kWC;        // 256 / 512
  static constexpr int kRowStep = kThreads / 8;        // rows one staging instruction covers: 32 / 64
  static constexpr int kStage = 2 * kTile * 128;       // bytes of one LDS buffer (candidate + query rows x 128 B)
  static constexpr int kSubsets = 8 * kWC * 2;         // disjoint candidate subsets per query in the sample: 32 / 64
};
constexpr int kSK = 64;   // bf16 per K chunk (128 B per row)
constexpr int kSplits = 8;
constexpr int kScreenBigMinRows = 49152;  // 256 x 256 tiles: slower at 20 000 rows (1.34 vs 1.17 ms), 10 % faster at 100 000
constexpr int kScreenMinRows = 1536;  // measured crossover against the fp32 kernel (tools/knn_crossover.py)
constexpr float kScreenEps = 0.0042f;                // 2^-8 + 2^-18 + slack for fp32 accumulation / norms
constexpr float kUnset = -2.0f;                      // below every cosine; list filler
constexpr float kMasked = -4.0f;                     // score given to padding candidates

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void knn_to_bf16_kernel(const float* __restrict__ Xn, int64_t ld, int N, int D,
                                                          uint16_t* __restrict__ Xb, int Np, int Dp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = Dp / 8;
  if (i >= (int64_t)Np * per_row) return;
  const int row = (int)(i / per_row), c0 = (int)(i - (int64_t)row * per_row) * 8;
  uint32_t h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float f = row < N && c0 + j < D ? Xn[(int64_t)row * ld + c0 + j] : 0.f;
    const uint32_t u = __float_as_uint(f);
    h[j] = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // round to nearest even (inputs are finite, |x| <= 1)
  }
  uint4 o;
  o.x = h[0] | (h[1] << 16);
  o.y = h[2] | (h[3] << 16);
  o.z = h[4] | (h[5] << 16);
  o.w = h[6] | (h[7] << 16);
  *reinterpret_cast<uint4*>(Xb + (int64_t)row * Dp + c0) = o;
}

struct ScreenArgs {
  const uint16_t* Xb;  // [Np][Dp] bf16
  int N, Np, Dp, k;       // Np: N rounded up to the tile
  int cap_r, cap_c;       // slots of one row-direction region / of the column-direction region of a query's buffer
  float* part_val;     // SAMPLE: [N][subsets][4] the four largest approx scores of each subset (kUnset: none)
  const float* tau0;   // EMIT: [N] lower bound of the k-th largest approx score
  // EMIT: a query's buffer = kSplits regions of cap_r slots, one per candidate split (filled by the one workgroup
  // that owns (query tile, split): LDS counters, count written at the end) + one region of cap_c slots for the
  // scores other workgroups offer it in the triangular sweep (a global counter, zeroed before the launch)
  int32_t* cnt;        // [N][kSplits + 1] entries offered (may exceed the region: overflow)
  int2* buf;           // [N][kSplits * cap_r + cap_c] (candidate id, approx bits)
  int sym;             // EMIT: 1 = triangular sweep (tile pairs j >= i, scores offered in both directions)
};

template <bool EMIT, bool BIG>
__global__ __launch_bounds__(Shape<BIG>::kThreads, 2) void knn_screen_kernel(ScreenArgs a) {
  using S = Shape<BIG>;
  constexpr int QS = S::kQS, kT = S::kTile, kStageBytes = S::kStage;
  extern __shared__ __align__(16) unsigned char screen_lds[];
  unsigned char* stage = screen_lds;                                             // [2][2 kT rows][128 B]
  uint32_t* cnt_sh = reinterpret_cast<uint32_t*>(screen_lds + 2 * kStageBytes);  // EMIT: [kT] entries of this (query, split) region
  float* ethr_c = reinterpret_cast<float*>(cnt_sh + kT);  // EMIT, triangular: [kT] emission thresholds of the candidate tile's rows

  // block -> (query tile, candidate split): XCD x gets blocks x, x + 8, ...; 64 consecutive ones of an
  // XCD form one super-tile (8 query tiles of query group G, 8 splits)
  const int n_tiles = a.Np / kT;
  const int b = (int)blockIdx.x, xcd = b & 7, j = b >> 3;
  const int inner = j & 63, G = (j >> 6) * 8 + xcd;
  const int q_tile = G * 8 + (inner & 7), split = inner >> 3;
  if (q_tile >= n_tiles) return;

  // candidate tiles of this workgroup: SAMPLE = every stride-th tile (stride 8; less when that would leave
  // a split without a tile), dealt round-robin to the splits; EMIT = the split's contiguous range
  int t0, tstep, nt;
  if (EMIT) {
    const int first = a.sym ? q_tile : 0;  // triangular sweep: candidate tiles from the diagonal on
    const int per = (n_tiles - first + kSplits - 1) / kSplits;
    t0 = first + split * per;
    tstep = 1;
    nt = t0 + per <= n_tiles ? per : (n_tiles > t0 ? n_tiles - t0 : 0);
  } else {
    const int stride = n_tiles >= 64 ? 8 : (n_tiles >= 8 ? n_tiles / 8 : 1);
    t0 = stride * split;
    tstep = stride * kSplits;
    nt = n_tiles > t0 ? (n_tiles - t0 + tstep - 1) / tstep : 0;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wq = wave & 1, wc = wave >> 1;
  const int Dp = a.Dp, nK = Dp / kSK;
  const int q_wave = q_tile * kT + wq * (32 * QS);  // first query of this wave

  // staging: thread -> 16-B slot sc of rows sr0 + kRowStep i (i < 4 candidates, i >= 4 queries)
  const int sc = tid & 7, sr0 = tid >> 3;
  const uint16_t* q_src = a.Xb + ((int64_t)q_tile * kT + sr0) * Dp + sc * 8;
  const uint16_t* c_src0 = a.Xb + (int64_t)sr0 * Dp + sc * 8;
  const int st_off0 = sr0 * 128 + ((sc ^ ((sr0 >> 1) & 7)) << 4);  // + kRowStep i rows: the swizzle term does not change
  // fragment reads: row (tile base + r), slot (2 ks + h) ^ swizzle(r)  (tile bases are multiples of 16)
  const int rd_row = r * 128, swz = (r >> 1) & 7;
  int rd_slot[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rd_slot[ks] = ((2 * ks + h) ^ swz) << 4;
  const int a_base = (wc * 64) * 128 + rd_row, b_base = (kT + wq * 32 * QS) * 128 + rd_row;

  float top[QS][4], ethr[QS];  // top[qs][0] >= ... >= top[qs][3]
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) top[qs][0] = top[qs][1] = top[qs][2] = top[qs][3] = kUnset, ethr[qs] = 4.0f;
  if (EMIT) {
    if (tid < kT) cnt_sh[tid] = 0;
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
      const int qg = q_wave + qs * 32 + r;
      if (qg < a.N) ethr[qs] = a.tau0[qg] - 2.f * kScreenEps;
    }
  }

  floatx16 acc[QS][2];
#pragma unroll
  for (int qs = 0; qs < QS; ++qs)
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[qs][cs][v] = 0.f;

  // next chunk's 8 slots, held in registers across the MFMAs (named, not an array: an indexed array
  // under the `more` predicate ends up in scratch)
  uint4 g0, g1, g2, g3, g4, g5, g6, g7;
  const int64_t rstep = (int64_t)S::kRowStep * Dp;
#define DGMI_SCREEN_GLOAD(tile_, kc_)                                                 \
  {                                                                                   \
    const uint16_t* cs_ = c_src0 + (int64_t)(tile_) * kT * Dp + (kc_) * kSK;          \
    const uint16_t* qs_ = q_src + (kc_) * kSK;                                        \
    g0 = *reinterpret_cast<const uint4*>(cs_);                                        \
    g1 = *reinterpret_cast<const uint4*>(cs_ + rstep);                                \
    g2 = *reinterpret_cast<const uint4*>(cs_ + 2 * rstep);                            \
    g3 = *reinterpret_cast<const uint4*>(cs_ + 3 * rstep);                            \
    g4 = *reinterpret_cast<const uint4*>(qs_);                                        \
    g5 = *reinterpret_cast<const uint4*>(qs_ + rstep);                                \
    g6 = *reinterpret_cast<const uint4*>(qs_ + 2 * rstep);                            \
    g7 = *reinterpret_cast<const uint4*>(qs_ + 3 * rstep);                            \
  }
#define DGMI_SCREEN_LSTORE(sel_)                                                      \
  {                                                                                   \
    unsigned char* base_ = stage + (sel_) * kStageBytes + st_off0;                    \
    constexpr int rs_ = S::kRowStep * 128;                                            \
    *reinterpret_cast<uint4*>(base_) = g0;                                            \
    *reinterpret_cast<uint4*>(base_ + rs_) = g1;                                      \
    *reinterpret_cast<uint4*>(base_ + 2 * rs_) = g2;                                  \
    *reinterpret_cast<uint4*>(base_ + 3 * rs_) = g3;                                  \
    *reinterpret_cast<uint4*>(base_ + 4 * rs_) = g4;                                  \
    *reinterpret_cast<uint4*>(base_ + 5 * rs_) = g5;                                  \
    *reinterpret_cast<uint4*>(base_ + 6 * rs_) = g6;                                  \
    *reinterpret_cast<uint4*>(base_ + 7 * rs_) = g7;                                  \
  }

  const int total = nt * nK;
  if (total > 0) {
    DGMI_SCREEN_GLOAD(t0, 0);
    DGMI_SCREEN_LSTORE(0);
  }
  __syncthreads();

  int tile = t0, kc = 0;
  for (int it = 0; it < total; ++it) {
    int ntile = tile, nkc = kc + 1;
    if (nkc == nK) {
      nkc = 0;
      ntile += tstep;
    }
    const bool more = it + 1 < total;
    if (more) DGMI_SCREEN_GLOAD(ntile, nkc);

    if (EMIT && kc == 0 && a.sym) {
      // thresholds of this candidate tile's rows, for the scores offered to them; the previous tile's epilogue
      // read the old ones before the barrier that ended its last chunk
      if (tid < kT) {
        const int cg = tile * kT + tid;
        ethr_c[tid] = cg < a.N ? a.tau0[cg] - 2.f * kScreenEps : 4.0f;
      }
      if (nK == 1) __syncthreads();  // otherwise a barrier separates this write from the epilogue's reads
    }
    const unsigned char* base = stage + (it & 1) * kStageBytes;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[2], fb[QS];
#pragma unroll
      for (int cs = 0; cs < 2; ++cs)
        fa[cs] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + a_base + cs * 32 * 128 + rd_slot[ks]));
#pragma unroll
      for (int qs = 0; qs < QS; ++qs)
        fb[qs] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + b_base + qs * 32 * 128 + rd_slot[ks]));
#pragma unroll
      for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) acc[qs][cs] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cs], fb[qs], acc[qs][cs], 0, 0, 0);
    }

    if (kc == nK - 1) {
      // acc[qs][cs][v] = approx <candidate c_base + 32 cs + 8 (v >> 2) + 4 h + (v & 3), query q_wave + 32 qs + r>
      const int c_base = tile * kT + wc * 64;
      if (c_base + 64 > a.N) {  // padding candidates (last tile only)
#pragma unroll
        for (int qs = 0; qs < QS; ++qs)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs)
#pragma unroll
            for (int v = 0; v < 16; ++v)
              if (c_base + 32 * cs + 8 * (v >> 2) + 4 * h + (v & 3) >= a.N) acc[qs][cs][v] = kMasked;
      }
#pragma unroll
      for (int qs = 0; qs < QS; ++qs) {
        float mx = kMasked;
#pragma unroll
        for (int cs = 0; cs < 2; ++cs)
#pragma unroll
          for (int v = 0; v < 16; ++v) mx = fmaxf(mx, acc[qs][cs][v]);
        if (EMIT) {
          const int qg = q_wave + qs * 32 + r;  // this lane's query
          const int64_t q_slots = (int64_t)kSplits * a.cap_r + a.cap_c;
          if (mx >= ethr[qs]) {  // (ethr = 4 for a padding query: never)
            const int ql = wq * (32 * QS) + qs * 32 + r;
            const int64_t row = (int64_t)qg * q_slots + (int64_t)split * a.cap_r;
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                const float s = acc[qs][cs][v];
                if (s >= ethr[qs]) {
                  const uint32_t slot = atomicAdd(&cnt_sh[ql], 1u);
                  if (slot < (uint32_t)a.cap_r)
                    a.buf[row + slot] = make_int2(c_base + 32 * cs + 8 * (v >> 2) + 4 * h + (v & 3), __float_as_int(s));
                }
              }
          }
          if (a.sym && tile != q_tile && qg < a.N) {
            // the same scores, offered to the queries that are this tile's candidate rows (padding rows carry
            // kMasked scores and a threshold of 4: never)
            const float* tc = ethr_c + wc * 64 + 4 * h;
            bool any = false;
            float4 th[2][4];
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4) {  // rows 32 cs + 8 g4 + 4 h + (0..3)
                th[cs][g4] = *reinterpret_cast<const float4*>(tc + 32 * cs + 8 * g4);
                any |= acc[qs][cs][4 * g4] >= th[cs][g4].x || acc[qs][cs][4 * g4 + 1] >= th[cs][g4].y ||
                       acc[qs][cs][4 * g4 + 2] >= th[cs][g4].z || acc[qs][cs][4 * g4 + 3] >= th[cs][g4].w;
              }
            if (any) {
#pragma unroll
              for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                  const float s = acc[qs][cs][v];
                  const float4 t4 = th[cs][v >> 2];
                  const float t = (v & 3) == 0 ? t4.x : (v & 3) == 1 ? t4.y : (v & 3) == 2 ? t4.z : t4.w;
                  if (s >= t) {
                    const int cq = c_base + 32 * cs + 8 * (v >> 2) + 4 * h + (v & 3);  // the row's query
                    const uint32_t slot = (uint32_t)atomicAdd(&a.cnt[(int64_t)cq * (kSplits + 1) + kSplits], 1);
                    if (slot < (uint32_t)a.cap_c)
                      a.buf[(int64_t)cq * q_slots + (int64_t)kSplits * a.cap_r + slot] = make_int2(qg, __float_as_int(s));
                  }
                }
            }
          }
        } else if (mx > top[qs][3]) {
#pragma unroll
          for (int cs = 0; cs < 2; ++cs)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const float s = acc[qs][cs][v];
              top[qs][3] = fmaxf(top[qs][3], fminf(top[qs][2], s));  // branch-free sorted insert, lowest first
              top[qs][2] = fmaxf(top[qs][2], fminf(top[qs][1], s));
              top[qs][1] = fmaxf(top[qs][1], fminf(top[qs][0], s));
              top[qs][0] = fmaxf(top[qs][0], s);
            }
        }
      }
#pragma unroll
      for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[qs][cs][v] = 0.f;
    }

    if (more) DGMI_SCREEN_LSTORE((it + 1) & 1);
    __syncthreads();
    tile = ntile;
    kc = nkc;
  }

  if (EMIT) {
    if (tid < kT && q_tile * kT + tid < a.N)
      a.cnt[(int64_t)(q_tile * kT + tid) * (kSplits + 1) + split] = (int32_t)cnt_sh[tid];
  } else {
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
      const int qg = q_wave + qs * 32 + r;
      if (qg < a.N)
        *reinterpret_cast<float4*>(a.part_val + ((int64_t)qg * S::kSubsets + (split * S::kWC + wc) * 2 + h) * 4) =
            make_float4(top[qs][0], top[qs][1], top[qs][2], top[qs][3]);
    }
  }
}

#undef DGMI_SCREEN_GLOAD
#undef DGMI_SCREEN_LSTORE

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// k-th largest of the values v[0..J) held per lane across the wave (destroys v)
template <int J>
__device__ __forceinline__ float wave_kth_largest(float (&v)[J], int k, int lane) {
  float m = kUnset;
  for (int round = 0; round < k; ++round) {
    float lm = kMasked;
#pragma unroll
    for (int i = 0; i < J; ++i) lm = fmaxf(lm, v[i]);
    m = wave_max(lm);
    const uint64_t holders = __ballot(lm == m);
    if (lane == __ffsll((long long)holders) - 1) {  // the first holder drops ONE instance
      bool done = false;
#pragma unroll
      for (int i = 0; i < J; ++i)
        if (!done && v[i] == m) {
          v[i] = kMasked;
          done = true;
        }
    }
  }
  return m;
}

// tau0[q] = k-th largest of the sample scores (top four of each subset: 128 or 256 values) of query q, one wave per query
template <int J>
__global__ __launch_bounds__(256) void knn_tau_kernel(const float* __restrict__ part_val, int N, int k, float* __restrict__ tau0) {
  const int lane = threadIdx.x & 63, q = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= N) return;
  float v[J];
#pragma unroll
  for (int i = 0; i < J; ++i) v[i] = part_val[(int64_t)q * (64 * J) + lane + 64 * i];
  const float t = wave_kth_largest<J>(v, k, lane);
  if (lane == 0) tau0[q] = t;
}

// k-th largest of the values v[0..J) held per lane across the wave, by bisection on the order-preserving integer
// image of the floats (32 counting rounds whatever k is; the extraction above costs k rounds).  Needs >= k values.
template <int J>
__device__ __forceinline__ float wave_kth_largest_bisect(const float (&v)[J], int k) {
  uint32_t key[J];
#pragma unroll
  for (int i = 0; i < J; ++i) {
    const uint32_t u = __float_as_uint(v[i]);
    key[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  uint32_t prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {  // the largest t with #{key >= t} >= k is the k-th largest key
    const uint32_t trial = prefix | (1u << bit);
    int c = 0;
#pragma unroll
    for (int i = 0; i < J; ++i) c += key[i] >= trial ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (c >= k) prefix = trial;
  }
  const uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
  return __uint_as_float(u);
}

// A wave's running top-k: lane i < k holds the i-th best (score desc, id asc).  Every lane passes the same (s, cid).
__device__ __forceinline__ void wave_topk_insert(float& best_s, int32_t& best_id, float s, int32_t cid, int k, int lane) {
  const bool better = best_s > s || (best_s == s && best_id < cid);
  const int pos = __popcll(__ballot(better && lane < k));
  const float up_s = __shfl_up(best_s, 1);
  const int32_t up_id = __shfl_up(best_id, 1);
  if (lane < k && lane > pos) {
    best_s = up_s;
    best_id = up_id;
  } else if (lane == pos && pos < k) {
    best_s = s;
    best_id = cid;
  }
}

// exact <Xn[q], Xn[c]>: 16 B per lane per 256 columns (qv: the query row, held the same way), fixed reduction tree
__device__ __forceinline__ float wave_dot(const float4 (&qv)[4], const float* __restrict__ crow_, int d4, int lane) {
  const float4* crow = reinterpret_cast<const float4*>(crow_);
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (lane + 64 * u < d4) {
      const float4 c = crow[lane + 64 * u];
      s = fmaf(qv[u].x, c.x, s);
      s = fmaf(qv[u].y, c.y, s);
      s = fmaf(qv[u].z, c.z, s);
      s = fmaf(qv[u].w, c.w, s);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);  // the same sum on every lane
  return s;
}

// One wave per query: exact top-k among the screened candidates.  kJ = buffer slots per lane
// ((kSplits * cap_r + cap_c) / 64).
template <int kJ>
__global__ __launch_bounds__(256) void knn_rescore_kernel(const float* __restrict__ Xn, int64_t ld, int N, int D, int k, int cap_r,
                                                          int cap_c, const int32_t* __restrict__ cnt,
                                                          const int2* __restrict__ buf, int32_t* __restrict__ nbr,
                                                          int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63, q = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= N) return;
  // entries: slot e = lane + 64 i of the query's buffer; region e / cap_r (the last one cap_c wide)
  int32_t e_id[kJ];
  float e_s[kJ], work[kJ];
  bool over = false;
  const int row_slots = kSplits * cap_r;
#pragma unroll
  for (int i = 0; i < kJ; ++i) {
    const int e = lane + 64 * i;
    const int region = e < row_slots ? e / cap_r : kSplits;
    const int at = e - region * cap_r;
    const int c = cnt[(int64_t)q * (kSplits + 1) + region];
    over |= c > (region < kSplits ? cap_r : cap_c);
    const bool live = at < c;
    const int2 rec = live ? buf[(int64_t)q * (row_slots + cap_c) + e] : make_int2(-1, 0);
    e_id[i] = rec.x;
    e_s[i] = live ? __int_as_float(rec.y) : kMasked;
    work[i] = e_s[i];
  }
  if (__ballot(over) != 0) {  // a buffer overflowed: the 32-query tile is recomputed exactly afterwards
    if (lane == 0) flags[q] = 1;
    return;
  }
  if (lane == 0) flags[q] = 0;
  const float tau = kJ >= 32 ? wave_kth_largest_bisect<kJ>(work, k) : wave_kth_largest<kJ>(work, k, lane);
  const float keep_from = tau - 2.f * kScreenEps;

  // the query row, 16 B per lane per 256 columns
  const int d4 = D / 4;
  const float4* qrow = reinterpret_cast<const float4*>(Xn + (int64_t)q * ld);
  float4 qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qv[i] = lane + 64 * i < d4 ? qrow[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);

  float best_s = kMasked;  // lane i < k: the i-th best (score desc, id asc)
  int32_t best_id = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < kJ; ++i) {
    uint64_t todo = __ballot(e_s[i] >= keep_from);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int32_t cid = __shfl(e_id[i], src);
      const float s = wave_dot(qv, Xn + (int64_t)cid * ld, d4, lane);
      wave_topk_insert(best_s, best_id, s, cid, k, lane);
    }
  }
  if (lane < k) nbr[(int64_t)q * k + lane] = best_id;
}

// Take-over for k > 16 (the fp32 tile kernel keeps its lists in LDS and stops at k = 16): one workgroup per FLAGGED
// query scores every candidate exactly, a candidate per wave at a time, and merges the four waves' top-k.
// Unflagged queries' workgroups leave at once.
__global__ __launch_bounds__(256) void knn_exact_rows_kernel(const float* __restrict__ Xn, int64_t ld, int N, int D, int k,
                                                             const int32_t* __restrict__ flags, int32_t* __restrict__ nbr) {
  const int q = (int)blockIdx.x;
  if (flags[q] == 0) return;
  __shared__ float m_s[4 * 64];
  __shared__ int32_t m_id[4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = D / 4;
  const float4* qrow = reinterpret_cast<const float4*>(Xn + (int64_t)q * ld);
  float4 qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qv[i] = lane + 64 * i < d4 ? qrow[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float best_s = kMasked;
  int32_t best_id = 0x7fffffff;
  for (int c = wave; c < N; c += 4) wave_topk_insert(best_s, best_id, wave_dot(qv, Xn + (int64_t)c * ld, d4, lane), c, k, lane);
  m_s[wave * 64 + lane] = best_s;
  m_id[wave * 64 + lane] = best_id;
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < 4; ++w)
    for (int j = 0; j < k; ++j) wave_topk_insert(best_s, best_id, m_s[w * 64 + j], m_id[w * 64 + j], k, lane);
  if (lane < k) nbr[(int64_t)q * k + lane] = best_id;
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// 256 x 256 tiles once N / 256 query tiles x 8 splits fill the chip several times over;
// DGMI_KNN_BIG_MIN_ROWS overrides the crossover (tools/knn_crossover.py)
bool screen_big(int64_t N) {
  static const int64_t min_rows = [] {
    const char* e = getenv("DGMI_KNN_BIG_MIN_ROWS");
    const long long v = e != nullptr ? atoll(e) : 0;
    return (int64_t)(v > 0 ? v : kScreenBigMinRows);
  }();
  return N >= min_rows;
}

// Triangular sweep (half the MFMA work, scores offered in both directions) once the workgroups of the first
// query tiles no longer set the run time: measured crossover (tools/knn_sym_ab.py) N ~ 20 000 at k = 4,
// ~ 40 000 at k = 16 (its second direction appends through global counters).  DGMI_KNN_SYM=0/1 overrides.
bool screen_sym(int64_t N, int k) {
  static const int forced = [] {
    const char* e = getenv("DGMI_KNN_SYM");
    return e == nullptr ? -1 : (e[0] == '0' ? 0 : 1);
  }();
  if (forced >= 0) return forced == 1;
  return N >= (k <= 8 ? 24576 : 40960);
}

struct ScreenLayout {
  bool big, sym;
  int Np, Dp, cap_r, cap_c;
  size_t xb, part, tau, cnt, buf, flags, total;
};

ScreenLayout screen_layout(int64_t N, int64_t D, int k) {
  ScreenLayout L;
  L.big = screen_big(N);
  const int tile = L.big ? Shape<true>::kTile : Shape<false>::kTile;
  L.Np = (int)((N + tile - 1) / tile * tile);
  L.Dp = (int)((D + kSK - 1) / kSK * kSK);
  L.cap_r = k <= 8 ? 64 : (k <= 16 ? 128 : 256);  // slots per (query, candidate split) region
  L.sym = screen_sym(N, k);
  L.cap_c = L.sym ? kSplits * L.cap_r : 0;  // slots of the column-direction region (triangular sweep only)
  size_t at = 0;
  L.xb = at, at += align256((size_t)L.Np * L.Dp * 2);
  L.part = at, at += align256((size_t)N * Shape<true>::kSubsets * 4 * 4);
  L.tau = at, at += align256((size_t)N * 4);
  L.cnt = at, at += align256((size_t)N * (kSplits + 1) * 4);
  L.buf = at, at += align256((size_t)N * (kSplits * L.cap_r + L.cap_c) * 8);
  L.flags = at, at += align256((size_t)N * 4);
  L.total = at;
  return L;
}

template <bool BIG>
hipError_t launch_screen(const ScreenArgs& a, int64_t N, int k, hipStream_t s) {
  using S = Shape<BIG>;
  const int n_groups = (a.Np / S::kTile + 7) / 8;
  const unsigned blocks = (unsigned)(8 * ((n_groups + 7) / 8) * 64);
  const size_t lds_sample = 2 * S::kStage, lds_emit = 2 * S::kStage + 2 * S::kTile * 4;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_screen_kernel<false, BIG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sample);
  if (err != hipSuccess) return err;
  err = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_screen_kernel<true, BIG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_emit);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((knn_screen_kernel<false, BIG>), dim3(blocks), dim3(S::kThreads), lds_sample, s, a);
  hipLaunchKernelGGL((knn_tau_kernel<S::kSubsets * 4 / 64>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, a.part_val, (int)N, k,
                     const_cast<float*>(a.tau0));
  err = hipMemsetAsync(a.cnt, 0, (size_t)N * (kSplits + 1) * 4, s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((knn_screen_kernel<true, BIG>), dim3(blocks), dim3(S::kThreads), lds_emit, s, a);
  return hipGetLastError();
}

}  // namespace

// the crossover to the screen: kScreenMinRows, or DGMI_KNN_SCREEN_MIN_ROWS (tools/knn_bench.py) — ONE value for every
// caller, so knn_supported() and knn_screen_supported() can never disagree about which kernel takes a shape
int64_t knn_screen_min_rows() {
  static const int64_t min_rows = [] {
    const char* e = getenv("DGMI_KNN_SCREEN_MIN_ROWS");
    const long long v = e != nullptr ? atoll(e) : 0;
    return (int64_t)(v > 0 ? v : kScreenMinRows);
  }();
  return min_rows;
}

bool knn_screen_supported(int64_t N, int64_t D, int64_t k) {
  // the rescoring holds a row in 4 float4 per lane; below the crossover the fp32 kernel alone is faster
  return knn_supported(N, D, k) && N >= knn_screen_min_rows() && N < (1 << 30) && D <= 1024;
}

size_t knn_screen_workspace_bytes(int64_t N, int64_t D, int k) { return screen_layout(N, D, k).total; }

hipError_t knn_cosine_topk_screened(const float* Xn, int64_t ld, int64_t N, int64_t D, int k, int32_t* nbr, void* workspace,
                                    hipStream_t s) {
  const ScreenLayout L = screen_layout(N, D, k);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  uint16_t* Xb = reinterpret_cast<uint16_t*>(ws + L.xb);
  float* part = reinterpret_cast<float*>(ws + L.part);
  float* tau0 = reinterpret_cast<float*>(ws + L.tau);
  int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
  int2* buf = reinterpret_cast<int2*>(ws + L.buf);
  int32_t* flags = reinterpret_cast<int32_t*>(ws + L.flags);

  const int64_t n8 = (int64_t)L.Np * (L.Dp / 8);
  hipLaunchKernelGGL(knn_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, Xn, ld, (int)N, (int)D, Xb, L.Np,
                     L.Dp);

  ScreenArgs a;
  a.Xb = Xb, a.N = (int)N, a.Np = L.Np, a.Dp = L.Dp, a.k = k, a.cap_r = L.cap_r, a.cap_c = L.cap_c;
  a.part_val = part, a.tau0 = tau0, a.cnt = cnt, a.buf = buf;
  a.sym = L.sym ? 1 : 0;
  hipError_t err = L.big ? launch_screen<true>(a, N, k, s) : launch_screen<false>(a, N, k, s);
  if (err != hipSuccess) return err;
  const dim3 rgrid((unsigned)((N + 3) / 4));
#define DGMI_RESCORE(J) \
  hipLaunchKernelGGL(knn_rescore_kernel<J>, rgrid, dim3(256), 0, s, Xn, ld, (int)N, (int)D, k, L.cap_r, L.cap_c, cnt, buf, nbr, flags)
  switch ((kSplits * L.cap_r + L.cap_c) / 64) {  // buffer slots per lane
    case 8: DGMI_RESCORE(8); break;
    case 16: DGMI_RESCORE(16); break;
    case 32: DGMI_RESCORE(32); break;
    default: DGMI_RESCORE(64); break;
  }
#undef DGMI_RESCORE
  err = hipGetLastError();
  if (err != hipSuccess) return err;
  if (k <= kKnnTileMaxK) return knn_cosine_topk_exact_tiles(Xn, ld, N, D, k, nbr, flags, s);
  hipLaunchKernelGGL(knn_exact_rows_kernel, dim3((unsigned)N), dim3(256), 0, s, Xn, ld, (int)N, (int)D, k, flags, nbr);
  return hipGetLastError();
}

}  // namespace dgmi
