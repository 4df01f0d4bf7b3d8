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
#include "dgmi_tuning.h"

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
  const float* thr;    // EMIT: [Np] emission thresholds tau0 - 2 eps (4 for the padding rows: never reached)
  // EMIT: a query's buffer = kSplits regions of cap_r slots, one per candidate split (filled by the one workgroup
  // that owns (query tile, split): LDS counters, count written at the end) + one region of cap_c slots for the
  // scores other workgroups offer it in the triangular sweep (a global counter, zeroed before the launch)
  int32_t* cnt;        // [N][kSplits + 1] entries offered (may exceed the region: overflow)
  int2* buf;           // [N][kSplits * cap_r + cap_c] (candidate id, approx bits)
  int sym;             // EMIT: 1 = triangular sweep (tile pairs j >= i, scores offered in both directions)
  // EMIT, 256 x 256 shape (knn_screen8_kernel): every kept pair is a 16-B record (query, candidate, approx bits, region) in a
  // pool of 256-record chunks that the waves draw from one cursor; knn_pool_scatter_kernel files them into the queries' buffers
  int4* pool;          // [pool_chunks][256]
  int32_t* pool_ctl;   // [0]: chunks handed out; [1 + c]: records in chunk c
  int pool_chunks;
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

// ---------------------------------------------------------------------------------------------------------------
// (r4) The 256 x 256 shape, rebuilt around LDS-DMA staging and a phase-interleaved schedule (`knn_screen8_kernel`).
//
// The first 256 x 256 kernel (above, BIG = true) staged through registers with ONE barrier per 64-deep K chunk: all 8 waves
// read fragments, multiplied, wrote the next chunk (8 `ds_write_b128` per lane = 830 LDS cycles per chunk at the 79 B/clk
// store rate, nothing else running) and met at the barrier together — MFMA pipe busy 35-38 % (profiles/r02_knn_pmc.csv).
// This one:
//   * stages with `global_load_lds_dwordx4` (no VGPRs, no store pass): the LDS image of a wave-instruction is lane-linear
//     (8 rows x 128 B), so the XOR swizzle sits on the per-lane SOURCE piece and on the fragment reads;
//   * cuts a K chunk into 4 UNITS of 16 KB — U0 / U2 = the candidate rows each wave needs for its first / second
//     32-candidate sub-tile, U1 / U3 = the query rows for its first / second 64-query sub-tile — and a chunk's work into 2
//     PHASES of 32 `v_mfma_f32_16x16x32_bf16`, two 64 x 32 quadrants of the wave's 128 x 64 tile each: phase A = the first
//     query sub-tile against both candidate sub-tiles (reads U0, U1, U2: 16 `ds_read_b128`), phase B = the second (reads U3: 8;
//     the candidate fragments stay in registers);
//   * every phase = [fragment reads, LDS-DMA of the NEXT chunk (its U0-U2 in phase A, its U3 in phase B), a counted
//     `s_waitcnt vmcnt`] barrier [32 MFMAs] barrier (at a tile's end: [DMA, the thresholding of the two quadrants finished one
//     phase ago, reads]).  Waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one wave multiplies while its partner
//     loads (MI355X_MICROARCH 'Two waves per SIMD'; measured: worth 10 %);
//   * a unit is issued two phases before its first read and waited for (counted, never 0 in the loop: `vmcnt(6)` after phase A's
//     issue leaves the next chunk's six instructions in flight across the barriers, `vmcnt(2)` after phase B's its U3) at the end
//     of the phase before that read; a slot is refilled no earlier than two phases after its last read (the staggered waves'
//     reads retire one barrier later).  The stream runs across tile boundaries; past the end it re-reads the last chunk, so the
//     counts never change, and it is drained before exit.  (First form: 4 phases of 16 MFMAs, units five phases ahead — 3 % slower;
//     the DMA instructions spread between the MFMAs, one per 32 cycles and CU: 5 % slower: profiles/r04_knn_lab_notes.md.)
// A = candidates, B = queries as before: a lane's 4 results per MFMA belong to ONE query (its column), 4 candidate rows.
constexpr int kUnitBytes = 16384;            // 128 rows x 128 B
constexpr int kBufBytes = 4 * kUnitBytes;    // one K chunk of both tiles
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define DGMI_GLDS16(src_, dst_)                                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_),                    \
                                   (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0)
// a raw barrier the compiler may move neither LDS accesses nor MFMAs across
#define DGMI_PHASE_BARRIER()          \
  {                                   \
    asm volatile("" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();     \
    __builtin_amdgcn_sched_barrier(0); \
    asm volatile("" ::: "memory");    \
  }

// LDS accesses of the bookkeeping arrays (thresholds, group lists) are asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every
// LDS access it can see while an LDS-DMA is in flight (it cannot tell the staging buffers from these arrays), which drained the
// four units in flight at every threshold read and every append.  "memory" keeps the compiler's own accesses in order; the
// hidden operations only make its counted `lgkmcnt` waits more conservative (LDS operations of a wave complete in order).
typedef int intx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ floatx4 lds_read_f4(uint32_t addr) {
  floatx4 r;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
  return r;
}

// What the EMIT pass keeps: per pair 1 in ~1400 (k = 4, N = 100 000: tau' comes from an eighth of the candidates), i.e. ~3 per
// wave and finished quadrant — not rare, so the appends are built for throughput, and so that the main loop sees NO vector-memory
// operation besides its LDS-DMA (the counted `vmcnt(8)` is in order: a store issued in an epilogue would have to complete within
// four phases, and a returning atomic drains the queue).  A lane whose 8 scores of one (query, quadrant) hold a candidate for
// either direction writes them as a GROUP (48 B: 8-B header + the two accumulator registers as they stand) into its WAVE's list in
// LDS: position by ballot rank, cursor in a scalar register, no atomic, no wait.  Once per tile the wave files its list: one
// lane per (group, score), exact comparison; a pair for one of the workgroup's OWN queries takes its slot from the (query, split)
// region's counter in LDS and goes straight to its place, a pair offered to a query of another tile is ranked by ballot into a
// 16-B record of the wave's current pool chunk (global, 256 records, drawn from one cursor with a returning atomic every ~40
// tiles) and knn_pool_scatter_kernel allocates its slot afterwards — two or three store instructions per tile and wave.
constexpr int kGrpCap = 64;     // groups per wave list (one epilogue position can add 64)
constexpr int kGrpBytes = 48;   // header (8 of 16 B) + 8 scores
constexpr int kChunk = 256;     // records per pool chunk
constexpr int kOverflowMark = 0x40000000;  // an entry count no region holds: the query is recomputed exactly

struct Screen8Lane {
  int wq, wc, fr, kg, lane;
  int q_tile, split, N, cap_r, cap_c;
  int64_t q_slots;
  uint32_t list;     // LDS byte address of this wave's group list
  uint32_t ethr_c;   // LDS byte address of the candidate rows' thresholds [2][256]
  uint32_t ethr_q;   // LDS byte address of the workgroup's own queries' thresholds [wq][fr][nt]: a lane's 8 are 32 B
  uint32_t cnt_sh;   // LDS byte address of the entry counters of the workgroup's 256 (query, split) regions
};
struct Screen8Wave {  // wave-uniform
  int g_cnt;          // groups in the list
  int chunk, used;    // current pool chunk (-1: none / pool exhausted) and records in it
};

// the wave's list -> pool records
__device__ __forceinline__ void screen8_flush(const ScreenArgs& a, const Screen8Lane& L, Screen8Wave& W) {
  const int total = W.g_cnt * 8;
  for (int base = 0; base < total; base += 64) {
    const int idx = base + L.lane, j = idx & 7;
    const bool valid = idx < total;
    const uint32_t g_at = L.list + (uint32_t)((valid ? idx : 0) >> 3) * kGrpBytes;
    int hq, hc;  // header: (direction << 31 | threshold parity << 30 | query), candidate of (mt = 0, j = 0)
    float s;
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:4\n\tds_read_b32 %2, %4 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(hq), "=&v"(hc), "=&v"(s)
                 : "v"(g_at), "v"(g_at + 4u * (uint32_t)j)
                 : "memory");
    const bool dir2 = hq < 0;
    const int qid = hq & 0x3fffffff, cand = hc + 16 * (j >> 2) + (j & 3);
    float thr;
    {
      // direction 2: the threshold of the candidate row (tile-local row = candidate & 255); direction 1: the query's own
      const int ql = qid & 255;
      const uint32_t t_at = dir2 ? L.ethr_c + (((uint32_t)hq >> 30) & 1u) * 1024u + 4u * (uint32_t)(cand & 255)
                                 : L.ethr_q + 4u * (uint32_t)((((ql >> 7) * 16 + (ql & 15)) << 3) + ((ql >> 4) & 7));
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(thr) : "v"(t_at) : "memory");
    }
    const bool hit = valid && s >= thr;  // (padding candidates carry kMasked scores, padding rows a threshold of 4)
    // direction 1: the pair belongs to one of the workgroup's own queries, whose (query, split) region no other workgroup
    // writes: the slot comes from the region's counter in LDS and the pair goes straight to its place
    if (__ballot(hit && !dir2) != 0) {
      if (hit && !dir2) {
        uint32_t slot;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(slot) : "v"(L.cnt_sh + 4u * (uint32_t)(qid & 255)), "v"(1u) : "memory");
        if (slot < (uint32_t)L.cap_r)
          a.buf[(int64_t)qid * L.q_slots + (int64_t)L.split * L.cap_r + slot] = make_int2(cand, __float_as_int(s));
      }
    }
    // direction 2: the pair belongs to a query of ANOTHER tile: a record in the wave's pool chunk, filed by the scatter kernel
    const bool hit2 = hit && dir2;
    const uint64_t bal = __ballot(hit2);
    if (bal != 0) {
      const int n_hit = __popcll(bal);
      if (W.chunk < 0 || W.used + n_hit > kChunk) {  // a new chunk (the rest of the old one stays empty)
        int c = -1;
        if (L.lane == 0) {
          if (W.chunk >= 0) a.pool_ctl[1 + W.chunk] = W.used;
          c = atomicAdd(&a.pool_ctl[0], 1);
        }
        c = __builtin_amdgcn_readfirstlane(c);
        W.chunk = c < a.pool_chunks ? c : -1;
        W.used = 0;
      }
      if (hit2) {
        if (W.chunk >= 0) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          a.pool[(int64_t)W.chunk * kChunk + W.used + rank] = make_int4(cand, qid, __float_as_int(s), kSplits);
        } else {  // pool exhausted: straight into the query's column region
          const uint32_t slot = (uint32_t)atomicAdd(&a.cnt[(int64_t)cand * (kSplits + 1) + kSplits], 1);
          if (slot < (uint32_t)L.cap_c)
            a.buf[(int64_t)cand * L.q_slots + (int64_t)kSplits * L.cap_r + slot] = make_int2(qid, __float_as_int(s));
        }
      }
      if (W.chunk >= 0) W.used += n_hit;
    }
  }
  W.g_cnt = 0;
}

// thresholding of one finished 64-query x 32-candidate quadrant (QQ, CQ) of tile `tl`, then the quadrant is cleared
template <bool EMIT, int QQ, int CQ>
__device__ __forceinline__ void screen8_quadrant(floatx4 (&acc)[8][4], float (&top)[8][4], const Screen8Lane& L, Screen8Wave& W,
                                                 uint32_t& ovf, const ScreenArgs& a, int tl, int par) {
  const int c0 = tl * 256 + L.wc * 64 + CQ * 32 + 4 * L.kg;  // candidate of (mt = 0, j = 0); (mt, j): + 16 mt + j
  if (c0 - 4 * L.kg + 32 > a.N) {                            // padding candidates (last tile only)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + 16 * m + j >= a.N) acc[4 * QQ + n][2 * CQ + m][j] = kMasked;
  }
#define DGMI_S8_MAX8(n_) fmaxf(fmaxf(fmaxf(acc[4 * QQ + (n_)][2 * CQ][0], acc[4 * QQ + (n_)][2 * CQ][1]), fmaxf(acc[4 * QQ + (n_)][2 * CQ][2], acc[4 * QQ + (n_)][2 * CQ][3])), \
                               fmaxf(fmaxf(acc[4 * QQ + (n_)][2 * CQ + 1][0], acc[4 * QQ + (n_)][2 * CQ + 1][1]), fmaxf(acc[4 * QQ + (n_)][2 * CQ + 1][2], acc[4 * QQ + (n_)][2 * CQ + 1][3])))
  if (EMIT) {
    // direction 1: the lane's own queries (thresholds in registers; 4 for a padding query: never).  Direction 2 (triangular
    // sweep, off the diagonal): the same scores offered to the queries that are this tile's candidate rows — pre-filter: the
    // lowest of the lane's 8 row thresholds.  One group serves both.
    float tmin = 4.0f;
    if (a.sym && tl != L.q_tile) {
      const uint32_t t_at = L.ethr_c + (uint32_t)par * 1024u + 4u * (uint32_t)(L.wc * 64 + CQ * 32 + 4 * L.kg);
      const floatx4 t0 = lds_read_f4(t_at);
      tmin = fminf(fminf(t0[0], t0[1]), fminf(t0[2], t0[3]));
      const floatx4 t1 = lds_read_f4(t_at + 64u);
      tmin = fminf(tmin, fminf(fminf(t1[0], t1[1]), fminf(t1[2], t1[3])));
    }
    const floatx4 ethr = lds_read_f4(L.ethr_q + 4u * (uint32_t)(((L.wq * 16 + L.fr) << 3) + 4 * QQ));  // (4 for a padding query)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int nt = 4 * QQ + n;
      const float mxn = DGMI_S8_MAX8(n);
      const bool d1 = mxn >= ethr[n], d2 = mxn >= tmin && ethr[n] < 3.5f;  // (not for a padding query)
      const uint64_t b1 = __ballot(d1), b2 = __ballot(d2);
      if ((b1 | b2) != 0) {
        int qt = L.q_tile;
        asm volatile("" : "+s"(qt));  // (computed here, not kept in a register across the loop)
        const int qg = qt * 256 + L.wq * 128 + nt * 16 + L.fr;
        // a lane may write two groups (one per direction): direction-1 groups first
        const int n1 = __popcll(b1), n2 = __popcll(b2);
        if (W.g_cnt + n1 + n2 > kGrpCap) screen8_flush(a, L, W);  // (a diagonal tile's self-matches alone are 64 groups)
        const int r1 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
        const int r2 = n1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
        if (d1) {
          const int g = W.g_cnt + r1;
          if (g < kGrpCap) {
            const uint32_t at = L.list + (uint32_t)g * kGrpBytes;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b128 %0, %3 offset:16\n\tds_write_b128 %0, %4 offset:32" ::"v"(at),
                         "v"(qg), "v"(c0), "v"(acc[nt][2 * CQ]), "v"(acc[nt][2 * CQ + 1])
                         : "memory");
          } else {
            ovf |= 1u << nt;
          }
        }
        if (d2) {
          const int g = W.g_cnt + r2;
          if (g < kGrpCap) {
            const uint32_t at = L.list + (uint32_t)g * kGrpBytes;
            const int hq = (int)(0x80000000u | ((uint32_t)par << 30) | (uint32_t)qg);
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b128 %0, %3 offset:16\n\tds_write_b128 %0, %4 offset:32" ::"v"(at),
                         "v"(hq), "v"(c0), "v"(acc[nt][2 * CQ]), "v"(acc[nt][2 * CQ + 1])
                         : "memory");
          } else {  // no room: the 8 rows' queries are recomputed exactly
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (c0 + 16 * m + j < a.N) a.cnt[(int64_t)(c0 + 16 * m + j) * (kSplits + 1) + kSplits] = kOverflowMark;
          }
        }
        const int g_new = W.g_cnt + n1 + n2;
        W.g_cnt = g_new < kGrpCap ? g_new : kGrpCap;
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int nt = 4 * QQ + n;
      if (DGMI_S8_MAX8(n) > top[nt][3]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float s = acc[nt][2 * CQ + m][j];
            top[nt][3] = fmaxf(top[nt][3], fminf(top[nt][2], s));  // branch-free sorted insert, lowest first
            top[nt][2] = fmaxf(top[nt][2], fminf(top[nt][1], s));
            top[nt][1] = fmaxf(top[nt][1], fminf(top[nt][0], s));
            top[nt][0] = fmaxf(top[nt][0], s);
          }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * QQ + n][2 * CQ + m][j] = 0.f;
}

template <bool EMIT>
__global__ __launch_bounds__(512, 2) void knn_screen8_kernel(ScreenArgs a) {
#define S8_BAR() DGMI_PHASE_BARRIER()
  extern __shared__ __align__(16) unsigned char screen_lds[];
  unsigned char* const smem = screen_lds;                                   // [2 buffers][4 units][128 rows][128 B]
  float* ethr_c = reinterpret_cast<float*>(smem + 2 * kBufBytes);           // EMIT, triangular: [2][256] thresholds of candidate rows
  float* ethr_q = ethr_c + 512;                                             // EMIT: [2][16][8] thresholds of the workgroup's queries
  uint32_t* cnt_sh = reinterpret_cast<uint32_t*>(ethr_q + 256);             // EMIT: [256] entries of the workgroup's (query, split) regions
  unsigned char* lists = smem + 2 * kBufBytes + 4096;                       // EMIT: [8 waves][kGrpCap][kGrpBytes]

  constexpr int kT = 256;
  const int n_tiles = a.Np / kT;
  const int b = (int)blockIdx.x, xcd = b & 7, jb = b >> 3;
  const int inner = jb & 63, G = (jb >> 6) * 8 + xcd;
  const int q_tile = G * 8 + (inner & 7), split = inner >> 3;
  if (q_tile >= n_tiles) return;
  int t0, tstep, nt;
  if (EMIT) {
    const int first = a.sym ? q_tile : 0;
    const int per = (n_tiles - first + kSplits - 1) / kSplits;
    t0 = first + split * per;
    tstep = 1;
    nt = t0 + per <= n_tiles ? per : (n_tiles > t0 ? n_tiles - t0 : 0);
  } else {
    // every 8th candidate tile; every 4th for k > 16: the sample is twice as long (2 -> 4 ms at N = 100 000) and tau' that much
    // closer to tau — half the pairs kept at k = 64 (1 000 -> 525 per query: EMIT 15.3 -> 12.7 ms, scatter 2.5 -> 1.5)
    const int stride = a.k > 16 && n_tiles >= 128 ? 4 : (n_tiles >= 64 ? 8 : (n_tiles >= 8 ? n_tiles / 8 : 1));
    t0 = stride * split;
    tstep = stride * kSplits;
    nt = n_tiles > t0 ? (n_tiles - t0 + tstep - 1) / tstep : 0;
    // the same number of tiles for every split (7 / 6 / 6 ... at N = 100 000 made the pass as long as its 7-tile workgroups):
    // the few sampled tiles beyond a multiple of the splits are left out — any sample of distinct candidates bounds tau
    const int sampled = (n_tiles + stride - 1) / stride;
    if (sampled >= 2 * kSplits && nt > sampled / kSplits) nt = sampled / kSplits;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Screen8Lane L;
  L.wq = wave & 1, L.wc = wave >> 1, L.fr = lane & 15, L.kg = lane >> 4;
  L.q_tile = q_tile, L.split = split, L.N = a.N, L.cap_r = a.cap_r, L.cap_c = a.cap_c;
  L.q_slots = (int64_t)kSplits * a.cap_r + a.cap_c;
  L.lane = lane;
  L.list = lds_addr(lists) + (uint32_t)wave * (kGrpCap * kGrpBytes);
  L.ethr_c = lds_addr(ethr_c), L.ethr_q = lds_addr(ethr_q), L.cnt_sh = lds_addr(cnt_sh);
  Screen8Wave W;
  W.g_cnt = 0, W.chunk = -1, W.used = 0;
  uint32_t ovf = 0;  // EMIT: bit nt = a direction-1 group of query sub-tile nt found the list full
  const int Dp = a.Dp, nK = Dp / kSK;
  const int total = nt * nK;

  float top[8][4];
#pragma unroll
  for (int n = 0; n < 8; ++n) top[n][0] = top[n][1] = top[n][2] = top[n][3] = kUnset;
  if (EMIT) {
    if (tid < kT) ethr_q[(((tid >> 7) * 16 + (tid & 15)) << 3) + ((tid >> 4) & 7)] = a.thr[q_tile * kT + tid], cnt_sh[tid] = 0;
    if (a.sym && nt > 0 && tid < kT) ethr_c[tid] = a.thr[t0 * kT + tid];
  }
  floatx4 acc[8][4];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = floatx4{0.f, 0.f, 0.f, 0.f};

  // staging: wave-instruction i of a unit fills image rows (8 i + wave) 8 .. + 8; lane -> row + (lane >> 3), slot lane & 7,
  // which holds source piece (lane & 7) ^ swizzle(image row), swizzle(r) = (r >> 1) & 7.  Source address = a running
  // scalar pointer (chunk, unit) + one of four per-lane 32-bit offsets (queries / candidates, instruction 0 / 1).
  const int64_t rowB = (int64_t)Dp * 2;
  const int piece = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const uint32_t voQ0 = (uint32_t)((wave * 8 + (lane >> 3)) * (int)rowB + piece * 16);
  const uint32_t voC0 = (uint32_t)(((wave >> 2) * 64 + (wave & 3) * 8 + (lane >> 3)) * (int)rowB + piece * 16);
  const int64_t half_rows = 128 * rowB;  // instruction 1 of a unit: 128 rows further on
  const unsigned char* const xb = reinterpret_cast<const unsigned char*>(a.Xb);
  const int64_t q_sub1 = 64 * rowB, c_sub1 = 32 * rowB;
  unsigned char* const st_dst = smem + wave * 1024;
  // instruction I (0 / 1) of unit U of the chunk at byte offsets (oq_, oc_) into buffer buf_
#define DGMI_S8_ISSUE1(U, I, oq_, oc_, buf_)                                                                     \
  {                                                                                                              \
    int64_t o_ = ((U)&1) ? (oq_) + ((U) == 3 ? q_sub1 : 0) : (oc_) + ((U) == 2 ? c_sub1 : 0);                     \
    const unsigned char* sb_ = xb + o_ + ((I) ? half_rows : 0);                                                  \
    asm volatile("" : "+s"(sb_));                                                                                \
    uint32_t vo_ = ((U)&1) ? voQ0 : voC0;                                                                        \
    asm volatile("" : "+v"(vo_));                                                                                \
    DGMI_GLDS16(sb_ + vo_, st_dst + (buf_) * kBufBytes + (U) * kUnitBytes + (I) * 8192);                          \
  }
#define DGMI_S8_ISSUE(U, oq_, oc_, buf_) \
  {                                      \
    DGMI_S8_ISSUE1(U, 0, oq_, oc_, buf_) \
    DGMI_S8_ISSUE1(U, 1, oq_, oc_, buf_) \
  }
  // fragment reads: image row (sub-tile base + 16 t + fr), slot (4 kk + kg) ^ swizzle(fr)
  const int swzr = (L.fr >> 1) & 7;
  const int q_rd0 = kUnitBytes + (L.wq * 64 + L.fr) * 128, c_rd0 = (L.wc * 32 + L.fr) * 128;
  const int sl0 = ((0 + L.kg) ^ swzr) << 4, sl1 = ((4 + L.kg) ^ swzr) << 4;
#define DGMI_S8_FRAG(off_) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + (off_)))
#define DGMI_S8_RD(dst_, off_) dst_ = DGMI_S8_FRAG(off_)
  // 16 MFMAs of one quadrant; with the LDS-DMA of this phase's unit between them when it is issued from the MFMA part
#define DGMI_S8_MMA(QB, CB, cf_)                                                                                             \
  {                                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int n = 0; n < 4; ++n)                           \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[QB + n][CB + m] =                                                  \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf_[m][kk], qf[n][kk], acc[QB + n][CB + m], 0, 0, 0);                    \
    __builtin_amdgcn_s_setprio(0);                                                                                           \
  }

  if (total > 0) {  // (workgroup-uniform)
  // chunk cursors: (tile, kc) = the chunk being multiplied; byte offsets of the next chunk's rows (o1q, o1c), clamped at the
  // last chunk
  int tile = t0, kc = 0, ci = 0;
  const int64_t oq0 = (int64_t)q_tile * kT * rowB, oc0 = (int64_t)t0 * kT * rowB;
  const int64_t tile_step = (int64_t)tstep * kT * rowB - (int64_t)(nK - 1) * 128;
  int k1 = 0;
  int64_t o1q = oq0, o1c = oc0;
#define DGMI_S8_ADVANCE(idx_)                                                   \
  {                                                                             \
    if ((idx_) + 1 < total) {                                                   \
      if (k1 + 1 == nK)                                                         \
        k1 = 0, o1q = o1q - (int64_t)(nK - 1) * 128, o1c = o1c + tile_step;     \
      else                                                                      \
        k1 = k1 + 1, o1q = o1q + 128, o1c = o1c + 128;                          \
    }                                                                           \
  }
  // prologue: chunk 0 whole; U0-U2 must have landed before the first reads, U3 before phase B
  DGMI_S8_ISSUE(0, oq0, oc0, 0);
  DGMI_S8_ISSUE(1, oq0, oc0, 0);
  DGMI_S8_ISSUE(2, oq0, oc0, 0);
  DGMI_S8_ISSUE(3, oq0, oc0, 0);
  DGMI_S8_ADVANCE(0);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  S8_BAR();
  if (wave >= 4) S8_BAR();  // waves 4-7 run one barrier behind

  bf16x8 qf[4][2], cf0[2][2], cf1[2][2];
  int tcount = 0;           // tiles finished by this workgroup
  int prev_tile = tile;     // the tile whose second-half quadrants are still to be thresholded
  for (; ci < total; ++ci) {
    const int bo = (ci & 1) * kBufBytes, bn = ((ci + 1) & 1);
    const bool first = kc == 0, last = kc == nK - 1;
    // A load part is [fragment reads, this phase's LDS-DMA] — the reads first: their latency passes while the DMA is issued.
    // Where quadrants were finished one phase ago (a tile's last chunk and the phase after it) the order is [DMA, thresholding
    // of those quadrants, reads]: the fragment registers are dead until the reads, and the epilogue with the list filing inside
    // it needs them (a spilled register costs a scratch reload, a vector-memory operation whose wait drains the DMA queue).
#define S8_FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define S8_READS_A()                                                               \
  {                                                                                \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                \
      DGMI_S8_RD(cf0[m][0], bo + c_rd0 + m * 2048 + sl0);                          \
      DGMI_S8_RD(cf0[m][1], bo + c_rd0 + m * 2048 + sl1);                          \
    }                                                                              \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                \
      DGMI_S8_RD(qf[n][0], bo + q_rd0 + n * 2048 + sl0);                           \
      DGMI_S8_RD(qf[n][1], bo + q_rd0 + n * 2048 + sl1);                           \
    }                                                                              \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                \
      DGMI_S8_RD(cf1[m][0], bo + 2 * kUnitBytes + c_rd0 + m * 2048 + sl0);         \
      DGMI_S8_RD(cf1[m][1], bo + 2 * kUnitBytes + c_rd0 + m * 2048 + sl1);         \
    }                                                                              \
  }
#define S8_READS_B()                                                               \
  {                                                                                \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                \
      DGMI_S8_RD(qf[n][0], bo + 2 * kUnitBytes + q_rd0 + n * 2048 + sl0);          \
      DGMI_S8_RD(qf[n][1], bo + 2 * kUnitBytes + q_rd0 + n * 2048 + sl1);          \
    }                                                                              \
  }
#define S8_ISSUE_A()                         \
  {                                          \
    DGMI_S8_ISSUE(0, o1q, o1c, bn);          \
    DGMI_S8_ISSUE(1, o1q, o1c, bn);          \
    DGMI_S8_ISSUE(2, o1q, o1c, bn);          \
  }
    // ---- phase A: quadrants (Q0, C0), (Q0, C1); reads U0, U1, U2; DMA of the next chunk's U0, U1, U2 ----
    if (first && ci > 0) {
      S8_ISSUE_A();
      screen8_quadrant<EMIT, 1, 1>(acc, top, L, W, ovf, a, prev_tile, (tcount - 1) & 1);
      screen8_quadrant<EMIT, 1, 0>(acc, top, L, W, ovf, a, prev_tile, (tcount - 1) & 1);
      if (EMIT && W.g_cnt > 0) screen8_flush(a, L, W);  // the finished tile's groups (its thresholds stay two more phases)
      S8_FENCE();
      S8_READS_A();
    } else {
      S8_READS_A();
      S8_ISSUE_A();
    }
    if (EMIT && kc == 1 && a.sym && wave < 4 && ci - 1 + nK < total) {
      // the NEXT tile's row thresholds, by LDS-DMA like everything else (4 B per lane, one instruction in each of waves 0-3);
      // their buffer was last read two phases ago, they are read from the next tile's last chunk on
      int ln_ = lane;
      asm volatile("" : "+v"(ln_));  // (the address is made here, not carried through the loop)
      const float* src_ = a.thr + (int64_t)(tile + tstep) * kT + wave * 64 + ln_;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,
                                       (__attribute__((address_space(3))) void*)(ethr_c + ((tcount + 1) & 1) * 256 + wave * 64), 4, 0, 0);
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");  // U3 of this chunk has landed; the six of the next and this one fly
    } else {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // U3 of this chunk has landed; the six of the next chunk fly
    }
    S8_BAR();
    DGMI_S8_MMA(0, 0, cf0);
    DGMI_S8_MMA(0, 2, cf1);
    S8_BAR();
    // ---- phase B: quadrants (Q1, C1), (Q1, C0); reads U3; DMA of the next chunk's U3 ----
    if (last) {
      DGMI_S8_ISSUE(3, o1q, o1c, bn);
      screen8_quadrant<EMIT, 0, 0>(acc, top, L, W, ovf, a, tile, tcount & 1);
      screen8_quadrant<EMIT, 0, 1>(acc, top, L, W, ovf, a, tile, tcount & 1);
      S8_FENCE();
      S8_READS_B();
    } else {
      S8_READS_B();
      DGMI_S8_ISSUE(3, o1q, o1c, bn);
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // U0-U2 of the next chunk have landed
    S8_BAR();
    DGMI_S8_MMA(4, 2, cf1);
    DGMI_S8_MMA(4, 0, cf0);
    S8_BAR();
#undef S8_FENCE
#undef S8_READS_A
#undef S8_READS_B
#undef S8_ISSUE_A

    if (last) prev_tile = tile, ++tcount;
    if (ci + 1 < total) {
      if (last) kc = 0, tile += tstep; else ++kc;
    }
    DGMI_S8_ADVANCE(ci + 1);
  }
  if (wave < 4) S8_BAR();  // waves 0-3 end one barrier ahead
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup's LDS is released
  screen8_quadrant<EMIT, 1, 1>(acc, top, L, W, ovf, a, prev_tile, (tcount - 1) & 1);
  screen8_quadrant<EMIT, 1, 0>(acc, top, L, W, ovf, a, prev_tile, (tcount - 1) & 1);
  if (EMIT && W.g_cnt > 0) screen8_flush(a, L, W);
  }

  if (EMIT) {
    if (W.chunk >= 0 && lane == 0) a.pool_ctl[1 + W.chunk] = W.used;
    if (ovf != 0) {  // a direction-1 group found the list full: its query takes the exact path
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if ((ovf >> n) & 1u) cnt_sh[L.wq * 128 + n * 16 + L.fr] = kOverflowMark;
    }
    __syncthreads();
    if (tid < kT && q_tile * kT + tid < a.N) a.cnt[(int64_t)(q_tile * kT + tid) * (kSplits + 1) + split] = (int32_t)cnt_sh[tid];
  } else {
    // a query's column is seen by the four lanes fr + 16 kg: lanes kg and kg ^ 1 merge their lists, so that the subsets
    // (and the layout of part_val) are those of the first 256 x 256 kernel: (split, wc, kg >> 1), four scores each
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(top[n][i], 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float s = o[i];
        top[n][3] = fmaxf(top[n][3], fminf(top[n][2], s));
        top[n][2] = fmaxf(top[n][2], fminf(top[n][1], s));
        top[n][1] = fmaxf(top[n][1], fminf(top[n][0], s));
        top[n][0] = fmaxf(top[n][0], s);
      }
      const int qg = q_tile * kT + L.wq * 128 + n * 16 + L.fr;
      if ((L.kg & 1) == 0 && qg < a.N)
        *reinterpret_cast<float4*>(a.part_val + ((int64_t)qg * Shape<true>::kSubsets + (split * 4 + L.wc) * 2 + (L.kg >> 1)) * 4) =
            make_float4(top[n][0], top[n][1], top[n][2], top[n][3]);
    }
  }
}
#undef S8_BAR
#undef DGMI_S8_ISSUE1
#undef DGMI_S8_ISSUE
#undef DGMI_S8_FRAG
#undef DGMI_S8_RD
#undef DGMI_S8_MMA
#undef DGMI_S8_ADVANCE
// the pool's records, filed into the buffers of the queries they belong to (region < kSplits: the candidate split's region
// of cap_r slots; kSplits: the column region of cap_c slots); a count beyond the region flags the query for the exact path
__global__ __launch_bounds__(256) void knn_pool_scatter_kernel(const int4* __restrict__ pool, const int32_t* __restrict__ pool_ctl,
                                                               int cap_r, int cap_c, int32_t* __restrict__ cnt, int2* __restrict__ buf) {
  const int chunk = (int)blockIdx.x, n = pool_ctl[1 + chunk];
  if ((int)threadIdx.x >= n) return;
  const int4 rec = pool[(int64_t)chunk * kChunk + threadIdx.x];
  const int64_t q_slots = (int64_t)kSplits * cap_r + cap_c;
  const uint32_t slot = (uint32_t)atomicAdd(&cnt[(int64_t)rec.x * (kSplits + 1) + rec.w], 1);
  if (slot < (uint32_t)(rec.w < kSplits ? cap_r : cap_c)) buf[(int64_t)rec.x * q_slots + (int64_t)rec.w * cap_r + slot] = make_int2(rec.y, rec.z);
}
#undef DGMI_GLDS16
#undef DGMI_PHASE_BARRIER

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

// tau0[q] = k-th largest of the sample scores (top four of each subset: 128 or 256 values) of query q, one wave per query
template <int J>
__global__ __launch_bounds__(256) void knn_tau_kernel(const float* __restrict__ part_val, int N, int Np, int k, float* __restrict__ tau0,
                                                      float* __restrict__ thr) {
  const int lane = threadIdx.x & 63, q = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= N) {
    if (q < Np && lane == 0) thr[q] = 4.0f;  // a padding row offers and is offered nothing
    return;
  }
  float v[J];
#pragma unroll
  for (int i = 0; i < J; ++i) v[i] = part_val[(int64_t)q * (64 * J) + lane + 64 * i];
  // (k rounds of extraction for small k; from k = 24 the 32 counting rounds of the bisection are fewer)
  const float t = k >= 24 ? wave_kth_largest_bisect<J>(v, k) : wave_kth_largest<J>(v, k, lane);
  if (lane == 0) tau0[q] = t, thr[q] = t - 2.f * kScreenEps;
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
  if constexpr (kJ < 32) {
    // short buffers (k <= 8): few registers, many waves per SIMD — they hide the latency of one entry after the other
    // (batched like the long ones this pass went 0.56 -> 0.90 ms at N = 100 000, k = 4)
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
  } else {
    // (r4) The entries to rescore, compacted into the wave's list in LDS, then their rows EIGHT at a time: one entry after the
    // other — row loads, dot product, insertion, next entry — paid a memory latency per entry with nothing else in flight
    // (k = 64, N = 100 000: ~140 entries per query, 24.3 ms for the pass; the insertion order does not matter: the list is
    // ordered by (score desc, id asc)).
    __shared__ int32_t keep_ids[4][kJ * 64];
    int32_t* mine = keep_ids[threadIdx.x >> 6];
    int n_keep = 0;
#pragma unroll
    for (int i = 0; i < kJ; ++i) {
      const bool pred = e_s[i] >= keep_from;
      const uint64_t b = __ballot(pred);
      if (b != 0) {
        if (pred) mine[n_keep + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = e_id[i];
        n_keep += __popcll(b);
      }
    }
    constexpr int kB = 8;
    for (int b0 = 0; b0 < n_keep; b0 += kB) {
      int32_t cid[kB];
      float4 c[kB][4];
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        cid[u] = mine[b0 + u < n_keep ? b0 + u : n_keep - 1];  // (one address for the wave: a broadcast)
        const float4* crow = reinterpret_cast<const float4*>(Xn + (int64_t)cid[u] * ld);
#pragma unroll
        for (int v = 0; v < 4; ++v) c[u][v] = lane + 64 * v < d4 ? crow[lane + 64 * v] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        if (b0 + u < n_keep) {  // (wave-uniform)
          float sc = 0.f;  // the arithmetic of wave_dot: the same sum whatever the batch
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (lane + 64 * v < d4) {
              sc = fmaf(qv[v].x, c[u][v].x, sc);
              sc = fmaf(qv[v].y, c[u][v].y, sc);
              sc = fmaf(qv[v].z, c[u][v].z, sc);
              sc = fmaf(qv[v].w, c[u][v].w, sc);
            }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
          wave_topk_insert(best_s, best_id, sc, cid[u], k, lane);
        }
      }
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
  int pool_chunks;  // 256-record chunks of the 256 x 256 EMIT kernel's pool (0: none)
  size_t xb, part, tau, thr, cnt, buf, flags, pool, pool_ctl, total;
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
  L.thr = at, at += align256((size_t)L.Np * 4);
  L.cnt = at, at += align256((size_t)N * (kSplits + 1) * 4);
  L.buf = at, at += align256((size_t)N * (kSplits * L.cap_r + L.cap_c) * 8);
  L.flags = at, at += align256((size_t)N * 4);
  // the phase-interleaved 256 x 256 kernel keeps ~ 18 k pairs per query (tau' comes from an eighth of the candidates, the
  // 2 eps margin); the half of them that its triangular sweep offers to the queries of OTHER tiles goes through the pool:
  // room for 16 k + 48 per query, in chunks (each wave of each workgroup holds one partly filled chunk: the 2048)
  const size_t want_chunks = ((size_t)N * (16 * k + 48) + kChunk - 1) / kChunk + 2048;
  L.pool_chunks = L.big ? (int)(want_chunks < ((size_t)1 << 30) ? want_chunks : ((size_t)1 << 30)) : 0;
  if (L.pool_chunks != 0 && tuning().knn_pool_chunks > 0) L.pool_chunks = (int)tuning().knn_pool_chunks;  // (test: a pool that runs out)
  L.pool = at, at += align256((size_t)L.pool_chunks * kChunk * 16);
  L.pool_ctl = at, at += align256(L.pool_chunks ? ((size_t)L.pool_chunks + 1) * 4 : 0);
  L.total = at;
  return L;
}

template <bool BIG>
hipError_t launch_screen(const ScreenArgs& a, int64_t N, int k, hipStream_t s) {
  using S = Shape<BIG>;
  const int n_groups = (a.Np / S::kTile + 7) / 8;
  const unsigned blocks = (unsigned)(8 * ((n_groups + 7) / 8) * 64);
  // the 256 x 256 shape runs the phase-interleaved LDS-DMA kernel (tuning knn_screen_first: the first kernel, for A/B tools; it
  // also keeps the one-chunk rows, D <= 64)
  const bool v1 = tuning().knn_screen_first != 0;
  const bool v8 = BIG && a.pool != nullptr && !v1 && a.Dp >= 2 * kSK;  // (one-chunk rows: the threshold DMA would arrive late)
  const size_t lds_sample = v8 ? 2 * kBufBytes : 2 * S::kStage;
  const size_t lds_emit = v8 ? 2 * kBufBytes + 4096 + 8 * kGrpCap * kGrpBytes : 2 * S::kStage + 2 * S::kTile * 4;
  const void* f_sample = v8 ? reinterpret_cast<const void*>(knn_screen8_kernel<false>)
                            : reinterpret_cast<const void*>(knn_screen_kernel<false, BIG>);
  const void* f_emit = v8 ? reinterpret_cast<const void*>(knn_screen8_kernel<true>)
                          : reinterpret_cast<const void*>(knn_screen_kernel<true, BIG>);
  hipError_t err = hipFuncSetAttribute(f_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sample);
  if (err != hipSuccess) return err;
  err = hipFuncSetAttribute(f_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_emit);
  if (err != hipSuccess) return err;
  if (v8)
    hipLaunchKernelGGL((knn_screen8_kernel<false>), dim3(blocks), dim3(512), lds_sample, s, a);
  else
    hipLaunchKernelGGL((knn_screen_kernel<false, BIG>), dim3(blocks), dim3(S::kThreads), lds_sample, s, a);
  hipLaunchKernelGGL((knn_tau_kernel<S::kSubsets * 4 / 64>), dim3((unsigned)((a.Np + 3) / 4)), dim3(256), 0, s, a.part_val, (int)N,
                     a.Np, k, const_cast<float*>(a.tau0), const_cast<float*>(a.thr));
  err = hipMemsetAsync(a.cnt, 0, (size_t)N * (kSplits + 1) * 4, s);
  if (err != hipSuccess) return err;
  if (v8) {
    err = hipMemsetAsync(a.pool_ctl, 0, ((size_t)a.pool_chunks + 1) * 4, s);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((knn_screen8_kernel<true>), dim3(blocks), dim3(512), lds_emit, s, a);
    hipLaunchKernelGGL(knn_pool_scatter_kernel, dim3((unsigned)a.pool_chunks), dim3(kChunk), 0, s, a.pool, a.pool_ctl, a.cap_r, a.cap_c,
                       a.cnt, a.buf);
  } else {
    hipLaunchKernelGGL((knn_screen_kernel<true, BIG>), dim3(blocks), dim3(S::kThreads), lds_emit, s, a);
  }
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
  a.thr = reinterpret_cast<const float*>(ws + L.thr);
  a.sym = L.sym ? 1 : 0;
  a.pool = L.pool_chunks ? reinterpret_cast<int4*>(ws + L.pool) : nullptr;
  a.pool_ctl = reinterpret_cast<int32_t*>(ws + L.pool_ctl);
  a.pool_chunks = L.pool_chunks;
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
