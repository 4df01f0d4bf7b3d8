// dgmi_sliced.hip — XCD-local CSR SpMM for gfx950 (MI355X).
//
// MI355X has 8 XCDs, each with a private 4 MiB L2; workgroups are dealt round-robin over the
// XCDs (block b runs on XCD b % 8 — a speed observation, never relied on for correctness).
// A uniformly random gather over a 25-50 MB feature table hits L2 only ~27 % of the time and
// runs at the Infinity-Cache rate (~8 TB/s).  If every workgroup on XCD s only ever gathers
// source rows from slice s of the table (1/8 of it: 3-6 MB, i.e. L2-sized), the same gather
// runs 2.4-3x faster (measured: 19-26 TB/s algorithmic).
//
// So the graph is stored as 8 column-blocked CSRs (dgmi_csr.hip, key = slice * n_rows + row)
// and block b processes rows of slice b % 8 only, writing a partial row into plane b % 8;
// a second, streaming kernel adds the 8 planes in slice order (deterministic) and applies
// dst_scale.  Extra traffic: 2 * 8 * N_dst * 4F bytes of plane write + read, against
// nnz * 4F bytes of gather that now come out of L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgmi_kernels.h"
#include "dgmi_segment.h"
#include "dgmi_tuning.h"

namespace dgmi {
namespace {

constexpr int64_t kColumnPassMinRows = 32768;  // column passes only when a pass still has >= ~8k waves
constexpr int kRowsPerGroup = 8;  // < LPR (row boundaries live one per lane of the group)
constexpr int kTouchLead = 24;   // worker blocks of a slice between a toucher and the blocks it touches for
constexpr int kTouchGroup = 8;   // worker blocks per toucher block

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_plane_row(float* p, const float4& v) {
  // one streaming 16-B store: the planes are write-once / read-once; keep them from evicting
  // the XCD's slice of X out of L2
  // (ordinary stores instead: the step of bench.py 3.03 ms against 2.85 ms, profiles/r03_swept_experiment/)
  v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// A (row, slice) segment is short (deg / n_slices: 12-25 edges at config 4).  In the slice-major
// layout the segments of consecutive rows of one slice are contiguous, so each LPR-lane *group*
// of the wave (F=128: a half-wave) streams the edges of kRowsPerGroup consecutive rows as ONE
// run — ids read LPR at a time by the group's own lanes, one gathered source row per group per
// wave-instruction, 8 gathers in flight per group — and cuts the running sum at the row
// boundaries (held one per lane, fetched with ds_bpermute when crossed).  No bubble between
// rows, no cross-group reduction.  Within a row the sum is sequential in edge order.
//
// grid.x = n_slices * (worker blocks + toucher blocks), worker blocks = ceil(rows / (4 waves * G groups * kRowsPerGroup));
// block b: slice = b % n_slices.
// Rows [row_begin, row_end) of every slice; plane row index is relative to row_begin.
// KEEP: edge dropout on the fly — an edge whose keep(eid[p]) fails (dgmi_keep.h) is flagged in the sign
// bit of its source id; its gather repeats the group's previous row (an L1 hit — never one fixed row,
// which would turn 10 % of all gathers into traffic on a single L2 channel) and its contribution is
// replaced by zeros.
// Source row `idx` of the feature table.  OFF32: the table is < 4 GiB, so the byte offset fits 32 bits — one
// v_mul_lo_u32 and a global_load with a scalar base and a 32-bit vector offset, instead of the six-instruction 64-bit
// multiply-add chain a (int64 ldx) product costs per gathered row.
template <bool OFF32>
__device__ __forceinline__ float4 ld_row(const float* __restrict__ X, const float* __restrict__ Xc, int idx, int64_t ldx,
                                         uint32_t row_bytes, uint32_t col_bytes) {
  // X is the (wave-uniform) table base, Xc = X + this lane's column
  if (OFF32) return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(X) + ((uint32_t)idx * row_bytes + col_bytes));
  return ld4(Xc + (int64_t)idx * ldx);
}

// VALS: 0 = unit edge values; 1 = vals[p]; 2 = small integer multiplicities carried in the id words themselves (bits
// kMultShift..30 hold m - 1): the reference's adjacencies are D^-1 (A + A^T + I) (data_loader.py:297-308, utils.py:11-17),
// i.e. value = row scale x multiplicity — the scale goes where dst_scale / src_scale go, the multiplicity rides with the
// id: no value stream (4 B / edge), no second cross-lane hand-off per edge.
template <int LPR, int VALS, bool HAS_SS, bool KEEP, bool OFF32>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void spmm_sliced_vec4_kernel(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ src_scale, float* __restrict__ planes, int64_t ldp, int64_t n_dst,
    int64_t row_begin, int64_t row_end, int F, int n_slices, const int32_t* __restrict__ eid,
    const KeepSeg* __restrict__ keep, int n_keep, int touch_lead, int rows_per_group, int touch_group) {
  constexpr int G = kWave / LPR;
  const int R = rows_per_group;  // < LPR: a group's row boundaries live one per lane
  constexpr bool HAS_VALS = VALS == 1;
  constexpr bool MULT = VALS == 2;
  constexpr bool WEIGHTED = HAS_VALS || HAS_SS || MULT;  // a per-edge factor exists
  constexpr bool W_LANE = HAS_VALS || HAS_SS;            // ... and travels in a register of its own
  constexpr int kIdMask = MULT ? (int)kMultIdMask : 0x7fffffff;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR, glane = lane % LPR, gbase = grp * LPR;
  const int slice = (int)(blockIdx.x % (unsigned)n_slices);
  int64_t block = blockIdx.x / (unsigned)n_slices;
  if (touch_group > 0) {
    // Touch-ahead.  The id stream and the row boundaries are read once, so a wave's first two loads (boundaries, then
    // ids — dependent) miss every cache, and it gathers nothing for two memory latencies of its ~20 us life: inside a
    // training step, where the other products have pushed this one's ids out of the Infinity Cache, that is 10-25 % of the
    // product.  Every (touch_group + 1)-th block of a slice therefore gathers nothing: it touches the boundaries and the id
    // lines (one word per 128-B line) of the touch_group worker blocks that start touch_lead worker blocks further on IN THE
    // SAME SLICE — same XCD, same L2, a few microseconds later — and leaves.  Nobody waits for these loads but the toucher
    // (vmcnt is in order: a worker that issued them would hold its own first gathers back).  A hint: results never depend on it.
    const int64_t t = block / (touch_group + 1);
    if (block % (touch_group + 1) == 0) {
      __shared__ int range[2];
      const int32_t* sp_s = segptr + (int64_t)slice * n_dst;
      const int64_t rows_blk = (int64_t)kWavesPerBlock * G * R;
      const int64_t r_first = row_begin + (t * touch_group + touch_lead) * rows_blk;
      if (r_first >= row_end) return;  // block-uniform
      const int64_t r_last = min(r_first + touch_group * rows_blk, row_end);
      int keepalive = 0;
      if (wave == 0) {  // lanes 0 / 1: the id range of those blocks; the others: one word per line of their boundaries
        const int64_t rp = lane == 0 ? r_first : (lane == 1 ? r_last : r_first + (int64_t)(lane - 1) * 32);
        if (rp <= r_last) {
          const int v = sp_s[rp];
          if (lane < 2) range[lane] = v;
          keepalive = v;
        }
      }
      __syncthreads();
      const int e0 = range[0], e1 = range[1];
      for (int64_t p = (int64_t)e0 + (int64_t)threadIdx.x * 32; p < e1; p += (int64_t)blockDim.x * 32) {
        keepalive ^= indices[p];
        if (HAS_VALS) keepalive ^= __float_as_int(vals[p]);
        if (KEEP) keepalive ^= eid[p];
      }
      asm volatile("" ::"v"(keepalive));  // the loads exist, and are waited for, without an instruction
      return;
    }
    block -= t + 1;  // worker blocks are numbered without the touchers
  }
  const int64_t row0 = row_begin + ((block * kWavesPerBlock + wave) * G + grp) * R;
  if (row0 >= row_end) return;  // whole group idle (lanes of other groups carry on)
  const int nr = (int)(row0 + R <= row_end ? R : row_end - row0);
  int col = ((int)blockIdx.y * LPR + glane) * 4;
  const bool col_ok = col < F;
  if (!col_ok) col = 0;
  const float* Xc = X + col;
  const uint32_t row_bytes = (uint32_t)ldx * 4u, col_bytes = (uint32_t)col * 4u;
  const int32_t* sp = segptr + (int64_t)slice * n_dst + row0;
  float* prow = planes + ((int64_t)slice * (row_end - row_begin) + (row0 - row_begin)) * ldp + col;

  const KeepPre first = first_seg<KEEP>(keep, n_keep);
  const int my_b = sp[glane < nr ? glane : nr];  // lane k holds boundary k (k <= nr)
  const int e_begin = __shfl(my_b, gbase, kWave);
  const int e_end = __shfl(my_b, gbase + nr, kWave);
  int r = 0;
  int next_b = __shfl(my_b, gbase + 1, kWave);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // ids (and weights) one batch ahead of the gathers that use them
  int nxt_idx = 0;
  float nxt_w = 0.f;
  int last_row = -1;  // KEEP: the row this group gathered last (where a dropped edge's load is parked)
  if (e_begin < e_end) {
    const int q = e_begin + glane < e_end ? e_begin + glane : e_begin;
    nxt_idx = fetch_id<KEEP>(indices, eid, first, keep, n_keep, q);
    if (W_LANE) {
      nxt_w = HAS_VALS ? vals[q] : 1.f;
      if (HAS_SS) nxt_w *= src_scale[(KEEP || MULT) ? nxt_idx & kIdMask : nxt_idx];
    }
  }
  for (int base = e_begin; base < e_end; base += LPR) {
    const int n = min(LPR, e_end - base);
    const int my_idx = nxt_idx;
    const float my_w = nxt_w;
    if (base + LPR < e_end) {
      const int nb = base + LPR;
      const int q = nb + glane < e_end ? nb + glane : nb;
      nxt_idx = fetch_id<KEEP>(indices, eid, first, keep, n_keep, q);
      if (W_LANE) {
        nxt_w = HAS_VALS ? vals[q] : 1.f;
        if (HAS_SS) nxt_w *= src_scale[(KEEP || MULT) ? nxt_idx & kIdMask : nxt_idx];
      }
    }
    for (int j = 0; j < n; j += kUnroll) {
      float4 v[kUnroll];
      float w[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int e = j + u;  // < LPR
        int idx = __shfl(my_idx, gbase + e, kWave);
        if (W_LANE) w[u] = __shfl(my_w, gbase + e, kWave);
        if (MULT) {  // the multiplicity arrived with the id
          const float m = (float)(((idx >> kMultShift) & kMultMax) + 1);
          w[u] = W_LANE ? w[u] * m : m;
        }
        const bool dropped = KEEP && idx < 0;
        if (KEEP) {
          idx = dropped && last_row >= 0 ? last_row : idx & kIdMask;
          last_row = idx;
        } else if (MULT) {
          idx &= kIdMask;
        }
        v[u] = ld_row<OFF32>(X, Xc, idx, ldx, row_bytes, col_bytes);
        if (dropped) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // fast path (group-uniform): all 8 edges belong to the current row -> balanced tree, no
      // per-edge boundary tests (a timing-only build that took it for EVERY batch gained 2-7 %: the slow path is not
      // what holds the kernel at 0.81 of the gather probe — profiles/r03_swept_experiment/README.md)
      if (base + j + kUnroll <= next_b) {
        if (WEIGHTED) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            v[u].x *= w[u];
            v[u].y *= w[u];
            v[u].z *= w[u];
            v[u].w *= w[u];
          }
        }
        tree_sum(v, kUnroll);
        acc.x += v[0].x;
        acc.y += v[0].y;
        acc.z += v[0].z;
        acc.w += v[0].w;
        continue;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int p = base + j + u;
        if (p < e_end) {  // group-uniform
          while (p >= next_b) {  // row(s) ended before this edge: emit them (empty rows emit zeros)
            if (col_ok) store_plane_row(prow + (int64_t)r * ldp, acc);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            ++r;
            next_b = __shfl(my_b, gbase + r + 1, kWave);
          }
          if (WEIGHTED) {
            acc.x = fmaf(w[u], v[u].x, acc.x);
            acc.y = fmaf(w[u], v[u].y, acc.y);
            acc.z = fmaf(w[u], v[u].z, acc.z);
            acc.w = fmaf(w[u], v[u].w, acc.w);
          } else {
            acc.x += v[u].x;
            acc.y += v[u].y;
            acc.z += v[u].z;
            acc.w += v[u].w;
          }
        }
      }
    }
  }
  for (; r < nr; ++r) {  // the last non-empty row, then any trailing empty rows
    if (col_ok) store_plane_row(prow + (int64_t)r * ldp, acc);
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Y[row] = dst_scale[row] * (plane_0[row] + plane_1[row] + ...) in slice order; one float4 per
// thread, all S plane loads of an element in flight together (S known at compile time for the
// usual 8 slices).  dst_scale and Y already point at the chunk's first row.
template <bool HAS_DS, int S>
__global__ __launch_bounds__(256) void reduce_planes_kernel(const float* __restrict__ planes, int64_t ldp,
                                                            int64_t rows, int F4, int n_slices,
                                                            const float* __restrict__ dst_scale,
                                                            float* __restrict__ Y, int64_t ldy, Epilogue ep) {
  const int64_t total = rows * F4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t plane_stride = rows * ldp;
  const bool dense = (ldp == 4 * (int64_t)F4) && (ldy == ldp);  // element t of a plane is at offset 4t
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    int64_t row, poff, yoff;
    if (dense) {
      row = t / F4;
      poff = yoff = 4 * t;
    } else {
      row = t / F4;
      const int c = (int)(t - row * F4) * 4;
      poff = row * ldp + c;
      yoff = row * ldy + c;
    }
    const float* p = planes + poff;
    float4 acc;
    if (S > 0) {
      float4 v[S > 0 ? S : 1];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p + s * plane_stride));
        v[s] = make_float4(t.x, t.y, t.z, t.w);
      }
      acc = v[0];
#pragma unroll
      for (int s = 1; s < S; ++s) {
        acc.x += v[s].x;
        acc.y += v[s].y;
        acc.z += v[s].z;
        acc.w += v[s].w;
      }
    } else {
      acc = *reinterpret_cast<const float4*>(p);
      for (int s = 1; s < n_slices; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p + s * plane_stride);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
    }
    if (HAS_DS) {
      const float d = dst_scale[row];
      acc.x *= d;
      acc.y *= d;
      acc.z *= d;
      acc.w *= d;
    }
    *reinterpret_cast<float4*>(Y + yoff) = epilogue4(ep, acc, row, (int)(yoff - row * ldy));
  }
}

template <int LPR>
hipError_t launch_sliced(const SlicedArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s) {
  constexpr int G = kWave / LPR;
  // Rows per lane group.  In the step (cold id stream, touch-ahead on), G edges/s at 4 / 6 / 8 / 12 / 15 rows: half-width
  // products 31.8 / 31.2 / 30.8 / 28.2 / 27.9, full-width ones 28.4 / 29.5 / 30.1 / 30.2 / 30.3, the step 30.8 / 31.0 / 31.0 /
  // 30.1 / 29.7.  One value for every width: where a group's run starts decides how its batches of 8 are cut, so a
  // width-dependent value would make the column passes round differently from the full-width pass (they are bit-identical,
  // test_xcd_sliced_column_passes).  Tuning::sliced_rows forces a value (tools).
  const Tuning& tune = tuning();
  const int rows_req = tune.sliced_rows > 0 ? tune.sliced_rows : kRowsPerGroup;
  const int R = rows_req < 1 ? 1 : (rows_req < LPR ? rows_req : LPR - 1);
  const int64_t per_block = (int64_t)kWavesPerBlock * G * R;
  const int64_t blocks = (row_end - row_begin + per_block - 1) / per_block;
  dim3 block(kWave * kWavesPerBlock);
  const int key = (a.vals ? 4 : (a.id_mult ? 8 : 0)) | (a.src_scale ? 2 : 0) | (a.n_keep > 0 ? 1 : 0);
  const bool off32 = !tune.sliced_no_off32 && (a.n_src * a.ldx + a.F) * 4 < ((int64_t)1 << 32);
  // Touch-ahead (see the kernel): one toucher per kTouchGroup worker blocks, kTouchLead worker blocks ahead.  An XCD starts
  // ~7 blocks of its slice per us, so 24 blocks are ~3.5 us of lead — a memory latency, and short enough for the touched
  // lines to still be in its L2.  Step of bench.py: no touching 2.724 ms; wave 0 of every block touching for the block 16 /
  // 24 / 32 further on 2.553 / 2.548 / 2.548; toucher blocks, one per 4 / 8 / 16 workers 2.526 / 2.528 / 2.530
  // (profiles/r03_touch_ahead/).  Tuning::sliced_touch_lead overrides the lead (tools/cold_ids_probe.py; 0 = no touchers).
  const int touch_lead = tune.sliced_touch_lead >= 0 ? tune.sliced_touch_lead : kTouchLead;
  const int touch_group = touch_lead > 0 ? kTouchGroup : 0;
  const int64_t touchers = touch_group > 0 ? (blocks + touch_group - 1) / touch_group : 0;
  dim3 grid((unsigned)((blocks + touchers) * a.n_slices), (unsigned)((a.F + 4 * LPR - 1) / (4 * LPR)));
#define DGMI_LAUNCH(V, S, K)                                                                                    \
  do {                                                                                                          \
    if (off32)                                                                                                  \
      hipLaunchKernelGGL((spmm_sliced_vec4_kernel<LPR, V, S, K, true>), grid, block, 0, s, a.segptr, a.indices,  \
                         a.vals, a.X, a.ldx, a.src_scale, a.planes, a.ldp, a.n_dst, row_begin, row_end,          \
                         (int)a.F, (int)a.n_slices, a.eid, static_cast<const KeepSeg*>(a.keep), a.n_keep,       \
                         touch_lead, R, touch_group);                                                             \
    else                                                                                                        \
      hipLaunchKernelGGL((spmm_sliced_vec4_kernel<LPR, V, S, K, false>), grid, block, 0, s, a.segptr, a.indices, \
                         a.vals, a.X, a.ldx, a.src_scale, a.planes, a.ldp, a.n_dst, row_begin, row_end,          \
                         (int)a.F, (int)a.n_slices, a.eid, static_cast<const KeepSeg*>(a.keep), a.n_keep,       \
                         touch_lead, R, touch_group);                                                             \
  } while (0)
  switch (key) {
    case 0: DGMI_LAUNCH(0, false, false); break;
    case 1: DGMI_LAUNCH(0, false, true); break;
    case 2: DGMI_LAUNCH(0, true, false); break;
    case 3: DGMI_LAUNCH(0, true, true); break;
    case 4: DGMI_LAUNCH(1, false, false); break;
    case 5: DGMI_LAUNCH(1, false, true); break;
    case 6: DGMI_LAUNCH(1, true, false); break;
    case 7: DGMI_LAUNCH(1, true, true); break;
    case 8: DGMI_LAUNCH(2, false, false); break;
    case 9: DGMI_LAUNCH(2, false, true); break;
    case 10: DGMI_LAUNCH(2, true, false); break;
    default: DGMI_LAUNCH(2, true, true); break;
  }
#undef DGMI_LAUNCH
  return hipGetLastError();
}

}  // namespace

hipError_t spmm_sliced_f32(const SlicedArgs& a, hipStream_t s) {
  if (a.n_dst == 0 || a.F == 0) return hipSuccess;
  const int F4 = (int)(a.F / 4);
  // Row chunks: the 8 partial planes of a chunk (chunk_rows * n_slices * 4F bytes, <= ~32 MB) are
  // written and read back while still resident in the 256 MiB Infinity Cache, and the same
  // plane buffer is reused by every chunk.
  const int64_t chunk = a.chunk_rows > 0 ? a.chunk_rows : a.n_dst;
  for (int64_t r0 = 0; r0 < a.n_dst; r0 += chunk) {
    const int64_t r1 = r0 + chunk < a.n_dst ? r0 + chunk : a.n_dst;
    hipError_t err;
    // Lane-group width = column tile.  Widest group whose last column tile is still >= 85 % used — unless
    // the slice of X one XCD gathers from (n_src / n_slices rows x 16 LPR bytes) is larger than its 4 MiB
    // L2: then half the width.  The column tiles are grid.y, dispatched one after the other, so the XCD
    // sweeps its slice twice at half the footprint.  Measured at F = 128: 100k-source table (6.4 -> 3.2 MB
    // per pass, bench.py) 0.379 -> 0.365 ms unweighted, 0.512 -> 0.464 ms kNN-64 weighted; config-5 shards
    // (tools/cfg5_forms_probe.py) 204 MB table 0.740 -> 0.663 ms, 409 MB table 0.741 -> 0.712 ms (the halves
    // of all 8 slices together fit the 256 MiB Infinity Cache; a quarter width gains nothing more).  A
    // 50k-source table (already 3.2 MB per slice) loses 10-18 % when halved, so the rule is tied to the
    // footprint; and a graph whose time is set by a few very long (virtual) rows pays their dependent gather
    // chain once per pass (Zipf(1.2) cut into 2048-edge virtual rows: 0.47 -> 0.62 ms; at the 512 edges
    // ops._SplitSliced uses the passes win again, 0.435 -> 0.418 ms): such a caller can ask for full width.
    // With edge dropout on the fly every pass re-evaluates keep(eid[p]) per edge (0.386 -> 0.408 ms): full width.
    // Half-width groups also mean half as many waves per pass (n_dst / 4 at F = 128): with few, long rows the
    // launch no longer fills the chip (config-5 edge-scaled shard, 6250 rows of 1600 edges: 0.382 -> 0.440 ms),
    // so the rule needs kColumnPassMinRows destination rows.
    // Tuning::sliced_lpr forces a width (tools).
    const int forced_lpr = tuning().sliced_lpr;
    int lpr = pick_lpr(a.F);
    if (lpr >= 32 && !a.full_width && a.n_keep == 0 && a.n_dst >= kColumnPassMinRows) {
      const int64_t width = 16 * (int64_t)lpr < 4 * a.F ? 16 * (int64_t)lpr : 4 * a.F;
      const int64_t slice_bytes = (a.n_src + a.n_slices - 1) / a.n_slices * width;
      if (slice_bytes > (4 << 20)) lpr /= 2;
    }
    if (forced_lpr == 8 || forced_lpr == 16 || forced_lpr == 32 || forced_lpr == 64) lpr = forced_lpr;
    switch (lpr) {
      case 8: err = launch_sliced<8>(a, r0, r1, s); break;
      case 16: err = launch_sliced<16>(a, r0, r1, s); break;
      case 32: err = launch_sliced<32>(a, r0, r1, s); break;
      default: err = launch_sliced<64>(a, r0, r1, s); break;
    }
    if (err != hipSuccess) return err;
    int64_t blocks = ((r1 - r0) * F4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const float* ds = a.dst_scale ? a.dst_scale + r0 : nullptr;
    float* y = a.Y + r0 * a.ldy;
    Epilogue ep = a.ep;  // rows of this chunk start at r0
    if (ep.mask != nullptr) ep.mask += r0 * ep.ldm;
#define DGMI_REDUCE(D, S)                                                                               \
  hipLaunchKernelGGL((reduce_planes_kernel<D, S>), dim3((unsigned)blocks), dim3(256), 0, s, a.planes, a.ldp, \
                     r1 - r0, F4, (int)a.n_slices, ds, y, a.ldy, ep)
    if (a.n_slices == 8) {
      if (ds) DGMI_REDUCE(true, 8); else DGMI_REDUCE(false, 8);
    } else {
      if (ds) DGMI_REDUCE(true, 0); else DGMI_REDUCE(false, 0);
    }
#undef DGMI_REDUCE
    err = hipGetLastError();
    if (err != hipSuccess) return err;
  }
  return hipSuccess;
}

}  // namespace dgmi
