// dgmi_kernels.h — internal launch interface between the C ABI (dgmi_api.hip)
// and the kernel translation units.  Not installed; the public contract is
// include/dgmi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dgmi {

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
};

// Y = diag(dst_scale) A diag(src_scale) X   (dgmi_spmm.hip)
hipError_t spmm_csr_f32(const SpmmArgs& a, hipStream_t s);

// Stable COO->CSR (dgmi_csr.hip).  workspace == nullptr: size query only.
hipError_t csr_from_coo_i32(const int32_t* row, const int32_t* col, int64_t E,
                            int64_t n_rows, int32_t* indptr, int32_t* indices,
                            int32_t* eid, void* workspace, size_t* workspace_bytes,
                            hipStream_t s);

hipError_t gather_f32(const float* in, const int32_t* perm, int64_t n, float* out,
                      hipStream_t s);

}  // namespace dgmi
