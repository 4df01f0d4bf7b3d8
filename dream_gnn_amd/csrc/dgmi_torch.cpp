// dgmi_torch.cpp — the torch operator layer of SURVEY.md §8(b): `dreamgnn_mi::*` dispatcher ops over
// the C ABI of libdgmi.so (include/dgmi.h).  Built for ROCm only (ROCm torch's HIP stream / device guard —
// its tensors carry the device type "cuda", hence the *MasqueradingAsCUDA names; there is no CUDA branch
// and no CPU kernel: the product path has no fallback).
//
// What they replace inside the dispatcher:
//   dreamgnn_mi::spmm_csr      graph.update_all(fn.copy_u('h','m'), fn.sum('m','h'))  (reference layers.py:229-232,
//                              with the cj / ci scalings of :224-225,234 fused) and th.spmm(adj, support)
//                              (layers.py:312); autograd registered: dX = diag(ss) A^T diag(ds) dY through the
//                              same kernels on the transposed CSR.
//   dreamgnn_mi::csr_from_coo  DGL's COO->CSR behind dgl.heterograph (data_loader.py:448, augmentation.py:65)
//   the *_raw ops              one C-ABI entry point each, no autograd: what dream_gnn_amd/ops.py (layout caches,
//                              kernel choice, graph-level autograd) is written in.
// Every op validates dtype / device / contiguity with TORCH_CHECK (-> RuntimeError), makes the inputs'
// device current, takes torch's current HIP stream, allocates outputs from the caching allocator and
// kernel scratch from a per-(device, stream) cache, and never synchronises.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <mutex>
#include <tuple>
#include <unordered_map>

#include "dgmi.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

void check_status(int rc, const char* what) {
  TORCH_CHECK(rc == DGMI_OK, what, " failed: ", dgmi_status_string(rc), " (status ", rc, ")");
}

void check_dev(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "dream_gnn_amd ops run on the MI355X only: ", name, " is a ", t.device().str(),
              " tensor. There is no CPU path (the CPU restatement under oracle/ is test infrastructure).");
}

void check(const Tensor& t, at::ScalarType dt, int64_t dim, const char* name, const Tensor& like) {
  check_dev(t, name);
  TORCH_CHECK(t.scalar_type() == dt, name, " must be ", c10::toString(dt), ", got ", c10::toString(t.scalar_type()));
  TORCH_CHECK(t.dim() == dim, name, " must be ", dim, "-D, got ", t.dim(), "-D");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK(t.device() == like.device(), "tensors on different devices: ", name, " on ", t.device().str(), " vs ",
              like.device().str());
}

const void* optptr(const OptTensor& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

void check_opt(const OptTensor& t, at::ScalarType dt, int64_t n, const char* name, const Tensor& like) {
  if (!t.has_value() || !t->defined()) return;
  check(*t, dt, 1, name, like);
  TORCH_CHECK(n < 0 || t->numel() == n, name, " has ", t->numel(), " entries, expected ", n);
}

dgmi_stream_t stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// Kernel scratch (chunk partials, partial planes, builder workspaces): one buffer per (device, stream,
// kind), grown on demand.  Products issued on one stream are ordered, so they can share it; nothing is
// allocated per call and contents are never read across calls.
Tensor scratch(const Tensor& like, size_t nbytes, int kind) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, Tensor> cache;
  const auto s = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(like.device().index());
  const uint64_t key = (reinterpret_cast<uint64_t>(s.stream()) * 31u + (uint64_t)like.device().index()) * 8u + (uint64_t)kind;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end() || (size_t)it->second.numel() < nbytes) {
    Tensor buf = at::empty({(int64_t)(nbytes < 16 ? 16 : nbytes)}, like.options().dtype(at::kByte));
    cache[key] = buf;
    return buf;
  }
  return it->second;
}
enum { kPartials = 0, kPlanes = 1, kBuilder = 2 };

struct Keep {
  const int32_t* eid = nullptr;
  const uint32_t* table = nullptr;
  int32_t n = 0;
};

Keep keep_of(const OptTensor& eid, const OptTensor& keep, int64_t nnz, const Tensor& like) {
  Keep k;
  if (!keep.has_value() || !keep->defined()) return k;
  TORCH_CHECK(keep->scalar_type() == at::kInt && keep->is_contiguous() && keep->dim() == 2 && keep->size(1) == 8,
              "keep must be a contiguous (n, 8) int32 tensor of subset descriptions");
  TORCH_CHECK(keep->size(0) <= 8, "at most 8 subset descriptions per product");
  TORCH_CHECK(eid.has_value() && eid->defined(), "edge dropout on the fly needs the layout's eid array");
  check(*eid, at::kInt, 1, "eid", like);
  check_dev(*keep, "keep");
  TORCH_CHECK(eid->numel() == nnz, "eid has ", eid->numel(), " entries, the layout has ", nnz, " edges");
  k.eid = eid->data_ptr<int32_t>();
  k.table = reinterpret_cast<const uint32_t*>(keep->data_ptr<int32_t>());
  k.n = (int32_t)keep->size(0);
  return k;
}

// output epilogue: Y = out_mask * mask_scale * act(dst_scale * sum); act 0 none, 1 leaky-relu(slope)
struct Epi {
  int32_t act = 0;
  float slope = 0.f;
  const float* mask = nullptr;
  int64_t ldm = 0;
  float mscale = 1.f;
};

Epi epi_of(int64_t act, double slope, const OptTensor& out_mask, double mask_scale, int64_t rows, int64_t F, const Tensor& like) {
  Epi e;
  TORCH_CHECK(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-relu)");
  e.act = (int32_t)act;
  e.slope = (float)slope;
  e.mscale = (float)mask_scale;
  if (out_mask.has_value() && out_mask->defined()) {
    check_dev(*out_mask, "out_mask");
    TORCH_CHECK(out_mask->scalar_type() == at::kFloat && out_mask->dim() == 2 && out_mask->size(0) == rows &&
                    out_mask->size(1) == F && out_mask->stride(1) == 1 && out_mask->device() == like.device(),
                "out_mask must be a float32 (", rows, ", ", F, ") tensor with contiguous rows on ", like.device().str());
    e.mask = out_mask->data_ptr<float>();
    e.ldm = rows > 1 ? out_mask->stride(0) : (F > 0 ? F : 1);
  }
  return e;
}

struct Dense {
  Tensor t;
  int64_t rows, F, ld;
};

// a 2-D fp32 matrix whose rows are contiguous (a row-strided view is consumed in place)
Dense dense_of(const Tensor& X, const char* name) {
  check_dev(X, name);
  TORCH_CHECK(X.scalar_type() == at::kFloat && X.dim() == 2, name, " must be a 2-D float32 tensor");
  Tensor t = X;
  if (X.stride(1) != 1 || (X.size(0) > 1 && X.stride(0) < X.size(1))) t = X.contiguous();
  const int64_t F = t.size(1);
  return {t, t.size(0), F, t.size(0) > 1 ? t.stride(0) : (F > 0 ? F : 1)};
}

Tensor out_of(const OptTensor& out, int64_t rows, int64_t F, const Tensor& like) {
  if (out.has_value() && out->defined()) {
    TORCH_CHECK(out->scalar_type() == at::kFloat && out->dim() == 2 && out->size(0) == rows && out->size(1) == F &&
                    out->is_contiguous() && out->device() == like.device(),
                "out must be a contiguous float32 (", rows, ", ", F, ") tensor on ", like.device().str());
    return *out;
  }
  return at::empty({rows, F}, like.options().dtype(at::kFloat));
}

// ---------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor> csr_from_coo(const Tensor& row, const Tensor& col, int64_t n_rows, int64_t n_cols) {
  check(row, at::kInt, 1, "row", row);
  check(col, at::kInt, 1, "col", row);
  TORCH_CHECK(row.numel() == col.numel(), "row/col length mismatch");
  TORCH_CHECK(n_rows >= 0, "n_rows must be >= 0");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(row.device());
  const int64_t E = row.numel();
  Tensor indptr = at::empty({n_rows + 1}, row.options()), indices = at::empty({E}, row.options()),
         eid = at::empty({E}, row.options());
  size_t need = 0;
  check_status(dgmi_csr_from_coo_i32(row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), E, n_rows, n_cols, nullptr, nullptr,
                                     nullptr, nullptr, &need, nullptr), "dgmi_csr_from_coo_i32(size query)");
  Tensor ws = scratch(row, need < 256 ? 256 : need, kBuilder);
  size_t have = (size_t)ws.numel();
  check_status(dgmi_csr_from_coo_i32(row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), E, n_rows, n_cols,
                                     indptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), eid.data_ptr<int32_t>(),
                                     ws.data_ptr(), &have, stream_of(row)), "dgmi_csr_from_coo_i32");
  // the range flag leaves the shared workspace before the next builder call overwrites it
  Tensor flag = ws.narrow(0, 0, 4).view(at::kInt).clone();
  return {indptr, indices, eid, flag};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> csr_sliced_from_coo(const Tensor& row, const Tensor& col, int64_t n_rows,
                                                               int64_t n_cols, int64_t n_slices) {
  check(row, at::kInt, 1, "row", row);
  check(col, at::kInt, 1, "col", row);
  TORCH_CHECK(row.numel() == col.numel(), "row/col length mismatch");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(row.device());
  const int64_t E = row.numel();
  Tensor segptr = at::empty({n_slices * n_rows + 1}, row.options()), indices = at::empty({E}, row.options()),
         eid = at::empty({E}, row.options());
  size_t need = 0;
  check_status(dgmi_csr_sliced_from_coo_i32(row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), E, n_rows, n_cols,
                                            (int32_t)n_slices, nullptr, nullptr, nullptr, nullptr, &need, nullptr),
               "dgmi_csr_sliced_from_coo_i32(size query)");
  Tensor ws = scratch(row, need < 256 ? 256 : need, kBuilder);
  size_t have = (size_t)ws.numel();
  check_status(dgmi_csr_sliced_from_coo_i32(row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), E, n_rows, n_cols,
                                            (int32_t)n_slices, segptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(),
                                            eid.data_ptr<int32_t>(), ws.data_ptr(), &have, stream_of(row)),
               "dgmi_csr_sliced_from_coo_i32");
  Tensor flag = ws.narrow(0, 0, 4).view(at::kInt).clone();
  return {segptr, indices, eid, flag};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> csr_sliced_from_csr(const Tensor& indptr, const Tensor& indices, const Tensor& eid,
                                                               int64_t n_cols, int64_t n_slices) {
  check(indptr, at::kInt, 1, "indptr", indptr);
  check(indices, at::kInt, 1, "indices", indptr);
  check(eid, at::kInt, 1, "eid", indptr);
  TORCH_CHECK(indices.numel() == eid.numel(), "indices/eid length mismatch");
  TORCH_CHECK(indptr.numel() >= 1, "indptr is empty");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indptr.device());
  const int64_t E = indices.numel(), n_rows = indptr.numel() - 1;
  Tensor segptr = at::empty({n_slices * n_rows + 1}, indptr.options()), s_indices = at::empty({E}, indptr.options()),
         s_eid = at::empty({E}, indptr.options());
  size_t need = 0;
  check_status(dgmi_csr_sliced_from_csr_i32(nullptr, nullptr, nullptr, E, n_rows, n_cols, (int32_t)n_slices, nullptr, nullptr,
                                            nullptr, nullptr, &need, nullptr),
               "dgmi_csr_sliced_from_csr_i32(size query)");
  Tensor ws = scratch(indptr, need < 256 ? 256 : need, kBuilder);
  size_t have = (size_t)ws.numel();
  check_status(dgmi_csr_sliced_from_csr_i32(indptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), eid.data_ptr<int32_t>(), E,
                                            n_rows, n_cols, (int32_t)n_slices, segptr.data_ptr<int32_t>(),
                                            s_indices.data_ptr<int32_t>(), s_eid.data_ptr<int32_t>(), ws.data_ptr(), &have,
                                            stream_of(indptr)),
               "dgmi_csr_sliced_from_csr_i32");
  Tensor flag = ws.narrow(0, 0, 4).view(at::kInt).clone();
  return {segptr, s_indices, s_eid, flag};
}

Tensor plan_build(const Tensor& indptr, int64_t nnz, int64_t chunk) {
  check(indptr, at::kInt, 1, "indptr", indptr);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indptr.device());
  const int64_t n_rows = indptr.numel() - 1;
  const size_t nbytes = dgmi_spmm_plan_bytes(n_rows, nnz, (int32_t)chunk);
  TORCH_CHECK(nbytes > 0, "invalid plan parameters (chunk=", chunk, ")");
  Tensor plan = at::empty({(int64_t)nbytes}, indptr.options().dtype(at::kByte));
  size_t need = 0;
  check_status(dgmi_spmm_plan_build(indptr.data_ptr<int32_t>(), n_rows, nnz, (int32_t)chunk, nullptr, 0, nullptr, &need, nullptr),
               "dgmi_spmm_plan_build(size query)");
  Tensor ws = scratch(indptr, need < 256 ? 256 : need, kBuilder);
  size_t have = (size_t)ws.numel();
  check_status(dgmi_spmm_plan_build(indptr.data_ptr<int32_t>(), n_rows, nnz, (int32_t)chunk, plan.data_ptr(), nbytes,
                                    ws.data_ptr(), &have, stream_of(indptr)), "dgmi_spmm_plan_build");
  return plan;
}

// plan undefined: dgmi_spmm_csr_f32 (a wave per row); else dgmi_spmm_csr_planned_f32
Tensor spmm_csr_raw(const Tensor& indptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid,
                    const OptTensor& keep, const Tensor& X, const OptTensor& src_scale, const OptTensor& dst_scale,
                    const OptTensor& plan, int64_t chunk, const OptTensor& out, int64_t act = 0, double slope = 0.0,
                    const OptTensor& out_mask = c10::nullopt, double mask_scale = 1.0) {
  check(indptr, at::kInt, 1, "indptr", indptr);
  check(indices, at::kInt, 1, "indices", indptr);
  Dense x = dense_of(X, "X");
  TORCH_CHECK(x.t.device() == indptr.device(), "X and the graph are on different devices");
  const int64_t n_dst = indptr.numel() - 1, nnz = indices.numel();
  check_opt(vals, at::kFloat, nnz, "vals", indptr);
  check_opt(src_scale, at::kFloat, x.rows, "src_scale", indptr);
  check_opt(dst_scale, at::kFloat, n_dst, "dst_scale", indptr);
  const Keep k = keep_of(eid, keep, nnz, indptr);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indptr.device());
  Tensor y = out_of(out, n_dst, x.F, indptr);
  const int64_t ldy = x.F > 0 ? x.F : 1;
  const Epi e = epi_of(act, slope, out_mask, mask_scale, n_dst, x.F, indptr);
  if (!plan.has_value() || !plan->defined()) {
    check_status(dgmi_spmm_csr_f32(indptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), (const float*)optptr(vals), k.eid,
                                   k.table, k.n, x.t.data_ptr<float>(), x.ld, (const float*)optptr(src_scale),
                                   (const float*)optptr(dst_scale), y.data_ptr<float>(), ldy, n_dst, x.rows, x.F, e.act,
                                   e.slope, e.mask, e.ldm, e.mscale, stream_of(indptr)), "dgmi_spmm_csr_f32");
    return y;
  }
  const size_t pbytes = dgmi_spmm_partials_bytes(nnz, (int32_t)chunk, x.F);
  Tensor partials = scratch(indptr, pbytes, kPartials);
  check_status(dgmi_spmm_csr_planned_f32(indptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), (const float*)optptr(vals),
                                         k.eid, k.table, k.n, x.t.data_ptr<float>(), x.ld, (const float*)optptr(src_scale),
                                         (const float*)optptr(dst_scale), y.data_ptr<float>(), ldy, n_dst, x.rows, x.F, nnz,
                                         (int32_t)chunk, plan->data_ptr(), partials.data_ptr(), pbytes, e.act, e.slope, e.mask,
                                         e.ldm, e.mscale, stream_of(indptr)),
               "dgmi_spmm_csr_planned_f32");
  return y;
}

Tensor spmm_sliced_raw(const Tensor& segptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid,
                       const OptTensor& keep, const Tensor& X, const OptTensor& src_scale, const OptTensor& dst_scale,
                       int64_t n_dst, int64_t n_slices, const OptTensor& out, int64_t act = 0, double slope = 0.0,
                       const OptTensor& out_mask = c10::nullopt, double mask_scale = 1.0, int64_t column_passes = 0,
                       int64_t id_mult = 0) {
  check(segptr, at::kInt, 1, "segptr", segptr);
  check(indices, at::kInt, 1, "indices", segptr);
  TORCH_CHECK(segptr.numel() == n_slices * n_dst + 1, "segptr has ", segptr.numel(), " entries, expected n_slices * n_dst + 1");
  Dense x = dense_of(X, "X");
  TORCH_CHECK(x.t.device() == segptr.device(), "X and the graph are on different devices");
  const int64_t nnz = indices.numel();
  check_opt(vals, at::kFloat, nnz, "vals", segptr);
  check_opt(src_scale, at::kFloat, x.rows, "src_scale", segptr);
  check_opt(dst_scale, at::kFloat, n_dst, "dst_scale", segptr);
  const Keep k = keep_of(eid, keep, nnz, segptr);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(segptr.device());
  Tensor y = out_of(out, n_dst, x.F, segptr);
  const Epi e = epi_of(act, slope, out_mask, mask_scale, n_dst, x.F, segptr);
  const size_t pbytes = dgmi_spmm_sliced_planes_bytes(n_dst, (int32_t)n_slices, x.F);
  Tensor planes = scratch(segptr, pbytes, kPlanes);
  check_status(dgmi_spmm_sliced_f32(segptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), (const float*)optptr(vals), k.eid,
                                    k.table, k.n, x.t.data_ptr<float>(), x.ld, (const float*)optptr(src_scale),
                                    (const float*)optptr(dst_scale), y.data_ptr<float>(), x.F, n_dst, x.rows, x.F,
                                    (int32_t)n_slices, (int32_t)column_passes, (int32_t)id_mult, planes.data_ptr(), pbytes, e.act,
                                    e.slope, e.mask,
                                    e.ldm, e.mscale, stream_of(segptr)), "dgmi_spmm_sliced_f32");
  return y;
}

Tensor gather_f32(const Tensor& values, const Tensor& perm) {
  check(values, at::kFloat, 1, "values", values);
  check(perm, at::kInt, 1, "perm", values);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(values.device());
  Tensor out = at::empty({perm.numel()}, values.options());
  check_status(dgmi_gather_f32(values.data_ptr<float>(), perm.data_ptr<int32_t>(), perm.numel(), out.data_ptr<float>(),
                               stream_of(values)), "dgmi_gather_f32");
  return out;
}

Tensor gather_concat_raw(const Tensor& src, const Tensor& dst, const Tensor& A, const Tensor& B) {
  check(src, at::kInt, 1, "src", src);
  check(dst, at::kInt, 1, "dst", src);
  TORCH_CHECK(src.numel() == dst.numel(), "src/dst length mismatch");
  Dense a = dense_of(A, "A"), b = dense_of(B, "B");
  TORCH_CHECK(a.t.device() == src.device() && b.t.device() == src.device(), "tensors on different devices");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  const int64_t E = src.numel(), W = a.F + b.F;
  Tensor out = at::empty({E, W}, a.t.options());
  check_status(dgmi_gather_concat_f32(src.data_ptr<int32_t>(), dst.data_ptr<int32_t>(), E, a.t.data_ptr<float>(), a.ld, a.F,
                                      b.t.data_ptr<float>(), b.ld, b.F, out.data_ptr<float>(), W > 0 ? W : 1, stream_of(src)),
               "dgmi_gather_concat_f32");
  return out;
}

Tensor gather_add_raw(const Tensor& src, const Tensor& dst, const Tensor& A, const Tensor& B, const OptTensor& bias, int64_t act) {
  TORCH_CHECK(act == 0 || act == 1, "act must be 0 (none) or 1 (relu)");
  check(src, at::kInt, 1, "src", src);
  check(dst, at::kInt, 1, "dst", src);
  TORCH_CHECK(src.numel() == dst.numel(), "src/dst length mismatch");
  Dense a = dense_of(A, "A"), b = dense_of(B, "B");
  TORCH_CHECK(a.F == b.F, "A and B must have the same width, got ", a.F, " and ", b.F);
  TORCH_CHECK(a.t.device() == src.device() && b.t.device() == src.device(), "tensors on different devices");
  check_opt(bias, at::kFloat, a.F, "bias", src);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  const int64_t E = src.numel();
  Tensor out = at::empty({E, a.F}, a.t.options());
  check_status(dgmi_gather_add_f32(src.data_ptr<int32_t>(), dst.data_ptr<int32_t>(), E, a.t.data_ptr<float>(), a.ld,
                                   b.t.data_ptr<float>(), b.ld, (const float*)optptr(bias), a.F, out.data_ptr<float>(),
                                   a.F > 0 ? a.F : 1, (int32_t)act, stream_of(src)), "dgmi_gather_add_f32");
  return out;
}

Tensor epilogue_backward(const Tensor& dY, const Tensor& Y, const OptTensor& mask, int64_t act, double slope, double mask_scale) {
  check_dev(dY, "dY");
  TORCH_CHECK(dY.scalar_type() == at::kFloat && dY.is_contiguous(), "dY must be contiguous float32");
  TORCH_CHECK(Y.scalar_type() == at::kFloat && Y.is_contiguous() && Y.numel() == dY.numel() && Y.device() == dY.device(),
              "Y must be contiguous float32 of dY's size, on its device");
  if (mask.has_value() && mask->defined())
    TORCH_CHECK(mask->scalar_type() == at::kFloat && mask->is_contiguous() && mask->numel() == dY.numel() &&
                    mask->device() == dY.device(), "mask must be contiguous float32 of dY's size, on its device");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dY.device());
  Tensor out = at::empty_like(dY);
  check_status(dgmi_epilogue_backward_f32(dY.data_ptr<float>(), Y.data_ptr<float>(), (const float*)optptr(mask), dY.numel(),
                                          (int32_t)act, (float)slope, (float)mask_scale, out.data_ptr<float>(), stream_of(dY)),
               "dgmi_epilogue_backward_f32");
  return out;
}

// diag(scale) X in one streaming pass (ahead of an XCD-local product, see include/dgmi.h)
Tensor scale_rows(const Tensor& X, const Tensor& scale) {
  Dense x = dense_of(X, "X");
  check(scale, at::kFloat, 1, "scale", x.t);
  TORCH_CHECK(scale.numel() == x.rows, "scale has ", scale.numel(), " entries, X has ", x.rows, " rows");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.t.device());
  Tensor out = at::empty({x.rows, x.F}, x.t.options());
  check_status(dgmi_scale_rows_f32(x.t.data_ptr<float>(), x.ld, scale.data_ptr<float>(), x.rows, x.F, out.data_ptr<float>(), x.F,
                                   stream_of(x.t)), "dgmi_scale_rows_f32");
  return out;
}

// (f3 complement form) rows [n*R, n*R + B) of `feat_ext` <- coef @ feat_ext[u*R + i0, :] (column sums per block), in place
void colsum_rows_(Tensor feat_ext, const Tensor& coef, int64_t n, int64_t R, int64_t i0) {
  check(feat_ext, at::kFloat, 2, "feat_ext", feat_ext);
  check(coef, at::kFloat, 2, "coef", feat_ext);
  const int64_t B = coef.size(0), W = feat_ext.size(1);
  TORCH_CHECK(coef.size(1) == n && feat_ext.size(0) == n * R + B && i0 >= 0 && i0 < R && B <= 64,
              "colsum_rows_: feat_ext must have n*R + B rows, coef (B, n)");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(feat_ext.device());
  float* base = feat_ext.data_ptr<float>();
  check_status(dgmi_weighted_colsum_f32(base + i0 * W, R * W, coef.data_ptr<float>(), n, n, W, (int32_t)B, base + n * R * W, W,
                                        stream_of(feat_ext)), "dgmi_weighted_colsum_f32");
}

// its backward: gf[u*R + i0, :] += coef[:, u]^T @ gs, in place on the (n*R, W) gradient of the transform
void colsum_rows_backward_(Tensor gf, const Tensor& coef, const Tensor& gs, int64_t n, int64_t R, int64_t i0) {
  check(gf, at::kFloat, 2, "gf", gf);
  check(coef, at::kFloat, 2, "coef", gf);
  check(gs, at::kFloat, 2, "gs", gf);
  const int64_t B = coef.size(0), W = gf.size(1);
  TORCH_CHECK(coef.size(1) == n && gf.size(0) == n * R && gs.size(0) == B && gs.size(1) == W && i0 >= 0 && i0 < R && B <= 64,
              "colsum_rows_backward_: shape mismatch");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(gf.device());
  check_status(dgmi_rank_add_f32(gf.data_ptr<float>() + i0 * W, R * W, coef.data_ptr<float>(), n, gs.data_ptr<float>(), W, n, W,
                                 (int32_t)B, stream_of(gf)), "dgmi_rank_add_f32");
}

// (f4) (N, k) int32 neighbours of the k largest cosine similarities per row of a row-normalised matrix
Tensor knn_cosine_topk(const Tensor& Xn, int64_t k) {
  Dense x = dense_of(Xn, "Xn");
  TORCH_CHECK(x.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x.t.data_ptr()) & 15) == 0, "Xn rows must be 16-B aligned");
  TORCH_CHECK(dgmi_knn_cosine_supported(x.rows, x.F, k) == 1, "shape not supported by the fused kNN kernel: N=", x.rows,
              " D=", x.F, " k=", k);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.t.device());
  Tensor nbr = at::empty({x.rows, k}, x.t.options().dtype(at::kInt));
  const size_t wbytes = dgmi_knn_cosine_workspace_bytes(x.rows, x.F, (int32_t)k);
  // per call, from the caching allocator (reused on the same stream, returned by empty_cache()): the screened search
  // needs 1-3.5 GB at N = 100 000, which a one-off graph construction must not pin for the life of the process
  Tensor ws = at::empty({(int64_t)(wbytes < 16 ? 16 : wbytes)}, x.t.options().dtype(at::kByte));
  check_status(dgmi_knn_cosine_topk_f32(x.t.data_ptr<float>(), x.ld, x.rows, x.F, (int32_t)k, nbr.data_ptr<int32_t>(),
                                        ws.data_ptr(), (size_t)ws.numel(), stream_of(x.t)), "dgmi_knn_cosine_topk_f32");
  return nbr;
}

// `like`: any tensor on the target device (the op needs a device to allocate on)
Tensor random_subset_select(const Tensor& like, int64_t E, int64_t keep, int64_t seed, int64_t e_offset) {
  check_dev(like, "like");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  Tensor desc = at::empty({8}, like.options().dtype(at::kInt));
  const size_t nbytes = dgmi_random_subset_workspace_bytes();
  Tensor ws = at::empty({(int64_t)nbytes}, like.options().dtype(at::kByte));  // own buffer: selections of one step overlap
  check_status(dgmi_random_subset_select(E, keep, (uint64_t)seed, (uint32_t)e_offset,
                                         reinterpret_cast<uint32_t*>(desc.data_ptr<int32_t>()), ws.data_ptr(), nbytes,
                                         stream_of(like)), "dgmi_random_subset_select");
  return desc;
}

// n <= 8 subsets with one series of launches; returns (n, 8) descriptions
Tensor random_subset_select_batch(const Tensor& like, at::IntArrayRef E, at::IntArrayRef keep, at::IntArrayRef seed,
                                  at::IntArrayRef e_offset) {
  check_dev(like, "like");
  const int64_t n = (int64_t)E.size();
  TORCH_CHECK(n >= 1 && n <= 8 && (int64_t)keep.size() == n && (int64_t)seed.size() == n &&
                  (e_offset.empty() || (int64_t)e_offset.size() == n),
              "random_subset_select_batch: 1..8 subsets, E / keep / seed (/ e_offset) of equal length");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  Tensor descs = at::empty({n, 8}, like.options().dtype(at::kInt));
  const size_t nbytes = dgmi_random_subset_workspace_bytes();
  Tensor ws = at::empty({(int64_t)nbytes}, like.options().dtype(at::kByte));
  uint64_t seeds[8];
  uint32_t offs[8];
  for (int64_t i = 0; i < n; ++i) {
    seeds[i] = (uint64_t)seed[i];
    offs[i] = e_offset.empty() ? 0u : (uint32_t)e_offset[i];
  }
  check_status(dgmi_random_subset_select_batch((int32_t)n, E.data(), keep.data(), seeds, offs,
                                               reinterpret_cast<uint32_t*>(descs.data_ptr<int32_t>()), ws.data_ptr(), nbytes,
                                               stream_of(like)), "dgmi_random_subset_select_batch");
  return descs;
}

// the same with the seeds in a device tensor (int64, n entries): nothing host-side goes into the launches but the
// list lengths, so the call can sit inside a captured HIP graph and still select new subsets on every replay
Tensor random_subset_select_batch_dseed(const Tensor& seeds, at::IntArrayRef E, at::IntArrayRef keep, at::IntArrayRef e_offset) {
  check_dev(seeds, "seeds");
  const int64_t n = (int64_t)E.size();
  TORCH_CHECK(n >= 1 && n <= 8 && (int64_t)keep.size() == n && (e_offset.empty() || (int64_t)e_offset.size() == n),
              "random_subset_select_batch_dseed: 1..8 subsets, E / keep (/ e_offset) of equal length");
  TORCH_CHECK(seeds.scalar_type() == at::kLong && seeds.is_contiguous() && seeds.numel() == n,
              "seeds must be a contiguous int64 tensor with one entry per subset");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(seeds.device());
  Tensor descs = at::empty({n, 8}, seeds.options().dtype(at::kInt));
  const size_t nbytes = dgmi_random_subset_workspace_bytes();
  Tensor ws = at::empty({(int64_t)nbytes}, seeds.options().dtype(at::kByte));
  uint32_t offs[8];
  for (int64_t i = 0; i < n; ++i) offs[i] = e_offset.empty() ? 0u : (uint32_t)e_offset[i];
  check_status(dgmi_random_subset_select_batch_dseed((int32_t)n, E.data(), keep.data(),
                                                     reinterpret_cast<const uint64_t*>(seeds.data_ptr<int64_t>()), offs,
                                                     reinterpret_cast<uint32_t*>(descs.data_ptr<int32_t>()), ws.data_ptr(), nbytes,
                                                     stream_of(seeds)), "dgmi_random_subset_select_batch_dseed");
  return descs;
}

Tensor keep_mask(const Tensor& keep, int64_t E) {
  check_dev(keep, "keep");
  TORCH_CHECK(keep.scalar_type() == at::kInt && keep.is_contiguous() && keep.dim() == 2 && keep.size(1) == 8 && keep.size(0) <= 8,
              "keep must be a contiguous (n <= 8, 8) int32 tensor of subset descriptions");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(keep.device());
  Tensor mask = at::empty({E}, keep.options().dtype(at::kFloat));
  check_status(dgmi_keep_mask_f32(reinterpret_cast<const uint32_t*>(keep.data_ptr<int32_t>()), (int32_t)keep.size(0), E,
                                  mask.data_ptr<float>(), stream_of(keep)), "dgmi_keep_mask_f32");
  return mask;
}

// (D3) the layout with the edges dropped under `keep` removed: (ptr_out, indices_out[, vals_out]); indices_out / vals_out
// keep the parent's length (the survivors fill a prefix: nothing is read back to size them)
std::tuple<Tensor, Tensor, Tensor> compact_layout(const Tensor& ptr, const Tensor& indices, const OptTensor& vals, const Tensor& eid,
                                                  const Tensor& keep) {
  check(ptr, at::kInt, 1, "ptr", ptr);
  check(indices, at::kInt, 1, "indices", ptr);
  const int64_t nnz = indices.numel();
  check_opt(vals, at::kFloat, nnz, "vals", ptr);
  const Keep k = keep_of(eid, keep, nnz, ptr);
  TORCH_CHECK(k.n > 0, "compact_layout needs at least one subset description");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(ptr.device());
  Tensor ptr_out = at::empty_like(ptr), indices_out = at::empty_like(indices);
  const bool has_vals = vals.has_value() && vals->defined();
  Tensor vals_out = has_vals ? at::empty_like(*vals) : Tensor();
  const size_t wbytes = dgmi_compact_layout_workspace_bytes(nnz);
  Tensor ws = scratch(ptr, wbytes < 256 ? 256 : wbytes, kBuilder);
  check_status(dgmi_compact_layout_i32(ptr.data_ptr<int32_t>(), ptr.numel(), indices.data_ptr<int32_t>(), (const float*)optptr(vals),
                                       k.eid, nnz, k.table, k.n, ptr_out.data_ptr<int32_t>(), indices_out.data_ptr<int32_t>(),
                                       has_vals ? vals_out.data_ptr<float>() : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                       stream_of(ptr)), "dgmi_compact_layout_i32");
  return {ptr_out, indices_out, has_vals ? vals_out : at::empty({0}, ptr.options().dtype(at::kFloat))};
}

// (D2) row scale x multiplicity form of a weighted CSR: (row_scale[n_rows], mult[nnz] = m - 1, fail[1])
std::tuple<Tensor, Tensor, Tensor> row_multiplicity(const Tensor& indptr, const Tensor& vals, double rel_tol) {
  check(indptr, at::kInt, 1, "indptr", indptr);
  check(vals, at::kFloat, 1, "vals", indptr);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indptr.device());
  const int64_t n_rows = indptr.numel() - 1, nnz = vals.numel();
  Tensor scale = at::empty({n_rows}, vals.options()), mult = at::empty({nnz}, indptr.options()), fail = at::empty({1}, indptr.options());
  check_status(dgmi_row_multiplicity_f32(indptr.data_ptr<int32_t>(), vals.data_ptr<float>(), n_rows, nnz, (float)rel_tol,
                                         scale.data_ptr<float>(), mult.data_ptr<int32_t>(), fail.data_ptr<int32_t>(), stream_of(indptr)),
               "dgmi_row_multiplicity_f32");
  return {scale, mult, fail};
}

Tensor spmm_csr_new(const Tensor& indptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid, const OptTensor& keep,
                    const Tensor& X, const OptTensor& ss, const OptTensor& ds, const OptTensor& plan, int64_t chunk,
                    int64_t act, double slope, const OptTensor& out_mask, double mask_scale) {
  return spmm_csr_raw(indptr, indices, vals, eid, keep, X, ss, ds, plan, chunk, c10::nullopt, act, slope, out_mask, mask_scale);
}
void spmm_csr_out(const Tensor& indptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid, const OptTensor& keep,
                  const Tensor& X, const OptTensor& ss, const OptTensor& ds, const OptTensor& plan, int64_t chunk, Tensor out,
                  int64_t act, double slope, const OptTensor& out_mask, double mask_scale) {
  spmm_csr_raw(indptr, indices, vals, eid, keep, X, ss, ds, plan, chunk, out, act, slope, out_mask, mask_scale);
}
Tensor spmm_sliced_new(const Tensor& segptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid,
                       const OptTensor& keep, const Tensor& X, const OptTensor& ss, const OptTensor& ds, int64_t n_dst,
                       int64_t n_slices, int64_t act, double slope, const OptTensor& out_mask, double mask_scale,
                       int64_t column_passes, int64_t id_mult) {
  return spmm_sliced_raw(segptr, indices, vals, eid, keep, X, ss, ds, n_dst, n_slices, c10::nullopt, act, slope, out_mask,
                         mask_scale, column_passes, id_mult);
}
void spmm_sliced_out(const Tensor& segptr, const Tensor& indices, const OptTensor& vals, const OptTensor& eid,
                     const OptTensor& keep, const Tensor& X, const OptTensor& ss, const OptTensor& ds, int64_t n_dst,
                     int64_t n_slices, Tensor out, int64_t act, double slope, const OptTensor& out_mask, double mask_scale,
                     int64_t column_passes, int64_t id_mult) {
  spmm_sliced_raw(segptr, indices, vals, eid, keep, X, ss, ds, n_dst, n_slices, out, act, slope, out_mask, mask_scale,
                  column_passes, id_mult);
}

// ---------------------------------------------------------------------------------------------
// dreamgnn_mi::spmm_csr — the functional op of §8(b), with autograd w.r.t. X.
Tensor spmm_csr_forward(const Tensor& indptr, const Tensor& indices, const OptTensor& vals, const Tensor& X,
                        const OptTensor& src_scale, const OptTensor& dst_scale) {
  return spmm_csr_raw(indptr, indices, vals, c10::nullopt, c10::nullopt, X, src_scale, dst_scale, c10::nullopt, 0, c10::nullopt);
}

class SpmmCsrFunction : public torch::autograd::Function<SpmmCsrFunction> {
 public:
  static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& indptr, const Tensor& indices,
                        const OptTensor& vals, const Tensor& X, const OptTensor& src_scale, const OptTensor& dst_scale) {
    TORCH_CHECK(!(src_scale.has_value() && src_scale->defined() && src_scale->requires_grad()) &&
                    !(dst_scale.has_value() && dst_scale->defined() && dst_scale->requires_grad()) &&
                    !(vals.has_value() && vals->defined() && vals->requires_grad()),
                "spmm_csr: gradients w.r.t. edge values and the diagonal scales are not part of the path "
                "(adjacency values are constants, ci / cj are non-learnable node data)");
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({indptr, indices, vals.value_or(Tensor()), src_scale.value_or(Tensor()), dst_scale.value_or(Tensor())});
    ctx->saved_data["n_src"] = X.size(0);
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dreamgnn_mi::spmm_csr", "").typed<decltype(spmm_csr_forward)>();
    return op.call(indptr, indices, vals, X, src_scale, dst_scale);
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &indptr = saved[0], &indices = saved[1], &vals = saved[2], &ss = saved[3], &ds = saved[4];
    const int64_t n_src = ctx->saved_data["n_src"].toInt(), n_dst = indptr.numel() - 1;
    Tensor dX;
    if (grads[0].defined()) {  // X is the only differentiable input (forward refuses the others)
      // dX = diag(ss) A^T diag(ds) dY: the same kernel on the CSR of the reversed edges (built here;
      // callers with a graph object — dream_gnn_amd.ops.CSRGraph — keep it cached instead)
      const Tensor deg = indptr.slice(0, 1) - indptr.slice(0, 0, n_dst);
      const Tensor rows = at::repeat_interleave(at::arange(n_dst, indptr.options()), deg, c10::nullopt, indices.numel());
      auto t = csr_from_coo(indices, rows.to(at::kInt), n_src, n_dst);
      OptTensor vals_t = vals.defined() ? OptTensor(gather_f32(vals, std::get<2>(t))) : c10::nullopt;
      dX = spmm_csr_raw(std::get<0>(t), std::get<1>(t), vals_t, c10::nullopt, c10::nullopt, grads[0].contiguous(),
                        ds.defined() ? OptTensor(ds) : c10::nullopt, ss.defined() ? OptTensor(ss) : c10::nullopt,
                        c10::nullopt, 0, c10::nullopt);
    }
    return {Tensor(), Tensor(), Tensor(), dX, Tensor(), Tensor()};
  }
};

Tensor spmm_csr_autograd(const Tensor& indptr, const Tensor& indices, const OptTensor& vals, const Tensor& X,
                         const OptTensor& src_scale, const OptTensor& dst_scale) {
  return SpmmCsrFunction::apply(indptr, indices, vals, X, src_scale, dst_scale);
}

}  // namespace

TORCH_LIBRARY(dreamgnn_mi, m) {
  m.def("csr_from_coo(Tensor row, Tensor col, int n_rows, int n_cols=0) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("csr_sliced_from_coo(Tensor row, Tensor col, int n_rows, int n_cols, int n_slices) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("csr_sliced_from_csr(Tensor indptr, Tensor indices, Tensor eid, int n_cols, int n_slices) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("plan_build(Tensor indptr, int nnz, int chunk) -> Tensor");
  m.def("spmm_csr(Tensor indptr, Tensor indices, Tensor? vals, Tensor X, Tensor? src_scale=None, Tensor? dst_scale=None) -> Tensor");
  m.def("spmm_csr_raw(Tensor indptr, Tensor indices, Tensor? vals, Tensor? eid, Tensor? keep, Tensor X, Tensor? src_scale, "
        "Tensor? dst_scale, Tensor? plan, int chunk, int act=0, float slope=0., Tensor? out_mask=None, float mask_scale=1.) -> Tensor");
  m.def("spmm_csr_out(Tensor indptr, Tensor indices, Tensor? vals, Tensor? eid, Tensor? keep, Tensor X, Tensor? src_scale, "
        "Tensor? dst_scale, Tensor? plan, int chunk, Tensor(a!) out, int act=0, float slope=0., Tensor? out_mask=None, "
        "float mask_scale=1.) -> ()");
  m.def("spmm_sliced_raw(Tensor segptr, Tensor indices, Tensor? vals, Tensor? eid, Tensor? keep, Tensor X, Tensor? src_scale, "
        "Tensor? dst_scale, int n_dst, int n_slices, int act=0, float slope=0., Tensor? out_mask=None, float mask_scale=1., "
        "int column_passes=0, int id_mult=0) -> Tensor");
  m.def("spmm_sliced_out(Tensor segptr, Tensor indices, Tensor? vals, Tensor? eid, Tensor? keep, Tensor X, Tensor? src_scale, "
        "Tensor? dst_scale, int n_dst, int n_slices, Tensor(a!) out, int act=0, float slope=0., Tensor? out_mask=None, "
        "float mask_scale=1., int column_passes=0, int id_mult=0) -> ()");
  m.def("epilogue_backward(Tensor dY, Tensor Y, Tensor? mask, int act, float slope, float mask_scale) -> Tensor");
  m.def("knn_cosine_topk(Tensor Xn, int k) -> Tensor");
  m.def("scale_rows(Tensor X, Tensor scale) -> Tensor");
  m.def("colsum_rows_(Tensor(a!) feat_ext, Tensor coef, int n, int R, int i0) -> ()");
  m.def("colsum_rows_backward_(Tensor(a!) gf, Tensor coef, Tensor gs, int n, int R, int i0) -> ()");
  m.def("gather_f32(Tensor values, Tensor perm) -> Tensor");
  m.def("gather_concat_raw(Tensor src, Tensor dst, Tensor A, Tensor B) -> Tensor");
  m.def("gather_add_raw(Tensor src, Tensor dst, Tensor A, Tensor B, Tensor? bias, int act=0) -> Tensor");
  m.def("random_subset_select(Tensor like, int E, int keep, int seed, int e_offset=0) -> Tensor");
  m.def("random_subset_select_batch(Tensor like, int[] E, int[] keep, int[] seed, int[] e_offset) -> Tensor");
  m.def("random_subset_select_batch_dseed(Tensor seeds, int[] E, int[] keep, int[] e_offset) -> Tensor");
  m.def("keep_mask(Tensor keep, int E) -> Tensor");
  m.def("row_multiplicity(Tensor indptr, Tensor vals, float rel_tol) -> (Tensor, Tensor, Tensor)");
  m.def("compact_layout(Tensor ptr, Tensor indices, Tensor? vals, Tensor eid, Tensor keep) -> (Tensor, Tensor, Tensor)");
}

// ROCm builds of torch dispatch HIP tensors under the CUDA key
TORCH_LIBRARY_IMPL(dreamgnn_mi, CUDA, m) {
  m.impl("csr_from_coo", csr_from_coo);
  m.impl("csr_sliced_from_coo", csr_sliced_from_coo);
  m.impl("csr_sliced_from_csr", csr_sliced_from_csr);
  m.impl("plan_build", plan_build);
  m.impl("spmm_csr", spmm_csr_forward);
  m.impl("spmm_csr_raw", spmm_csr_new);
  m.impl("spmm_csr_out", spmm_csr_out);
  m.impl("spmm_sliced_raw", spmm_sliced_new);
  m.impl("spmm_sliced_out", spmm_sliced_out);
  m.impl("epilogue_backward", epilogue_backward);
  m.impl("knn_cosine_topk", knn_cosine_topk);
  m.impl("scale_rows", scale_rows);
  m.impl("colsum_rows_", colsum_rows_);
  m.impl("colsum_rows_backward_", colsum_rows_backward_);
  m.impl("gather_f32", gather_f32);
  m.impl("gather_concat_raw", gather_concat_raw);
  m.impl("gather_add_raw", gather_add_raw);
  m.impl("random_subset_select", random_subset_select);
  m.impl("random_subset_select_batch", random_subset_select_batch);
  m.impl("random_subset_select_batch_dseed", random_subset_select_batch_dseed);
  m.impl("keep_mask", keep_mask);
  m.impl("row_multiplicity", row_multiplicity);
  m.impl("compact_layout", compact_layout);
}

TORCH_LIBRARY_IMPL(dreamgnn_mi, Autograd, m) { m.impl("spmm_csr", spmm_csr_autograd); }
