"""The C-ABI library loads on the build box and exports exactly what include/dgmi.h declares.
No compute call is made here (no GPU): only version / string / argument-validation paths that
return before any HIP launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dgmi.h")).read()
    return sorted(set(re.findall(r"DGMI_API\s+[\w\s\*]+?\b(dgmi_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    assert _declared() == ["dgmi_abi_version", "dgmi_compact_layout_i32", "dgmi_compact_layout_workspace_bytes", "dgmi_csr_from_coo_i32", "dgmi_csr_sliced_from_coo_i32", "dgmi_csr_sliced_from_csr_i32",
                           "dgmi_device_ok", "dgmi_epilogue_backward_f32", "dgmi_gather_add_f32", "dgmi_gather_concat_f32",
                           "dgmi_gather_f32", "dgmi_keep_mask_f32", "dgmi_knn_cosine_supported", "dgmi_knn_cosine_topk_f32",
                           "dgmi_knn_cosine_workspace_bytes",
                           "dgmi_probe_row_gather_f32",
                           "dgmi_random_subset_mask_f32", "dgmi_random_subset_select", "dgmi_random_subset_select_batch",
                           "dgmi_random_subset_select_batch_dseed",
                           "dgmi_random_subset_workspace_bytes", "dgmi_rank_add_f32", "dgmi_row_multiplicity_f32", "dgmi_scale_rows_f32",
                           "dgmi_set_tuning", "dgmi_spmm_csr_f32", "dgmi_spmm_csr_planned_f32", "dgmi_spmm_default_chunk",
                           "dgmi_spmm_partials_bytes", "dgmi_spmm_plan_build", "dgmi_spmm_plan_bytes",
                           "dgmi_spmm_sliced_f32", "dgmi_spmm_sliced_planes_bytes", "dgmi_status_string", "dgmi_weighted_colsum_f32"]


def test_library_exports_every_declared_symbol():
    from dream_gnn_amd import _lib

    for name in _declared():
        assert hasattr(_lib.lib, name), name
        assert name in _lib.SIGNATURES, "ctypes binding missing for " + name
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_status_strings():
    from dream_gnn_amd import _lib

    text = open(os.path.join(ROOT, "include", "dgmi.h")).read()
    assert _lib.lib.dgmi_abi_version() == int(re.search(r"#define DGMI_ABI_VERSION (\d+)", text).group(1))
    assert _lib.lib.dgmi_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert len(_lib.lib.dgmi_status_string(code)) > 3
    assert b"unknown" in _lib.lib.dgmi_status_string(-99)


def test_argument_validation_returns_codes_without_a_gpu():
    from dream_gnn_amd import _lib

    L = _lib.lib
    need = ctypes.c_size_t(0)
    assert L.dgmi_csr_from_coo_i32(None, None, -1, 4, 0, None, None, None, None, ctypes.byref(need), None) == -1
    assert L.dgmi_csr_from_coo_i32(None, None, 1, 4, 0, None, None, None, None, None, None) == -1
    assert L.dgmi_csr_from_coo_i32(None, None, 2 ** 31, 4, 0, None, None, None, None, ctypes.byref(need), None) == -2
    K0 = (None, None, 0)  # no edge dropout on the fly: eid, keep, n_keep
    E0 = (0, 0.0, None, 0, 1.0)  # no epilogue: act, act_slope, out_mask, ld_mask, out_mask_scale
    assert L.dgmi_spmm_csr_f32(None, None, None, *K0, None, 4, None, None, None, 4, -1, 1, 4, *E0, None) == -1
    assert L.dgmi_spmm_csr_f32(None, None, None, *K0, None, 4, None, None, None, 4, 2, 2, 4, *E0, None) == -1  # null indptr/Y
    assert L.dgmi_spmm_csr_f32(None, None, None, *K0, None, 4, None, None, None, 4, 0, 0, 4, *E0, None) == 0   # empty problem
    assert L.dgmi_spmm_csr_f32(16, 16, None, *K0, 16, 2, None, None, 32, 4, 2, 2, 4, *E0, None) == -1          # ldx < F
    assert L.dgmi_spmm_csr_f32(16, 16, None, *K0, 64, 4, None, None, 64, 4, 2, 2, 4, *E0, None) == -1          # Y aliases X
    assert L.dgmi_spmm_csr_f32(16, 16, None, None, 16, 1, 16, 4, None, None, 32, 4, 2, 2, 4, *E0, None) == -1  # keep without eid
    assert L.dgmi_spmm_csr_f32(16, 16, None, 16, 16, 9, 16, 4, None, None, 32, 4, 2, 2, 4, *E0, None) == -1    # > 8 descriptions
    assert L.dgmi_random_subset_select(10, 11, 0, 0, 16, 16, 1 << 20, None) == -1                         # keep > E
    assert L.dgmi_random_subset_select(10, 5, 0, 0, 16, 16, 8, None) == -3                                # workspace too small
    assert L.dgmi_random_subset_select(10, 5, 0, 2 ** 31 - 5, 16, 16, 1 << 20, None) == -2                # offset + E overflows
    assert L.dgmi_keep_mask_f32(None, 1, 5, 16, None) == -1 and L.dgmi_keep_mask_f32(None, 0, 0, None, None) == 0
    assert L.dgmi_probe_row_gather_f32(16, 8, 6, 8, 8, 8, 0, 16, None) == -1                              # F % 4
    # per-step layout compaction: size query is host arithmetic; argument errors come back before any launch
    assert L.dgmi_compact_layout_workspace_bytes(0) >= 0 and L.dgmi_compact_layout_workspace_bytes(10_000_000) < 4 << 20
    assert L.dgmi_compact_layout_i32(16, 5, 16, None, 16, 100, None, 1, 16, 16, None, 16, 1 << 20, None) == -1  # keep missing
    assert L.dgmi_compact_layout_i32(16, 5, 16, 16, 16, 100, 16, 1, 16, 16, None, 16, 1 << 20, None) == -1      # vals without vals_out
    assert L.dgmi_compact_layout_i32(16, 5, 16, None, 16, 100, 16, 1, 16, 16, None, 16, 8, None) == -3          # workspace too small
    assert L.dgmi_compact_layout_i32(None, 0, None, None, None, 0, 16, 1, None, None, None, None, 0, None) == 0   # empty layout
    assert L.dgmi_compact_layout_i32(16, 5, 16, None, 16, 100, None, 0, 16, 16, None, 16, 1 << 20, None) == -1  # nothing dropped
    # launch-parameter overrides: known names only; the library reads no environment on a launch path
    assert L.dgmi_set_tuning(b"sliced_rows", 0) == 0 and L.dgmi_set_tuning(b"no_such_knob", 1) == -1
    assert L.dgmi_set_tuning(None, 1) == -1
    # plan sizing is host arithmetic: items <= n_rows + nnz/chunk, long rows <= nnz/(chunk+1)
    assert L.dgmi_spmm_default_chunk(50_000, 10_000_000) == 512
    assert L.dgmi_spmm_default_chunk(681, 465_000) == 128
    assert L.dgmi_spmm_default_chunk(763, 6_800) == 64
    assert L.dgmi_spmm_plan_bytes(10, 100, 64) == 4 * (16 + 4 * (10 + 1 + 1))
    assert L.dgmi_spmm_plan_bytes(10, 100, 3) == 0  # chunk out of range
    assert L.dgmi_spmm_partials_bytes(1000, 64, 128) == (15 + 15) * 128 * 4
    assert L.dgmi_spmm_partials_bytes(1000, 64, 341) == (15 + 15) * 344 * 4
    assert L.dgmi_spmm_partials_bytes(0, 64, 128) == 16
    assert L.dgmi_spmm_plan_build(None, -1, 0, 64, None, 0, None, ctypes.byref(need), None) == -1
    assert L.dgmi_spmm_csr_planned_f32(16, 16, None, *K0, 16, 4, None, None, 32, 4, 2, 2, 4, 8, 7, 16, 16, 1 << 20, *E0, None) == -1
    assert L.dgmi_spmm_csr_planned_f32(16, 16, None, *K0, 16, 4, None, None, 32, 4, 2, 2, 4, 8, 64, 16, 16, 0, *E0, None) == -3
    assert L.dgmi_gather_concat_f32(None, None, -1, None, 4, 4, None, 4, 4, None, 8, None) == -1
    assert L.dgmi_gather_concat_f32(16, 16, 3, 16, 4, 4, 16, 4, 4, 16, 7, None) == -1  # ldo < Fa+Fb
    assert L.dgmi_gather_concat_f32(None, None, 0, None, 4, 4, None, 4, 4, None, 8, None) == 0
    assert L.dgmi_gather_f32(None, None, -3, None, None) == -1
    assert L.dgmi_gather_f32(None, None, 0, None, None) == 0
    # sliced product: column_passes is 0 (by footprint) or 1 (one full-width pass)
    assert L.dgmi_spmm_sliced_f32(16, 16, None, *K0, 16, 4, None, None, 32, 4, 2, 2, 4, 8, 2, 0, 48, 1 << 20, *E0, None) == -1
    assert L.dgmi_spmm_sliced_f32(16, 16, None, *K0, 16, 4, None, None, 32, 4, 2, 2, 4, 8, 1, 0, 48, 0, *E0, None) == -3  # planes too small
    assert L.dgmi_spmm_sliced_f32(16, 16, 16, *K0, 16, 4, None, None, 32, 4, 2, 2, 4, 8, 1, 1, 48, 1 << 20, *E0, None) == -1  # vals AND multiplicities
    assert L.dgmi_row_multiplicity_f32(16, 16, 4, 10, 2.4e-7, 16, 16, None, None) == -1  # no flag word
    # (f4) cosine kNN: shapes the kernels take, workspace sizing, argument checks — host arithmetic only
    assert L.dgmi_knn_cosine_supported(763, 768, 4) == 1 and L.dgmi_knn_cosine_supported(100_000, 768, 16) == 1
    assert L.dgmi_knn_cosine_supported(763, 770, 4) == 0    # D % 8
    assert L.dgmi_knn_cosine_supported(763, 768, 17) == 0   # k > 16 only through the screen (N >= 1536)
    assert L.dgmi_knn_cosine_supported(100_000, 768, 64) == 1 and L.dgmi_knn_cosine_supported(100_000, 768, 65) == 0
    assert L.dgmi_knn_cosine_supported(3, 768, 4) == 0      # k > N
    assert L.dgmi_knn_cosine_supported(763, 2048, 4) == 0   # a 32-query tile of 2048 columns does not fit the LDS
    small, mid, big = (L.dgmi_knn_cosine_workspace_bytes(n, 768, 4) for n in (763, 8192, 100_000))
    assert 0 < small < mid < big and big >= 100_000 * (768 * 2 + 16 * 64 * 8)  # bf16 copy + 16 regions of 64 slots
    assert L.dgmi_knn_cosine_workspace_bytes(763, 770, 4) == 0
    assert L.dgmi_knn_cosine_topk_f32(None, 768, 0, 768, 4, None, None, 0, None) == 0           # empty problem
    assert L.dgmi_knn_cosine_topk_f32(None, 768, 763, 768, 4, None, None, 0, None) == -1        # null pointers
    assert L.dgmi_knn_cosine_topk_f32(16, 760, 763, 768, 4, 16, None, 0, None) == -1            # ld < D
    assert L.dgmi_knn_cosine_topk_f32(16, 768, 763, 768, 17, 16, None, 0, None) == -1           # unsupported k at this N
    assert L.dgmi_knn_cosine_topk_f32(16, 768, 763, 768, 4, 16, None, 0, None) == -3            # workspace missing


def test_torch_operator_library_is_registered():
    """libdgmi_torch.so loads on the build box and registers the dreamgnn_mi ops of SURVEY §8(b)
    for the HIP ("CUDA") dispatch key only — no CPU kernels exist."""
    import torch

    from dream_gnn_amd import _lib

    assert os.path.exists(_lib.TORCH_LIB_PATH)
    for name in ("csr_from_coo", "csr_sliced_from_coo", "plan_build", "spmm_csr", "spmm_csr_raw", "spmm_csr_out",
                 "spmm_sliced_raw", "spmm_sliced_out", "csr_sliced_from_csr", "epilogue_backward", "knn_cosine_topk", "gather_f32", "gather_concat_raw", "gather_add_raw",
                 "random_subset_select", "random_subset_select_batch", "keep_mask"):
        assert hasattr(torch.ops.dreamgnn_mi, name), name
    schema = torch.ops.dreamgnn_mi.spmm_csr.default._schema
    assert [a.name for a in schema.arguments] == ["indptr", "indices", "vals", "X", "src_scale", "dst_scale"]
    with pytest.raises(NotImplementedError):
        torch.ops.dreamgnn_mi.spmm_csr(torch.zeros(2, dtype=torch.int32), torch.zeros(1, dtype=torch.int32), None,
                                       torch.zeros(1, 4))


def test_product_path_refuses_cpu_tensors():
    import torch

    from dream_gnn_amd import ops

    with pytest.raises(RuntimeError, match="MI355X only"):
        ops.csr_from_coo(torch.zeros(3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), 2)
    with pytest.raises(RuntimeError, match="MI355X only"):
        ops.CSRGraph(torch.zeros(3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), 2, 2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dream_gnn_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libdgmi.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib._load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dream_gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libdgmi_oracle" not in src, f


def test_integration_md_stub_matches_the_header_signatures():
    """INTEGRATION.md §2's ctypes stub (executed verbatim on the GPU by tests/test_gpu_cabi.py) declares, for every
    entry point it binds, exactly the argument list of dream_gnn_amd/_lib.py — which is checked against the header."""
    from dream_gnn_amd import _lib

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# dgmi_binding\.py.*?)```", text, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(block, "INTEGRATION.md:dgmi_binding.py", "exec"), ns)
    finally:
        os.chdir(cwd)
    bound = [n for n in _lib.SIGNATURES if getattr(getattr(ns["lib"], n), "argtypes", None)]
    assert {"dgmi_spmm_csr_f32", "dgmi_csr_from_coo_i32", "dgmi_csr_sliced_from_coo_i32", "dgmi_spmm_sliced_f32",
            "dgmi_spmm_sliced_planes_bytes"} <= set(bound)
    for n in bound:
        res, args = _lib.SIGNATURES[n]
        got = list(getattr(ns["lib"], n).argtypes)
        assert [ctypes.sizeof(a) for a in got] == [ctypes.sizeof(a) for a in args], n
        assert getattr(ns["lib"], n).restype is res or ctypes.sizeof(getattr(ns["lib"], n).restype) == ctypes.sizeof(res), n


def test_no_launch_path_reads_the_environment():
    """the environment is read once (dgmi_api.hip `tuning()`, the kNN crossovers' static
    initialisers), never per launch; the product kernels carry no experiment switches."""
    csrc = os.path.join(ROOT, "dream_gnn_amd", "csrc")
    for name in ("dgmi_sliced.hip", "dgmi_select.hip", "dgmi_spmm.hip", "dgmi_plan.hip", "dgmi_compact.hip", "dgmi_csr.hip",
                 "dgmi_sort.hip", "dgmi_edge.hip"):
        text = open(os.path.join(csrc, name)).read()
        assert "getenv" not in text, name
        assert "DGMI_EXPERIMENT" not in text and "#ifdef" not in text, name
    api = open(os.path.join(csrc, "dgmi_api.hip")).read()
    assert api.count("getenv") == 2 and "static Tuning t" in api  # env_ll + the presence test, both inside tuning()
