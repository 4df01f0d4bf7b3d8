"""dream_gnn_amd — MI355X-native message passing for DREAM-GNN's GCMC/FGCN hot path.

The package holds only what the path needs:

* ``csrc/`` + ``libdgmi.so`` — hand-written gfx950 HIP kernels behind the C ABI of
  ``include/dgmi.h`` (CSR SpMM with fused diagonal scalings in three launch forms — wave-per-row,
  nnz-balanced plan, XCD-local source slices —, device COO->CSR, decoder edge gather, edge-dropout
  subset selection);
* ``ops``      — torch-facing wrappers (device pointers + current stream -> C ABI) and the
  autograd formula;
* ``graph``    — the light graph containers the modules consume in place of DGL graphs;
* ``layers``   — drop-in ``GCMCGraphConv`` / ``GCMCLayer`` / ``GraphConvolution`` / ``GCN`` /
  ``FGCN`` with the reference's signatures and ``state_dict`` keys (reference layers.py);
* ``model`` / ``harness`` — the reference's callers (Net, decoder, train / eval step) restated so the
  path can be driven end to end without DGL or the reference;
* ``synth``    — device-side generators for the synthetic configs;
* ``shard``    — nnz-balanced edge partition + RCCL exchange for the multi-GPU configs.

There is no CPU fallback: every op raises if ``libdgmi.so`` is missing or a tensor is
not on a HIP device.
"""
from . import _lib  # noqa: F401  (fails loudly if the extension is not built)
from .ops import (CSRGraph, EdgePairs, SlicedCSR, SpmmPlan, csr_from_coo, gather_add, gather_concat,  # noqa: F401
                  random_subset_mask, spmm_csr)

__all__ = ["CSRGraph", "EdgePairs", "SlicedCSR", "SpmmPlan", "csr_from_coo", "gather_add", "gather_concat",
           "random_subset_mask", "spmm_csr"]
