"""Host-side logic of the drop-in modules on the build box (no GPU): the three ops are
replaced, explicitly and only here, by the oracle-backed functions of tests/_cpu_backend.py;
expected values are the golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

import _cases as C
import _cpu_backend

CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def _backend(oracle):
    with _cpu_backend.patched():
        yield


@pytest.mark.parametrize("name", ["random", "dups", "empty_rows", "single_row", "E0"])
def test_csr(name):
    C.case_csr(CPU, name)


@pytest.mark.parametrize("F", [1, 3, 4, 127, 128, 341, 768])
def test_spmm_autograd(F):
    C.case_spmm(CPU, F)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gcmc_graph_conv(mode):
    C.case_gcmc_conv(CPU, mode)


@pytest.mark.parametrize("name", ["shared_ini", "shared_noini", "unshared", "shareflag_dimdiff"])
def test_gcmc_layer(name):
    C.case_gcmc_layer(CPU, name)


@pytest.mark.parametrize("name", ["both", "simonly"])
def test_fgcn(name):
    C.case_fgcn(CPU, name)


def test_graphconv_nobias():
    C.case_graphconv_nobias(CPU)


def test_net_forward_loss_grads():
    C.case_net(CPU)


@pytest.mark.parametrize("symm", [1, 0])
def test_enc_graph_format(symm):
    C.case_encgraph(CPU, symm)


def test_external_and_own_weight_conflict():
    from dream_gnn_amd import graph as G, layers as L

    hg = G.HeteroGraph({("drug", "0", "disease"): (torch.tensor([0, 1]), torch.tensor([1, 0]))},
                       {"drug": 2, "disease": 2}).int()
    hg.nodes["drug"].data["cj"] = torch.ones(2, 1)
    hg.nodes["disease"].data["ci"] = torch.ones(2, 1)
    conv = L.GCMCGraphConv(4, 4, weight=True)
    with pytest.raises(L.DGMIError):  # layers.py:214-216
        conv(hg["0"], torch.randn(2, 4), weight=torch.randn(4, 4))


def test_dot_or_identity_three_column_branch():
    from dream_gnn_amd.layers import dot_or_identity

    B = torch.arange(20.0).view(5, 4)
    A = torch.tensor([[0, 1, 2], [4, 4, 3]])
    out = dot_or_identity(A, B)  # layers.py:385-389
    assert out.shape == (2, 12) and torch.equal(out[1, :4], B[4]) and torch.equal(out[1, 8:], B[3])
    assert dot_or_identity(None, B) is B


def test_state_dict_keys_match_reference_layout():
    from dream_gnn_amd import layers as L

    shared = L.GCMCLayer([0, 1], 12, 12, 24, 6, agg="sum", share_user_item_param=True)
    assert sorted(shared.state_dict()) == ["att", "basis", "ifc.bias", "ifc.weight", "ufc.bias", "ufc.weight"]
    assert shared.ifc is shared.ufc and shared.msg_units == 8 and shared.W_r is not None
    own = L.GCMCLayer([0, 1], 12, 10, 24, 6, agg="sum", share_user_item_param=False)
    keys = set(own.state_dict())
    assert {"conv.mods.0.weight", "conv.mods.rev-0.weight", "conv.mods.1.weight", "conv.mods.rev-1.weight",
            "att", "basis", "ifc.weight", "ufc.weight"} <= keys
    assert own.W_r is None and own.conv.mods["rev-0"].weight.shape == (10, 8)
    with pytest.raises(AssertionError):  # layers.py:53
        L.GCMCLayer([0, 1], 12, 12, 25, 6, agg="stack")
    f = L.FGCN(7, 5, 16, 8, 0.1)
    assert "FGCN_drug.gc1.weight" in f.state_dict() and "disease_fusion.bias" in f.state_dict()


def test_adjacency_cache_is_per_tensor_and_version():
    from dream_gnn_amd import layers as L

    idx = torch.tensor([[0, 1, 1], [1, 0, 1]])
    adj = torch.sparse_coo_tensor(idx, torch.tensor([1.0, 2.0, 3.0]), (2, 2))
    g1 = L.adjacency_csr(adj)
    assert L.adjacency_csr(adj) is g1
    adj2 = torch.sparse_coo_tensor(idx, torch.tensor([1.0, 2.0, 3.0]), (2, 2))
    assert L.adjacency_csr(adj2) is not g1
    assert L.adjacency_csr(g1) is g1


def test_uncoalesced_duplicates_sum_like_torch_spmm():
    from dream_gnn_amd import layers as L

    idx = torch.tensor([[0, 0, 0, 2], [1, 1, 1, 0]])
    val = torch.tensor([1.0, 2.0, 4.0, 5.0])
    adj = torch.sparse_coo_tensor(idx, val, (3, 2))
    x = torch.randn(2, 6)
    gc = L.GraphConvolution(6, 6, bias=False)
    gc.weight.data = torch.eye(6)
    assert torch.allclose(gc(x, adj), torch.spmm(adj, x), atol=1e-6)
