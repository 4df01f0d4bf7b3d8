"""GPU parity of the drop-in modules: the HIP path (C ABI) against the golden vectors the
reference produced (tests/golden), on cuda:0."""
import pytest

import _cases as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["random", "dups", "empty_rows", "single_row", "E0"])
def test_csr(dev, name):
    C.case_csr(dev, name)


@pytest.mark.parametrize("F", [1, 3, 4, 127, 128, 341, 768])
def test_spmm_autograd(dev, F):
    C.case_spmm(dev, F)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gcmc_graph_conv(dev, mode):
    C.case_gcmc_conv(dev, mode)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("name", ["shared_ini", "shared_noini", "unshared", "shareflag_dimdiff"])
def test_gcmc_layer(dev, name, fuse):
    C.case_gcmc_layer(dev, name, fuse)


@pytest.mark.parametrize("name", ["shared_ini", "unshared"])
def test_gcmc_layer_with_device_argument(dev, name):
    """The reference's Net always passes device=args.device to its layers (model.py:19,40): the
    fused relation path (f3) must engage in that configuration too, not only with device=None
    (`case_gcmc_layer` asserts that the fused CSRs were built and used)."""
    import torch

    C.case_gcmc_layer(dev, name, True, device_arg=torch.device("cuda:0"))
    C.case_gcmc_layer(dev, name, True, device_arg="cuda:0")


@pytest.mark.parametrize("name", ["both", "simonly"])
def test_fgcn(dev, name):
    C.case_fgcn(dev, name)


def test_graphconv_nobias(dev):
    C.case_graphconv_nobias(dev)


@pytest.mark.parametrize("fuse_decoder", [True, False])
def test_net_forward_loss_grads(dev, fuse_decoder):
    C.case_net(dev, fuse_decoder)


@pytest.mark.parametrize("symm", [1, 0])
def test_enc_graph_format(dev, symm):
    C.case_encgraph(dev, symm)


def test_native_library_is_what_runs(dev):
    """The ops must be backed by the in-tree libdgmi.so bound to torch's HIP runtime."""
    import os

    from dream_gnn_amd import _lib

    assert _lib.lib.dgmi_device_ok() == 1
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.LIB_PATH) in maps
    hip = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hip) == 1, "two HIP runtimes loaded: %s" % hip


@pytest.mark.parametrize("name", ["n30_k4", "n12_k20"])
def test_similarity_graph_builder(dev, name):
    C.case_similarity_graph(dev, name)
