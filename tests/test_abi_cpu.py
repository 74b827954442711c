"""CPU-side checks of the boundary: the C-ABI library is built, loads, and exports every
symbol include/bnpvi.h declares (no compute calls -- there is no GPU here); the ctypes
binding table matches the header; the drop-in classes keep the reference's surface."""
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "bnpvi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bnpvi_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from bnp_vae_amd import _lib
    names = header_functions()
    assert len(names) >= 15
    so = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), "libbnpvi.so does not export %s" % n
    assert sorted(_lib.EXPORTS.keys()) == names, "ctypes table and header disagree"
    assert _lib.lib.bnpvi_abi_version() == 1


def test_padding_rules():
    from bnp_vae_amd import _lib
    assert [_lib.lib.bnpvi_padded_dim(d) for d in (1, 8, 9, 32, 33, 64, 200, 256, 257)] == \
        [32, 32, 32, 32, 64, 64, 256, 256, -1]
    assert [_lib.lib.bnpvi_padded_leaves(l) for l in (1, 25, 32, 33, 64, 512, 7776)] == \
        [32, 32, 32, 64, 64, 512, 7776]


def test_error_reporting_without_gpu():
    import ctypes as C
    from bnp_vae_amd import _lib
    h = C.c_void_p()
    rc = _lib.lib.bnpvi_ctx_create(0, C.byref(h))
    if rc == 0:                       # a GPU is present: fine
        _lib.lib.bnpvi_ctx_destroy(h)
        return
    assert _lib.lib.bnpvi_last_error()
    with pytest.raises(_lib.BnpviError):
        _lib.check(rc, "bnpvi_ctx_create")


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bnp_vae_amd import VarInf
    vi = VarInf(latent_code_size=4)
    lc = {"v": {"0": np.zeros(4, np.float32), "1": np.ones(4, np.float32)}}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vi.update_variational_parameters(lc)


def test_surface_matches_reference():
    """Names and argument lists of src/var_inf.py:12-491 / src/node.py:7."""
    from bnp_vae_amd import VarInf, Node
    want = {
        "update_variational_parameters": ["self", "latent_codes"],
        "get_true_path_mean": ["self", "x_annot_batch"],
        "get_matrix_from_dict": ["self", "latent_codes"],
        "compute_sigma_alpha": ["self", "latent_codes_matrix"],
        "compute_gamma": ["self"],
        "split_nodes": ["self", "node", "latent_codes_matrix", "split_thr"],
        "merge_nodes": ["self", "node", "latent_codes_matrix", "split_thr"],
        "weighted_kmeans": ["data", "weights", "K"],
        "normalize": ["p"],
        "write_alpha": ["self", "filename"],
        "write_sigma": ["self", "filename"],
        "write_nodes": ["self", "filename"],
        "get_nodes_list": ["self", "node", "result"],
        "get_alpha_leaves": ["self"],
        "get_phi_leaves": ["self", "idx"],
        "initialize_phi": ["self", "node", "init_values"],
    }
    for name, args in want.items():
        got = list(inspect.signature(getattr(VarInf, name)).parameters)
        assert got[:len(args)] == args, (name, got)
    assert list(inspect.signature(VarInf.compute_phi).parameters)[:2] == ["self", "latent_codes_matrix"]
    vi = VarInf()
    assert vi.root is None and vi.decay_coeff == 0.0
    assert vi.vidid_frameid_to_idx == [] and vi.vidid_to_idx == []
    assert list(inspect.signature(Node.__init__).parameters) == \
        ["self", "node_id", "num_vids", "num_frames", "parent", "LATENT_CODE_SIZE", "GAMMA"]
    np.random.seed(1)
    n = Node("0", 3, 5, None, 4, 10.0)
    np.random.seed(1)
    assert np.array_equal(n.alpha, np.random.normal(size=4))
    assert n.isLeafNode and n.parent is None and n.children == [] and n.sigmasqr_inv == 1.0
    assert n.phi.shape == (5, 1) and np.all(n.phi == 1.0)
    assert n.gamma.shape == (3, 2) and np.all(n.gamma[:, 0] == 1.0) and np.all(n.gamma[:, 1] == 10.0)
    assert np.array_equal(VarInf.normalize([0.0, 0.0]), [0.5, 0.5])
    assert vi.get_true_path_mean([("a", "b")] * 3).shape == (3, 32)


def test_config_resolution(monkeypatch):
    import sys
    import types
    from bnp_vae_amd import VarInf, config
    util = types.ModuleType("util")
    util.LATENT_CODE_SIZE, util.GAMMA, util.STDEV_THR = 16, 5.0, 7
    util.SIGMA_B_sqrinv, util.SIGMA_Z_sqrinv = 2.0, 3.0
    util.ALPHA = np.ones(16)
    util.args = types.SimpleNamespace(num_splits=3)
    monkeypatch.setitem(sys.modules, "util", util)
    vi = VarInf()
    assert (vi.LATENT_CODE_SIZE, vi.GAMMA, vi.STDEV_THR, vi.num_splits) == (16, 5.0, 7.0, 3)
    assert vi.SIGMA_B_sqrinv == 2.0 and vi.SIGMA_Z_sqrinv == 3.0 and np.all(vi.ALPHA == 1.0)
    monkeypatch.setattr(config, "STDEV_THR", 11, raising=False)
    assert VarInf().STDEV_THR == 11.0
    assert VarInf(stdev_thr=2).STDEV_THR == 2.0
