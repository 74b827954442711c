"""Pins the CPU oracle (oracle/vi_oracle.py) against the REAL reference, run live through
the in-memory lib2to3 loader.  Only possible in the build container (the reference does
not travel to the GPU box) -- skipped elsewhere; the committed fixtures in tests/golden/
carry the same pins to the GPU box (tests/test_oracle_golden.py).
"""
import numpy as np
import pytest

from oracle import ref_loader
from oracle.vi_oracle import OracleVarInf, structure, make_latent_dict, gaussian_mixture

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(),
                                reason="/root/reference not present")


def _ref(D, thr, ns):
    vi_mod, node_mod = ref_loader.load_reference(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    return vi_mod, node_mod


def _collect(vi):
    nodes = vi.get_nodes_list(vi.root, [])
    return nodes


def _assert_same_tree(ref_vi, ora_vi, tol=0.0):
    assert structure(ref_vi) == structure(ora_vi)
    for a, b in zip(_collect(ref_vi), _collect(ora_vi)):
        np.testing.assert_allclose(b.alpha, a.alpha, rtol=tol, atol=tol)
        assert abs(b.sigmasqr_inv - a.sigmasqr_inv) <= tol * max(1.0, abs(a.sigmasqr_inv))
        if a.gamma is None:
            assert b.gamma is None
        else:
            np.testing.assert_allclose(b.gamma, a.gamma, rtol=tol, atol=tol)
        if a.isLeafNode:
            np.testing.assert_allclose(b.phi, a.phi, rtol=tol, atol=tol)


@pytest.mark.parametrize("D,V,fpv,thr,ns,epochs", [
    (8, 4, [25], 32.0, 2, 3),
    (6, 5, [7, 13, 1, 30, 9], 1.0, 3, 4),     # low threshold: splits every epoch
    (4, 1, [60], 0.5, 2, 3),                  # single video ("pure GMM" variant)
])
def test_trajectory_matches_reference(D, V, fpv, thr, ns, epochs):
    vi_mod, _ = _ref(D, thr, ns)
    N = sum(fpv) if len(fpv) == V else fpv[0] * V
    rng = np.random.RandomState(5)
    ref_vi = vi_mod.VarInf()
    ora_vi = OracleVarInf(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    for ep in range(epochs):
        Z = (gaussian_mixture(N, D, 6, seed=100 + ep, spread=3.0)
             + 0.01 * rng.normal(size=(N, D))).astype(np.float32)
        lc = make_latent_dict(Z, fpv)
        assert len(lc) == V
        np.random.seed(13 + ep)
        ref_vi.update_variational_parameters(lc)
        np.random.seed(13 + ep)
        ora_vi.update_variational_parameters(lc)
        _assert_same_tree(ref_vi, ora_vi, tol=0.0)      # bit-for-bit, fp64
        assert ref_vi.decay_coeff == ora_vi.decay_coeff


def test_phases_match_reference_on_irregular_tree():
    """Per-phase equality (M, gamma, prior tables, E) after the tree has been reshaped."""
    D, thr, ns = 5, 0.8, 3
    vi_mod, _ = _ref(D, thr, ns)
    fpv = [11, 4, 17, 9]
    N = sum(fpv)
    Z = gaussian_mixture(N, D, 5, seed=3, spread=3.0)
    lc = make_latent_dict(Z, fpv)
    ref_vi, ora_vi = vi_mod.VarInf(), OracleVarInf(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    for s, vi in ((1, ref_vi), (1, ora_vi)):
        np.random.seed(s)
        vi.update_variational_parameters(lc)
        vi.update_variational_parameters(lc)
    assert structure(ref_vi) == structure(ora_vi)
    Z2 = gaussian_mixture(N, D, 5, seed=4, spread=3.0)
    ref_vi.compute_sigma_alpha(Z2); ora_vi.compute_sigma_alpha(Z2)
    _assert_same_tree(ref_vi, ora_vi)
    ref_vi.compute_gamma(); ora_vi.compute_gamma()
    _assert_same_tree(ref_vi, ora_vi)
    on_r = ref_vi.compute_gamma_sum_on(ref_vi.root, np.zeros(len(fpv)), {})
    _, bf_r = ref_vi.compute_gamma_sum_before(ref_vi.root, np.zeros(len(fpv)), {})
    on_o, bf_o = ora_vi.prior_tables()
    assert [k.node_id for k in on_r] == [k.node_id for k in on_o]
    for (ka, va), (kb, vb) in zip(on_r.items(), on_o.items()):
        np.testing.assert_array_equal(va, vb)
    for (ka, va), (kb, vb) in zip(bf_r.items(), bf_o.items()):
        np.testing.assert_array_equal(va, vb)
    ref_vi.compute_phi(Z2); ora_vi.compute_phi(Z2)
    _assert_same_tree(ref_vi, ora_vi)


def test_weighted_kmeans_and_normalize_match_reference():
    vi_mod, _ = _ref(4, 1.0, 2)
    rng = np.random.RandomState(0)
    data = rng.normal(size=(50, 4)).astype(np.float32)
    w = rng.uniform(size=50)
    a = vi_mod.VarInf.weighted_kmeans(data, w, 4)
    b = OracleVarInf.weighted_kmeans(data, w, 4)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(vi_mod.VarInf.normalize([0.0, 0.0]), OracleVarInf.normalize([0.0, 0.0]))
    np.testing.assert_array_equal(vi_mod.VarInf.normalize([1.0, 3.0]), OracleVarInf.normalize([1.0, 3.0]))


def test_writers_match_reference(tmp_path):
    D = 4
    vi_mod, _ = _ref(D, 32.0, 2)
    Z = gaussian_mixture(40, D, 3, seed=9)
    lc = make_latent_dict(Z, [10])
    ref_vi, ora_vi = vi_mod.VarInf(), OracleVarInf(latent_code_size=D)
    for vi in (ref_vi, ora_vi):
        np.random.seed(2)
        vi.update_variational_parameters(lc)
    pa, ps = tmp_path / "a.txt", tmp_path / "s.txt"
    ref_vi.write_alpha(str(pa)); ref_vi.write_sigma(str(ps))
    assert pa.read_text() == "".join(ora_vi.alpha_lines())
    assert ps.read_text() == "".join(ora_vi.sigma_lines())
    np.testing.assert_array_equal(ref_vi.get_true_path_mean([("a", "b")] * 3),
                                  ora_vi.get_true_path_mean([("a", "b")] * 3))
