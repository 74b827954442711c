"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by RUNNING THE REAL REFERENCE.

Run in the build container only (`python -m oracle.make_golden`): the reference's
`var_inf.py` / `node.py` are imported through the in-memory lib2to3 route of
`oracle/ref_loader.py` and driven on seeded synthetic inputs; only inputs and
the reference's outputs (plain arrays) are written.  The reference ships no fixtures of its
own, so these files are what pins the oracle -- and through it the HIP
path -- on the GPU box, where /root/reference does not exist.

Versions used for the committed fixtures are stored inside every file (`meta`).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_loader                                            # noqa: E402
from oracle.vi_oracle import snapshot, make_latent_dict, gaussian_mixture  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _meta(**kw):
    kw.update(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
              generator="oracle/make_golden.py (real reference via lib2to3)")
    return np.asarray(json.dumps(kw))


def _flat(prefix, snap):
    return {prefix + k: v for k, v in snap.items()}


def _leaves(vi):
    return [n for n in vi.get_nodes_list(vi.root, []) if n.isLeafNode]


def trajectory(name, D, fpv, V, thr, ns, epochs, spread, ncomp, seed):
    """Several calls of update_variational_parameters with a new Z per epoch."""
    vi_mod, _ = ref_loader.load_reference(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    N = sum(fpv) if len(fpv) == V else fpv[0] * V
    vi = vi_mod.VarInf()
    out = dict(meta=_meta(kind="trajectory", D=D, thr=thr, num_splits=ns, epochs=epochs),
               fpv=np.asarray(fpv), V=np.asarray(V), N=np.asarray(N))
    rng = np.random.RandomState(seed)
    for ep in range(epochs):
        Z = (gaussian_mixture(N, D, ncomp, seed=seed + ep, spread=spread)
             + 0.05 * rng.normal(size=(N, D))).astype(np.float32)
        lc = make_latent_dict(Z, fpv)
        np.random.seed(13 + ep)                       # src/train.py:37 uses 13
        vi.update_variational_parameters(lc)
        out["Z%d" % ep] = Z
        out.update(_flat("e%d_" % ep, snapshot(vi)))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "K per epoch:", [len(out["e%d_ids" % e]) for e in range(epochs)])


def phases(name, D, fpv, thr, ns, warm_calls, seed):
    """Per-phase outputs (M, gamma, prior tables, logits, phi) on a reshaped (irregular) tree."""
    vi_mod, _ = ref_loader.load_reference(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    V, N = len(fpv), sum(fpv)
    vi = vi_mod.VarInf()
    Z = gaussian_mixture(N, D, 5, seed=seed, spread=3.0)
    lc = make_latent_dict(Z, fpv)
    np.random.seed(seed)
    for _ in range(warm_calls):
        vi.update_variational_parameters(lc)
    out = dict(meta=_meta(kind="phases", D=D, thr=thr, num_splits=ns), fpv=np.asarray(fpv))
    out.update(_flat("in_", snapshot(vi)))
    Z2 = gaussian_mixture(N, D, 5, seed=seed + 1, spread=3.0)
    out["Z"] = Z2
    vi.compute_sigma_alpha(Z2)
    out.update(_flat("m_", snapshot(vi)))
    vi.compute_gamma()
    out.update(_flat("g_", snapshot(vi)))
    on = vi.compute_gamma_sum_on(vi.root, np.zeros(V), {})
    _, before = vi.compute_gamma_sum_before(vi.root, np.zeros(V), {})
    leaves = _leaves(vi)
    out["on"] = np.asarray([on[l] for l in leaves])            # (L, V)
    out["before"] = np.asarray([before[l] for l in leaves])    # (L, V)
    vi.compute_phi_node(vi.root, Z2, on, before)
    out["logits"] = np.concatenate([l.phi for l in leaves], axis=1)
    # finish the softmax exactly as compute_phi does (src/var_inf.py:389-393)
    top = vi.get_phi_min(vi.root)
    vi.exponentiate_phi(vi.root, top)
    tot = vi.get_phi_sum(vi.root)
    vi.normalize_phi(vi.root, tot)
    out["log_phi"] = (out["logits"] - top) - np.log(tot)
    out.update(_flat("e_", snapshot(vi)))
    # split diagnostics on this state (src/var_inf.py:217-220)
    out["leaf_mean_phi"] = np.asarray([np.mean(l.phi) for l in leaves])
    out["leaf_stdev"] = np.asarray([
        np.sqrt(np.linalg.norm(np.sqrt(l.phi) * (Z2 - l.alpha)) / np.sum(l.phi)) for l in leaves])
    split_thr = float(np.median(out["leaf_stdev"][out["leaf_mean_phi"] >= 1e-2]))
    out["split_thr"] = np.asarray(split_thr)
    np.random.seed(seed + 7)
    out["split_ret"] = np.asarray(bool(vi.split_nodes(vi.root, Z2, split_thr)))
    out.update(_flat("s_", snapshot(vi)))
    vi.compute_phi(Z2)
    vi.merge_nodes(vi.root, Z2, split_thr)
    out.update(_flat("x_", snapshot(vi)))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "K in/split/merge:", len(out["in_ids"]), len(out["s_ids"]), len(out["x_ids"]))


def kmeans(name, seed):
    vi_mod, _ = ref_loader.load_reference(latent_code_size=6)
    rng = np.random.RandomState(seed)
    out = dict(meta=_meta(kind="weighted_kmeans"))
    for i, (N, D, K) in enumerate([(200, 6, 4), (64, 3, 2), (33, 10, 5)]):
        data = (rng.normal(size=(N, D)) * 3).astype(np.float32)
        w = rng.uniform(size=N) ** 3
        if i == 1:
            w[::2] = 0.0
        out["data%d" % i], out["w%d" % i], out["K%d" % i] = data, w, np.asarray(K)
        out["centers%d" % i] = vi_mod.VarInf.weighted_kmeans(data, w, K)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name)


def main():
    os.makedirs(OUT, exist_ok=True)
    # the reference's own shape: 4-ary, 85 nodes, D=32 default (src/var_inf.py:88-118, train.py:21)
    trajectory("traj_ref_shape.npz", D=32, fpv=[40], V=8, thr=32.0, ns=2, epochs=3,
               spread=4.0, ncomp=12, seed=21)
    # low threshold -> splits and merges every epoch, ragged videos
    trajectory("traj_dynamic.npz", D=6, fpv=[7, 13, 1, 30, 9, 22], V=6, thr=1.0, ns=3, epochs=4,
               spread=3.0, ncomp=6, seed=31)
    trajectory("traj_single_video.npz", D=4, fpv=[90], V=1, thr=0.5, ns=2, epochs=3,
               spread=3.0, ncomp=4, seed=41)
    phases("phases_irregular.npz", D=5, fpv=[11, 4, 17, 9, 1, 25], thr=0.8, ns=3, warm_calls=2, seed=3)
    phases("phases_ref_shape.npz", D=32, fpv=[33, 31, 40, 16], thr=32.0, ns=2, warm_calls=1, seed=8)
    kmeans("kmeans.npz", seed=0)


if __name__ == "__main__":
    main()
