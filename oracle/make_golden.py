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
from oracle.vi_oracle import snapshot, make_latent_dict, gaussian_mixture, drifting_latents  # noqa: E402

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


def trajectory_config5(name, D, fpv, thr, ns, epochs, spread, ncomp, seed):
    """BASELINE config-5 shape at fixture size: D = 128, num_splits = 4, a threshold low enough that
    the reference splits AND prunes in every epoch (src/var_inf.py:139-147,191-239).  Only the tree
    structure, node parameters, gamma and per-leaf column sums of phi are stored per epoch (phi
    itself is N x L fp64 per epoch)."""
    vi_mod, _ = ref_loader.load_reference(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    N = int(sum(fpv))
    vi = vi_mod.VarInf()
    out = dict(meta=_meta(kind="trajectory_config5", D=D, thr=thr, num_splits=ns, epochs=epochs,
                          spread=spread, ncomp=ncomp, seed=seed),
               fpv=np.asarray(fpv), N=np.asarray(N))
    for ep in range(epochs):
        Z = drifting_latents(N, D, ncomp, seed, ep, spread)
        lc = make_latent_dict(Z, fpv)
        np.random.seed(13 + ep)
        vi.update_variational_parameters(lc)
        snap = snapshot(vi)
        phi = snap.pop("phi")
        snap["phi_colsum"] = phi.sum(axis=0)
        snap["phi_rows"] = phi[:: max(1, N // 64)]          # a few whole rows
        snap["z_checksum"] = np.asarray(float(np.sum(Z.astype(np.float64))))
        out.update(_flat("e%d_" % ep, snap))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "K per epoch:", [len(out["e%d_ids" % e]) for e in range(epochs)])


def _write_tool_inputs(tmp, vi, node_mod, z_train, train_labels, z_test, test_labels):
    """Files in the formats the offline tools read: write_nodes pickle (src/var_inf.py:474-478),
    z files as `vidid frameid z...` rows (src/model.py:267-272), 1-based label files."""
    nodes = os.path.join(tmp, "nodes.pkl")
    sys.modules["node"] = node_mod                 # pickle stores the class by its module name
    try:
        vi.write_nodes(nodes)
    finally:
        sys.modules.pop("node", None)

    def zfile(path, z):
        np.savetxt(path, np.concatenate([np.zeros((len(z), 2)), z], axis=1))
    zfile(os.path.join(tmp, "z_train.txt"), z_train)
    zfile(os.path.join(tmp, "z_test.txt"), z_test)
    np.savetxt(os.path.join(tmp, "train_labels.txt"), np.asarray(train_labels) + 1)
    np.savetxt(os.path.join(tmp, "test_labels.txt"), np.asarray(test_labels) + 1)
    return nodes


def _grown_reference_tree(D, seed):
    """A tree grown by the real reference (splits + merges) to run the offline tools on."""
    vi_mod, node_mod = ref_loader.load_reference(latent_code_size=D, stdev_thr=1.0, num_splits=3)
    fpv = [7, 13, 1, 30, 9, 22]
    N = sum(fpv)
    vi = vi_mod.VarInf()
    for ep in range(3):
        Z = gaussian_mixture(N, D, 6, seed=seed + ep, spread=3.0)
        np.random.seed(13 + ep)
        vi.update_variational_parameters(make_latent_dict(Z, fpv))
    return vi, vi_mod, node_mod


def _tree_arrays(vi):
    snap = snapshot(vi)
    return dict(ids=snap["ids"], parent=snap["parent"], is_leaf=snap["is_leaf"], alpha=snap["alpha"])


def offline_tools(name, seed):
    """Runs the reference's two offline consumers of the tree, UNMODIFIED, on files written in
    their input formats: src/compute_acc_frame_avg_dynamic.py main() (nearest-leaf assignment,
    tree-decayed class vote, :31-146) and src/create_hierarchy.py main() (per-node closest frames,
    :40-59,63-91; its image montage writes into a scratch directory and is discarded).  The
    results are read from the tools' own local variables when main() returns (sys.setprofile) and
    from what create_hierarchy prints per node."""
    import contextlib
    import io
    import pickle
    import tempfile
    import types
    D = 6
    vi, vi_mod, node_mod = _grown_reference_tree(D, seed)
    rng = np.random.RandomState(seed)
    n_test, C = 60, 15
    # create_hierarchy indexes K file names per node without checking (src/create_hierarchy.py:33-35): every
    # leaf must hold at least K frames, so 5 frames are placed next to each leaf mean, the rest is a mixture
    leaf_alpha = np.asarray([n.alpha for n in vi.get_nodes_list(vi.root, []) if n.isLeafNode])
    near = np.repeat(leaf_alpha, 5, axis=0) + 0.01 * rng.normal(size=(5 * len(leaf_alpha), D))
    z_train = np.concatenate([near, gaussian_mixture(60, D, 6, seed=seed + 2, spread=3.0)]).astype(np.float64)
    z_train = z_train[rng.permutation(len(z_train))]
    z_train[10:16] = z_train[0:6]                                   # exact duplicates: distance ties
    n_train = len(z_train)
    z_test = gaussian_mixture(n_test, D, 6, seed=seed + 2, spread=3.0).astype(np.float64) \
        + 0.3 * rng.normal(size=(n_test, D))
    train_labels = np.concatenate([np.arange(C), rng.randint(0, C, size=n_train - C)])
    test_labels = np.concatenate([np.arange(C), rng.randint(0, C, size=n_test - C)])
    out = dict(meta=_meta(kind="offline_tools", D=D), z_train=z_train, z_test=z_test,
               train_labels=train_labels, test_labels=test_labels)
    out.update(_flat("tree_", _tree_arrays(vi)))

    def load_tool(fname):
        src = ref_loader._to_py3(os.path.join(ref_loader.REF_SRC, fname))
        mod = types.ModuleType(fname[:-3])
        saved = sys.modules.get("node")
        sys.modules["node"] = node_mod
        try:
            exec(compile(src, "<reference %s (py3 transform)>" % fname, "exec"), mod.__dict__)
        finally:
            if saved is None:
                sys.modules.pop("node", None)
            else:
                sys.modules["node"] = saved
        return mod

    def run_main(mod, *args):
        got = {}

        def prof(frame, event, arg):
            if event == "return" and frame.f_code.co_name == "main":
                got.update(frame.f_locals)
        buf = io.StringIO()
        sys.modules["node"] = node_mod                               # pickle looks the class up by module name
        sys.setprofile(prof)
        try:
            with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
                mod.main(*args)
        finally:
            sys.setprofile(None)
            sys.modules.pop("node", None)
        return got, buf.getvalue()

    with tempfile.TemporaryDirectory() as tmp:
        nodes = _write_tool_inputs(tmp, vi, node_mod, z_train, train_labels, z_test, test_labels)
        acc = load_tool("compute_acc_frame_avg_dynamic.py")
        # sparse variant: 25 training frames only, so some leaves get no vote and borrow the class of
        # the nearest voted leaf (src/compute_acc_frame_avg_dynamic.py:114-122)
        n_sparse = 25
        out["n_sparse"] = np.asarray(n_sparse)
        np.savetxt(os.path.join(tmp, "z_sparse.txt"),
                   np.concatenate([np.zeros((n_sparse, 2)), z_train[:n_sparse]], axis=1))
        np.savetxt(os.path.join(tmp, "sparse_labels.txt"), np.asarray(train_labels[:n_sparse]) + 1)
        for decay, sparse in ((0.0, False), (0.5, False), (0.0, True), (0.5, True)):
            loc, printed = run_main(acc, nodes, os.path.join(tmp, "sparse_labels.txt" if sparse else "train_labels.txt"),
                                    os.path.join(tmp, "z_sparse.txt" if sparse else "z_train.txt"),
                                    os.path.join(tmp, "test_labels.txt"), os.path.join(tmp, "z_test.txt"), decay)
            tag = "vote%d%s_" % (int(decay * 10), "s" if sparse else "")
            out[tag + "leaf_ids"] = np.asarray(list(loc["leaf_nodes_id_list"]))
            out[tag + "path_assignments"] = np.asarray(loc["path_assignments"], dtype=np.int64)
            out[tag + "counts_total"] = np.asarray(loc["counts_total"])
            out[tag + "path2class"] = np.asarray(loc["path2class"], dtype=np.int64)
            out[tag + "predicted"] = np.asarray(loc["predicted_labels"], dtype=np.int64)
            out[tag + "precision"] = np.asarray(float(printed.strip().splitlines()[-1]))
        # create_hierarchy: needs image files named per frame and ./output_imgs4/ in the cwd
        from PIL import Image
        names = os.path.join(tmp, "filenames.txt")
        with open(names, "w") as f:
            for i in range(n_train):
                path = os.path.join(tmp, "f%03d.jpg" % i)
                Image.new("RGB", (8, 8), (i, i, i)).save(path)
                f.write(path + "\n")
        os.makedirs(os.path.join(tmp, "output_imgs4"))
        hier = load_tool("create_hierarchy.py")
        cwd = os.getcwd()
        for K in (1, 4):
            os.chdir(tmp)
            try:
                loc, printed = run_main(hier, nodes, os.path.join(tmp, "train_labels.txt"),
                                        os.path.join(tmp, "z_train.txt"), os.path.join(tmp, "test_labels.txt"),
                                        os.path.join(tmp, "z_test.txt"), 0.0, names, K)
            finally:
                os.chdir(cwd)
            ids, frames = [], []
            for line in printed.splitlines():
                parts = line.split("\t")
                if len(parts) != 3:
                    continue
                ids.append(parts[0].strip())
                files = eval(parts[2].strip()) if parts[2].strip().startswith("[") else list(parts[2])
                frames.append([int(os.path.basename(x)[1:4]) for x in files] + [-1] * (K - len(files)))
            out["top%d_ids" % K] = np.asarray(ids)
            out["top%d_frames" % K] = np.asarray(frames, dtype=np.int64).reshape(len(ids), K)
            out["top%d_assign" % K] = np.asarray(loc["path_assignments"], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "leaves:", len(out["vote0_leaf_ids"]), "precision:", float(out["vote0_precision"]),
          float(out["vote5_precision"]), "nodes ranked:", len(out["top4_ids"]))


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
    trajectory_config5("traj_config5_shape.npz", D=128, fpv=[300, 177, 256, 640, 411, 160, 512, 616],
                       thr=-98.9, ns=4, epochs=5, spread=3.0, ncomp=24, seed=62)
    offline_tools("offline_tools.npz", seed=5)


if __name__ == "__main__":
    main()
