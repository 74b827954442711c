"""GPU parity tests added in round 2 (all through the C ABI): the configurations and entry points
round 1 left unpinned -- a BASELINE config-5-shaped dynamic trajectory against the real reference's
output, the R^T Z^2 statistic (want_q) and the split test that runs on it, bnpvi_gamma_prior, the
text writers, and the consumer side against the reference's offline tools.

Tolerances are BASELINE.json's: node parameters rel 1e-4, log-responsibilities abs 1e-4, hard
assignments equal except at near-ties (|dlogp| < 1e-5).
"""
import os

import numpy as np
import pytest

from oracle.vi_oracle import (OracleVarInf, snapshot, make_latent_dict, drifting_latents, tree_from_arrays)
from tests.test_gpu_parity import (load, sub, meta, make_vi, close_params, close_sigma, oracle_from_snapshot, _regular_case,
                                   PARAM_RTOL, PHI_ATOL, LOGP_ATOL)

pytestmark = pytest.mark.gpu


def test_config5_shape_trajectory_against_golden():
    """D = 128, num_splits = 4, five epochs in which the reference both splits and prunes
    (tests/golden/traj_config5_shape.npz, written by the real reference): same tree after every epoch,
    parameters within tolerance.  Reference: src/var_inf.py:139-147, 213-239, 191-211."""
    g = load("traj_config5_shape.npz")
    m = meta(g)
    fpv, N = list(g['fpv']), int(g['N'])
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    for ep in range(m['epochs']):
        Z = drifting_latents(N, m['D'], m['ncomp'], m['seed'], ep, m['spread'])
        want = sub(g, 'e%d_' % ep)
        assert float(np.sum(Z.astype(np.float64))) == float(want['z_checksum'])
        np.random.seed(13 + ep)
        vi.update_variational_parameters(make_latent_dict(Z, fpv))
        got = snapshot(vi)
        assert list(got['ids']) == list(want['ids']), "tree differs after epoch %d" % ep
        np.testing.assert_array_equal(got['is_leaf'], want['is_leaf'])
        np.testing.assert_array_equal(got['parent'], want['parent'])
        close_params(got['alpha'], want['alpha'], 'alpha, epoch %d' % ep)
        close_sigma(got['sigma'], want['sigma'], 'sigma, epoch %d' % ep)
        np.testing.assert_allclose(got['gamma'], want['gamma'], rtol=PARAM_RTOL, atol=1e-3, equal_nan=True)
        np.testing.assert_allclose(got['phi'].sum(axis=0), want['phi_colsum'], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(got['phi'][:: max(1, N // 64)], want['phi_rows'], rtol=0, atol=2e-4)
        assert float(got['decay']) == float(want['decay'])


@pytest.mark.parametrize("N,D,L,fpv", [
    (30_000, 64, 64, 256),       # config-2 shape: (1,2) tiles with Q
    (9_001, 32, 25, 100),        # config-1 shape, ragged N: (1,1) tiles with Q
    (5_000, 200, 343, 77),       # padded D and L
])
def test_leaf_stats_with_q_against_fp64(N, D, L, fpv):
    """bnpvi_leaf_stats(want_q=1): [S0 | S1 | Q = R^T Z^2] against torch fp64 (src/var_inf.py:219-220 needs
    sum phi ||z - alpha||^2, which the host forms from Q, S1 and S0)."""
    import torch
    from bnp_vae_amd.engine import DeviceVI
    parent = np.concatenate([[-1], np.zeros(L, np.int32)]).astype(np.int32)
    is_leaf = np.concatenate([[0], np.ones(L, np.uint8)]).astype(np.uint8)
    dev = DeviceVI(device=0)
    g = torch.Generator(device=dev.device)
    g.manual_seed(11)
    Z = torch.randn((N, D), generator=g, device=dev.device) * 3.0 + 1.0
    vid = (torch.arange(N, device=dev.device) // fpv).to(torch.int32)
    V = int(vid[-1]) + 1
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.zeros((L + 1, D)), np.ones(L + 1))
    dev.alloc_phi()
    dev.phi[:, :L] = torch.softmax(torch.randn((N, L), generator=g, device=dev.device) * 3.0, dim=1)
    st = dev.leaf_stats(want_q=True, want_mass=True).clone()
    torch.cuda.synchronize()
    assert tuple(st.shape) == (L, 1 + 2 * D)
    phi64, z64 = dev.phi[:, :L].to(torch.float64), Z.to(torch.float64)
    s0, s1, q = phi64.sum(dim=0), phi64.T @ z64, phi64.T @ (z64 * z64)
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), s0.cpu().numpy(), rtol=2e-6, atol=1e-6)
    assert float((st[:, 1:1 + D] - s1).abs().max()) <= 2e-6 * float(s1.abs().max())
    assert float((st[:, 1 + D:] - q).abs().max()) <= 2e-6 * float(q.abs().max())
    mass = torch.zeros((V, L), dtype=torch.float64, device=dev.device).index_add_(0, vid.long(), phi64)
    np.testing.assert_allclose(dev.mass.cpu().numpy(), mass.cpu().numpy(), rtol=2e-6, atol=1e-7)
    # the statistics without Q (the LDS-DMA kernel) agree with the Q kernel's S0 | S1
    st2 = dev.leaf_stats(want_q=False, want_mass=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(st2.cpu().numpy(), st[:, :1 + D].cpu().numpy(), rtol=1e-6, atol=1e-6 * float(s1.abs().max()))
    dev.close()


def test_split_decision_from_q_statistics():
    """split_nodes on responsibilities that did NOT come from the last E-step (a caller assigned
    Node.phi): the stale-statistics route recomputes S0 / S2 from one bnpvi_leaf_stats(want_q=1)
    pass (bnp_vae_amd/var_inf.py:_leaf_moments).  Same stdev per leaf and the same split decisions,
    children and k-means centres as the oracle (src/var_inf.py:213-239)."""
    N, D, bf, levels = 3000, 16, 3, 2
    ora, vi, Z = _regular_case(N, D, bf, levels, [300, 177], seed=5)
    ora.num_splits = vi.num_splits = 3
    for it in range(2):
        ora.compute_sigma_alpha(Z); vi.compute_sigma_alpha(Z)
        ora.compute_gamma(); vi.compute_gamma()
        ora.compute_phi(Z); vi.compute_phi(Z)
    # overwrite every leaf's phi with a smoothed copy of the oracle's (a caller-side assignment)
    po = snapshot(ora)['phi']
    smooth = 0.7 * po + 0.3 / po.shape[1]
    for leaf, col in zip(ora.leaves(), smooth.T):
        leaf.phi = col.reshape(-1, 1).copy()
    for leaf, col in zip(vi._dfs_leaves(), smooth.T):
        leaf.phi = col.reshape(-1, 1).copy()
    S0, S2 = vi._leaf_moments()
    assert vi._dev.want_q, "the stale-statistics route must have run the Q kernel"
    want_sd = np.asarray([ora.leaf_stdev(l.phi, Z, l.alpha) for l in ora.leaves()])
    got_sd = np.sqrt(np.sqrt(S2) / S0)
    np.testing.assert_allclose(got_sd, want_sd, rtol=1e-4)
    np.testing.assert_allclose(S0, smooth.sum(axis=0), rtol=1e-5)
    thr = float(0.5 * (np.sort(want_sd)[len(want_sd) // 2 - 1] + np.sort(want_sd)[len(want_sd) // 2]))
    np.random.seed(3)
    r_o = ora.split_nodes(ora.root, Z, thr)
    np.random.seed(3)
    r_p = vi.split_nodes(vi.root, Z, thr)
    assert bool(r_o) == bool(r_p) and bool(r_o)
    so, sp = snapshot(ora), snapshot(vi)
    assert list(sp['ids']) == list(so['ids'])
    np.testing.assert_array_equal(sp['is_leaf'], so['is_leaf'])
    close_params(sp['alpha'], so['alpha'], 'alpha after split (k-means centres)')
    np.testing.assert_allclose(sp['phi'], so['phi'], rtol=0, atol=PHI_ATOL)


def test_gamma_prior_entry_point():
    """bnpvi_gamma_prior (gamma-step + prior table in one launch) == bnpvi_gamma_update followed by
    bnpvi_prior_table, and both match the oracle's gamma (src/var_inf.py:395-416) and its on + before
    tables (:359-381) up to the per-row shift the table is stored with."""
    import torch
    N, D, bf, levels = 4000, 32, 4, 3
    ora, vi, Z = _regular_case(N, D, bf, levels, [500, 123], seed=17)
    for leaf, col in zip(vi._dfs_leaves(), snapshot(ora)['phi'].T):
        leaf.phi = col.reshape(-1, 1)
    ora.compute_sigma_alpha(Z); vi.compute_sigma_alpha(Z)
    dev = vi._dev
    dev.leaf_stats(want_q=False, want_mass=True)
    mass = dev.mass.clone()
    dev.gamma_update()
    dev.prior_table()
    torch.cuda.synchronize()
    g_sep, p_sep = dev.gamma.clone(), dev.prior.clone()
    dev.gamma.zero_(); dev.prior.zero_()
    dev.mass.copy_(mass)
    dev.gamma_prior()
    torch.cuda.synchronize()
    np.testing.assert_allclose(dev.gamma.cpu().numpy(), g_sep.cpu().numpy(), rtol=1e-12, atol=1e-12)
    fin = torch.isfinite(p_sep)
    assert torch.equal(fin, torch.isfinite(dev.prior))
    np.testing.assert_allclose(dev.prior[fin].cpu().numpy(), p_sep[fin].cpu().numpy(), rtol=0, atol=2e-6)
    # against the oracle
    ora.compute_gamma()
    want_gamma = snapshot(ora)['gamma']                                    # (K, V, 2)
    np.testing.assert_allclose(np.transpose(dev.gamma.cpu().numpy(), (1, 0, 2))[1:], want_gamma[1:], rtol=1e-4, atol=1e-4)
    on, before = ora.prior_tables()
    leaves = ora.leaves()
    L = len(leaves)
    tab = np.asarray([on[l] + before[l] - 0.5 / l.sigmasqr_inv for l in leaves]).T      # (V, L), natural log
    tab = (tab - tab.max(axis=1, keepdims=True)) * np.log2(np.e)                         # stored shifted, log2 units
    np.testing.assert_allclose(dev.prior.cpu().numpy()[:, :L], tab, rtol=0, atol=1e-4)


def test_writer_text_equals_oracle_lines(tmp_path):
    """write_alpha / write_sigma (src/var_inf.py:439-463): the files are, character for character,
    the oracle's lines for the same node values (2-decimal rounding through str(np.round(x, 2)),
    pre-order, `node_id v1 v2 ...`)."""
    g = load("traj_dynamic.npz")
    m = meta(g)
    fpv = list(g['fpv'])
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    for ep in range(2):
        np.random.seed(13 + ep)
        vi.update_variational_parameters(make_latent_dict(g['Z%d' % ep], fpv))
    vi.write_alpha(str(tmp_path / "alpha.txt"))
    vi.write_sigma(str(tmp_path / "sigma.txt"))
    ora = oracle_from_snapshot(snapshot(vi), m['D'], m['thr'], m['num_splits'], fpv)
    assert (tmp_path / "alpha.txt").read_text() == "".join(ora.alpha_lines())
    assert (tmp_path / "sigma.txt").read_text() == "".join(ora.sigma_lines())
    # and the values themselves are the golden trajectory's (to the parameter tolerance)
    close_params(snapshot(vi)['alpha'], g['e1_alpha'], 'alpha')


def _offline_nodes(g):
    return tree_from_arrays(g['tree_ids'], g['tree_parent'], g['tree_is_leaf'], g['tree_alpha'])


@pytest.mark.parametrize("K", [1, 4])
def test_closest_frames_against_reference_tool(K):
    """hierarchy.closest_frames against what src/create_hierarchy.py printed per node when run
    unmodified (tests/golden/offline_tools.npz), duplicated frames (ties) included."""
    from bnp_vae_amd.hierarchy import closest_frames
    g = load("offline_tools.npz")
    nodes = _offline_nodes(g)
    assign, top = closest_frames(nodes, g['z_train'].astype(np.float32), K)
    np.testing.assert_array_equal(assign, g['top%d_assign' % K])
    by_id = {n.node_id: n for n in nodes}
    for nid, frames in zip(g['top%d_ids' % K], g['top%d_frames' % K]):
        want = [int(f) for f in frames if f >= 0]
        got = top[by_id[str(nid)]]
        if got != want:          # fp32 latents on the device: only exact-distance ties may reorder
            z = g['z_train']
            d = np.linalg.norm(by_id[str(nid)].alpha - z, axis=1)
            np.testing.assert_allclose(np.sort(d[got]), np.sort(d[want]), rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag,decay,sparse", [("vote0_", 0.0, False), ("vote5_", 0.5, False),
                                              ("vote0s_", 0.0, True), ("vote5s_", 0.5, True)])
def test_class_vote_against_reference_tool(tag, decay, sparse):
    """hierarchy.class_vote against src/compute_acc_frame_avg_dynamic.py run unmodified: the
    path -> class map and the predicted class of every test frame."""
    from bnp_vae_amd.hierarchy import class_vote
    g = load("offline_tools.npz")
    nodes = _offline_nodes(g)
    n = int(g['n_sparse']) if sparse else len(g['z_train'])
    p2c, pred = class_vote(nodes, g['z_train'][:n].astype(np.float32), g['train_labels'][:n],
                           g['z_test'].astype(np.float32), decay_factor=decay)
    np.testing.assert_array_equal(p2c, g[tag + 'path2class'])
    np.testing.assert_array_equal(pred, g[tag + 'predicted'])


@pytest.mark.parametrize("cfg,N", [("config2", 999_936), ("config3", 262_144), ("config1", 60_000)])
def test_s0_is_exact_on_one_hot_responsibilities(cfg, N):
    """With one-hot phi every partial sum of S0 is an integer below 2^24, so S0 must equal the
    per-leaf frame counts bit for bit and add up to N exactly -- a deterministic check of the S0
    accumulation of the M-step kernels (mstep_dma.hip, mstep_wide.hip), which once lost k-steps."""
    import torch
    from bench import CONFIGS, regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    _n, D, bf, levels = CONFIGS[cfg]
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    _, assign = dev.estep(mode=_lib.ESTEP_HARD, want_stats=False, want_assign=True)
    counts = torch.bincount(assign.to(torch.int64), minlength=L).to(torch.float64)
    for _ in range(20):
        dev.leaf_stats(want_q=False, want_mass=True)
        assert torch.equal(dev.stats[:, 0], counts)
    assert float(dev.stats[:, 0].sum()) == float(N)
    assert torch.equal(dev.mass.sum(dim=0), counts)
    dev.close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("golden,world", [("traj_dynamic.npz", 2), ("traj_dynamic.npz", 3),
                                          ("traj_config5_shape.npz", 2)])
def test_multi_rank_trajectory_matches_single_rank_golden(golden, world, tmp_path):
    """The product with a process group, end to end: `world` fresh processes (all on cuda:0, gloo with
    host-staged collectives -- the one-GPU rehearsal of the one-process-per-GPU RCCL layout), frames
    sharded on video boundaries (dist.shard_videos), VarInf(group=...) run over the golden trajectory.
    Every rank must hold the golden's tree and node parameters after every epoch (the tree is
    replicated; alpha starts from rank 0's draws), its own rows of phi and its own videos' gamma.
    Covers the statistics all-reduce, the split/merge statistics reduce, the global frame count and
    the arg-max exchange + row broadcast of weighted k-means."""
    import subprocess
    import sys
    g = load(golden)
    m = meta(g)
    port = str(_free_port())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_worker.py"), str(r), str(world),
                               port, str(tmp_path), golden], cwd=root) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    _ranks_hold_the_golden(g, m, tmp_path, world)


def test_rccl_collectives_on_a_single_rank_group(tmp_path):
    """The RCCL branch of bnp_vae_amd/dist.py on the hardware this suite has: ONE rank, process group "nccl"
    (= RCCL), BNPVI_DIST_FORCE_COLLECTIVES=1 so that the group of one still runs every collective of the path on
    DEVICE tensors -- the fp64 all-reduces of the leaf and split/merge statistics and of the centre table, the
    all-gathers of the frame count and of the [J, 2] arg-max table, the broadcast of the initial means -- in the
    stream order the multi-GPU run has.  Sums and gathers over one rank change nothing, so the rank must hold the
    single-rank golden.  (More than one rank needs more than one GPU: two ranks cannot share a device under RCCL.)"""
    import subprocess
    import sys
    golden = "traj_dynamic.npz"
    g = load(golden)
    m = meta(g)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BNPVI_WORKER_BACKEND="nccl", BNPVI_DIST_FORCE_COLLECTIVES="1")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_worker.py"), "0", "1", str(_free_port()),
                        str(tmp_path), golden], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode == 77:
        pytest.skip("RCCL could not set up a one-rank communicator on this machine: " + p.stdout[-300:])
    assert p.returncode == 0, p.stderr[-3000:]
    assert "collectives run: " in p.stdout and "backend nccl" in p.stdout, p.stdout[-2000:]
    counts = dict(kv.split("=") for kv in p.stdout.split("collectives run: ")[1].split()[:3])
    assert int(counts["all_reduce"]) >= 2 * m['epochs'] and int(counts["all_gather"]) >= 1 and int(counts["broadcast"]) >= 1
    _ranks_hold_the_golden(g, m, tmp_path, 1)


def _ranks_hold_the_golden(g, m, tmp_path, world):
    N = None
    for r in range(world):
        got = dict(np.load(str(tmp_path / ("rank%d.npz" % r)), allow_pickle=False))
        f0, f1, v0, v1 = int(got['f0']), int(got['f1']), int(got['v0']), int(got['v1'])
        for ep in range(m['epochs']):
            want = sub(g, 'e%d_' % ep)
            have = sub(got, 'e%d_' % ep)
            assert list(have['ids']) == list(want['ids']), "rank %d epoch %d: tree differs" % (r, ep)
            np.testing.assert_array_equal(have['is_leaf'], want['is_leaf'])
            close_params(have['alpha'], want['alpha'], 'alpha rank %d epoch %d' % (r, ep))
            close_sigma(have['sigma'], want['sigma'], 'sigma rank %d epoch %d' % (r, ep))
            assert float(have['decay']) == float(want['decay'])
            np.testing.assert_allclose(have['gamma'], want['gamma'][:, v0:v1], rtol=PARAM_RTOL, atol=1e-3, equal_nan=True)
            if 'phi' in want:
                np.testing.assert_allclose(have['phi'], want['phi'][f0:f1], rtol=0, atol=2e-4)
            else:
                N = int(g['N'])
                step = max(1, N // 64)
                rows = [i for i in range(0, N, step) if f0 <= i < f1]
                np.testing.assert_allclose(have['phi'][[i - f0 for i in rows]],
                                           want['phi_rows'][[i // step for i in rows]], rtol=0, atol=2e-4)


@pytest.mark.parametrize("N,D,bf,levels", [
    (4000, 64, 8, 2),      # L=64: fused kernel
    (3000, 64, 8, 3),      # L=512: two-pass row kernel
    (2500, 32, 5, 2),      # D=32: the row group's lanes 8-15 hold no part of a row
    (1500, 256, 6, 2),     # D=256: four float4 per lane and row
    (1800, 128, 4, 3),     # the reference tree's shape
])
def test_tight_far_clusters_take_the_exact_distance_path(N, D, bf, levels):
    """Tight clusters far from the origin: ||z||^2 ~ 64 * 32^2 while a frame is ~4 away from its own leaf, so the
    fp32 expansion ||z||^2 - 2 z.a + ||a||^2 cannot give that distance to the 2e-5 the log-responsibilities
    need -- EVERY frame's own leaf (and, in places, neighbouring leaves: several flagged pairs per lane and per
    16-lane row group) goes through the direct-difference recomputation.  log phi must still match fp64."""
    ora, vi, _ = _regular_case(N, D, bf, levels, [256], seed=5)
    oleaves, pleaves = ora.leaves(), vi._dfs_leaves()
    L = len(oleaves)
    rng = np.random.RandomState(3)
    centres = 32.0 * rng.normal(size=(L, D))
    # a few leaves close together (distance ~ 32 * 2): their neighbours are flagged as well
    centres[1::7] = centres[0::7][:len(centres[1::7])] + 64.0 * rng.normal(size=(len(centres[1::7]), D)) / np.sqrt(D)
    Z = (centres[np.arange(N) % L] + 0.5 * rng.normal(size=(N, D))).astype(np.float32)
    for k, (lo_, lp) in enumerate(zip(oleaves, pleaves)):
        lo_.alpha = centres[k].copy()
        lp.alpha = centres[k].copy()
    ora.compute_phi(Z)
    logphi, _ = vi.compute_phi(Z, want_logphi=True)
    got = logphi.cpu().numpy()[:, :L]
    want = ora.last_log_phi
    # the own leaf and everything within reach of the tolerance's meaning: entries below -80 are exp-underflow
    # territory on both sides, compare them only loosely
    near = want > -80.0
    np.testing.assert_allclose(got[near], want[near], rtol=0, atol=LOGP_ATOL)
    np.testing.assert_allclose(got[~near], want[~near], rtol=1e-5, atol=LOGP_ATOL)
    np.testing.assert_allclose(snapshot(vi)['phi'], snapshot(ora)['phi'], rtol=0, atol=PHI_ATOL)
    # statistics of the same E-step: S2 = sum phi ||z - alpha||^2 is dominated by the refined distances
    S0, S2 = vi._leaf_moments()
    po = snapshot(ora)['phi']
    d2 = ((Z[:, None, :].astype(np.float64) - centres[None]) ** 2).sum(-1)
    np.testing.assert_allclose(S0, po.sum(0), rtol=1e-4, atol=1e-4)
    from tests.test_gpu_parity import check_gated_s2
    check_gated_s2(S0, S2, (po * d2).sum(0), len(Z))


@pytest.mark.parametrize("N,D,L,K,sharp", [
    (3000, 32, 20, 3, 1.0),       # D=32: 8 lanes share a row -> 20 columns go in three batches
    (2500, 64, 40, 4, 1.0),       # 16 lanes per row, three batches
    (2000, 128, 33, 4, 1.0),      # 32 lanes per row (crosses a DPP row), two batches
    (1500, 200, 70, 2, 1.0),      # D=200 -> 256: one wave per row, two batches
    (4000, 64, 12, 4, 40.0),      # saturated: most frames have fp32 phi == 1.0 on their leaf
    (4000, 128, 9, 3, 25.0),
    (777, 48, 5, 6, 3.0),         # ragged N, D=48 -> 64, more centres than columns
])
def test_batched_kmeans_seeding_against_oracle(N, D, L, K, sharp):
    """`DeviceVI.kmeans_seed_multi` (all splitting leaves of one split_nodes walk seeded together, one pass
    over Z per centre index) against the oracle's weighted_kmeans (src/var_inf.py:23-36) run column by column
    on the fp64 responsibilities the fp32 device matrix was rounded from.  `sharp` scales the logits: at
    25-40 most frames sit at phi = 1.0 in fp32 and the first centre is decided by the fp64 tie rule."""
    import torch
    from bnp_vae_amd.engine import DeviceVI
    rng = np.random.RandomState(N + D)
    Z = rng.normal(size=(N, D)).astype(np.float32)
    logits = sharp * rng.normal(size=(N, L))
    phi64 = np.exp(logits - logits.max(1, keepdims=True))
    phi64 /= phi64.sum(1, keepdims=True)
    dev = DeviceVI(device=0)
    try:
        parent = np.asarray([-1] + [0] * L, dtype=np.int32)
        is_leaf = np.asarray([0] + [1] * L, dtype=np.uint8)
        dev.set_latents(Z, np.zeros(N, dtype=np.int32), 1)
        dev.set_tree(parent, is_leaf, np.zeros((L + 1, D)), np.ones(L + 1))
        dev.alloc_phi()
        dev.phi.zero_()
        dev.phi[:, :L].copy_(torch.from_numpy(phi64.astype(np.float32)))
        if sharp >= 25.0:
            sat = (dev.phi[:, :L] >= 1.0 - 2e-7).sum(0).cpu().numpy()
            assert (sat >= 2).sum() >= L // 2, "the case is meant to have tied fp32 maxima"
        cols = list(range(L))
        rows, centres = dev.kmeans_seed_multi(cols, K)
        single_rows, single_centres = dev.kmeans_seed(L // 2, K)
    finally:
        dev.close()
    assert np.asarray(centres).shape == (L, K, D)
    for j in cols:
        want = OracleVarInf.weighted_kmeans(Z.astype(np.float64), phi64[:, j], K)
        np.testing.assert_array_equal(centres[j], want, err_msg="column %d" % j)
        np.testing.assert_array_equal(Z[rows[j]].astype(np.float64), want)
    assert single_rows == rows[L // 2]
    np.testing.assert_array_equal(single_centres, centres[L // 2])


def _bare_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (the driver's N > 1 form without a launcher): the parent
    starts two fresh ranks before anything touches the GPU.  On a one-GPU box this runs as an explicit
    rehearsal (BNPVI_BENCH_REHEARSAL=1: both ranks on cuda:0 over gloo) whose record says n_gpus = 1 and
    value = null; without the switch the command refuses instead of passing a one-GPU run off as two.
    Checks the one-JSON-line contract, not the numbers."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    one_gpu = torch.cuda.device_count() < 2
    if one_gpu:
        refused = subprocess.run(cmd, cwd=root, env=_bare_env(), capture_output=True, text=True, timeout=110)
        assert refused.returncode != 0 and "BNPVI_BENCH_REHEARSAL" in refused.stderr
        assert not [l for l in refused.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run(cmd, cwd=root, env=_bare_env(**({"BNPVI_BENCH_REHEARSAL": "1"} if one_gpu else {})),
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["ranks"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["unit"] == "samples/s" and rec["higher_is_better"] is True and rec["scaling"] == "weak"
    value = rec["rehearsal_value"] if one_gpu else rec["value"]
    assert value > 0 and abs(value - 2 * 999936 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * value
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rec["roofline"]
    assert "cpu_baseline" not in rec           # rank 0 at N = 1 only
    strong = rec["config3_strong"]
    assert strong["scaling"] == "strong" and strong["frames_per_rank"] * 2 == strong["frames_total"]
    assert strong["allreduce_bytes_per_step"] == 512 * 65 * 8 + 512 * 2 * 8
    assert rec["config"]["ranks_seen"] == 2 and strong["ranks_seen"] == 2
    if one_gpu:
        assert rec["n_gpus"] == 1 and rec["value"] is None and strong["value"] is None
        assert rec["config"]["rehearsal_on_one_gpu"] is True and rec["config"]["collective_backend"] == "gloo"
        assert strong["collective_backend"] == "gloo"
    else:
        assert rec["n_gpus"] == 2 and rec["config"]["collective_backend"] == "nccl"


def test_bench_watchdog_ends_all_ranks_when_one_fails():
    """A rank that dies before the rendezvous (BNPVI_BENCH_FAIL_RANK) must not leave the others waiting: the
    parent notices the first non-zero exit, terminates the rest, exits non-zero within seconds, prints no JSON
    line and keeps every rank's stderr in a file of its own."""
    import subprocess
    import sys
    import tempfile
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as logdir:
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"],
                             cwd=root, env=_bare_env(BNPVI_BENCH_REHEARSAL="1", BNPVI_BENCH_FAIL_RANK="1",
                                                     BNPVI_BENCH_LOGDIR=logdir),
                             capture_output=True, text=True, timeout=110)
        took = time.time() - t0
        assert out.returncode != 0
        assert took < 60, took
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert "rank 1 exited with code" in out.stderr and "failure injected" in out.stderr
        assert sorted(os.listdir(logdir)) == ["rank%d.%s" % (r, x) for r in range(3) for x in ("err", "out")]
        assert "failure injected" in open(os.path.join(logdir, "rank1.err")).read()


def test_c_abi_argument_checks_of_round_two():
    """Error behaviour of the entry points touched in round 2: too many k-means columns per call, the frame-count
    bound, and the zeroed-mass bookkeeping of bnpvi_params_prior (a mass buffer of another size is memset
    again; the recorded one is not)."""
    import ctypes as C
    import torch
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI, _ptr
    N, D, bf, levels = 4096, 32, 4, 2
    rng = np.random.RandomState(0)
    dev = DeviceVI(device=0)
    try:
        parent, is_leaf = [-1], [0]          # regular tree in pre-order
        def build(depth):
            me = len(parent) - 1
            if depth == levels:
                is_leaf[me] = 1
                return
            for _c in range(bf):
                parent.append(me); is_leaf.append(0)
                build(depth + 1)
        build(0)
        K, L = len(parent), int(sum(is_leaf))
        Z = rng.normal(size=(N, D)).astype(np.float32)
        vid = (np.arange(N) // 256).astype(np.int32)
        V = int(vid[-1]) + 1
        dev.set_latents(Z, vid, V)
        dev.set_tree(np.asarray(parent, np.int32), np.asarray(is_leaf, np.uint8), rng.normal(size=(K, D)), np.ones(K))
        dev.default_gamma()
        dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
        dev.iteration()
        torch.cuda.synchronize()
        # (1) more columns than lanes share a row
        jmax = int(_lib.lib.bnpvi_kmeans_max_columns(D))
        assert jmax == 8
        cols = torch.zeros((jmax + 1,), dtype=torch.int32, device=dev.device)
        rows = torch.zeros((jmax + 1,), dtype=torch.int64, device=dev.device)
        score = torch.zeros((jmax + 1,), dtype=torch.float64, device=dev.device)
        rc = _lib.lib.bnpvi_kmeans_pass_multi(dev.ctx, _ptr(dev.Z), dev.Dp, N, D, _ptr(dev.phi), int(dev.phi.shape[1]),
                                              L, _ptr(cols), jmax + 1, None, 1, None, _ptr(rows), _ptr(score), None)
        assert rc != 0 and b"columns" in _lib.lib.bnpvi_last_error()
        # (2) frame-count bound
        rc = _lib.lib.bnpvi_leaf_stats(dev.ctx, _ptr(dev.Z), dev.Dp, _ptr(dev.phi), int(dev.phi.shape[1]), _ptr(dev.vid),
                                       1 << 33, D, L, V, 0, _ptr(dev.stats), _ptr(dev.mass), None)
        assert rc != 0 and b"2^33" in _lib.lib.bnpvi_last_error()
        # (3) after params_prior the recorded mass buffer is known to be zero; a buffer of another size
        #     (here: fewer videos, carved out of the same allocation, deliberately dirty) is zeroed again
        dev.leaf_stats(want_q=False, want_mass=True)
        ref_mass = dev.mass.clone()
        dev.params_prior()                                        # consumes dev.mass, records it as zeroed
        torch.cuda.synchronize()
        assert float(dev.mass.abs().max()) == 0.0
        dev.mass.fill_(7.0)                                       # a caller scribbles over it ...
        half = N // 2
        Vh = int(vid[half - 1]) + 1
        small = dev.mass[:Vh]                                     # ... and passes the same pointer with a smaller V
        _lib.check(_lib.lib.bnpvi_leaf_stats(dev.ctx, _ptr(dev.Z), dev.Dp, _ptr(dev.phi), int(dev.phi.shape[1]),
                                             _ptr(dev.vid), half, D, L, Vh, 0, _ptr(dev.stats), _ptr(small), None))
        torch.cuda.synchronize()
        np.testing.assert_allclose(small.cpu().numpy(), ref_mass[:Vh].cpu().numpy(), rtol=1e-12, atol=1e-12)
    finally:
        dev.close()
