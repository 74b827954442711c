"""World-size-2 gloo tests (CPU) of the N>1 path's host logic (bnp_vae_amd/dist.py): video-
boundary sharding, the single all-reduce of the per-leaf statistics, the global frame count /
row offsets, and the arg-max exchange of weighted k-means -- checked against the CPU oracle on
the full data.  The kernels themselves need a GPU and are covered by tests/test_gpu_parity.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bnp_vae_amd import dist as vdist
from oracle.vi_oracle import OracleVarInf, make_latent_dict, gaussian_mixture


def test_shard_videos_properties():
    rng = np.random.RandomState(0)
    for world in (1, 2, 3, 8):
        for _ in range(20):
            fpv = rng.randint(1, 300, size=rng.randint(1, 40))
            sh = vdist.shard_videos(fpv, world)
            assert len(sh) == world
            assert sh[0][0] == 0 and sh[0][2] == 0
            assert sh[-1][1] == len(fpv) and sh[-1][3] == int(fpv.sum())
            for (a0, a1, f0, f1), (b0, b1, g0, g1) in zip(sh, sh[1:]):
                assert a1 == b0 and f1 == g0 and a0 <= a1
            for v0, v1, f0, f1 in sh:
                assert f1 - f0 == int(fpv[v0:v1].sum())          # cuts on video boundaries only
    # balanced when videos are equal-sized
    sh = vdist.shard_videos([256] * 64, 8)
    assert all(f1 - f0 == 256 * 8 for _v0, _v1, f0, f1 in sh)
    # fewer videos than ranks: some ranks are empty, nothing is lost
    sh = vdist.shard_videos([10, 20], 4)
    assert sum(f1 - f0 for *_v, f0, f1 in sh) == 30


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD
    try:
        N, D, bf, levels = 3000, 16, 3, 2
        fpv_len = 170
        Z = gaussian_mixture(N, D, bf ** levels, seed=5, spread=4.0)
        lc = make_latent_dict(Z, [fpv_len])
        np.random.seed(5)
        ora = OracleVarInf(latent_code_size=D, branching=bf, levels=levels, stdev_thr=1e9)
        Zm = ora.get_matrix_from_dict(lc)
        ora.build_initial_tree(); ora.build_video_index(); ora.one_hot_init(Zm)
        for _ in range(2):                       # soft phi
            ora.compute_sigma_alpha(Zm); ora.compute_gamma(); ora.compute_phi(Zm)
        leaves = ora.leaves()
        phi = np.concatenate([l.phi for l in leaves], axis=1)             # (N, L) fp64
        parents_alpha = np.asarray([l.parent.alpha for l in leaves])      # OLD parent alpha
        fpv = [len(v) for v in lc.values()]

        # --- sharding + the single all-reduce of [S0 | S1]
        shards = vdist.shard_videos(fpv, world)
        v0, v1, f0, f1 = shards[rank]
        Zl, pl = Zm[f0:f1].astype(np.float64), phi[f0:f1]
        stats = torch.from_numpy(np.concatenate([pl.sum(0)[:, None], pl.T @ Zl], axis=1))
        vdist.allreduce_sum_(stats, group)
        full = np.concatenate([phi.sum(0)[:, None], phi.T @ Zm.astype(np.float64)], axis=1)
        np.testing.assert_allclose(stats.numpy(), full, rtol=1e-12, atol=1e-9)

        # every rank derives the same node parameters from the reduced buffer (src/var_inf.py:265-297)
        S0, S1 = stats.numpy()[:, 0], stats.numpy()[:, 1:]
        sig = 1.0 + S0
        alpha = (parents_alpha * 1.0 + S1) / sig[:, None]
        ora.compute_sigma_alpha(Zm)
        np.testing.assert_allclose(alpha, np.asarray([l.alpha for l in ora.leaves()]), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(sig, [l.sigmasqr_inv for l in ora.leaves()], rtol=1e-12)

        # --- global frame count and row offsets
        total, off = vdist.global_count(f1 - f0, group, torch.device("cpu"))
        assert total == N and off == f0

        # --- arg-max exchange (weighted k-means): best score wins, lowest global row on ties,
        #     empty ranks lose, all-zero scores give global row 0
        w = phi[:, 0]
        lr = int(np.argmax(w[f0:f1]))
        grow, owner = vdist.select_argmax(float(w[f0 + lr]), f0 + lr, group, torch.device("cpu"))
        assert grow == int(np.argmax(w))
        assert shards[owner][2] <= grow < shards[owner][3]
        grow, owner = vdist.select_argmax(0.5, 100 - 90 * rank, group, torch.device("cpu"))       # tie
        assert (grow, owner) == (10, 1)
        grow, owner = vdist.select_argmax(-1.0 if rank == 0 else 0.25, 7 + rank, group, torch.device("cpu"))
        assert (grow, owner) == (8, 1)
        grow, owner = vdist.select_argmax(0.0, f0, group, torch.device("cpu"))
        assert (grow, owner) == (0, 0)
        c = torch.full((4,), float(rank))
        vdist.broadcast_from(c, 1, group)
        assert torch.all(c == 1.0)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.isfile(str(tmp_path / ("ok%d" % r))) for r in range(world))
