"""One rank of the multi-rank drop-in run used by tests/test_gpu_round2.py::test_two_rank_trajectory
(not collected by pytest).  Every rank is a FRESH process (started before it touches the GPU),
all ranks share cuda:0 and talk through gloo with host-staged collectives -- the rehearsal form
of the one-process-per-GPU RCCL layout (bnp_vae_amd/dist.py).  The rank takes its videos from
dist.shard_videos, runs the golden trajectory through VarInf(group=...) and writes what it
holds after every epoch.

  python tests/dist_worker.py RANK WORLD PORT OUTDIR GOLDEN [config5]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir, golden = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import json
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from bnp_vae_amd import VarInf
    from bnp_vae_amd import dist as vdist
    from oracle.vi_oracle import snapshot, drifting_latents
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", golden), allow_pickle=False))
    m = json.loads(str(g["meta"]))
    fpv = [int(x) for x in g["fpv"]]
    if "V" in g and len(fpv) != int(g["V"]):
        fpv = fpv * int(g["V"])
    N = int(sum(fpv))
    # videos as make_latent_dict lays them out (cycling through fpv until N frames are used)
    v0, v1, f0, f1 = vdist.shard_videos(fpv, world)[rank]
    starts = np.concatenate([[0], np.cumsum(fpv)])
    np.random.seed(1000 + rank)                    # ranks deliberately NOT seeded alike: alpha comes from rank 0
    vi = VarInf(latent_code_size=m["D"], stdev_thr=m["thr"], num_splits=m["num_splits"], group=dist.group.WORLD)
    out = {"f0": f0, "f1": f1, "v0": v0, "v1": v1}
    for ep in range(m["epochs"]):
        if m.get("kind") == "trajectory_config5":
            Z = drifting_latents(N, m["D"], m["ncomp"], m["seed"], ep, m["spread"])
        else:
            Z = g["Z%d" % ep]
        lc = {}
        for v in range(v0, v1):
            lc["vid%05d" % v] = {"%06d" % f: Z[starts[v] + f] for f in range(fpv[v])}
        if rank == 0:
            np.random.seed(13 + ep)                # the single-rank run's seed, on rank 0 only
        vi.update_variational_parameters(lc)
        snap = snapshot(vi)
        for k in ("ids", "parent", "is_leaf", "alpha", "sigma", "gamma", "phi", "decay"):
            out["e%d_%s" % (ep, k)] = snap[k]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
