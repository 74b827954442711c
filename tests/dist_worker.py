"""One rank of the multi-rank drop-in run used by tests/test_gpu_round2.py::test_two_rank_trajectory
(not collected by pytest).  Every rank is a FRESH process (started before it touches the GPU),
all ranks share cuda:0 and talk through gloo with host-staged collectives -- the rehearsal form
of the one-process-per-GPU RCCL layout (bnp_vae_amd/dist.py).  The rank takes its videos from
dist.shard_videos, runs the golden trajectory through VarInf(group=...) and writes what it
holds after every epoch.

  python tests/dist_worker.py RANK WORLD PORT OUTDIR GOLDEN
  python tests/dist_worker.py RANK WORLD PORT OUTDIR fuzz:SEED
      the random trajectory tests/traj_fuzz.py draws for SEED, this rank's videos of it (tests/dist_fuzz.py)
  python tests/dist_worker.py RANK WORLD PORT OUTDIR synth:N:D:COMPS:CALLS:BF:LEVELS
      full-size synthetic run (test_gpu_round3.py): the rank's share of N clustered frames, built on the
      device from a formula of the GLOBAL row index (the same data whatever the sharding); writes the tree ids
      per call and a few size-independent checks instead of the matrices.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


FPV_SYNTH = 256


def synth(rank, world, out_dir, spec, group):
    import numpy as np
    import torch
    from bnp_vae_amd import VarInf
    from bnp_vae_amd import dist as vdist
    N, D, comps, calls, bf, levels = [int(x) for x in spec.split(":")[1:]]
    dev = torch.device("cuda", 0)
    V = (N + FPV_SYNTH - 1) // FPV_SYNTH
    fpv = [min(FPV_SYNTH, N - v * FPV_SYNTH) for v in range(V)]
    v0, v1, f0, f1 = vdist.shard_videos(fpv, world)[rank]
    mu = torch.from_numpy(np.random.RandomState(7).normal(scale=3.0, size=(comps, D)).astype(np.float32)).to(dev)
    keys = [(n // FPV_SYNTH, n % FPV_SYNTH) for n in range(f0, f1)]
    np.random.seed(1000 + rank)
    vi = VarInf(latent_code_size=D, stdev_thr=-100.0, num_splits=4, branching=bf, levels=levels, group=group)
    out = {"f0": f0, "f1": f1, "K0": 0}
    for call in range(calls):
        g = torch.Generator(device=dev)
        g.manual_seed(100 + call)
        noise = torch.randn((N, D), generator=g, device=dev)[f0:f1]        # global stream, this rank's rows
        rows = torch.arange(f0, f1, device=dev)
        Z = mu[(rows * 7919) % comps] * (1.0 + 0.01 * call) + 0.3 * noise
        del noise
        if rank == 0:
            np.random.seed(13 + call)
        vi.update_variational_parameters_matrix(Z, keys=keys if call == 0 else None)
        nodes = vi.get_nodes_list(vi.root, [])
        out["c%d_ids" % call] = np.asarray([n.node_id for n in nodes])
        out["c%d_is_leaf" % call] = np.asarray([bool(n.isLeafNode) for n in nodes])
        out["c%d_alpha" % call] = np.asarray([np.asarray(n.alpha) for n in nodes])
        # size-independent properties of a fresh E-step on the tree this call left behind
        vi.compute_sigma_alpha(Z)
        vi.compute_gamma()
        vi.compute_phi(Z)
        d = vi._dev
        L = d.tree.L
        rowsum = d.phi[:, :L].sum(dim=1)
        out["c%d_rowerr" % call] = float((rowsum - 1).abs().max()) if d.N else 0.0
        out["c%d_S0sum" % call] = float(d.estats[:, 0].sum())              # all-reduced: the global frame count
        out["c%d_colerr" % call] = float(((d.phi[:, :L].double().sum(dim=0) - 0) .sum() - d.N).abs())
        out["c%d_K" % call] = len(nodes)
        torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)


def fuzz(rank, world, out_dir, spec, group):
    import numpy as np
    from bnp_vae_amd import VarInf
    from bnp_vae_amd import dist as vdist
    from tests.traj_fuzz import config, latents
    seed = int(spec.split(":")[1])
    kw, N, D, mu, noise, fpv, _tag = config(seed)
    v0, v1, f0, f1 = vdist.shard_videos(fpv, world)[rank]
    starts = np.concatenate([[0], np.cumsum(fpv)])
    np.random.seed(1000 + rank)
    vi = VarInf(group=group, **kw)
    out = {"f0": f0, "f1": f1}
    for ep in range(5):
        Z = latents(N, D, mu, ep, seed, noise)
        lc = {}
        for v in range(v0, v1):
            lc["vid%05d" % v] = {"%06d" % f: Z[starts[v] + f] for f in range(fpv[v])}
        if rank == 0:
            np.random.seed(13 + ep)
        try:
            vi.update_variational_parameters(lc)
        except TypeError:
            out["e%d_raised" % ep] = 1                       # the tree collapsed to its root a call earlier
            break
        nodes = vi.get_nodes_list(vi.root, [])
        out["e%d_ids" % ep] = np.asarray([n.node_id for n in nodes])
        out["e%d_is_leaf" % ep] = np.asarray([bool(n.isLeafNode) for n in nodes])
        out["e%d_alpha" % ep] = np.asarray([np.asarray(n.alpha) for n in nodes])
        out["e%d_sigma" % ep] = np.asarray([float(n.sigmasqr_inv) for n in nodes])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)


def leave(dist):
    """The rank's results are on disk: wait for the others, then end the process without the interpreter's and the
    process group's teardown (one rank in ~70 runs of the sweep aborted there -- "terminate called without an active
    exception" after its last collective -- which the parent would count as a failed rank)."""
    import torch
    torch.cuda.synchronize()
    dist.barrier()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def main():
    rank, world, port, out_dir, golden = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import json
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    backend = os.environ.get("BNPVI_WORKER_BACKEND", "gloo")       # "nccl" (= RCCL): one rank per device only
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
            probe = torch.ones(4, dtype=torch.float64, device="cuda")
            dist.all_reduce(probe)                                  # RCCL builds its communicator on first use
            torch.cuda.synchronize()
        except Exception as e:                                      # the machine's RCCL, not the code under test
            print("nccl unavailable: %r" % (e,), flush=True)
            os._exit(77)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"all_reduce": 0, "all_gather": 0, "broadcast": 0}
    for name in calls:                                            # count what the path really sends
        def counted(*a, _f=getattr(dist, name), _n=name, **kw):
            calls[_n] += 1
            return _f(*a, **kw)
        setattr(dist, name, counted)
    from bnp_vae_amd import VarInf
    from bnp_vae_amd import dist as vdist
    from oracle.vi_oracle import snapshot, drifting_latents
    if golden.startswith("fuzz:"):
        fuzz(rank, world, out_dir, golden, dist.group.WORLD if world > 1 else None)
        leave(dist)
    if golden.startswith("synth:"):
        synth(rank, world, out_dir, golden, dist.group.WORLD if world > 1 else None)
        leave(dist)
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
    if rank == 0:
        print("collectives run: all_reduce=%d all_gather=%d broadcast=%d backend %s" % (
            calls["all_reduce"], calls["all_gather"], calls["broadcast"], dist.get_backend()), flush=True)
    leave(dist)


if __name__ == "__main__":
    main()
