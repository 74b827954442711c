#!/usr/bin/env python
"""Benchmark of the VI iteration (M-step -> gamma-step -> prior table -> E-step, the body of
/root/reference/src/var_inf.py:135-137) on synthetic Gaussian-mixture latents.

  python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torchrun)

A "step" is one full VI iteration over the rank's resident latents (Z, phi, video ids already
in HBM).  Default workload = BASELINE.json configs[1]: N=1M frames, D=64, 8-ary tree with
K=73 nodes / L=64 leaves, videos of 256 frames.  With N ranks every rank holds its own 1M
frames (weak scaling) and the only collective is one fp64 all-reduce of the leaf statistics
per iteration (RCCL).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N per rank, D, branching, levels below root)   -- BASELINE.json configs / SURVEY.md section 0
    "config1": (60_000, 32, 5, 2),        # K=31,  L=25
    "config2": (1_000_000, 64, 8, 2),     # K=73,  L=64   (headline, memory-bound E/M-step)
    "config3": (16_000_000, 64, 8, 3),    # K=585, L=512  (total N; divided over ranks)
    "config4": (4_000_000, 256, 6, 5),    # K=9331, L=7776 (total N; divided over ranks)
}
FRAMES_PER_VIDEO = 256
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def regular_tree(bf, levels):
    """Pre-order parent / is_leaf arrays of the full bf-ary tree."""
    parent, is_leaf = [], []

    def rec(p, depth):
        k = len(parent)
        parent.append(p)
        is_leaf.append(1 if depth == levels else 0)
        if depth < levels:
            for _ in range(bf):
                rec(k, depth + 1)

    rec(-1, 0)
    return np.asarray(parent, np.int32), np.asarray(is_leaf, np.uint8)


def make_latents(torch, device, N, D, L, rank):
    """L centres ~ N(0, 4^2 I), uniform component, unit noise, fp32 (device RNG)."""
    mu = torch.from_numpy(np.random.RandomState(13).normal(scale=4.0, size=(L, D)).astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(13 + 1000 * rank)
    comp = torch.randint(0, L, (N,), generator=g, device=device)
    Z = mu[comp] + torch.randn((N, D), generator=g, device=device, dtype=torch.float32)
    vid = (torch.arange(N, device=device) // FRAMES_PER_VIDEO).to(torch.int32)
    V = int((N + FRAMES_PER_VIDEO - 1) // FRAMES_PER_VIDEO)
    return Z, vid, V


def cpu_baseline(D, bf, levels, sample_n):
    """The reference-faithful CPU port (oracle with the reference's per-frame loop and
    list.index lookups, single thread like the reference) on a bounded sample."""
    from oracle.vi_oracle import OracleVarInf, make_latent_dict, gaussian_mixture
    L = bf ** levels
    Z = gaussian_mixture(sample_n, D, L, seed=13, spread=4.0)
    lc = make_latent_dict(Z, [FRAMES_PER_VIDEO])
    out = {}
    for faithful in (True, False):
        np.random.seed(13)
        vi = OracleVarInf(latent_code_size=D, branching=bf, levels=levels, faithful_loops=faithful)
        Zm = vi.get_matrix_from_dict(lc)
        vi.build_initial_tree()
        vi.build_video_index()
        vi.one_hot_init(Zm)
        t0 = time.perf_counter()
        vi.compute_sigma_alpha(Zm)
        vi.compute_gamma()
        vi.compute_phi(Zm)
        out[faithful] = sample_n / (time.perf_counter() - t0)
    return {
        "value": out[True], "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "first %d frames (%d videos) of the workload, one M+gamma+E iteration of "
                  "oracle/vi_oracle.py with the reference's per-frame loop" % (sample_n, len(lc)),
        "vectorized_value": out[False],
        "vectorized_note": "same oracle with the per-frame loop vectorised (NumPy, %d host cores visible)"
                           % (os.cpu_count() or 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per rank (override; not `--n`: torch.distributed.run reads that as its own)")
    ap.add_argument("--no-estats", action="store_true",
                    help="skip the split/merge statistics epilogue of the E-step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=60_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (never the measured configuration): BNPVI_BENCH_REHEARSAL=1 puts
    # every rank on cuda:0 and uses gloo for the collective, so the N>1 control flow can be run
    # where RCCL (one device per rank) cannot.
    rehearsal = world > 1 and os.environ.get("BNPVI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD

    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI

    N, D, bf, levels = CONFIGS[args.config]
    scaling = "weak"
    if args.config in ("config3", "config4"):
        N = N // world                       # fixed total problem: strong scaling
        scaling = "strong"
    if args.frames:
        N = args.frames
    N = (N // FRAMES_PER_VIDEO) * FRAMES_PER_VIDEO or N     # shards cut on video boundaries
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())

    dev = DeviceVI(device=local_rank, group=group)
    Z, vid, V = make_latents(torch, device, N, D, L, rank)
    dev.set_latents(Z, vid, V)
    del Z
    alpha0 = np.random.RandomState(13).normal(size=(K, D))      # Node.__init__ draw, src/node.py:12
    dev.set_tree(parent, is_leaf, alpha0, np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)            # phi_in = one-hot nearest leaf
    want_estats = not args.no_estats

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dev.iteration(want_estats=want_estats)
    import ctypes as C
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dev.iteration(want_estats=want_estats)
    barrier()
    elapsed = time.perf_counter() - t0

    # Kernel durations for the roofline: a second, UNTIMED run of the same K steps with the
    # library's HIP-event pairs around the two data-pass kernels (the event records cost a few
    # microseconds per launch, so they stay out of the timed region above).
    def read(which):
        ms, cnt = C.c_double(), C.c_int()
        _lib.check(_lib.lib.bnpvi_profile_read(dev.ctx, which, C.byref(ms), C.byref(cnt)))
        return (ms.value / cnt.value) if cnt.value else float("nan")

    _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 1))
    for _ in range(args.steps):
        dev.iteration(want_estats=want_estats)
    barrier()
    e_ms, m_ms = read(0), read(1)
    _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 0))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # sanity of the state the timed loop left behind (cheap, outside the timed region)
    rows = dev.phi[: min(N, 4096), :L].sum(dim=1)
    assert torch.allclose(rows, torch.ones_like(rows), atol=1e-4), "phi rows do not sum to 1"

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = N * world / (elapsed / args.steps)
        kern_bytes = N * 4.0 * (D + L)            # per launch: E reads Z + writes phi; M reads Z + phi
        dom, dom_ms = ("estep", e_ms) if e_ms >= m_ms else ("mstep_stats", m_ms)
        achieved = kern_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(pmc):
            try:
                rec = json.load(open(pmc)).get("%s:%s" % (args.config, dom))
                if rec and rec.get("N") == N:
                    traffic = rec["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "latent samples/sec per VI E+M iteration (whole node)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: N=%d frames/rank, D=%d, %d-ary tree K=%d nodes L=%d leaves, "
                                   "%d-frame videos (V=%d), Gaussian-mixture latents, estats=%s"
                                   % (args.config, N, D, bf, K, L, FRAMES_PER_VIDEO, V, want_estats),
                       "parallelism": "sample-sharded x%d, 1 fp64 all-reduce of L x (1+D) stats/iter" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": kern_bytes, "avg_launch_ms": dom_ms,
                         "estep_ms": e_ms, "mstep_stats_ms": m_ms,
                         "iteration_frac": N * 8.0 * (D + L) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, bf, levels, min(args.cpu_sample, N))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
