#!/usr/bin/env python
"""Wall-clock of the drop-in `VarInf.update_variational_parameters_matrix` (device tensor in, host-visible
tree out) at the five BASELINE shapes, next to the device time of the same step through
`DeviceVI.iteration` -- i.e. how much the Python side (Node walks, tree flattening, parameter
round trips, split / merge bookkeeping) adds on top of the kernels.

  python tools/dropin_profile.py [config ...] [--profile]      (default: all five)

Static shapes (configs 1-4), three figures each:
  host   : steady wall-clock of a drop-in call on a SMALL frame count (8 frames per leaf) with the tree held at
           its K nodes -- kernels are a few tens of microseconds there, so this is the Python side's cost at K;
  device : `DeviceVI.iteration` at the full frame count on bench.py's mixture (the kernels' time);
  call   : steady wall-clock of a drop-in call at the full frame count (kernels and Python overlap where no
           read-back forces a wait).
Config 5 is dynamic (splits and merges every call): per-call wall-clock and K.
Calls 0-1 warm up (the first builds the tree and the one-hot phi); fresh Z every call (src/model.py:180-188).
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bnp_vae_amd import VarInf  # noqa: E402

PROFILE_CALL = -1   # --profile-call=N: which call --profile looks at (default: the last)
COMPS = 0           # --comps=N: mixture components of config 5's workload (default 24)
SHAPES = {
    # name: (N, D, branching, levels, stdev_thr, num_splits, calls, frames per video)
    "config1": (60_000, 32, 5, 2, 1e9, 2, 6, 1000),
    "config2": (999_936, 64, 8, 2, 1e9, 2, 6, 3906),
    "config3": (2_000_128, 64, 8, 3, 1e9, 2, 5, 15626),     # one rank's share of the 16M frames
    "config4": (131_072, 256, 6, 5, 1e9, 2, 4, 256),
    "config5": (2_000_128, 128, 4, 3, -98.9, 4, 6, 256),    # dynamic: splits and merges every call
    # dynamic from a tree in the thousands (K = 1365, L = 1024; 800 mixture components): the calls walk the
    # streamed -> one-pass row -> fused E-step and wide -> DMA-ring M-step regimes on the way down (the
    # reference's prune rule, mean(phi) < 1e-3, bounds the leaves that survive a call by 1000)
    "config5_large": (2_000_128, 128, 4, 5, -100.0, 4, 8, 256),
}
DYNAMIC = {"config5": 24, "config5_large": 800}     # mixture components


def latents(N, D, mu, ep, device, noise):
    """Mixture around the rows of `mu` (equal shares, so every component keeps its share of the frames)."""
    g = torch.Generator(device=device)
    g.manual_seed(100 + ep)
    comp = (torch.arange(N, device=device) * 7919) % mu.shape[0]
    return mu[comp] * (1.0 + 0.01 * ep) + noise * torch.randn((N, D), generator=g, device=device)


def far_leaf_means(vi, N, FPV, keys, dev):
    """The static shapes: latents sit far apart, around 32x the leaves' initial means (the draws of
    Node.__init__): the stick priors favour the first children by tens of nats (~-20 for the last leaf of
    config 1, more in deeper trees) and the likelihood term is only -0.5*||z - alpha|| (un-squared), so closer
    mixtures drain into the first leaves within a few calls and the tree collapses -- the reference's own
    dynamics, but not the fixed-K workload this table is about.  At 32x the distance gap (~180) beats the
    priors, every leaf keeps its 1/L of the frames and the tree keeps its K nodes -- as long as 1/L clears the
    reference's prune threshold of 1e-3 (src/var_inf.py:194), which config 4's 7776 leaves cannot: its tree
    shrinks, as the reference's would.  (Such tight far-from-origin clusters also put every frame's own leaf
    on the E-step's exact-distance path: the full-N `call` figure of this workload is NOT the bench figure.)"""
    st = np.random.get_state()
    vi.vidid_to_idx, vi.vidid_frameid_to_idx = list(range((N + FPV - 1) // FPV)), keys
    vi._build_initial_tree()
    np.random.set_state(st)
    mu = torch.from_numpy(np.asarray([n.alpha for n in vi._dfs_leaves()], dtype=np.float32)).to(dev) * 32.0
    vi.root = None
    vi.vidid_to_idx, vi.vidid_frameid_to_idx = [], []
    return mu


def dropin_calls(name, N, calls, profile, dev):
    _N, D, bf, levels, thr, ns, _calls, FPV = SHAPES[name]
    keys = [(n // FPV, n % FPV) for n in range(N)]
    vi = VarInf(latent_code_size=D, branching=bf, levels=levels, stdev_thr=thr, num_splits=ns)
    np.random.seed(1)
    if name in DYNAMIC:
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        comps = COMPS if (name == "config5" and COMPS) else DYNAMIC[name]
        mu, noise = torch.randn((comps, D), generator=g, device=dev) * 3.0, 1.0
    else:
        mu, noise = far_leaf_means(vi, N, FPV, keys, dev), 0.5
    times, ks = [], []
    pr = None
    for call in range(calls):
        Z = latents(N, D, mu, call, dev, noise)
        torch.cuda.synchronize()
        if profile and call == (PROFILE_CALL if PROFILE_CALL >= 0 else calls - 1):
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        vi.update_variational_parameters_matrix(Z, keys=keys if call == 0 else None)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if pr is not None:
            pr.disable()
        ks.append(len(vi._preorder()))
    if pr is not None:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
    del vi
    torch.cuda.empty_cache()
    return times, ks


def device_iteration(name, dev):
    """`DeviceVI.iteration` (M-step, gamma, prior table, E-step with its statistics) on bench.py's mixture."""
    from bench import regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    N, D, bf, levels = SHAPES[name][:4]
    d = DeviceVI(device=0)
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    Z, vid, V = make_latents(torch, d.device, N, D, L, 0)
    d.set_latents(Z, vid, V)
    d.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    d.default_gamma()
    d.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for _ in range(3):
        d.iteration(want_estats=True)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        d.iteration(want_estats=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    d.close()
    torch.cuda.empty_cache()
    return ms, K


def median(x):
    return sorted(x)[len(x) // 2]


def run(name, profile):
    N, D, bf, levels, thr, ns, calls, FPV = SHAPES[name]
    dev = torch.device("cuda", 0)
    if name in DYNAMIC:
        times, ks = dropin_calls(name, N, calls, profile, dev)
        print("%-8s N=%-8d D=%-3d dynamic: K per call %s | call ms %s | median after warm-up %.2f ms"
              % (name, N, D, ks, " ".join("%.1f" % t for t in times), median(times[2:])), flush=True)
        return
    L = bf ** (levels + 1)
    n_small = min(N, 8 * L)
    host_t, host_k = dropin_calls(name, n_small, 8, profile, dev)
    dev_ms, K = device_iteration(name, dev)
    call_t, call_k = dropin_calls(name, N, calls, False, dev)
    host = median(host_t[2:])
    print("%-8s N=%-8d D=%-3d K=%-5d host %.2f ms (N=%d, K per call %s) | device %.2f ms | "
          "(host+device)/device %.2f | full-N far-cluster calls ms %s (K %s)"
          % (name, N, D, K, host, n_small, host_k, dev_ms, (host + dev_ms) / dev_ms,
             " ".join("%.1f" % t for t in call_t), call_k), flush=True)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--comps="):
            COMPS = int(a.split("=")[1])
        if a.startswith("--profile-call="):
            PROFILE_CALL = int(a.split("=")[1])
        if a.startswith("--calls="):                      # e.g. --calls=16: a longer dynamic run of config 5
            SHAPES["config5"] = SHAPES["config5"][:6] + (int(a.split("=")[1]),) + SHAPES["config5"][7:]
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name in (args or [n for n in SHAPES if n != "config5_large"]):
        run(name, "--profile" in sys.argv)
