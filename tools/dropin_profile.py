#!/usr/bin/env python
"""Wall-clock of the drop-in `VarInf.update_variational_parameters_matrix` (device tensor in, host-visible
tree out) at the five BASELINE shapes, next to the device time of the same step through
`DeviceVI.iteration` -- i.e. how much the Python side (Node walks, tree flattening, parameter
round trips, split / merge bookkeeping) adds on top of the kernels.

  python tools/dropin_profile.py [config ...] [--profile]      (default: all five)

Per config: calls 0-1 warm up (first call builds the tree and the one-hot phi), then the median of
the next calls with a fresh Z each (src/model.py:180-188: new latents every epoch).
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

SHAPES = {
    # name: (N, D, branching, levels, stdev_thr, num_splits, calls)
    "config1": (60_000, 32, 5, 2, 1e9, 2, 6),
    "config2": (999_936, 64, 8, 2, 1e9, 2, 6),
    "config3": (2_000_128, 64, 8, 3, 1e9, 2, 5),        # one rank's share of the 16M frames
    "config4": (131_072, 256, 6, 5, 1e9, 2, 4),
    "config5": (2_000_128, 128, 4, 3, -98.9, 4, 6),     # dynamic: splits and merges every call
}
FPV = 256


def latents(N, D, mu, ep, device, noise):
    """Mixture around the rows of `mu` (equal shares, so every component keeps its share of the frames)."""
    g = torch.Generator(device=device)
    g.manual_seed(100 + ep)
    comp = (torch.arange(N, device=device) * 7919) % mu.shape[0]
    return mu[comp] * (1.0 + 0.01 * ep) + noise * torch.randn((N, D), generator=g, device=device)


def run(name, profile):
    N, D, bf, levels, thr, ns, calls = SHAPES[name]
    dev = torch.device("cuda", 0)
    keys = [(n // FPV, n % FPV) for n in range(N)]
    vi = VarInf(latent_code_size=D, branching=bf, levels=levels, stdev_thr=thr, num_splits=ns)
    np.random.seed(1)
    if name == "config5":
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        mu, noise = torch.randn((24, D), generator=g, device=dev) * 3.0, 1.0
    else:
        # the static shapes: latents sit around the leaves' initial means (the draws of Node.__init__), so every
        # leaf keeps its 1/L of the frames and the tree keeps its K nodes -- as long as 1/L clears the
        # reference's own prune threshold of 1e-3 (src/var_inf.py:194), which config 4's 7776 leaves cannot
        st = np.random.get_state()
        vi.vidid_to_idx, vi.vidid_frameid_to_idx = list(range((N + FPV - 1) // FPV)), keys
        vi._build_initial_tree()
        np.random.set_state(st)
        mu = torch.from_numpy(np.asarray([n.alpha for n in vi._dfs_leaves()], dtype=np.float32)).to(dev)
        vi.root, noise = None, 0.2
        vi.vidid_to_idx, vi.vidid_frameid_to_idx = [], []
    times, ks = [], []
    pr = None
    for call in range(calls):
        Z = latents(N, D, mu, call, dev, noise)
        torch.cuda.synchronize()
        if profile and call == calls - 1:
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        vi.update_variational_parameters_matrix(Z, keys=keys if call == 0 else None)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if pr is not None:
            pr.disable()
        ks.append(len(vi._preorder()))
    # device time of one iteration at the final tree's shape
    d = vi._dev
    vi._ensure_tree(); vi._sync_phi(); vi._sync_gamma()
    for _ in range(2):
        d.iteration(want_estats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        d.iteration(want_estats=True)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) * 1e3 / reps
    steady = sorted(times[2:])[len(times[2:]) // 2]
    print("%-8s N=%-8d D=%-3d K per call %s | call ms %s | steady %.2f ms, device iteration %.2f ms, ratio %.2f"
          % (name, N, D, ks, " ".join("%.1f" % t for t in times), steady, dev_ms, steady / dev_ms), flush=True)
    if pr is not None:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    del vi
    torch.cuda.empty_cache()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name in (args or list(SHAPES)):
        run(name, "--profile" in sys.argv)
