#!/usr/bin/env python
"""Large-N sanity of the data passes (phi has more than 2^32 elements at config 3's 16M frames;
`bign_check.py config4 1048576` does the same for the streamed E-step and the wide M-step):
one hard E-step + two iterations, then row sums at both ends of phi, total mass and S1 of the
last rows against torch fp64.  Development check, run by hand on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(cfg="config3", N=None):
    import torch
    from bench import CONFIGS, regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    n_cfg, D, bf, levels = CONFIGS[cfg]
    N = int(N) if N else n_cfg
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    dev.set_latents(Z, vid, V)
    del Z
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    torch.cuda.synchronize()
    print("phi elements: %.3e (2^32 = %.3e)" % (dev.phi.numel(), 2.0 ** 32), flush=True)
    t0 = time.perf_counter()
    for _ in range(2):
        dev.iteration(want_estats=True)
    torch.cuda.synchronize()
    print("2 iterations: %.1f ms each" % (0.5e3 * (time.perf_counter() - t0)), flush=True)
    for name, sl in (("first", slice(0, 4096)), ("last", slice(N - 4096, N)), ("middle", slice(N // 2, N // 2 + 4096))):
        rows = dev.phi[sl, :L].sum(dim=1, dtype=torch.float64)
        err = float((rows - 1).abs().max())
        print("%s rows: max |sum - 1| = %.2e" % (name, err), flush=True)
        assert err < 1e-5
    S0e = float(dev.estats[:, 0].sum())
    print("E-step total mass %.3f vs N %d" % (S0e, N), flush=True)
    assert abs(S0e - N) <= 1e-6 * N
    dev.leaf_stats(want_q=False, want_mass=True)
    torch.cuda.synchronize()
    st = dev.stats
    assert abs(float(st[:, 0].sum()) - N) <= 1e-6 * N
    assert abs(float(dev.mass.sum()) - N) <= 1e-6 * N
    # S1 through linearity: statistics of the last 1M frames alone against torch fp64
    import ctypes as C
    lo = N - min(N, 1_000_192, ((1 << 30) // L) // 256 * 256)        # fp64 copy of the slice stays under 8 GB
    out = torch.empty_like(st)
    _lib.check(_lib.lib.bnpvi_leaf_stats(
        dev.ctx, C.c_void_p(dev.Z.data_ptr() + 4 * lo * dev.Dp), dev.Dp,
        C.c_void_p(dev.phi.data_ptr() + 4 * lo * dev.tree.Lp), dev.tree.Lp,
        C.c_void_p(dev.vid.data_ptr() + 4 * lo), N - lo, D, L, dev.V, 0,
        C.c_void_p(out.data_ptr()), C.c_void_p(0), dev._stream()))
    torch.cuda.synchronize()
    ref = dev.phi[lo:, :L].to(torch.float64).T @ dev.Z[lo:, :D].to(torch.float64)
    err = float((out[:, 1:1 + D] - ref).abs().max()) / float(ref.abs().max())
    print("S1 of the last %d frames: rel err %.2e" % (N - lo, err), flush=True)
    assert err < 2e-6
    print("large-N check ok")
    dev.close()


if __name__ == "__main__":
    main(*sys.argv[1:3])
