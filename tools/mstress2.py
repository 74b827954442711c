#!/usr/bin/env python
"""Development check: after every VI iteration compare the M-step's S0/S1 of the new phi with
torch fp64 sums; on a mismatch run the M-step again on the same phi."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CONFIGS, regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI
cfg, N = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
_n, D, bf, levels = CONFIGS[cfg]
parent, is_leaf = regular_tree(bf, levels)
K, L = len(parent), int(is_leaf.sum())
bad = 0
for trial in range(reps):
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, trial)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for it in range(3):
        dev.iteration(want_estats=True)
    col = dev.phi[:, :L].sum(dim=0, dtype=torch.float64)
    dev.leaf_stats(want_q=False, want_mass=True)
    st = dev.stats.clone()
    tol = 2e-6 * col.abs() + 1e-4
    off = torch.nonzero((st[:, 0] - col).abs() > tol)
    if off.numel():
        bad += 1
        dev.leaf_stats(want_q=False, want_mass=True)
        st2 = dev.stats.clone()
        dd = (st - st2).abs()
        nz = torch.nonzero(dd > 0)
        cols = sorted(set(nz[:, 1].tolist()))
        rows = sorted(set(nz[:, 0].tolist()))
        print("   first vs second run: %d entries differ; leaves %s; columns %s%s; max |diff| %.3e (S0 col: %.3e)"
              % (nz.shape[0], rows[:12], cols[:12], "..." if len(cols) > 12 else "", float(dd.max()), float(dd[:, 0].max())))
        col2 = dev.phi[:, :L].sum(dim=0, dtype=torch.float64)
        l = int(off[0])
        print("trial %d: %d leaves off; leaf %d: S0=%.9g col=%.9g | again: S0=%.9g col=%.9g | estats S0=%.9g"
              % (trial, off.numel(), l, float(st[l, 0]), float(col[l]), float(st2[l, 0]), float(col2[l]),
                 float(dev.estats[l, 0])))
    dev.close()
print("%s N=%d: %d of %d trials off" % (cfg, N, bad, reps))
