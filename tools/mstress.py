#!/usr/bin/env python
"""Development check: run-to-run reproducibility of bnpvi_leaf_stats (S0/S1 must be bit-identical)."""
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
dev = DeviceVI(device=0)
Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
dev.set_latents(Z, vid, V)
dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
dev.default_gamma()
dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
for it in range(3):
    dev.iteration(want_estats=True)
torch.cuda.synchronize()
col = dev.phi[:, :L].sum(dim=0, dtype=torch.float64)
ref = dev.phi[:, :L].to(torch.float64).T @ dev.Z[:, :D].to(torch.float64)
first = None
bad = 0
for r in range(reps):
    dev.leaf_stats(want_q=False, want_mass=(r % 2 == 0))
    torch.cuda.synchronize()
    st = dev.stats.clone()
    if first is None:
        first = st
    same = torch.equal(st, first)
    e0 = float((st[:, 0] - col).abs().max())
    e1 = float((st[:, 1:1 + D] - ref).abs().max()) / float(ref.abs().max())
    if not same or e0 > 1e-3 or e1 > 2e-6:
        bad += 1
        d = (st - first).abs()
        idx = torch.nonzero(d > 0)
        print("rep %d: identical=%s  |S0-col|max=%.3e  S1 rel err=%.3e  differing entries=%d  first=%s"
              % (r, same, e0, e1, idx.shape[0], idx[:6].tolist()))
print("%s N=%d: %d of %d repetitions off; S0 err of first %.3e" % (cfg, N, bad, reps, float((first[:, 0] - col).abs().max())))
