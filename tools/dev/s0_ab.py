"""A/B run for the S0 accumulation (mstep_common.h): `launches` M-steps on one-hot responsibilities, S0 must equal
the per-leaf frame counts bit for bit in every one.  BNPVI_LIB selects the library (shipped, or a `make variant`)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import CONFIGS, regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI

cfg, N, launches = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
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
bad, worst = 0, 0.0
t0 = time.time()
for it in range(launches):
    dev.leaf_stats(want_q=False, want_mass=True)
    d = (dev.stats[:, 0] - counts).abs().max()
    if float(d) != 0.0:
        bad += 1
        worst = max(worst, float(d))
print("%s N=%d lib=%s: %d of %d launches with a wrong S0 (largest deficit %.0f frames), %.1f s"
      % (cfg, N, _lib.LIB_PATH, bad, launches, worst, time.time() - t0), flush=True)
