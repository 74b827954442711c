#!/usr/bin/env python
"""Development check: E-step then M-step, repeated on fixed parameters; every M-step result must be
bit-identical to the first (phi is rewritten with identical values each time)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CONFIGS, regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI
cfg, N, reps, want_mass = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
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
first, bad = None, 0
for r in range(reps):
    dev.estep(want_stats=True)
    dev.leaf_stats(want_q=False, want_mass=want_mass)
    st = dev.stats.clone()
    if want_mass:
        mk = dev.mass.clone()
        if first is None:
            mass0 = mk
        else:
            md = float((mk - mass0).abs().max())
            if md > 1e-9:
                bad_mass = globals().get("bad_mass", 0) + 1
                globals()["bad_mass"] = bad_mass
                if bad_mass <= 4:
                    nz = torch.nonzero((mk - mass0).abs() > 1e-9)
                    print("rep %d: per-video masses differ by up to %.3e in %d entries, e.g. (video, leaf) %s"
                          % (r, md, nz.shape[0], nz[:6].tolist()))
    if first is None:
        first = st
    elif not torch.equal(st, first):
        bad += 1
        dd = (st - first).abs()
        nz = torch.nonzero(dd > 0)
        if bad <= 4:
            print("rep %d: %d entries differ; leaves %s; columns %s; max %.3e" % (
                r, nz.shape[0], sorted(set(nz[:, 0].tolist()))[:20], sorted(set(nz[:, 1].tolist()))[:8], float(dd.max())))
print("%s N=%d mass=%s order=%s: %d of %d repetitions differ from the first; masses off in %d" % (
    cfg, N, want_mass, os.environ.get("BNPVI_MSTEP_ORDER", "default"), bad, reps, globals().get("bad_mass", 0)))
