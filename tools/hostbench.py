#!/usr/bin/env python
"""Development check: host enqueue time vs device time of DeviceVI.iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = DeviceVI(device=0)
parent, is_leaf = regular_tree(8, 2)
K, L = len(parent), int(is_leaf.sum())
Z, vid, V = make_latents(torch, dev.device, n, 64, L, 0)
dev.set_latents(Z, vid, V)
dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, 64)), np.ones(K))
dev.default_gamma()
dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
for _ in range(5):
    dev.iteration()
torch.cuda.synchronize()
for reps in (20, 100):
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.iteration()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("N=%d reps=%d: host enqueue %.1f us/iter, total %.1f us/iter" % (n, reps, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6))
