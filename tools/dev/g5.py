import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI
N, D = 2000128, 64
parent, is_leaf = regular_tree(8, 3)
K, L = len(parent), int(is_leaf.sum())
dev = DeviceVI(device=0)
Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
dev.set_latents(Z, vid, V)
dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
dev.default_gamma()
dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
for _ in range(3):
    dev.iteration(want_estats=True, n_total=N)
torch.cuda.synchronize()
print("count", dev.gate_count.cpu().numpy(), "S0 max", float(dev.estats[:, 0].max()), "gate", 1e-2 * N)
for name, fn in (("gate+s2", lambda: dev.gated_s2(N)),):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    print(name, sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))[10] * 1e3, "us")
