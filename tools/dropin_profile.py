#!/usr/bin/env python
"""Development check: wall-clock of the drop-in VarInf.update_variational_parameters (host
marshalling + device work) on a dict-of-dicts of latents, with a cProfile of the second call."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bnp_vae_amd import VarInf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
D, FPV = 64, 256
rs = np.random.RandomState(0)
mu = rs.normal(scale=4.0, size=(64, D))
Z = (mu[rs.randint(0, 64, N)] + rs.normal(size=(N, D))).astype(np.float32)
lc = {}
for n in range(N):
    lc.setdefault(n // FPV, {})[n % FPV] = Z[n]
vi = VarInf(latent_code_size=D, branching=8, levels=2, stdev_thr=1e9)
np.random.seed(1)
for call in range(3):
    t0 = time.perf_counter()
    if call == 2:
        pr = cProfile.Profile(); pr.enable()
    vi.update_variational_parameters(lc)
    torch.cuda.synchronize()
    if call == 2:
        pr.disable()
    print("call %d: %.1f ms for N=%d" % (call, (time.perf_counter() - t0) * 1e3, N))
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
keys = [(v, f) for v in lc for f in lc[v]]
for name, mat in (("numpy matrix", Z), ("device tensor", torch.from_numpy(Z).to("cuda:0"))):
    vm = VarInf(latent_code_size=D, branching=8, levels=2, stdev_thr=1e9)
    np.random.seed(1)
    for call in range(3):
        t0 = time.perf_counter()
        vm.update_variational_parameters_matrix(mat, keys=keys)
        torch.cuda.synchronize()
        print("%s, call %d: %.2f ms" % (name, call, (time.perf_counter() - t0) * 1e3))
