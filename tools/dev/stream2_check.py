"""New pass A of the streamed E-step (estep_stream2.h) against the old kernel: bit-identical phi / statistics.
Needs a library built with -DBNPVI_DEV_KNOBS (BNPVI_ESTEP_STREAM2=off selects the old kernel)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI

for (N, D, bf, levels) in ((5000, 256, 6, 5), (4099, 128, 4, 6), (3001, 256, 6, 4)):
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    out = {}
    for mode in ("on", "off"):
        os.environ["BNPVI_ESTEP_STREAM2"] = mode
        dev = DeviceVI(device=0)
        Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
        dev.set_latents(Z, vid, V)
        dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
        dev.default_gamma()
        dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
        dev.iteration(want_estats=True)
        logphi, _ = dev.estep(want_logphi=True, want_stats=True, s0_only=False)
        torch.cuda.synchronize()
        out[mode] = (dev.phi.clone(), dev.estats.clone(), None if logphi is None else logphi.clone())
        dev.close()
    same_phi = torch.equal(out["on"][0], out["off"][0])
    same_st = torch.equal(out["on"][1], out["off"][1])
    print("N=%d D=%d L=%d: phi identical %s, estats identical %s, rows sum %.6f..%.6f" %
          (N, D, L, same_phi, same_st, float(out["on"][0].sum(1).min()), float(out["on"][0].sum(1).max())), flush=True)
    assert same_phi and same_st
