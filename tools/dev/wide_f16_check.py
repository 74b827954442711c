"""Wide M-step, F16 mode (two f16 pieces, three products) against the exact bf16 x 3 split and against torch fp64.
Needs a library built with -DBNPVI_DEV_KNOBS (BNPVI_WIDE_F16=off / on)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import regular_tree, make_latents
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI

for (N, D, bf, levels, zscale) in ((20000, 256, 6, 5, 1.0), (16411, 256, 6, 4, 300.0), (9000, 200, 4, 5, 1e-3)):
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    out = {}
    os.environ["BNPVI_WIDE_F16"] = "off"
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    Z = Z * zscale
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, zscale * np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    dev.iteration(want_estats=False)          # soft phi
    for mode in ("on", "off"):
        os.environ["BNPVI_WIDE_F16"] = mode
        dev.leaf_stats(want_q=False, want_mass=True)
        torch.cuda.synchronize()
        out[mode] = (dev.stats.clone(), dev.mass.clone())
    ref = dev.phi[:, :L].to(torch.float64).T @ dev.Z[:, :D].to(torch.float64)
    dev.close()
    s_on, s_off = out["on"][0], out["off"][0]
    scale = ref.abs().max()
    e_on = float((s_on[:, 1:] - ref).abs().max() / scale)
    e_off = float((s_off[:, 1:] - ref).abs().max() / scale)
    rel_on = float(((s_on[:, 1:] - ref).abs() / (ref.abs() + 1e-3 * scale)).max())
    print("N=%d D=%d L=%d zscale=%g: S1 max err / max|S1|: f16 %.2e, exact split %.2e; elementwise (floor 1e-3 max) f16 %.2e; "
          "S0 identical %s, masses identical %s" % (N, D, L, zscale, e_on, e_off, rel_on,
          torch.equal(s_on[:, 0], s_off[:, 0]), torch.equal(out["on"][1], out["off"][1])), flush=True)
    assert e_on < 2e-6 and torch.equal(s_on[:, 0], s_off[:, 0])
