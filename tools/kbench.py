#!/usr/bin/env python
"""Development micro-benchmark: times the individual C-ABI calls of one VI iteration with
HIP events (torch events on the launch stream).  BNPVI_LIB=<path> selects an alternative
build of the library (e.g. the ablation variants of `make -C bnp_vae_amd/csrc variants`)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--bf", type=int, default=8)
    ap.add_argument("--levels", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tag", default="")
    ap.add_argument("--only", default="", help="comma-separated phases to run (default: all)")
    ap.add_argument("--far", type=float, default=0.0,
                    help="tight far-from-origin clusters: z = FAR*alpha_leaf + 0.5*noise with the tree's alpha scaled "
                         "alike (every frame's own leaf, and some neighbours, take the E-step's exact-distance path)")
    args = ap.parse_args()
    import torch
    from bench import regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    dev = DeviceVI(device=0)
    parent, is_leaf = regular_tree(args.bf, args.levels)
    K, L = len(parent), int(is_leaf.sum())
    Z, vid, V = make_latents(torch, dev.device, args.n, args.d, L, 0)
    alpha = np.random.RandomState(13).normal(size=(K, args.d))
    if args.far > 0:
        alpha *= args.far
        mu = torch.from_numpy(alpha[is_leaf.astype(bool)].astype(np.float32)).to(dev.device)
        Z = mu[(torch.arange(args.n, device=dev.device) * 7919) % L] + 0.5 * torch.randn_like(Z)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, alpha, np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for _ in range(3):
        dev.iteration(want_estats=False)
    phases = [
        ("leaf_stats", lambda: dev.leaf_stats(False, True)),
        ("leaf_stats_q", lambda: dev.leaf_stats(True, True)),
        ("node_params", dev.node_params),
        ("gamma_update", dev.gamma_update),
        ("prior_table", dev.prior_table),
        ("gamma_prior", dev.gamma_prior),
        ("params_prior", dev.params_prior),
        ("estep", lambda: dev.estep(want_stats=False)),
        ("estep_stats", lambda: dev.estep(want_stats=True)),
        ("estep_stats_s2", lambda: dev.estep(want_stats=True, s0_only=False)),   # S2 for every leaf in the kernel
        ("gated_s2", lambda: dev.gated_s2(dev.N)),
        ("estep_logphi", lambda: dev.estep(want_stats=False, want_logphi=True)),
        ("iteration", lambda: dev.iteration(want_estats=False)),
    ]
    import ctypes as C

    def kern_us(which):
        ms, cnt = C.c_double(), C.c_int()
        _lib.check(_lib.lib.bnpvi_profile_read(dev.ctx, which, C.byref(ms), C.byref(cnt)))
        return 1e3 * ms.value / cnt.value if cnt.value else float("nan")

    out = []
    if args.only:
        phases = [p for p in phases if p[0] in args.only.split(",")]
    for name, fn in phases:
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
        ev[0].record()
        for i in range(args.iters):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
        out.append("%s=%.1f" % (name, 1e3 * ts[len(ts) // 2]))
        if name in ("leaf_stats", "leaf_stats_q", "estep", "estep_stats", "estep_stats_s2", "estep_logphi"):
            # the main kernel alone (HIP events recorded around it inside the library)
            _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 1))
            for i in range(args.iters):
                fn()
            torch.cuda.synchronize()
            out[-1] += "(k%.1f)" % kern_us(1 if name.startswith("leaf") else 0)
            _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 0))
    print("%s N=%d D=%d L=%d  [us median] %s" % (args.tag or os.environ.get("BNPVI_LIB", "default").split("_")[-1],
                                                args.n, args.d, L, " ".join(out)))


if __name__ == "__main__":
    main()
