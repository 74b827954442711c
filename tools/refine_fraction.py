#!/usr/bin/env python
"""How many (frame, leaf) pairs take the E-step's exact-distance path, and how good the logits are.

For each workload the pairs are classified in fp64 (torch, on the GPU) by the rule of
bnp_vae_amd/csrc/estep_kernels.h: with s = ||z||^2 + ||a||^2,
    absolute rule (rounds 1-2):  d^2 < 6.25e-6 * s^2
    shipped rule:                d^2 < s * min(6.25e-6 * s, 0.42)
and the product's log phi / phi (soft E-step, fp32) are compared with the fp64 evaluation of the same
logits (the device's fp32 prior table taken as given), so the figures show what the rule costs in accuracy.

    python tools/refine_fraction.py [--n 131072]

Workloads: bench.py's mixture at config 2's and config 3's shapes after three iterations (the state the
benchmark times), the same with noise 0.5 instead of 1.0, and `kbench --far 32` (tight clusters 32x further
out: z = 32 * alpha_leaf + 0.5 * noise) at both shapes.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(torch, dev, name, N, D, bf, levels):
    from bench import regular_tree, make_latents
    from bnp_vae_amd import _lib
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    alpha = np.random.RandomState(13).normal(size=(K, D))
    iters = 3
    if name == "noise0.5":
        mu = torch.from_numpy(np.random.RandomState(13).normal(scale=4.0, size=(L, D)).astype(np.float32)).to(dev.device)
        g = torch.Generator(device=dev.device)
        g.manual_seed(13)
        comp = torch.randint(0, L, (N,), generator=g, device=dev.device)
        Z = mu[comp] + 0.5 * torch.randn((N, D), generator=g, device=dev.device, dtype=torch.float32)
    elif name == "far32":
        alpha = alpha * 32.0
        mu = torch.from_numpy(alpha[is_leaf.astype(bool)].astype(np.float32)).to(dev.device)
        Z = mu[(torch.arange(N, device=dev.device) * 7919) % L] + 0.5 * torch.randn_like(Z)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, alpha, np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for _ in range(iters):
        dev.iteration(want_estats=False)
    dev.leaf_stats(False, True)
    dev.params_prior()
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    args = ap.parse_args()
    import torch
    from bnp_vae_amd.engine import DeviceVI
    print("%-10s %-8s %9s | pairs on the exact path: absolute rule, shipped rule | frames with >= 1 such pair | "
          "max |log phi - fp64| (where phi >= 1e-6), max |phi - fp64|" % ("workload", "shape", "median s"))
    for shape, (D, bf, levels) in (("config2", (64, 8, 2)), ("config3", (64, 8, 3))):
        for name in ("bench", "noise0.5", "far32"):
            dev = DeviceVI(device=0)
            N = args.n
            L = workload(torch, dev, name, N, D, bf, levels)
            logphi, _ = dev.estep(want_logphi=True, want_stats=False)
            nol = torch.from_numpy(np.asarray(dev.tree.node_of_leaf, dtype=np.int64)).to(dev.device)
            a = dev.alpha[nol]
            a32 = a.to(torch.float32).to(torch.float64)                       # the kernel's alpha is fp32-rounded
            aa = (a32 * a32).sum(1)
            old_pairs = new_pairs = old_frames = new_frames = 0
            err_lp = err_phi = 0.0
            s_med = []
            step = 8192
            for lo in range(0, N, step):
                z = dev.Z[lo:lo + step, :D].to(torch.float64)
                zz = (z * z).sum(1)
                d2 = torch.empty((z.shape[0], L), dtype=torch.float64, device=dev.device)
                for l0 in range(0, L, 64):                                        # direct differences in fp64
                    d2[:, l0:l0 + 64] = ((z[:, None, :] - a32[None, l0:l0 + 64, :]) ** 2).sum(2)
                s = zz[:, None] + aa[None, :]
                old = d2 < 6.25e-6 * s * s
                new = d2 < s * torch.minimum(6.25e-6 * s, torch.full_like(s, 0.42))
                old_pairs += int(old.sum()); new_pairs += int(new.sum())
                old_frames += int(old.any(1).sum()); new_frames += int(new.any(1).sum())
                s_med.append(float(s.median()))
                prior = dev.prior[dev.vid[lo:lo + step].long(), :L].to(torch.float64) * np.log(2.0)
                ref = torch.log_softmax(prior - 0.5 * dev.sZ * d2.sqrt(), dim=1)
                got = logphi[lo:lo + step, :L].to(torch.float64)
                live = ref.exp() >= 1e-6
                err_lp = max(err_lp, float((got - ref)[live].abs().max()))
                err_phi = max(err_phi, float((dev.phi[lo:lo + step, :L].to(torch.float64) - ref.exp()).abs().max()))
            tot = float(N) * L
            print("%-10s %-8s %9.0f | %.3e  %.3e | %.4f  %.4f | %.2e  %.2e"
                  % (name, "%s L=%d" % (shape, L), float(np.median(s_med)), old_pairs / tot, new_pairs / tot,
                     old_frames / N, new_frames / N, err_lp, err_phi), flush=True)
            dev.close()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
