#!/usr/bin/env python
"""Randomised shape sweep over the kernel regimes (development tool, GPU): for each random (N, D, tree, video
lengths) one hard E-step, two iterations, then the invariants and fp64 comparisons the parity tests use --
rows of phi sum to one, S0 adds up to N, log phi against the fp64 evaluation of the same logits (1e-4 + four ulps of
the largest distance, where phi > 1e-6), the M-step's [S0 | S1] and per-video masses against torch fp64, the E-step's S0' against the column sums.

    python tools/fuzz_shapes.py [--cases 100] [--seed 1] [--max-elems 60000000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-elems", type=int, default=60_000_000, help="cap on N * L")
    args = ap.parse_args()
    import torch
    from bench import regular_tree
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    rng = np.random.RandomState(args.seed)
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        D = int(rng.choice([3, 8, 17, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256]))
        bf = int(rng.randint(2, 9))
        levels = int(rng.randint(1, 6))
        while bf ** levels > 8000:
            levels -= 1
        parent, is_leaf = regular_tree(bf, levels)
        K, L = len(parent), int(is_leaf.sum())
        nmax = max(1, min(40000, args.max_elems // max(L, 1)))
        N = int(rng.randint(1, nmax + 1))
        # videos of random lengths
        lens = []
        while sum(lens) < N:
            lens.append(int(rng.randint(1, 400)))
        lens[-1] -= sum(lens) - N
        if lens[-1] == 0:
            lens.pop()
        vid = np.repeat(np.arange(len(lens)), lens).astype(np.int32)
        V = len(lens)
        scale = float(rng.choice([0.05, 1.0, 1.0, 4.0, 40.0]))
        mu = rng.normal(scale=4.0 * scale, size=(L, D))
        Z = (mu[rng.randint(0, L, size=N)] + scale * rng.normal(size=(N, D))).astype(np.float32)
        tag = "case %d: N=%d D=%d bf=%d levels=%d (K=%d, L=%d) V=%d scale=%g" % (case, N, D, bf, levels, K, L, V, scale)
        try:
            dev = DeviceVI(device=0)
            dev.set_latents(Z, vid, V)
            dev.set_tree(parent, is_leaf, scale * rng.normal(size=(K, D)), np.ones(K))
            dev.default_gamma()
            dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
            for _ in range(2):
                dev.iteration(want_estats=True)
            logphi, _ = dev.estep(want_logphi=True, want_stats=True, s0_only=False)
            torch.cuda.synchronize()
            phi = dev.phi[:, :L].to(torch.float64)
            rows = phi.sum(1)
            assert float((rows - 1).abs().max()) < 2e-5, "rows of phi: %g" % float((rows - 1).abs().max())
            assert bool(torch.all(dev.phi[:, L:] == 0)), "padding columns"
            col = phi.sum(0)
            est = dev.estats[:, 0]
            assert float((est - col).abs().max()) <= 2e-5 * max(1.0, float(col.max())) + 1e-3, "S0' vs columns"
            nol = torch.from_numpy(np.asarray(dev.tree.node_of_leaf, dtype=np.int64)).to(dev.device)
            a64 = dev.alpha[nol].to(torch.float32).to(torch.float64)
            R = min(N, 2048)
            z64 = dev.Z[:R, :D].to(torch.float64)
            d = torch.cdist(z64, a64)
            pri = dev.prior[dev.vid[:R].long(), :L].to(torch.float64) * np.log(2.0)
            ref = torch.log_softmax(pri - 0.5 * dev.sZ * d, dim=1)
            live = ref.exp() > 1e-6
            err = float((logphi[:R, :L].to(torch.float64) - ref)[live].abs().max())
            # 1e-4 absolute (the parity bar) plus what fp32 cannot resolve: the kernels accept a distance good to
            # 2^-22 of itself (csrc/estep_kernels.h), i.e. 0.5 * 2^-22 * d in the logit -- 3.6e-4 at d = 3000
            dmax = float(d[live].max())
            assert err < 1e-4 + 0.5 * 2.0 ** -22 * dmax, "log phi vs fp64: %g (largest live distance %g)" % (err, dmax)
            s2ref = (ref.exp() * d * d).sum(0) if R == N else None
            if s2ref is not None:
                s2 = dev.estats[:, 1]
                assert float((s2 - s2ref).abs().max()) <= 2e-4 * max(1.0, float(s2ref.max())), "S2 vs fp64"
            dev.leaf_stats(want_q=False, want_mass=True)
            torch.cuda.synchronize()
            s1 = phi.T @ dev.Z[:, :D].to(torch.float64)
            assert float((dev.stats[:, 1:1 + D] - s1).abs().max()) <= 4e-6 * max(float(s1.abs().max()), 1e-30), "S1"
            assert float((dev.stats[:, 0] - col).abs().max()) <= 2e-6 * max(1.0, float(col.max())) + 1e-5, "S0"
            pv = torch.zeros((V, L), dtype=torch.float64, device=dev.device).index_add_(0, dev.vid.long(), phi)
            assert float((dev.mass[:, :L] - pv).abs().max()) <= 2e-6 * max(1.0, float(pv.max())) + 1e-5, "masses"
            dev.close()
        except AssertionError as e:
            bad += 1
            print("FAIL " + tag + ": " + str(e), flush=True)
            continue
        if case % 10 == 9:
            print("%d cases, %d failed, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
        torch.cuda.empty_cache()
    print("done: %d cases, %d failed" % (args.cases, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
