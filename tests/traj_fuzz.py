"""Random small drop-in trajectories against OracleVarInf (helper of tests/test_gpu_round3.py::
test_random_trajectories_against_oracle, and a sweep on its own: `python tests/traj_fuzz.py [first_seed] [count]`).

A seed fixes D, the initial tree, num_splits, the mixture, N, the video lengths and the split threshold; five calls
with fresh drifting latents; after every call the product must have the oracle's tree, node parameters (1e-4 of
the largest alpha, 1e-4 of each sigma) and phi (2e-4).  When the tree collapses to its root the reference raises
TypeError on the next call (src/var_inf.py:372-373) and so must the product.
Sweep of 500 seeds (round 3): 3 differ, each at a near-tie of a k-means arg-max inside the accuracy of the fp32
responsibilities (relative gap between the two frames' scores 4e-7, 4e-7 and 1.7e-6: seeds 175, 222, 362); none in
the tree logic.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.vi_oracle import OracleVarInf, snapshot, make_latent_dict   # noqa: E402


def latents(N, D, mu, ep, seed, noise):
    comp = np.random.RandomState(seed).permutation(N) % mu.shape[0]
    eps = np.random.RandomState(seed + 1000 + ep).normal(size=(N, D)) * noise
    return (mu[comp] * (1.0 + 0.02 * ep) + eps).astype(np.float32)


def one(seed, verbose=True):
    from bnp_vae_amd import VarInf
    rng = np.random.RandomState(seed)
    D = int(rng.choice([8, 32, 64, 100, 128]))
    bf, levels, ns = int(rng.randint(2, 6)), int(rng.randint(1, 4)), int(rng.randint(2, 5))
    ncomp = int(rng.randint(5, 60))
    N = int(rng.randint(1500, 5000))
    noise = float(rng.choice([0.3, 0.6, 1.0]))
    spread = float(rng.choice([2.0, 3.0, 5.0]))
    thr = float(rng.choice([-100.0, noise * np.sqrt(D) * 0.7, noise * np.sqrt(D) * 1.5]))
    fpv = []
    while sum(fpv) < N:
        fpv.append(int(rng.randint(50, 700)))
    fpv[-1] -= sum(fpv) - N
    if fpv[-1] == 0:
        fpv.pop()
    mu = rng.normal(scale=spread, size=(ncomp, D))
    kw = dict(latent_code_size=D, stdev_thr=thr, num_splits=ns, branching=bf, levels=levels)
    ora, vi = OracleVarInf(**kw), VarInf(**kw)
    tag = "seed %d: N=%d D=%d bf=%d levels=%d ns=%d comps=%d noise=%g thr=%.3g" % (seed, N, D, bf, levels, ns, ncomp, noise, thr)
    ks = []
    for ep in range(5):
        Z = latents(N, D, mu, ep, seed, noise)
        lc = make_latent_dict(Z, fpv)
        np.random.seed(13 + ep)
        try:
            ora.update_variational_parameters(lc)
        except Exception as e:                      # the reference's own failure modes (restated by the oracle)
            try:
                np.random.seed(13 + ep)
                vi.update_variational_parameters(lc)
                return tag, "ERROR BEHAVIOUR differs: oracle raised %s in call %d, product did not (K %s)" % (type(e).__name__, ep, ks)
            except Exception as e2:
                if type(e2) is not type(e):
                    return tag, "ERROR BEHAVIOUR differs: oracle %s, product %s" % (type(e).__name__, type(e2).__name__)
            return tag, "ok (both raise %s in call %d; K %s)" % (type(e).__name__, ep, ks)
        np.random.seed(13 + ep)
        vi.update_variational_parameters(lc)
        want, got = snapshot(ora), snapshot(vi)
        ks.append(len(want['ids']))
        if list(got['ids']) != list(want['ids']) or not np.array_equal(got['is_leaf'], want['is_leaf']):
            return tag, "STRUCTURE differs after call %d (oracle K=%d, product K=%d; K so far %s)" % (ep, len(want['ids']), len(got['ids']), ks)
        sc = max(1.0, float(np.abs(want['alpha']).max()))
        da = float(np.abs(got['alpha'] - want['alpha']).max()) / sc
        ds = float(np.abs(got['sigma'] / want['sigma'] - 1).max())
        dp = float(np.abs(got['phi'] - want['phi']).max())
        if da > 1e-4 or ds > 1e-4 or dp > 2e-4:
            return tag, "VALUES differ after call %d: alpha %.2e sigma %.2e phi %.2e (K %s)" % (ep, da, ds, dp, ks)
    return tag, "ok (K per call %s)" % ks


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    for s in range(first, first + count):
        tag, res = one(s)
        bad += not res.startswith("ok")
        print(tag + " -> " + res, flush=True)
    print("%d of %d differ" % (bad, count))
