"""Debug one seed of tests/traj_fuzz.py: where do product and oracle part ways?"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import traj_fuzz as F
from oracle.vi_oracle import OracleVarInf, snapshot, make_latent_dict
from bnp_vae_amd import VarInf

seed = int(sys.argv[1])
rng = np.random.RandomState(seed)
D = int(rng.choice([8, 32, 64, 100, 128]))
bf, levels, ns = int(rng.randint(2, 6)), int(rng.randint(1, 4)), int(rng.randint(2, 5))
ncomp = int(rng.randint(5, 60)); N = int(rng.randint(1500, 5000))
noise = float(rng.choice([0.3, 0.6, 1.0])); spread = float(rng.choice([2.0, 3.0, 5.0]))
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
for ep in range(5):
    Z = F.latents(N, D, mu, ep, seed, noise)
    lc = make_latent_dict(Z, fpv)
    np.random.seed(13 + ep); ora.update_variational_parameters(lc)
    np.random.seed(13 + ep); vi.update_variational_parameters(lc)
    want, got = snapshot(ora), snapshot(vi)
    ids = list(want['ids'])
    print("call", ep, "K", len(ids), "same ids", ids == list(got['ids']))
    if ids != list(got['ids']):
        break
    da = np.abs(got['alpha'] - want['alpha']).max(1)
    badn = np.nonzero(da > 1e-3)[0]
    print("  nodes with alpha mismatch:", len(badn), [(ids[i], bool(want['is_leaf'][i]), float(da[i])) for i in badn[:12]])
    if len(badn):
        # is a mismatching leaf's alpha (an initial k-means centre) some row of Z on both sides?
        for i in badn[:6]:
            for name, s in (("oracle", want), ("product", got)):
                a = s['alpha'][i]
                d = np.abs(Z.astype(np.float64) - a).max(1)
                print("   ", ids[i], name, "nearest frame", int(d.argmin()), "max|diff| %.3g" % d.min())
        break

# ---- second part: rerun up to the diverging call and compare every pick of every seeding
print("---- seeding detail")
bad_call = ep
ora, vi = OracleVarInf(**kw), VarInf(**kw)
import bnp_vae_amd.engine as E
rec_o, rec_p = [], []
orig_wk = OracleVarInf.weighted_kmeans
def wk(data, weights, K, return_indices=False):
    cen, picked = orig_wk(data, weights, K, True)
    rec_o.append((np.array(weights, dtype=np.float64).copy(), list(picked)))
    return (cen, picked) if return_indices else cen
OracleVarInf.weighted_kmeans = staticmethod(wk)
orig_seed = E.DeviceVI.kmeans_seed_multi
def seed_multi(self, cols, num_centres, responsibilities=True):
    rows, cen = orig_seed(self, cols, num_centres, responsibilities)
    if len(cols) <= int(E.lib.bnpvi_kmeans_max_columns(self.D)):
        for c, r in zip(cols, np.asarray(rows)):
            rec_p.append((int(c), [int(x) for x in r]))
    return rows, cen
E.DeviceVI.kmeans_seed_multi = seed_multi
for e2 in range(bad_call + 1):
    Z = F.latents(N, D, mu, e2, seed, noise)
    lc = make_latent_dict(Z, fpv)
    rec_o.clear(); rec_p.clear()
    np.random.seed(13 + e2); ora.update_variational_parameters(lc)
    np.random.seed(13 + e2); vi.update_variational_parameters(lc)
print("seedings in call %d: oracle %d, product %d" % (bad_call, len(rec_o), len(rec_p)))
Z64 = Z.astype(np.float64)
for j, ((w, po), (col, pp)) in enumerate(zip(rec_o, rec_p)):
    if po != pp:
        k = [a == b for a, b in zip(po, pp)].index(False)
        print("split %d (column %d): oracle picks %s, product picks %s; first difference at centre %d" % (j, col, po, pp, k))
        wn = w / w.sum()
        if k == 0:
            print("    weights: oracle's %.17g, product's %.17g" % (w[po[0]], w[pp[0]]))
        else:
            cen = Z64[po[:k]]
            closest = np.min(np.linalg.norm(Z64[:, None, :] - cen[None], axis=2), axis=1)
            sc = wn * closest
            for r in (po[k], pp[k]):
                print("    frame %d: weight %.17g closest %.17g score %.17g" % (r, w[r], closest[r], sc[r]))
            print("    relative score gap %.3g" % (abs(sc[po[k]] - sc[pp[k]]) / sc[po[k]]))
