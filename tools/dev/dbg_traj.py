import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.vi_oracle import OracleVarInf, snapshot, make_latent_dict
from tests.test_gpu_round3 import _clustered_latents
from bnp_vae_amd import VarInf
N, D, ncomp, seed = 6450, 128, 800, 7
fpv = [300, 177, 256, 640, 411, 160, 512, 616, 3378]
kw = dict(latent_code_size=D, stdev_thr=-100.0, num_splits=4, branching=4, levels=5)
ora = OracleVarInf(**kw); vi = VarInf(**kw)
for ep in range(6):
    Z = _clustered_latents(N, D, ncomp, seed, ep, 3.0, 0.3)
    lc = make_latent_dict(Z, fpv)
    if ep == 5:
        # step through call 5 by hand
        Zo = ora.get_matrix_from_dict(lc)
        np.random.seed(13 + ep)
        for v in (ora, vi):
            v.compute_sigma_alpha(Zo); v.compute_gamma(); v.compute_phi(Zo)
        po, pp = snapshot(ora)['phi'], snapshot(vi)['phi']
        print("phi diff before split", np.abs(po - pp).max())
        leaves = ora.leaves()
        for j, l in enumerate(leaves):
            if np.mean(l.phi) >= 1e-2:
                w = l.phi[:, 0]
                cen = ora.weighted_kmeans(Zo, w, 4)
                rows_o = [int(np.argmin(np.abs(Zo - c).sum(1))) for c in cen]
                rows_p, _ = vi._dev.kmeans_seed(j, 4)
                flag = "" if rows_o == list(rows_p) else "  <-- differs"
                print(l.node_id, "mass %.3f" % w.sum(), rows_o, list(rows_p), flag)
                if flag:
                    for r in (rows_o[0], rows_p[0]):
                        others_o = np.delete(po[r], j).sum()
                        others_p = np.delete(pp[r].astype(np.float64), j).sum()
                        print("   row", r, "oracle phi", repr(w[r]), "1-phi", 1.0 - w[r], "oracle c", others_o,
                              "product phi", repr(pp[r, j]), "product c", others_p)
                    ones = np.nonzero(w == 1.0)[0]
                    print("   oracle rows with phi == 1.0 exactly:", len(ones), ones[:8])
                    cp = pp.astype(np.float64).sum(1) - pp[:, j].astype(np.float64)
                    sp = 1.0 / (1.0 + cp)
                    onesp = np.nonzero((sp == 1.0) & (pp[:, j] >= 0.99999))[0]
                    print("   product rows with 1/(1+c) == 1.0:", len(onesp), onesp[:8])
        break
    np.random.seed(13 + ep); ora.update_variational_parameters(lc)
    np.random.seed(13 + ep); vi.update_variational_parameters(lc)
    print(ep, int(snapshot(ora)['is_leaf'].sum()), list(snapshot(ora)['ids']) == list(snapshot(vi)['ids']))
