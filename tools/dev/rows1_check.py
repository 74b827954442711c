#!/usr/bin/env python
"""Development check: the one-pass row E-step (S0' only) + gated S2 against the two-pass row kernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bnp_vae_amd import _lib
from bnp_vae_amd.engine import DeviceVI


def tree_with_leaves(L):
    parent, is_leaf = [-1], [0]
    for _ in range(L):
        parent.append(0); is_leaf.append(1)
    return np.asarray(parent, np.int32), np.asarray(is_leaf, np.uint8)


def run(N, D, L, fpv, far=0.0, seed=0):
    rs = np.random.RandomState(seed)
    parent, is_leaf = tree_with_leaves(L)
    K = len(parent)
    alpha = rs.normal(size=(K, D)) * (far if far > 0 else 1.0)
    mu = alpha[1:]
    comp = rs.randint(0, L, size=N)
    comp[: N // 3] = rs.randint(0, max(1, L // 50), size=N // 3)          # a few heavy leaves: the gate fires
    Z = (mu[comp] + (0.5 if far > 0 else 3.0) * rs.normal(size=(N, D))).astype(np.float32)
    if far > 0:
        Z[:7] = mu[comp[:7]].astype(np.float32)                             # coinciding frames
    vid = (np.arange(N) // fpv).astype(np.int32)
    V = int(vid.max()) + 1
    dev = DeviceVI(device=0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, alpha, np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    dev.leaf_stats(False, True); dev.params_prior()
    dev.estep(want_stats=True, s0_only=False)            # two-pass rows / fused: S0' and S2
    phi_ref = dev.phi.clone(); est_ref = dev.estats.clone()
    dev.phi.zero_()
    dev.estep(want_stats=True, s0_only=True)
    torch.cuda.synchronize()
    dphi = float((dev.phi - phi_ref).abs().max())
    ds0 = float((dev.estats[:, 0] - est_ref[:, 0]).abs().max() / est_ref[:, 0].abs().max())
    dev.gated_s2(N)
    cnt = dev.gate_count.cpu().numpy(); cols = dev.gate_cols.cpu().numpy()[:cnt[0]]
    s2 = dev.gate_s2.cpu().numpy()[:cnt[0]]
    want = est_ref[:, 1].cpu().numpy()[cols]
    gate_ref = np.nonzero(est_ref[:, 0].cpu().numpy() >= 1e-2 * N)[0]
    ok_gate = np.array_equal(gate_ref, cols)
    ds2 = float(np.max(np.abs(s2 - want) / np.maximum(want, 1e-30))) if len(cols) else 0.0
    # fp64 reference of S2 for the gated columns
    if len(cols):
        ph = dev.phi[:, torch.from_numpy(cols).long().to(dev.device)].double()
        A = dev.alpha[1:][torch.from_numpy(cols).long().to(dev.device)].float().double()
        d2 = ((dev.Z[:, None, :D].double() - A[None]) ** 2).sum(-1) if N * len(cols) * D < 2e8 else None
        ds2x = float(((ph * d2).sum(0).cpu().numpy() - s2).__abs__().max() / s2.max()) if d2 is not None else -1
    else:
        ds2x = 0.0
    rows = dev.phi[:, :L].sum(1)
    drow = float((rows - 1).abs().max())
    print("N=%d D=%d L=%d fpv=%d far=%g: max|dphi|=%.2e relS0=%.2e gated=%d gate_ok=%s relS2(vs 2pass)=%.2e relS2(fp64)=%.2e rows=%.1e pad=%.1e"
          % (N, D, L, fpv, far, dphi, ds0, len(cols), ok_gate, ds2, ds2x, drow,
             float(dev.phi[:, L:].abs().max()) if dev.tree.Lp > L else 0.0))
    bad = dphi > 2e-6 or ds0 > 1e-6 or not ok_gate or ds2x > 1e-5 or drow > 1e-5
    dev.close()
    return bad


if __name__ == "__main__":
    bad = False
    for (N, D, L, fpv, far) in [(4099, 64, 512, 256, 0), (5000, 64, 100 + 1, 7, 0), (3000, 64, 300, 33, 0),
                               (3001, 32, 512, 256, 0), (2000, 32, 130, 5, 0), (2500, 128, 256, 100, 0),
                               (1999, 128, 200, 9, 0), (1500, 256, 128, 64, 0), (1501, 256, 100 + 20, 3, 0),
                               (4096, 64, 512, 256, 32.0), (3000, 64, 216, 17, 32.0), (2000, 17, 480, 40, 0),
                               (300000, 64, 512, 256, 0)]:
        bad |= run(N, D, L, fpv, far)
    print("FAIL" if bad else "OK")
    sys.exit(1 if bad else 0)
