#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (build container, needs /root/reference): wall-clock of the TRUE reference
`VarInf.update_variational_parameters` at BASELINE config 1's data shape (N = 60 000 frames, D = 32, 256-frame
videos), for BASELINE.md section 4.3.  The reference's tree is hard-coded (4-ary, 3 levels below the root: K = 85,
L = 64, src/var_inf.py:90-118), so that is the tree timed -- not config 1's 5-ary K = 31.

  python oracle/time_reference.py [N] [--second-call]

Prints the time of the first call (tree construction + one-hot phi + one VI iteration + split + merge) and, when
asked, of a second call (the steady-state step).  Nothing here is imported by the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_loader import load_reference, quiet_stdout          # noqa: E402
from oracle.vi_oracle import gaussian_mixture, make_latent_dict     # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 60_000
    D = 32
    vi_mod, _node_mod = load_reference(latent_code_size=D, stdev_thr=1e9, num_splits=2)
    Z = gaussian_mixture(N, D, 25, seed=13, spread=4.0)
    lc = make_latent_dict(Z, [256])
    np.random.seed(1)
    vi = vi_mod.VarInf()
    t0 = time.perf_counter()
    with quiet_stdout():
        vi.update_variational_parameters(lc)
    t1 = time.perf_counter()
    print("reference, N=%d D=%d V=%d, hard-coded 4-ary tree: first call %.1f s" % (N, D, len(lc), t1 - t0), flush=True)
    if "--second-call" in sys.argv:
        with quiet_stdout():
            vi.update_variational_parameters(lc)
        t2 = time.perf_counter()
        print("second call %.1f s = %.0f frames/s" % (t2 - t1, N / (t2 - t1)), flush=True)


if __name__ == "__main__":
    main()
