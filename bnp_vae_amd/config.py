"""Hyper-parameters of the VI step.

The reference reads them from module globals that `from util import *` leaks into
`var_inf.py` (/root/reference/src/util.py:10-31) and from `args.num_splits`
(/root/reference/src/var_inf.py:223, default in src/train.py:23).  `VarInf()` takes no
arguments there, so the drop-in resolves each value in this order:

  1. keyword argument of `VarInf(...)`
  2. an attribute set on this module (`bnp_vae_amd.config.STDEV_THR = 16`)
  3. a `util` module already imported by the host application (the reference's own way)
  4. the reference's defaults (src/train.py:20-23, src/util.py:23-30)
"""
import sys

import numpy as np

DEFAULTS = dict(
    LATENT_CODE_SIZE=32,      # src/train.py:21
    STDEV_THR=32.0,           # src/train.py:22
    NUM_SPLITS=4,             # src/train.py:23
    GAMMA=10.0,               # src/util.py:24
    SIGMA_B_sqrinv=1.0,       # src/util.py:27,29
    SIGMA_Z_sqrinv=1.0,       # src/util.py:28,30
    BRANCHING_FACTOR=4,       # src/var_inf.py:92 (hard-coded there)
    NUM_LEVELS=3,             # src/var_inf.py:93-109 (three levels below the root)
)

_THIS = sys.modules[__name__]


def resolve(name, explicit=None):
    if explicit is not None:
        return explicit
    if hasattr(_THIS, name):
        return getattr(_THIS, name)
    util = sys.modules.get("util")
    if util is not None:
        if name == "NUM_SPLITS":
            args = getattr(util, "args", None)
            if args is not None and hasattr(args, "num_splits"):
                return args.num_splits
        elif hasattr(util, name):
            return getattr(util, name)
    return DEFAULTS[name]


def resolve_alpha(D, explicit=None):
    if explicit is not None:
        return np.asarray(explicit, dtype=np.float64)
    if hasattr(_THIS, "ALPHA"):
        return np.asarray(getattr(_THIS, "ALPHA"), dtype=np.float64)
    util = sys.modules.get("util")
    if util is not None and hasattr(util, "ALPHA") and np.size(util.ALPHA) == D:
        return np.asarray(util.ALPHA, dtype=np.float64)
    return np.zeros(D)                      # src/util.py:23
