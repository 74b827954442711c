"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (build container only).

The reference (`/root/reference/src/var_inf.py`, `node.py`) is Python-2 source that cannot
be imported as committed.  This module turns the two files into
Python-3 *in memory* with `lib2to3` (nothing is written anywhere, no reference text is
copied into this repository), installs a stub `util` module that provides the names
`from util import *` must supply (`src/util.py:10-31`), and returns live `VarInf` / `Node`
classes.  It is used ONLY by `oracle/make_golden.py` (to emit fixtures under
`tests/golden/`) and by `tests/test_oracle_vs_reference.py` (skipped when
`/root/reference` is absent, e.g. on the GPU box).  Nothing in the product path imports it.
"""
from __future__ import annotations

import io
import os
import sys
import types
import contextlib

REF_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_SRC, "var_inf.py"))


def _to_py3(path: str) -> str:
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3.refactor import RefactoringTool, get_fixers_from_package
        tool = RefactoringTool(get_fixers_from_package("lib2to3.fixes"))
        with open(path, "r") as f:
            src = f.read()
        if not src.endswith("\n"):
            src += "\n"
        return str(tool.refactor_string(src, os.path.basename(path)))


class _Args:
    def __init__(self, num_splits):
        self.num_splits = num_splits


def load_reference(latent_code_size=32, stdev_thr=32.0, num_splits=2, quiet=True):
    """Return (var_inf_module, node_module) of the transformed reference.

    Hyper-parameters mirror `src/util.py:23-31` / `src/train.py:21-23`.
    Residual Py2-ism: `l1_nodes[n/BF]` (`src/var_inf.py:100,106`) is true division under
    Py3; the two expressions are rewritten to `//` in the transformed text.
    """
    import numpy as np
    import resource

    util = types.ModuleType("util")
    util.np = np
    util.sys = sys
    util.resource = resource
    util.args = _Args(num_splits)
    util.batch_size = 64
    util.LATENT_CODE_SIZE = latent_code_size
    util.ALPHA = np.zeros(shape=latent_code_size)
    util.GAMMA = 10.0
    util.SIGMA_B = 1.0
    util.SIGMA_Z = 1.0
    util.SIGMA_B_sqrinv = 1.0
    util.SIGMA_Z_sqrinv = 1.0
    util.STDEV_THR = stdev_thr

    saved = {k: sys.modules.get(k) for k in ("util", "node", "var_inf")}
    try:
        sys.modules["util"] = util
        node_mod = types.ModuleType("node")
        node_src = _to_py3(os.path.join(REF_SRC, "node.py"))
        exec(compile(node_src, "<reference node.py (py3 transform)>", "exec"), node_mod.__dict__)
        sys.modules["node"] = node_mod

        vi_src = _to_py3(os.path.join(REF_SRC, "var_inf.py"))
        vi_src = vi_src.replace("l1_nodes[n/BF]", "l1_nodes[n//BF]")
        vi_src = vi_src.replace("l2_nodes[n/BF]", "l2_nodes[n//BF]")
        vi_mod = types.ModuleType("var_inf")
        exec(compile(vi_src, "<reference var_inf.py (py3 transform)>", "exec"), vi_mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    if quiet:
        _silence(vi_mod)
        _silence(node_mod)
    vi_mod._util = util
    return vi_mod, node_mod


def _silence(mod):
    """The reference prints from every recursion step; route its `print` to a sink."""
    mod.__dict__["print"] = lambda *a, **k: None


@contextlib.contextmanager
def quiet_stdout():
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        yield
    finally:
        sys.stdout = old
