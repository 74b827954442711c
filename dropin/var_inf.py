"""Shim: put this directory first on sys.path (or copy the two files over the reference's
src/var_inf.py and src/node.py) and `from var_inf import VarInf` in model.py resolves to the
MI355X-native implementation.  See INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnp_vae_amd.var_inf import VarInf  # noqa: E402,F401
from node import Node                   # noqa: E402,F401
