"""MI355X-native (gfx950) drop-in for the variational-inference step of bnp-vae.

    from bnp_vae_amd import VarInf, Node        # same surface as src/var_inf.py, src/node.py

Importing the package is cheap and GPU-free; the HIP library (libbnpvi.so, C ABI in
include/bnpvi.h) is loaded on first device use and its absence is a hard error.
"""
from .node import Node            # noqa: F401
from .var_inf import VarInf       # noqa: F401

__all__ = ["VarInf", "Node"]
