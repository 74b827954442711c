"""Shim for `from node import Node` (offline tools unpickle `node.Node` objects written by
VarInf.write_nodes, /root/reference/src/compute_acc_frame_avg_dynamic.py:5,33-35)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnp_vae_amd.node import Node  # noqa: E402

Node.__module__ = "node"           # pickles then resolve through this module name
