"""ctypes binding of libbnpvi.so (the C ABI declared in include/bnpvi.h).

There is NO CPU fallback: importing this module fails loudly when the HIP library has not
been built (`python -c 'import __graft_entry__ as g; g.build()'` or
`make -C bnp_vae_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BNPVI_LIB") or os.path.join(_HERE, "libbnpvi.so")   # BNPVI_LIB: dev builds

OK = 0
ESTEP_SOFT = 0
ESTEP_HARD = 1
ESTEP_OPERANDS_READY = 0x100
ESTEP_STATS_S0 = 0x200
EUNSUPPORTED = 4          # BNPVI_EUNSUPPORTED: the entry point does not take this shape (the caller has another way)


class BnpviError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "bnp_vae_amd: %s is missing -- the gfx950 HIP library must be built first "
            "(make -C bnp_vae_amd/csrc); there is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and loads it
    # by absolute path; if libbnpvi is loaded first it pulls in /opt/rocm's copy of the same
    # SONAME and the second runtime to initialise reports "no ROCm-capable device".  Importing
    # torch first makes the dynamic linker resolve libbnpvi's NEEDED entry to torch's copy.
    import torch  # noqa: F401
    return C.CDLL(LIB_PATH)


lib = _load()

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double

EXPORTS = {
    # name: (restype, argtypes)
    "bnpvi_abi_version": (_i, []),
    "bnpvi_last_error": (C.c_char_p, []),
    "bnpvi_padded_dim": (_i, [_i]),
    "bnpvi_padded_leaves": (_i, [_i]),
    "bnpvi_ctx_create": (_i, [_i, C.POINTER(_p)]),
    "bnpvi_ctx_destroy": (_i, [_p]),
    "bnpvi_profile_enable": (_i, [_p, _i]),
    "bnpvi_profile_read": (_i, [_p, _i, C.POINTER(_d), C.POINTER(_i)]),
    "bnpvi_tree_create": (_i, [_p, _i, _p, _p, C.POINTER(_p)]),
    "bnpvi_tree_destroy": (_i, [_p]),
    "bnpvi_tree_num_leaves": (_i, [_p]),
    "bnpvi_tree_leaf_nodes": (_i, [_p, _p]),
    "bnpvi_leaf_stats": (_i, [_p, _p, _i, _p, _i, _p, _i64, _i, _i, _i, _i, _p, _p, _p]),
    "bnpvi_leaf_stats_noalloc": (_i, [_p, _p, _i, _p, _i, _p, _i64, _i, _i, _i, _i, _p, _p, _p]),
    "bnpvi_node_params": (_i, [_p, _p, _p, _i, _i, _d, _d, _p, _p, _p, _p]),
    "bnpvi_gamma_update": (_i, [_p, _p, _p, _i, _d, _p, _p]),
    "bnpvi_gamma_remap": (_i, [_p, _p, _i, _p, _i, _i, _p, _d, _p]),
    "bnpvi_prior_table": (_i, [_p, _p, _p, _p, _i, _d, _p, _p, _p, _p]),
    "bnpvi_gamma_prior": (_i, [_p, _p, _p, _p, _i, _d, _d, _p, _p, _p]),
    "bnpvi_params_prior": (_i, [_p, _p, _p, _i, _i, _d, _d, _p, _p, _p, _p, _i, _d, _p, _p, _p]),
    "bnpvi_estep": (_i, [_p, _p, _p, _i, _p, _i64, _i, _p, _p, _d, _i, _p, _p, _p, _p, _p]),
    "bnpvi_split_max_columns": (_i, []),
    "bnpvi_split_gate": (_i, [_p, _p, _i, _d, _p, _i, _p, _p]),
    "bnpvi_leaf_s2": (_i, [_p, _p, _p, _i, _i64, _i, _p, _p, _i, _p, _p, _i, _p, _p]),
    "bnpvi_kmeans_pass_multi": (_i, [_p, _p, _i, _i64, _i, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
    "bnpvi_kmeans_max_columns": (_i, [_i]),
    "bnpvi_kmeans_first_nested": (_i, [_p, _p, _p, _i, _i64, _p, _i, _p, _p, _p]),
    "bnpvi_path_distances": (_i, [_p, _p, _p, _i, _i64, _i, _p, _p, _i, _p, _p, _p]),
    "bnpvi_remap_columns": (_i, [_p, _p, _i, _p, _i, _i64, _i, _p, _p, _p]),
}

for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc, what=""):
    if rc != OK:
        msg = lib.bnpvi_last_error()
        raise BnpviError("%s failed (code %d): %s" % (what or "libbnpvi call", rc,
                                                      msg.decode() if msg else "?"))


def padded_dim(D):
    r = lib.bnpvi_padded_dim(int(D))
    if r < 0:
        raise BnpviError("latent dim %d unsupported (1..256)" % D)
    return r


def padded_leaves(L):
    return lib.bnpvi_padded_leaves(int(L))
