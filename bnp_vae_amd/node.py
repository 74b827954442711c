"""Tree node record with the attribute surface of /root/reference/src/node.py:6-17.

`node_id`, `isLeafNode`, `parent`, `alpha (D,) f64`, `sigmasqr_inv`, `children` are plain
host attributes.  `phi (N,1) f64` and `gamma (V,2) f64` look the same to a caller but are
backed by the device-resident matrices of the owning `VarInf` (a column of the N x L
responsibility matrix, a row block of the V x K x 2 stick table): reading them downloads
on demand, assigning them stores a host value that is pushed to the device before the next
kernel that needs it.  The reference keeps an (N,1) fp64 `phi` on EVERY node, read only on
leaves -- K*N*8 bytes, 75 GB at BASELINE config 3; here an internal node's
`phi` is materialised only if somebody asks for it.
"""
import numpy as np

_DEFAULT = object()


class Node(object):
    def __init__(self, node_id, num_vids, num_frames, parent, LATENT_CODE_SIZE, GAMMA):
        self.node_id = node_id
        self.isLeafNode = True
        self.parent = parent
        self._owner = None          # VarInf that holds the device state
        # same draw from the global NumPy stream as src/node.py:12 (keeps seeded runs aligned)
        self._alpha = np.random.normal(size=LATENT_CODE_SIZE)
        self._sigmasqr_inv = 1.0
        self.children = []
        self._num_vids = num_vids
        self._num_frames = num_frames
        self._gamma_hyper = GAMMA
        self._phi_host = None       # host value (authoritative when not None)
        self._phi_dev = None        # (generation, column, scale) into the owner's device phi
        self._gamma_host = _DEFAULT  # host value, None (root-as-leaf), or _DEFAULT
        self._gamma_dev = None      # (generation, node row) into the owner's device gamma

    # ------------------------------------------------------------------ alpha / sigmasqr_inv
    # Plain host values, as in the reference -- but after an M-step the owner's device copy is newer; it
    # is downloaded (all nodes at once) the first time anybody looks, not after every call.
    @property
    def alpha(self):
        o = self.__dict__.get('_owner')
        if o is not None and o._params_stale:
            o._pull_params()
        return self._alpha

    @alpha.setter
    def alpha(self, value):
        o = self.__dict__.get('_owner')
        if o is not None:
            if o._params_stale:
                o._pull_params()
            o._host_params_ver += 1
        self._alpha = value

    @property
    def sigmasqr_inv(self):
        o = self.__dict__.get('_owner')
        if o is not None and o._params_stale:
            o._pull_params()
        return self._sigmasqr_inv

    @sigmasqr_inv.setter
    def sigmasqr_inv(self, value):
        o = self.__dict__.get('_owner')
        if o is not None:
            if o._params_stale:
                o._pull_params()
            o._host_params_ver += 1
        self._sigmasqr_inv = value

    # ------------------------------------------------------------------ phi
    @property
    def phi(self):
        if self._phi_host is not None:
            return self._phi_host
        if self._phi_dev is not None and self._owner is not None:
            got = self._owner._fetch_phi(self)
            if got is not None:
                return got
        # src/node.py:14 default
        self._phi_host = np.ones(shape=(self._num_frames, 1))
        return self._phi_host

    @phi.setter
    def phi(self, value):
        self._phi_host = None if value is None else np.asarray(value, dtype=np.float64).reshape(-1, 1)
        self._phi_dev = None
        if self._owner is not None:
            self._owner._note_phi_changed()

    # ------------------------------------------------------------------ gamma
    @property
    def gamma(self):
        if self._gamma_host is None:
            return None
        if self._gamma_host is not _DEFAULT:
            return self._gamma_host
        if self._gamma_dev is not None and self._owner is not None:
            got = self._owner._fetch_gamma(self)
            if got is not None:
                return got
        g = np.ones(shape=(self._num_vids, 2))        # src/node.py:15-16 default
        g[:, 1] *= self._gamma_hyper
        self._gamma_host = g
        return g

    @gamma.setter
    def gamma(self, value):
        self._gamma_host = None if value is None else np.asarray(value, dtype=np.float64)
        self._gamma_dev = None

    # ------------------------------------------------------------------ pickling
    # write_nodes pickles the pre-order node list (src/var_inf.py:474-478) for the offline
    # tools, which read .parent/.isLeafNode/.children/.alpha/.node_id.  Device-backed values
    # are materialised; the owner reference is dropped.
    def __getstate__(self):
        self.alpha                   # refresh the host values if the device copy is newer
        d = dict(self.__dict__)
        keep_phi = True if self._owner is None else self._owner.pickle_phi
        d['_phi_host'] = self.phi if keep_phi else None
        g = self.gamma
        d['_gamma_host'] = g
        d['_phi_dev'] = None
        d['_gamma_dev'] = None
        d['_owner'] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        if 'alpha' in self.__dict__:                 # pickles written before alpha became a property
            self._alpha = self.__dict__.pop('alpha')
            self._sigmasqr_inv = self.__dict__.pop('sigmasqr_inv', 1.0)
        if self.__dict__.get('_gamma_host', _DEFAULT) is not None and \
                not isinstance(self.__dict__.get('_gamma_host'), np.ndarray):
            self._gamma_host = _DEFAULT
