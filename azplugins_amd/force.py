"""Common base of the force classes: per-type parameter dictionaries and the
per-particle result arrays HOOMD's ``Force`` exposes (``forces``, ``energies``,
``torques``, ``virials``, ``energy``)."""

import numpy as np

from . import _lib


class TypeParameter:
    """``hoomd.data.typeparam.TypeParameter`` + ``TypeParameterDict`` reduced to
    what the reference's pair/bond classes use: a dict keyed by a type name
    (``len_keys=1``) or an unordered pair of type names (``len_keys=2``) whose
    values are validated against a schema ``{key: type | default}``."""

    def __init__(self, name, schema, len_keys, on_change=None, readback=None):
        self.name = name
        self.schema = dict(schema)
        self.len_keys = len_keys
        self._data = {}
        self._on_change = on_change
        self._readback = readback

    def _key(self, key):
        if self.len_keys == 1:
            if not isinstance(key, str):
                raise KeyError("%s keys are single type names, got %r" % (self.name, key))
            return key
        if not (isinstance(key, (tuple, list)) and len(key) == 2 and all(isinstance(k, str) for k in key)):
            raise KeyError("%s keys are pairs of type names, got %r" % (self.name, key))
        return tuple(sorted(key))

    def _validate(self, value):
        if not isinstance(self.schema, dict) or not self.schema:
            return value
        if not isinstance(value, dict):
            raise TypeError("%s values must be dicts" % self.name)
        unknown = set(value) - set(self.schema)
        if unknown:
            raise ValueError("%s: unknown keys %s (expected %s)" % (self.name, sorted(unknown), sorted(self.schema)))
        out = {}
        for k, spec in self.schema.items():
            if k in value:
                typ = spec if isinstance(spec, type) else type(spec)
                try:
                    out[k] = typ(value[k])
                except (TypeError, ValueError):
                    raise TypeError("%s[%r] must be convertible to %s" % (self.name, k, typ.__name__))
            elif isinstance(spec, type):
                raise ValueError("%s: missing required key %r" % (self.name, k))
            else:
                out[k] = spec
        return out

    def __setitem__(self, key, value):
        self._data[self._key(key)] = self._validate(value)
        if self._on_change:
            self._on_change()

    def __getitem__(self, key):
        k = self._key(key)
        if self._readback is not None:
            rb = self._readback(k)
            if rb is not None:
                return _ParamView(self, k, rb)
        if k not in self._data:
            # HOOMD hands out a (partially filled) dict for a type that has no
            # parameters yet, e.g. ``barrier.params["A"].update(...)``
            return _ParamView(self, k, {})
        return _ParamView(self, k, self._data[k]) if isinstance(self._data[k], dict) else self._data[k]

    def __contains__(self, key):
        return self._key(key) in self._data

    def keys(self):
        return self._data.keys()

    def get_raw(self, key, default=None):
        return self._data.get(self._key(key), default)


class _ParamView(dict):
    """dict returned by ``params[key]``; ``update`` / item assignment write through
    (with validation), as HOOMD's TypeParameterDict entries do."""

    def __init__(self, owner, key, values):
        super().__init__(values)
        self._owner = owner
        self._key_ = key

    def update(self, *args, **kwargs):
        merged = dict(self)
        merged.update(*args, **kwargs)
        self._owner[self._key_] = merged
        super().update(self._owner._data[self._owner._key(self._key_)])

    def __setitem__(self, k, v):
        self.update({k: v})


class ScalarTypeParameter(TypeParameter):
    """Per-type-pair scalar (``r_cut`` / ``r_on``) with a default."""

    def __init__(self, name, default, on_change=None):
        super().__init__(name, {}, 2, on_change)
        self.default = default

    def _validate(self, value):
        return float(value)

    def __getitem__(self, key):
        k = self._key(key)
        if k in self._data:
            return self._data[k]
        if self.default is None:
            raise KeyError("%s[%r] is not set and there is no default" % (self.name, key))
        return float(self.default)


class Force:
    """Holds the result buffers of one force compute (HOOMD ``ForceCompute``):
    force (N,4) = (fx, fy, fz, energy), virial (6, N), torque (N,4)."""

    def __init__(self):
        self._state = None
        self._force = None
        self._virial = None
        self._torque = None
        self.compute_virial = False
        self._computed_generation = None

    @property
    def _attached(self):
        return self._state is not None

    def _attach(self, sim):
        import torch

        self._sim = sim
        self._state = sim.state
        N = self._state.N
        dev = self._state.device
        self._force = torch.zeros((N, 4), dtype=torch.float64, device=dev)
        self._torque = torch.zeros((N, 4), dtype=torch.float64, device=dev)
        self._virial = torch.zeros((6, N), dtype=torch.float64, device=dev)
        self._dirty = True
        self._computed_generation = None

    def _require(self):
        if not self._attached:
            raise _lib.AzpError("%s is not attached to a simulation; call sim.run(0) first" % type(self).__name__)

    def _ensure_buffers(self):
        """Result buffers follow the number of local particles (it changes when particles
        migrate between the ranks of a decomposed run)."""
        import torch

        N = self._state.N
        if self._force.shape[0] != N:
            dev = self._state.device
            self._force = torch.zeros((N, 4), dtype=torch.float64, device=dev)
            self._torque = torch.zeros((N, 4), dtype=torch.float64, device=dev)
            self._virial = torch.zeros((6, N), dtype=torch.float64, device=dev)

    def compute(self, timestep=None):
        self._require()
        raise NotImplementedError

    # -- HOOMD Force properties ---------------------------------------------
    @property
    def forces(self):
        self._require()
        return self._force[:, :3].cpu().numpy()

    @property
    def energies(self):
        self._require()
        return self._force[:, 3].cpu().numpy()

    @property
    def energy(self):
        self._require()
        return float(self._force[:, 3].sum().item())

    @property
    def torques(self):
        self._require()
        return self._torque[:, :3].cpu().numpy()

    @property
    def virials(self):
        self._require()
        if not self.compute_virial:
            return None
        return self._virial.t().cpu().numpy()

    # device views, for integrators and benchmarks
    @property
    def force_tensor(self):
        return self._force

    @property
    def torque_tensor(self):
        return self._torque
