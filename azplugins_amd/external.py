"""One-body external potentials: mirror of ``hoomd.azplugins.external``
(reference ``src/external.py``) on libazp's barrier kernels."""

import ctypes as C

import numpy as np

from . import _lib
from .force import Force, TypeParameter


class HarmonicBarrier(Force):
    """Purely repulsive harmonic barrier that may move in time (reference
    ``src/external.py:18-93``). ``location`` is a float or a callable of the
    timestep (HOOMD ``Variant``); ``params[type] = dict(k=..., offset=...)``."""

    _entry = None
    _valid = None
    _cpp_class_name = None

    def __init__(self, location):
        super().__init__()
        self.location = location
        self.params = TypeParameter("params", dict(k=float, offset=float), 1, self._mark_dirty)
        self._tables = None

    def _mark_dirty(self):
        self._tables = None

    def _location_at(self, timestep):
        return float(self.location(timestep)) if callable(self.location) else float(self.location)

    def _attach(self, sim):
        super()._attach(sim)
        self._tables = None

    def compute(self, timestep=None):
        import torch

        self._require()
        st = self._state
        self._ensure_buffers()
        if self._tables is None:
            raw = np.zeros((len(st.types), 2))
            for i, t in enumerate(st.types):
                d = self.params.get_raw(t)
                if d is None:
                    raise _lib.AzpError("%s.params[%r] is not set" % (type(self).__name__, t))
                raw[i] = (d["k"], d["offset"])
            self._tables = torch.from_numpy(raw).to(st.device)
        ts = self._sim.timestep if timestep is None else timestep
        loc = self._location_at(ts)
        box = st.box.to_c()
        lib = _lib.lib()
        if not getattr(lib, self._valid)(loc, C.byref(box)):
            raise _lib.AzpError("Barrier position is invalid")  # src/HarmonicBarrier.h:124-127
        a = _lib.BarrierArgs()
        a.d_force = self._force.data_ptr()
        a.d_virial = self._virial.data_ptr()
        a.virial_pitch = st.N
        a.N = st.N
        a.ntypes = len(st.types)
        a.d_pos = st.pos.data_ptr()
        a.box = box
        a.d_params = self._tables.data_ptr()
        a.location = loc
        stream = _lib.raw_stream(st.device)
        _lib.check(getattr(lib, self._entry)(C.byref(a), stream), self._entry)


class PlanarHarmonicBarrier(HarmonicBarrier):
    """Barrier at y = location (reference ``src/external.py:84-122``)."""

    _entry = "azp_external_planar_harmonic_barrier"
    _valid = "azp_planar_barrier_valid"
    _cpp_class_name = "PlanarHarmonicBarrier"


class SphericalHarmonicBarrier(HarmonicBarrier):
    """Barrier at radius location (reference ``src/external.py:123-155``)."""

    _entry = "azp_external_spherical_harmonic_barrier"
    _valid = "azp_spherical_barrier_valid"
    _cpp_class_name = "SphericalHarmonicBarrier"


__all__ = ["PlanarHarmonicBarrier", "SphericalHarmonicBarrier"]
