"""Bond potentials: drop-in mirror of ``hoomd.azplugins.bond`` (reference
``src/bond.py``) on libazp's gfx950 bond kernel."""

import ctypes as C

import numpy as np

from . import _lib
from .force import Force, TypeParameter


class Bond(Force):
    """Reduced ``hoomd.md.bond.Bond``: per-bond-type ``params``."""

    _cpp_class_name = None
    _entry = None
    _schema = {}
    _param_doubles = 4

    def __init__(self):
        super().__init__()
        self.params = TypeParameter("params", self._schema, 1, self._mark_dirty, self._readback)
        self._tables = None
        self.block_size = 0

    def _mark_dirty(self):
        self._tables = None
        self._computed_generation = None

    def _readback(self, key):
        """After attaching, ``params[...]`` returns what the C++ object holds (HOOMD:
        getParams -> asDict; src/pytest/test_bond.py:223)."""
        if not self._attached or key not in self.params._data:
            return None
        self._sync_cpp()
        return dict(self._cpp.getParams(key))

    def _attach(self, sim):
        import torch

        super()._attach(sim)
        self._tables = None
        self._cpp = None
        self._flags = torch.zeros(1, dtype=torch.int32, device=self._state.device)

    def _sync_cpp(self):
        """The ``_azplugins`` C++ object (class name + "GPU") with the current parameters."""
        types = list(self._state.bond_types) or ["A-A"]
        if getattr(self, "_cpp", None) is None:
            self._cpp = getattr(_lib.ext_module(), self._cpp_class_name + "GPU")(types)
        for t in self._state.bond_types:
            d = self.params.get_raw(t)
            if d is not None:
                self._cpp.setParams(t, d)

    def _build_tables(self):
        import torch

        types = self._state.bond_types
        for t in types:
            if self.params.get_raw(t) is None:
                raise _lib.AzpError("%s.params[%r] is not set" % (type(self).__name__, t))
        self._sync_cpp()
        raw = np.frombuffer(self._cpp.params_bytes(), dtype=np.float64).reshape(max(len(types), 1), -1).copy()
        assert raw.shape[1] == self._param_doubles
        self._tables = dict(params=torch.from_numpy(raw).to(self._state.device))

    def compute(self, timestep=None):
        import torch

        self._require()
        st = self._state
        self._ensure_buffers()
        if self._tables is None:
            self._build_tables()
            self._flags.zero_()  # (new parameters: the sticky "rejected" flag starts over, on the device and on the host)
            self._flag_pending = None
            if getattr(self, "_flag_host", None) is not None:
                getattr(self, "_flag_side").synchronize()
                self._flag_host.zero_()
        tab = st.bond_table()
        a = _lib.BondArgs()
        a.d_force = self._force.data_ptr()
        a.d_virial = self._virial.data_ptr()
        a.virial_pitch = st.N
        a.N = st.N
        a.n_max = st.n_max
        a.d_pos = st.pos.data_ptr()
        a.box = st.box.to_c()
        a.d_gpu_bondlist = tab["table"].data_ptr()
        a.d_gpu_bond_pos = tab["bond_pos"].data_ptr()
        a.d_gpu_n_bonds = tab["n_bonds"].data_ptr()
        a.pitch = tab["pitch"]
        a.n_bond_types = max(len(st.bond_types), 1)
        a.compute_virial = 1 if self.compute_virial else 0
        a.block_size = self.block_size
        # The evaluator's "rejected its parameters" flag (HOOMD: "bond.<name>: bond out of bounds") is sticky on the
        # device (the kernel only ever sets it) and travels to the host on a side stream behind each launch; it is
        # LOOKED AT when the next launch is queued, by which time it has long arrived -- a readback right behind
        # the launch would idle the GPU for a host round trip every step. check_flags() looks now.
        self.check_flags(wait=False)
        stream = _lib.raw_stream(st.device)
        fn = getattr(_lib.lib(), self._entry)
        _lib.check(fn(C.byref(a), self._tables["params"].data_ptr(), self._flags.data_ptr(), stream), self._entry)
        if getattr(self, "_flag_side", None) is None:
            self._flag_side = torch.cuda.Stream(device=st.device)
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        done = torch.cuda.Event()
        done.record()
        with torch.cuda.stream(self._flag_side):
            self._flag_side.wait_event(done)
            self._flag_host.copy_(self._flags, non_blocking=True)
        self._flag_pending = torch.cuda.Event()
        self._flag_pending.record(self._flag_side)
        self._computed_generation = st.position_generation
        if not self.defer_flag_check:
            self.check_flags(wait=True)

    defer_flag_check = False  # True inside Simulation.run: the flag of step k is examined when step k + 1 is queued

    def check_flags(self, wait=True):
        """Raise if an evaluator rejected its parameters in a launch whose flag has reached the host
        (``wait=True``: of every launch so far)."""
        ev = getattr(self, "_flag_pending", None)
        if ev is None:
            return
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        if int(self._flag_host[0]) != 0:
            # HOOMD: "bond.<name>: bond out of bounds" when the evaluator returns false
            raise _lib.AzpError("bond.%s: bond out of bounds (evaluator rejected its parameters)" % type(self).__name__)


class DoubleWell(Bond):
    """Double well bond potential (reference ``src/bond.py:13-65``)."""

    _cpp_class_name = "PotentialBondDoubleWell"
    _entry = "azp_bond_forces_double_well"
    _schema = dict(r_0=float, r_1=float, U_1=float, U_tilt=float)

    def _pack(self, d):
        out = np.zeros(4)
        _lib.lib().azp_dw_params_make(d["r_0"], d["r_1"], d["U_1"], d["U_tilt"], out.ctypes.data)
        return out

    def _unpack(self, raw):
        v = [C.c_double() for _ in range(4)]
        _lib.lib().azp_dw_params_unpack(raw.ctypes.data, *[C.byref(x) for x in v])
        return dict(r_0=v[0].value, r_1=v[1].value, U_1=v[2].value, U_tilt=v[3].value)


class Quartic(Bond):
    """Quartic bond potential (reference ``src/bond.py:68-157``; ``delta``
    defaults to 0, ``src/bond.py:153``)."""

    _cpp_class_name = "PotentialBondQuartic"
    _entry = "azp_bond_forces_quartic"
    _schema = dict(k=float, r_0=float, b_1=float, b_2=float, U_0=float, sigma=float, epsilon=float, delta=0.0)
    _param_doubles = 8

    def _pack(self, d):
        out = np.zeros(8)
        _lib.lib().azp_quartic_params_make(d["k"], d["r_0"], d["b_1"], d["b_2"], d["U_0"], d["sigma"], d["epsilon"],
                                           d["delta"], out.ctypes.data)
        return out

    def _unpack(self, raw):
        v = [C.c_double() for _ in range(8)]
        _lib.lib().azp_quartic_params_unpack(raw.ctypes.data, *[C.byref(x) for x in v])
        keys = ("k", "r_0", "b_1", "b_2", "U_0", "sigma", "epsilon", "delta")
        return {k: x.value for k, x in zip(keys, v)}


__all__ = ["DoubleWell", "Quartic"]
