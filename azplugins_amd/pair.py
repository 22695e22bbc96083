"""Pair potentials: drop-in mirror of ``hoomd.azplugins.pair`` (reference
``src/pair.py``) driving libazp's gfx950 kernels through the C ABI.

Class names, constructor arguments, parameter keys, accepted ``mode`` values and
the parameter round trip follow the reference (file:line in each class). What
the reference delegates to HOOMD's ``hoomd.md.pair.Pair`` -- ``r_cut`` / ``r_on``
type parameters, ``mode``, attaching -- is restated here in reduced form.
"""

import ctypes as C

import numpy as np

from . import _lib
from .force import Force, ScalarTypeParameter, TypeParameter

_SHIFT = {"none": 0, "shift": 1, "xplor": 2}


class Pair(Force):
    """Reduced ``hoomd.md.pair.Pair``: neighbor list, per-type-pair ``params``,
    ``r_cut``, ``r_on`` and ``mode``."""

    _cpp_class_name = None          # name the reference registers in _azplugins
    _entry = None                   # libazp entry point
    _planned_entry = None           # tile-plan entry point (isotropic pairs)
    _plan_balance = False           # plans from cells: rows to lanes by in-range length (expensive, ragged pair blocks)
    _schema = {}
    _param_doubles = 4              # size of the raw param struct in doubles
    _accepted_modes = ("none", "shift", "xplor")
    _halo_fields = ("pos",)         # per-particle arrays whose ghost rows the kernel reads (decomposed runs)

    def __init__(self, nlist, default_r_cut=None, default_r_on=0.0, mode="none"):
        super().__init__()
        if mode not in self._accepted_modes:
            raise ValueError("mode must be one of %s, got %r" % (self._accepted_modes, mode))
        self.nlist = nlist
        self._mode = mode
        self.params = TypeParameter("params", self._schema, 2, self._mark_dirty, self._readback)
        self.r_cut = ScalarTypeParameter("r_cut", default_r_cut, self._r_cut_changed)
        self.r_on = ScalarTypeParameter("r_on", default_r_on, self._mark_dirty)
        self.threads_per_particle = 0   # 0 = library heuristic (HOOMD autotunes this)
        self.block_size = 0
        self.use_plan = True            # LDS-staged tile kernel when the neighbor list can be tiled
        self.use_displacement_bound = True  # let it stop rows early while particles have barely moved (exact)
        # ... by the displacements of each tile's own particles instead of the global maximum: pays when a few
        # particles move much farther than the rest; on a liquid in equilibrium it costs 1 %, so opt-in
        self.use_local_displacement = False
        self.plan_bank_order = None     # None: bank-aware rows only for long-lived lists (below); True / False: always / never
        self.use_fused_plan = True      # sole consumer of its list: compile the plan straight from the binned particles
        self.use_speculative_launch = True  # queue the force kernel behind the list's distance check (_compute_speculative)
        self._plan_valid = False
        self._plan_ids = None
        self._plan = None
        self._plan_builds = None
        self._tables = None
        self._cpp = None            # the _azplugins C++ object (created on attach)
        self._cpp_synced = False
        nlist._add_consumer(self)

    # -- mode ----------------------------------------------------------------
    @property
    def mode(self):
        return self._mode

    @mode.setter
    def mode(self, m):
        if m not in self._accepted_modes:
            raise ValueError("mode must be one of %s, got %r" % (self._accepted_modes, m))
        self._mode = m
        self._mark_dirty()

    def _mark_dirty(self):
        self._tables = None
        self._cpp_synced = False
        self._computed_generation = None
        # the tile plan orders and classifies rows against the cutoffs / inner radii it was
        # built with: recompile it with the new tables
        self._plan_builds = None

    def _r_cut_changed(self):
        """HOOMD rebuilds the neighbor list when a consumer's r_cut matrix changes: the
        list on hand was built for the old r_cut + buffer."""
        self._mark_dirty()
        self.nlist._consumers_changed()

    # -- raw parameter structs (host side of the C ABI) ------------------------
    def _pack(self, d):
        raise NotImplementedError

    def _unpack(self, raw):
        raise NotImplementedError

    def _readback(self, key):
        """After attaching, ``params[...]`` returns what the C++ side holds
        (HOOMD: getParams -> asDict), as the reference tests check
        (src/pytest/test_pair.py:349)."""
        if not self._attached or key not in self.params._data:
            return None
        self._sync_cpp()
        return dict(self._cpp.getParams(*key))

    def _make_cpp(self):
        """The ``_azplugins`` class of this potential for a GPU device: ``_cpp_class_name +
        "GPU"`` (hoomd's _attach_hook; the conservative DPD potential only has a CPU class)."""
        ext = _lib.ext_module()
        cls = getattr(ext, self._cpp_class_name + "GPU", None) or getattr(ext, self._cpp_class_name)
        return cls(list(self._state.types))

    def _sync_cpp(self):
        """Push the Python-side type parameters into the C++ object (setParams / setRCut /
        setROn / mode), as HOOMD's TypeParameter machinery does on attach and on every change."""
        if self._cpp is None:
            self._cpp = self._make_cpp()
            self._cpp_synced = False
        if self._cpp_synced:
            return
        types = self._state.types
        for i, a in enumerate(types):
            for b in types[i:]:
                d = self.params.get_raw((a, b))
                if d is not None:
                    self._cpp.setParams(a, b, d)
                self._cpp.setRCut(a, b, self.r_cut[(a, b)])
                self._cpp.setROn(a, b, self.r_on[(a, b)])
        self._cpp.mode = self._mode
        self._cpp_synced = True

    def _types(self):
        return self._state.types if self._attached else None

    def _r_cut_matrix(self):
        self._require()
        types = self._state.types
        T = len(types)
        rc = np.zeros((T, T))
        for i, a in enumerate(types):
            for j, b in enumerate(types):
                rc[i, j] = self.r_cut[(a, b)]
        return rc

    def _attach(self, sim):
        super()._attach(sim)
        self._tables = None
        self._cpp = None
        self._cpp_synced = False

    def _build_tables(self):
        """Device tables from the C++ object's host tables (param_type structs, r_cut^2, r_on^2)."""
        import torch

        types = self._state.types
        T = len(types)
        for a in types:
            for b in types:
                if self.params.get_raw((a, b)) is None:
                    raise _lib.AzpError("%s.params[(%r, %r)] is not set" % (type(self).__name__, a, b))
        self._sync_cpp()
        raw = np.frombuffer(self._cpp.params_bytes(), dtype=np.float64).reshape(T * T, -1).copy()
        assert raw.shape[1] == self._param_doubles
        rc = np.asarray(self._cpp.rcutsq())
        ro = np.asarray(self._cpp.ronsq())
        dev = self._state.device
        self._spec_cache = None  # (argument structs kept by _compute_speculative point into the old tables)
        self._tables = dict(
            params=torch.from_numpy(raw).to(dev), rcutsq=torch.from_numpy(rc).to(dev), ronsq=torch.from_numpy(ro).to(dev)
        )
        # optional row-ordering hint for the tile plan (azp_pair_args.d_rinnersq)
        inner = np.zeros(T * T)
        for i, a in enumerate(types):
            for j, b in enumerate(types):
                inner[i * T + j] = self._inner_radius(self.params.get_raw((a, b))) ** 2
        if inner.any():
            self._tables["rinnersq"] = torch.from_numpy(inner).to(dev)

    def _inner_radius(self, d):
        """Radius inside which the potential takes a (rare) short-range branch; the
        tile plan lists such pairs first. 0 = no such branch."""
        return 0.0

    def _pair_args(self, for_launch=False):
        """azp_pair_args for the current state. Outside a launch the HOOMD-format list is always
        there (a fused list is materialized first): the struct can be handed to any entry point."""
        st = self._state
        nl = self.nlist
        a = _lib.PairArgs()
        a.d_force = self._force.data_ptr()
        a.d_virial = self._virial.data_ptr()
        a.virial_pitch = st.N
        a.N = st.N
        a.n_max = st.n_max
        a.d_pos = st.pos.data_ptr()
        a.box = st.box.to_c()
        a.d_n_neigh = nl.n_neigh.data_ptr()
        fused = getattr(nl, "_fused_active", False)
        if fused and not (for_launch and self.use_plan and self._planned_entry is not None and self.use_fused_plan):
            nl.leave_fused_mode()  # this launch needs the HOOMD-format list (generic kernel)
            fused = False
        if fused:
            # no HOOMD-format list: the tile plan compiled from the cells is its own list (ids set
            # by _prepare_plan; placeholders until then)
            ids = self._plan_ids if self._plan_builds == self._plan_key() else None
            a.d_nlist, a.d_head_list = ids if ids else (nl.n_neigh.data_ptr(), nl.n_neigh.data_ptr())
        else:
            if self._plan_ids is not None:
                # the plan on hand was compiled from the cells and is its own list: recompile from the u32 list
                self._plan_ids = None
                self._plan_builds = None
            a.d_nlist = nl.nlist.data_ptr()
            a.d_head_list = nl.head_list.data_ptr()
        a.d_rcutsq = self._tables["rcutsq"].data_ptr()
        a.d_ronsq = self._tables["ronsq"].data_ptr()
        if "rinnersq" in self._tables:
            a.d_rinnersq = self._tables["rinnersq"].data_ptr()
        a.size_nlist = 0 if fused else nl.size
        a.ntypes = len(st.types)
        a.shift_mode = _SHIFT[self._mode]
        a.compute_virial = 1 if self.compute_virial else 0
        a.block_size = self.block_size
        a.threads_per_particle = self.threads_per_particle
        if not self.use_plan:
            a.flags = _lib.PAIR_FLAG_NO_AUTO_PLAN  # use_plan = False means the generic kernel, not libazp's own plan cache
        a.r_list_max = nl.r_list_max
        # displacement of any particle since the PLAN was built <= displacement since the list was
        # built (now) + the same quantity at the time the plan was built (0 in the usual flow)
        bound = nl.displacement_bound(st) if hasattr(nl, "displacement_bound") else None
        d0 = getattr(self, "_plan_disp0", None)
        if bound is not None and d0 is not None and self.use_displacement_bound:
            a.has_displacement_bound, a.displacement_bound = 1, bound + d0
            darr = nl.displacements(st) if (self.use_local_displacement and hasattr(nl, "displacements")) else None
            if darr is not None and darr.shape[0] == st.n_max:
                self._disp_keepalive = darr
                a.d_displacement, a.displacement_bound_extra = darr.data_ptr(), d0
        rng = getattr(self, "_range", None)
        if rng is not None:
            a.range_first, a.range_count = int(rng[0]), int(rng[1])
        return a

    def compute(self, timestep=None, particle_range=None):
        """Evaluate the forces. ``particle_range=(first, count)`` restricts the
        launch to a sub-range of the local particles (domain-decomposed runs compute
        the interior while the halo exchange is in flight)."""
        import torch

        self._require()
        st = self._state
        if particle_range is None and self._compute_speculative(timestep):
            return
        self.nlist.compute(st)
        self._ensure_buffers()  # after the list: a rebuild of a decomposed run migrates particles
        if self._tables is None:
            self._build_tables()
        stream = _lib.raw_stream(st.device)
        self._range = particle_range
        self._launch(stream, timestep)
        self._range = None
        self._computed_generation = st.position_generation

    def _compute_speculative(self, timestep):
        """The common MD step -- the list was built earlier, the particles have moved, the plan is current: queue the
        neighbor list's distance check AND the force kernel behind it before asking for the check's result. The
        kernel reads the displacement bound from the check's device words and leaves at once if the check asks for
        a rebuild (azp_pair_args.d_stale_flag); the host then waits for the check alone while the GPU is already
        computing forces, instead of idling it for a readback and a launch every step. Returns False when the
        ordinary path has to run (first call, rebuild, no plan, decomposed run)."""
        nl, st = self.nlist, self._state
        if not (self.use_speculative_launch and self.use_plan and self._planned_entry is not None and self.use_displacement_bound
                and self._plan is not None and self._tables is not None and hasattr(nl, "begin_check")):
            return False
        if not (nl.built and nl.reduce_flag is None and nl.before_rebuild is None and self._plan_builds == self._plan_key()
                and nl._built_consumer_version == nl._consumer_version and nl._built_generation != st.position_generation
                and getattr(st, "order_generation", 0) == getattr(nl, "_order_generation", 0)
                and getattr(self, "_plan_disp0", None) is not None and self._force.shape[0] == st.N):
            return False
        if not self._plan_valid:
            return False
        token = nl.begin_check(st)
        # the argument struct of the previous step, while nothing it points to has moved (the host side of a step has to
        # stay shorter than NVE + distance check on the GPU, ~50 us at N = 2^20, or the force kernel waits for its launch)
        boxc = st.box.to_c()
        sig = (self._plan_builds, self._force.data_ptr(), self._virial.data_ptr(), st.pos.data_ptr(), nl.n_neigh.data_ptr(), id(self._tables),
               self._mode, self.compute_virial, self._plan_disp0, st.N, st.n_max, self.block_size)
        cache = self.__dict__.get("_spec_cache")
        if cache is not None and cache[0] == sig and cache[1] is boxc:
            a = cache[2]
        else:
            a = self._pair_args(for_launch=True)
            a.has_displacement_bound, a.displacement_bound, a.d_displacement = 0, 0.0, None
            a.displacement_bound_extra = self._plan_disp0
            self._spec_cache = (sig, boxc, a)
        a.d_stale_flag, a.d_displacement_sq_bits = token["flag_ptr"], token["bits_ptr"]
        args = self._wrap_args(a, timestep)
        stream = _lib.raw_stream(st.device)
        fn = getattr(_lib.lib(), self._planned_entry)
        _lib.check(fn(self._plan.handle, C.byref(args), self._tables["params"].data_ptr(), stream), self._planned_entry)
        self._calls_since_plan = getattr(self, "_calls_since_plan", 0) + 1
        if nl.end_check(token):
            return False  # rebuild: the kernel left without writing; the ordinary path takes over (it has the verdict)
        nl._built_generation = st.position_generation
        self._computed_generation = st.position_generation
        return True

    def _prepare_plan(self, a, stream):
        """Compile the tile plan when the neighbor list was rebuilt since the last compile and
        fill in the displacement fields of ``a`` for this launch."""
        if self._plan is None:
            self._plan = _lib.PairPlan()
        key = self._plan_key()
        nl = self.nlist
        if self._plan_builds != key and getattr(nl, "_fused_active", False):
            # the plan straight from the binned particles: no u32 list, no hash set, one kernel
            first, count = a.range_first, a.range_count
            a.range_first = a.range_count = 0
            cap = getattr(nl, "_plan_row_capacity", 0) or 160
            info = None
            self._plan.set_balance(self._plan_balance)
            for _ in range(4):
                self._plan.build_from_cells(nl.cells_args(cap), a, stream)
                info = self._plan.info()
                if info["valid"]:
                    if nl._cells.cell_subdivision == 2:
                        nl._half_failures = 0
                    break
                if info["invalid_reason"] == 3:
                    cap = (int(info["max_row"] * 1.06) + 4 + 7) // 8 * 8  # a row overflowed: longer rows (HOOMD's protocol)
                elif nl._cells.cell_subdivision == 2:
                    # half-width cells refused (a tile's members too spread out, a very dense run of cells): bin the
                    # same positions into cells of the full list radius and compile from those
                    nl._half_failures = getattr(nl, "_half_failures", 0) + 1
                    nl.rebin_full()
                else:
                    break
            a.range_first, a.range_count = first, count
            self._plan_valid = bool(info["valid"])
            if info["valid"]:
                nl._fused_failures = 0
                # the members of the tiles drift apart between two particle sorts; past 128 cells under one tile the
                # compile is refused: ask for a sort when that comes near (and the count has grown since the last sort)
                mc = int(info.get("max_member_cells", 0))
                if getattr(nl, "_member_cells_key", None) != getattr(self._state, "order_generation", 0):
                    nl._member_cells_key, nl._member_cells_sorted = getattr(self._state, "order_generation", 0), mc
                if mc > 108 and mc > 1.15 * nl._member_cells_sorted:
                    nl._sort_wanted = True
                nl._plan_row_capacity = max((int(info["max_row"] * 1.06) + 4 + 7) // 8 * 8, 32)
                nl._fused_counts_ready = True
                self._plan_ids = (info["list_id"], info["head_id"])
                a.d_nlist, a.d_head_list = self._plan_ids
                self._plan_builds = key
                self._calls_since_plan = 0
                # row classes were cut at the positions the LIST was built from, whenever this compile runs
                self._plan_disp0 = 0.0
                b = nl.displacement_bound(self._state)
                known = b is not None and self.use_displacement_bound
                a.has_displacement_bound, a.displacement_bound = (1, b) if known else (0, 0.0)
                a.displacement_bound_extra = 0.0
            else:
                # particles not spatially sorted / a tile stages too much: the list-based path. A list
                # whose tiles fail twice in a row (e.g. the thin boundary shells of a decomposed DPD
                # fluid: 256 particles in more than 128 cells) stops trying at every rebuild
                nl._fused_failures = getattr(nl, "_fused_failures", 0) + 1
                if nl._fused_failures >= 2:
                    nl.fused = False
                    nl._fused_auto_off = True   # (a particle sort switches it back on: Simulation.run)
                if info["invalid_reason"] in (4, 5):
                    # the members of some tile have drifted apart (a fast-diffusing fluid between two particle sorts):
                    # ask for a sort now rather than at the sorter's next period
                    nl._sort_wanted = True
                nl.leave_fused_mode()
                a.d_nlist = nl.nlist.data_ptr()
                a.d_head_list = nl.head_list.data_ptr()
                a.size_nlist = nl.size
        if self._plan_builds != key:
            # recompile the plan only when the neighbor list was rebuilt
            first, count = a.range_first, a.range_count
            a.range_first = a.range_count = 0
            # bank-aware rows pay off only for lists that live long (include/azp.h): keep them
            # for the first build and whenever the previous list served >= 50 force calls
            calls = getattr(self, "_calls_since_plan", None)
            self._plan.set_bank_order((calls is None or calls >= 50) if self.plan_bank_order is None else self.plan_bank_order)
            self._calls_since_plan = 0
            self._plan.build(a, stream)
            a.range_first, a.range_count = first, count
            self._plan_valid = bool(self._plan.info()["valid"])
            self._plan_builds = key
            self._plan_disp0 = self.nlist.displacement_bound(self._state)
            # this launch sees exactly the positions the plan was built from
            a.has_displacement_bound, a.displacement_bound = (1 if self.use_displacement_bound else 0), 0.0
            a.d_displacement = None  # (those refer to the positions the LIST was built from)
        self._calls_since_plan = getattr(self, "_calls_since_plan", 0) + 1

    def _plan_key(self):
        return (id(self.nlist), self.nlist.num_builds, self.threads_per_particle)

    def _wrap_args(self, a, timestep):
        """The argument struct of this potential's entry points around the common pair args."""
        return a

    def _launch(self, stream, timestep):
        a = self._pair_args(for_launch=True)
        planned = self.use_plan and self._planned_entry is not None
        if planned:
            self._prepare_plan(a, stream)
        args = self._wrap_args(a, timestep)
        params = self._tables["params"].data_ptr()
        if planned:
            fn = getattr(_lib.lib(), self._planned_entry)
            _lib.check(fn(self._plan.handle, C.byref(args), params, stream), self._planned_entry)
            return
        fn = getattr(_lib.lib(), self._entry)
        _lib.check(fn(C.byref(args), params, stream), self._entry)

    @property
    def plan_info(self):
        return self._plan.info() if self._plan is not None else None


def _d(x):
    return C.c_double(x)


class Colloid(Pair):
    """Colloid pair potential (reference ``src/pair.py:14-118``)."""

    _cpp_class_name = "PotentialPairColloid"
    _entry = "azp_pair_forces_colloid"
    _planned_entry = "azp_pair_forces_planned_colloid"
    _schema = dict(A=float, a_1=float, a_2=float, sigma=float)

    def _pack(self, d):
        out = np.zeros(4)
        _lib.lib().azp_colloid_params_make(d["A"], d["a_1"], d["a_2"], d["sigma"], out.ctypes.data)
        return out

    def _unpack(self, raw):
        v = [C.c_double() for _ in range(4)]
        _lib.lib().azp_colloid_params_unpack(raw.ctypes.data, *[C.byref(x) for x in v])
        return dict(A=v[0].value, a_1=v[1].value, a_2=v[2].value, sigma=v[3].value)


class ExpandedYukawa(Pair):
    """Expanded Yukawa pair potential (reference ``src/pair.py:242-297``)."""

    _cpp_class_name = "PotentialPairExpandedYukawa"
    _entry = "azp_pair_forces_expanded_yukawa"
    _planned_entry = "azp_pair_forces_planned_expanded_yukawa"
    _schema = dict(epsilon=float, kappa=float, delta=float)

    def _pack(self, d):
        return np.array([d["epsilon"], d["kappa"], d["delta"], 0.0])

    def _unpack(self, raw):
        return dict(epsilon=float(raw[0]), kappa=float(raw[1]), delta=float(raw[2]))


class Hertz(Pair):
    """Hertz pair potential (reference ``src/pair.py:300-351``)."""

    _cpp_class_name = "PotentialPairHertz"
    _entry = "azp_pair_forces_hertz"
    _planned_entry = "azp_pair_forces_planned_hertz"
    _schema = dict(epsilon=float)
    _param_doubles = 1

    def _pack(self, d):
        return np.array([d["epsilon"]])

    def _unpack(self, raw):
        return dict(epsilon=float(raw[0]))


class PerturbedLennardJones(Pair):
    """Perturbed Lennard-Jones pair potential (reference ``src/pair.py:354-426``)."""

    _cpp_class_name = "PotentialPairPerturbedLennardJones"
    _entry = "azp_pair_forces_perturbed_lennard_jones"
    _planned_entry = "azp_pair_forces_planned_perturbed_lennard_jones"
    _schema = dict(epsilon=float, sigma=float, attraction_scale_factor=float)

    def _pack(self, d):
        out = np.zeros(4)
        _lib.lib().azp_plj_params_make(d["epsilon"], d["sigma"], d["attraction_scale_factor"], out.ctypes.data)
        return out

    def _unpack(self, raw):
        v = [C.c_double() for _ in range(3)]
        _lib.lib().azp_plj_params_unpack(raw.ctypes.data, *[C.byref(x) for x in v])
        return dict(epsilon=v[0].value, sigma=v[1].value, attraction_scale_factor=v[2].value)

    def _inner_radius(self, d):
        # the WCA core (r < 2^(1/6) sigma), plus the distance a pair can close before
        # the next neighbor-list rebuild
        # (+ 1e-3: the tile kernel drops its core test beyond the first chunks of a row only while
        # r_wca + 2 x displacement bound clears this radius by the plan's single-precision margin)
        return 2.0 ** (1.0 / 6.0) * d["sigma"] + self.nlist.buffer + 1e-3


class DPDGeneralWeight(Pair):
    """DPD thermostat with generalized weight function (reference
    ``src/pair.py:121-239``). ``kT`` may be a float or a callable of the
    timestep (HOOMD ``Variant``). ``mode`` is always ``"none"``."""

    _cpp_class_name = "PotentialPairDPDThermoGeneralWeight"
    _halo_fields = ("pos", "vel")   # (ghost tags change only when the ghosts are re-selected)
    _entry = "azp_dpd_forces_general_weight"
    _planned_entry = "azp_dpd_forces_planned_general_weight"
    _plan_balance = True
    _schema = dict(A=float, gamma=float, s=float)
    _accepted_modes = ("none",)

    def __init__(self, nlist, kT, default_r_cut=None):
        super().__init__(nlist, default_r_cut=default_r_cut, default_r_on=0.0, mode="none")
        self.kT = kT

    def _pack(self, d):
        return np.array([d["A"], d["gamma"], d["s"], 0.0])

    def _unpack(self, raw):
        return dict(A=float(raw[0]), gamma=float(raw[1]), s=float(raw[2]))

    def _attach(self, sim):
        # DPD uses RNGs: warn if the seed was not set (src/pair.py:236-239)
        sim._warn_if_seed_unset()
        super()._attach(sim)

    def _wrap_args(self, a, timestep):
        st = self._state
        d = _lib.DPDArgs()
        d.pair = a
        d.d_vel = st.vel.data_ptr()
        d.d_tag = st.tag.data_ptr()
        ts = self._sim.timestep if timestep is None else timestep
        d.timestep = int(ts)
        d.deltaT = float(self._sim.dt)
        d.T = float(self.kT(ts)) if callable(self.kT) else float(self.kT)
        d.seed = int(self._sim.seed) & 0xFFFF
        return d


class DPDConservativeGeneralWeight(Pair):
    """Conservative part only: the reference also registers
    ``PotentialPairConservativeGeneralWeight``
    (src/export_PotentialPairDPDThermo.cc.inc:33-35)."""

    _cpp_class_name = "PotentialPairConservativeGeneralWeight"
    _entry = "azp_pair_forces_dpd_conservative"
    _planned_entry = "azp_pair_forces_planned_dpd_conservative"
    _schema = dict(A=float, gamma=float, s=float)
    _accepted_modes = ("none",)

    def __init__(self, nlist, default_r_cut=None):
        super().__init__(nlist, default_r_cut=default_r_cut, default_r_on=0.0, mode="none")

    _pack = DPDGeneralWeight._pack
    _unpack = DPDGeneralWeight._unpack


class TwoPatchMorse(Pair):
    """Two-patch Morse anisotropic pair potential (reference
    ``src/pair.py:429-525``; HOOMD ``AnisotropicPair``: no ``r_on``, modes
    ``"none"`` / ``"shift"``)."""

    _cpp_class_name = "AnisoPotentialPairTwoPatchMorse"
    _halo_fields = ("pos", "orientation")
    _entry = "azp_aniso_forces_two_patch_morse"
    _planned_entry = "azp_aniso_forces_planned_two_patch_morse"
    _plan_balance = True
    _schema = dict(M_d=float, M_r=float, r_eq=float, omega=float, alpha=float, repulsion=bool)
    _param_doubles = 6
    _accepted_modes = ("none", "shift")

    def __init__(self, nlist, default_r_cut=None, mode="none"):
        super().__init__(nlist, default_r_cut=default_r_cut, default_r_on=0.0, mode=mode)

    def _pack(self, d):
        out = np.zeros(6)
        _lib.lib().azp_tpm_params_make(d["M_d"], d["M_r"], d["r_eq"], d["omega"], d["alpha"], int(d["repulsion"]),
                                       out.ctypes.data)
        return out

    def _unpack(self, raw):
        v = [C.c_double() for _ in range(5)]
        rep = C.c_int()
        _lib.lib().azp_tpm_params_unpack(raw.ctypes.data, *[C.byref(x) for x in v], C.byref(rep))
        return dict(M_d=v[0].value, M_r=v[1].value, r_eq=v[2].value, omega=v[3].value, alpha=v[4].value,
                    repulsion=bool(rep.value))

    def _wrap_args(self, a, timestep):
        st = self._state
        g = _lib.AnisoArgs()
        g.pair = a
        g.d_orientation = st.orientation.data_ptr()
        g.d_torque = self._torque.data_ptr()
        return g


__all__ = ["Colloid", "DPDGeneralWeight", "DPDConservativeGeneralWeight", "ExpandedYukawa", "Hertz",
           "PerturbedLennardJones", "TwoPatchMorse"]
